#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done

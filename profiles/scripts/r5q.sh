#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5q; mkdir -p $O; cd $R
echo "== --strong 1024 as the main line"; timeout -k 10 300 python bench.py --strong 1024 --steps 3 --warmup 1 --no-cpu --no-host > $O/strong.json 2>$O/strong.err; echo rc=$?; python3 -c "
import json; d=json.loads(open('$O/strong.json').read().strip().splitlines()[-1]); d.pop('details',None); print(json.dumps(d)[:1500])"
bash profiles/scripts/r5_kstats.sh

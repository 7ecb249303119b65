#!/bin/bash
# round 5: the wide fused y+z kernels with ONE workgroup per CU (unused dynamic LDS), with and without the hold
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for cfg in "0 0" "0 40000" "1 40000" "1 0"; do set -- $cfg
echo "== sched $1 yzpad $2"; SIFT3D_AMD_SCHED=$1 SIFT3D_AMD_YZPAD=$2 SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done; done

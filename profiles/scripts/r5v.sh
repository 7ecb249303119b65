#!/bin/bash
# round 5: z segments of the extrema sweep (diag build: SIFT3D_AMD_SWEEP_WGS = workgroups aimed at)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for w in 2048 2304 3072 4608 1536; do
echo "== sweep workgroups $w"; SIFT3D_AMD_SWEEP_WGS=$w SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg --no-pyramid-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['candidates'], {k:round(1e3*v,3) for k,v in d['stage_s'].items() if k in ('extrema','detect_wall','detect_dev','gauss')})"
done; done

#!/bin/bash
# round 5: the hold of octave 0's last blurs (diag build: SIFT3D_AMD_SCHED 0 = x pass not held, 4 = held as in round 4,
# 1 = no hold at all)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
echo "== golden tests"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden" > $O/t.log 2>&1; tail -2 $O/t.log
for rep in 1 2 3; do for m in 0 4 1; do
echo "== sched $m"; SIFT3D_AMD_SCHED=$m SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done; done

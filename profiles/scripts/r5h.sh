#!/bin/bash
# round 5: k_orient_sums with the window's planes staged in LDS (diag build: SIFT3D_AMD_ORI_ABLATE=4 = the gather from memory)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
echo "== tests (orientation, goldens, sharded)"; timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "orient or golden or vs_oracle or sharded or slab or config4" > $O/t.log 2>&1; tail -4 $O/t.log
for rep in 1 2 3; do for m in 0 4; do
echo "== ORI_ABLATE=$m"; SIFT3D_AMD_ORI_ABLATE=$m SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['candidates'], d['keypoints'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done; done

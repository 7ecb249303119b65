#!/bin/bash
# round 5: octave 0's tail on a CU-masked stream (diag build: SIFT3D_AMD_CUMASK = CUs left to the other streams)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
echo "== new tests"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wider or golden" > $O/t.log 2>&1; tail -4 $O/t.log
for rep in 1 2; do for m in 0 16 32 64; do for sc in 0 1; do
echo "== cumask $m sched $sc"; SIFT3D_AMD_CUMASK=$m SIFT3D_AMD_SCHED=$sc SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['candidates'], d['keypoints'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done; done; done

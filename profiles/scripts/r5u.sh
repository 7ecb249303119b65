#!/bin/bash
# round 5: descriptor windows summed in four parts (work items of a quarter of the size: the persistent kernel's drain)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5u; mkdir -p $O; cd $R
echo "== tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t.log 2>&1; tail -4 $O/t.log
echo "== tail, new"; timeout -k 10 300 python profiles/microbench/describe_tail.py 2>&1 | tail -1
echo "== tail, prev"; SIFT3D_AMD_LIB=$R/scratch/prev.so timeout -k 10 300 python profiles/microbench/describe_tail.py 2>&1 | tail -1
for rep in 1 2 3; do for lib in scratch/prev.so sift3d_amd/libsift3d_amd.so; do
echo "== $lib"; SIFT3D_AMD_LIB=$R/$lib timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg --no-pyramid-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items() if k in ('describe','describe_wall','detect_wall')})"
done; done

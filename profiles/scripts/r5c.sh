#!/bin/bash
# round 5: schedule experiments (diag build: SIFT3D_AMD_SCHED 0 = shipped, 1 = no hold, 2 = round-4 joins, 3 = both)
# + the new tests (wide filters, sigma0 = 8) + a kernel trace of the shipped schedule
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
echo "== new tests"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wider or sigma8 or golden" > $O/t.log 2>&1; tail -4 $O/t.log
for rep in 1 2; do for m in 0 1 2 3; do
echo "== sched $m"; SIFT3D_AMD_SCHED=$m SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done; done
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
SIFT3D_AMD_SCHED=$m SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/kt$m -o run --output-format csv -- python3 $R/bench.py --no-cpu --no-host --no-micro --no-strong-leg --steps 10 --warmup 3 > $O/kt$m.json 2> $O/kt$m.err
cd $R; f=$(ls $O/kt$m/*/run_kernel_trace.csv $O/kt$m/run_kernel_trace.csv 2>/dev/null | head -1); echo "trace $f"; python3 profiles/timeline.py $f 8 > $O/timeline$m.txt 2>&1; cd /tmp
done

#!/bin/bash
# round 5: the new API test + the N = 2 process path of bench.py on one device (gloo rehearsal transport)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
echo "== test"; timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pyramid_only" 2>&1 | tail -5
echo "== bench --gpus 2 (rehearsal: two processes share device 0, exchanges staged through gloo)"
SIFT3D_AMD_REHEARSE=1 timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu > $O/bench_n2.json 2> $O/bench_n2.err; echo rc=$?; tail -c 2500 $O/bench_n2.json; echo; tail -5 $O/bench_n2.err

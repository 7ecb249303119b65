#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
python3 profiles/microbench/desc_trace.py /tmp/desc_trace.bin > $O/trace.log 2>&1
cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_f64_atomic lds_f64_atomic.hip && cd $R &&
timeout -k 10 300 /tmp/lds_f64_atomic /tmp/desc_trace.bin > $O/lds_f64_atomic.txt 2>&1
cat $O/lds_f64_atomic.txt
echo "== tests"; timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); d.pop('details',None); print(json.dumps(d)[:3000])
PY

#!/bin/bash
# what the driver runs at round end: the gpu suite, smoke(), the default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5check; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > $O/bench_default.json 2>$O/bench_default.err; echo rc=$?; python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); d.pop('details',None); d.pop('cpu_baseline',None); print(json.dumps(d)[:1200])"

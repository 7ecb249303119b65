#!/bin/bash
# the in-step kernel trace alone (as in r5_profiles.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5final; mkdir -p $O; rm -rf $O/kstats
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kstats -o run --output-format csv -- python3 $R/bench.py --no-cpu --no-host --no-micro --no-strong-leg --no-pyramid-leg --steps 10 --warmup 3 > $O/kstats_bench.json 2> $O/kstats.err
tail -c 600 $O/kstats_bench.json

#!/bin/bash
# round 5: new stream schedule (octave 0's sweep starts when octave 0 ends; one scan + emission launch) -- tests,
# the new bench line, and a same-box A/B of the step against the build before it (scratch/base, not committed)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
echo "== tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
echo "== bench"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; tail -3 $O/bench.err; python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); det=d.pop('details',None); print(json.dumps(d)[:4500])
if det: print(json.dumps(det.get('in_step_launches'))[:3000])
PY
if [ -d scratch/base ]; then
for i in 1 2; do
echo "== A/B base"; (cd scratch/base && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_s'])")
echo "== A/B new"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_s'])"
done
fi

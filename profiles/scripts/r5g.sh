#!/bin/bash
# round 5: orient_serial with its samples requested a batch ahead: full gpu suite + step A/B against scratch/base
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
echo "== tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
for rep in 1 2; do
echo "== new"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
echo "== serial orientation (every candidate through orient_serial)"; timeout -k 10 300 python - <<'PY'
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from sift3d_amd import api, hip
n = 256
vol = torch.empty((n, n, n), device="cuda"); hip.synth_lattice(vol, 0, 11); torch.cuda.synchronize()
det = api.Detector(); kp = api.KeypointStore()
api.lib().sift3d_amd_detector_set_serial_orientation(det.h, 1)
for i in range(3):
    assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
print("256^3 serial orientation stage: %.3f ms, %d candidates -> %d keypoints" % (1e3 * det.timings()["orient"], det.num_candidates(), len(kp)))
PY
done

#!/usr/bin/env python3
"""describe_tail.py -- what part of k_describe's time does not scale with its list (the drain of the persistent
kernel: DESIGN.md 3.3, "A window in four parts").

    python3 profiles/microbench/describe_tail.py          # on an MI355X

The 512^3 bench workload is detected once; its keypoint list -- whole, and the keypoints of one level alone -- is
described 1x and 3x replicated: time = per-list part * replicas + fixed part.  Round 5: whole list 25.97 ms per
list + 1.06 ms fixed with whole windows as work items (level s = 0 / 1 / 2 alone: fixed 1.25 / 2.00 / 3.65 ms =
0.8 x the level's window time), 25.86 + 0.39 ms with every window summed in four parts.
"""
import os
import sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sift3d_amd import api, hip
n = 512
vol = torch.empty((n, n, n), device="cuda"); hip.synth_lattice(vol, 0, 11); torch.cuda.synchronize()
det, kp, desc = api.Detector(), api.KeypointStore(), api.DescriptorStore()
assert det.detect_keypoints_device(vol.data_ptr(), n, n, n, kp) == 0
recs = kp.records()
def t(r, reps=5):
    k2 = api.KeypointStore(); k2.set_records(r)
    d2 = api.DescriptorStore()
    ts = []
    for _ in range(reps):
        assert det.extract_descriptors(k2, d2) == 0
        ts.append(det.timings()["describe"])
    return 1e3 * float(np.median(ts[1:]))
t1, t3 = t(recs), t(np.concatenate([recs] * 3))
print("whole list: %d keypoints: 1x %.3f ms, 3x %.3f ms -> per list %.3f ms, fixed part %.3f ms"
      % (len(recs), t1, t3, (t3 - t1) / 2, t1 - (t3 - t1) / 2))
for s in (0, 1, 2):
    sub = recs[recs["s"] == s]
    t1, t3 = t(sub), t(np.concatenate([sub] * 3))
    print("level s=%d: %d keypoints: 1x %.3f ms, 3x %.3f ms -> per list %.3f ms (%.2f us per keypoint and wave slot), fixed part %.3f ms"
          % (s, len(sub), t1, t3, (t3 - t1) / 2, 1e3 * (t3 - t1) / 2 * 4096 / len(sub), t1 - (t3 - t1) / 2))

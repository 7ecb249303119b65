#!/bin/bash
# round 5: k_fir_yz_dma with two rows per thread (diag build: SIFT3D_AMD_YZR2 = bit mask of half widths)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
echo "== fused yz tests, every instance with two rows per thread"
SIFT3D_AMD_YZR2=510 SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_yz or golden or thinnest" > $O/t.log 2>&1; tail -4 $O/t.log
cat > /tmp/yzb.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from sift3d_amd import api, hip
n = 512
sig = [0.5387011637869722, 0.9732939207323564, 1.2262734984654078, 1.5450077936447955, 1.9465878414647133, 2.4525469969308156]
src = torch.empty((n, n, n), device="cuda"); dst = torch.empty_like(src)
hip.synth_lattice(src, 0, 11)
out = []
for s in sig:
    taps = api.gauss_filter(s)
    hip.fir_yz(src, dst, taps); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for r in range(20):
        hip.fir_yz(src, dst, taps); ev[r + 1].record()
    torch.cuda.synchronize()
    t = np.sort([ev[r].elapsed_time(ev[r + 1]) for r in range(20)])
    out.append("%d taps: median %.4f min %.4f ms" % (len(taps), np.median(t), t[0]))
print("; ".join(out))
PY
for rep in 1 2; do for m in 0 510; do
echo "== YZR2=$m"; SIFT3D_AMD_YZR2=$m SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 200 python /tmp/yzb.py $R 2>/dev/null
done; done
for rep in 1 2; do for m in 0 510 320; do
echo "== step YZR2=$m"; SIFT3D_AMD_YZR2=$m SIFT3D_AMD_LIB=$R/scratch/diag.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-micro --no-host --no-strong-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], {k:round(1e3*v,3) for k,v in d['stage_s'].items()})"
done; done

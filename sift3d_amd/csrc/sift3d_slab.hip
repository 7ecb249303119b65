// sift3d_slab.hip -- small device stages of the Z-slab driver (sift3d_sharded.c) that are not part of
// the single-GPU path: the reduction step of the thread transport's all-reduce, and (below) the kernels
// that build the global keypoint list on the device from the gathered per-rank lists.
//
// The reference has no counterpart (no collective call sites; its only parallelism on this
// path is the OpenMP loop over keypoints, sift.c:1117); what these kernels must reproduce is the ORDER of
// its lists: (octave, level) major, then the scan order (z, y, x) inside a level (sift.c:835-871), and the
// stale-strength quirk of the in-place compaction (sift.c:372-384, 1148-1162).
#include "sift3d_kernels_common.h"

// ---- all-reduce(max) of the thread transport: dst[i] = max over rows ------------------------------
// (maxima of non-negative floats -- max|v| and the per-level max|DoG| --: order-free, hence exact)
__global__ __launch_bounds__(256) void k_max_rows(float *__restrict__ dst, const float *__restrict__ rows,
                                                  int nrows, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    float m = rows[i];
    for (int r = 1; r < nrows; r++)
        m = fmaxf(m, rows[(size_t)r * n + i]);
    dst[i] = m;
}

extern "C" {

int sift3d_hip_max_rows(float *d_dst, const float *d_rows, int nrows, int n, void *stream)
{
    if (n < 1 || nrows < 1)
        return SIFT3D_SUCCESS;
    hipLaunchKernelGGL(k_max_rows, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_dst, d_rows,
                       nrows, n);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

} // extern "C"

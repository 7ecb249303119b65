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

// ---- the global keypoint list, built on the device ------------------------------------------------
// After the orientation stage a rank holds its candidates (d_cand, in the reference's scan order), their
// keep flags and matrices.  What crosses ranks: per-(octave, level) counts, the candidates'
// |DoG| values and the ORIENTED keypoints.  k_slab_count forms the counts; k_slab_pack this rank's block
// for the one all-gather (values of all candidates, then the kept records in order -- an exclusive scan of
// the keep flags); k_slab_build, on every rank, the global list from the gathered blocks: position g of the
// global order belongs to segment (key, rank) -- keys ascending, ranks in slab order inside a key, a rank's
// own order inside a segment: the reference's (o, s, z, y, x) order (sift.c:835-871) -- and carries the
// strength of GLOBAL CANDIDATE g, because the reference's in-place compaction copies everything but
// `strength` (sift.c:372-384, 1148-1162: quirk Q2).
struct SlabRec {          // an oriented keypoint as exchanged between ranks (= sh_gkp of sift3d_sharded.c)
    int32_t o, s, x, y, z;
    float R[9];
};
struct SlabOut {          // ... and as delivered to the host
    int32_t o, s, x, y, z;
    float R[9];
    float strength;
    int32_t pad;
};
static_assert(sizeof(SlabRec) == 56 && sizeof(SlabOut) == 64, "record layouts");

__global__ __launch_bounds__(256) void k_slab_count(const sift3d_hip_cand *__restrict__ cand,
                                                    const int32_t *__restrict__ keep, uint32_t n, int ngl, int K,
                                                    int nkey, int32_t *__restrict__ cnt)
{
    __shared__ int32_t h[2 * 256];
    for (int i = threadIdx.x; i < 2 * nkey; i += 256)
        h[i] = 0;
    __syncthreads();
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < n; q += gridDim.x * 256) {
        const int tag = cand[q].tag;
        const int key = (tag / ngl) * K + (tag % ngl - 1);
        if (key >= 0 && key < nkey) {
            atomicAdd(&h[2 * key], 1);
            if (keep[q])
                atomicAdd(&h[2 * key + 1], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nkey; i += 256)
        if (h[i])
            atomicAdd(&cnt[i], h[i]);
}

// exclusive scan over the 256 threads of a block (value per thread); total in *tot
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *tot)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += u;
    }
    if (lane == 63)
        wsum[w] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < w)
            base += wsum[i];
        all += wsum[i];
    }
    __syncthreads();
    *tot = all;
    return base + incl - v;
}

constexpr int SLAB_PER_BLOCK = 1024;      // candidates per block of the scan / pack kernels

__global__ __launch_bounds__(256) void k_slab_scan1(const int32_t *__restrict__ keep, uint32_t n,
                                                    uint32_t *__restrict__ bsum)
{
    const uint32_t q0 = blockIdx.x * SLAB_PER_BLOCK + threadIdx.x * 4;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        v += q0 + i < n && keep[q0 + i] != 0;
    uint32_t tot;
    (void)block_excl_scan(v, &tot);
    if (threadIdx.x == 0)
        bsum[blockIdx.x] = tot;
}

// exclusive scan of the block sums in place (one block; nb <= 256 * 64)
__global__ __launch_bounds__(256) void k_slab_scan2(uint32_t *__restrict__ bsum, uint32_t nb)
{
    __shared__ uint32_t carry;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, &tot);
        const uint32_t c = carry;
        if (i < nb)
            bsum[i] = c + ex;
        __syncthreads();
        if (threadIdx.x == 0)
            carry = c + tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_slab_pack(const sift3d_hip_level *__restrict__ levels,
                                                   const sift3d_hip_cand *__restrict__ cand,
                                                   const int32_t *__restrict__ keep, const float *__restrict__ R,
                                                   uint32_t n, int ngl, const uint32_t *__restrict__ boff,
                                                   float *__restrict__ vals, SlabRec *__restrict__ recs)
{
    const uint32_t q0 = blockIdx.x * SLAB_PER_BLOCK + threadIdx.x * 4;
    uint32_t v = 0;
    bool k[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        k[i] = q0 + i < n && keep[q0 + i] != 0;
        v += k[i];
    }
    uint32_t tot;
    uint32_t pos = boff[blockIdx.x] + block_excl_scan(v, &tot);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t q = q0 + i;
        if (q >= n)
            break;
        const sift3d_hip_cand c = cand[q];
        vals[q] = c.val;
        if (!k[i])
            continue;
        const sift3d_hip_level L = levels[c.tag];
        const uint32_t plane = (uint32_t)L.nx * (uint32_t)L.ny;      // (a level has < 2^32 voxels)
        const uint32_t zq = c.idx / plane, rem = c.idx - zq * plane;
        SlabRec r;
        r.o = c.tag / ngl;
        r.s = c.tag % ngl - 1;
        r.x = (int32_t)(rem % (uint32_t)L.nx);
        r.y = (int32_t)(rem / (uint32_t)L.nx);
        r.z = (int32_t)zq + L.z_off;
#pragma unroll
        for (int j = 0; j < 9; j++)
            r.R[j] = R[9 * (size_t)q + j];
        recs[pos++] = r;
    }
}

// tab: [segk: nseg + 1][segc: nseg + 1][ck: world * (nkey + 1)][cc: world * (nkey + 1)], uint32
__global__ __launch_bounds__(256) void k_slab_build(const char *__restrict__ all, size_t blk_bytes, size_t roff,
                                                    size_t toff, int world, int nkey,
                                                    const uint32_t *__restrict__ tab, uint32_t tot_k,
                                                    uint32_t tot_c, char *__restrict__ out)
{
    const int nseg = nkey * world;
    const uint32_t *segk = tab, *segc = tab + nseg + 1, *ck = segc + nseg + 1, *cc = ck + world * (nkey + 1);
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g == 0) {
        // header: the status words the ranks put behind their blocks (a rank-local failure is everybody's)
        int32_t st = 0;
        for (int r = 0; r < world; r++) {
            const int32_t v = *reinterpret_cast<const int32_t *>(all + (size_t)r * blk_bytes + toff);
            st = v != 0 ? v : st;
        }
        reinterpret_cast<int32_t *>(out)[0] = st;
    }
    if (g >= tot_k)
        return;
    // the segment that holds position g: the last one that starts at or before g (empty ones start there too
    // and end there: the search runs on the segment ENDS)
    auto find = [&](const uint32_t *seg) {
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg[mid + 1] > g)
                hi = mid;
            else
                lo = mid + 1;
        }
        return lo;
    };
    const int sk = find(segk);
    const int kk = sk / world, rk = sk % world;
    const SlabRec *recs = reinterpret_cast<const SlabRec *>(all + (size_t)rk * blk_bytes + roff);
    const SlabRec rec = recs[ck[rk * (nkey + 1) + kk] + (g - segk[sk])];
    float strength = 0.0f;
    if (g < tot_c) {
        const int sc = find(segc);
        const int kc = sc / world, rc = sc % world;
        const float *vals = reinterpret_cast<const float *>(all + (size_t)rc * blk_bytes);
        strength = vals[cc[rc * (nkey + 1) + kc] + (g - segc[sc])];
    }
    SlabOut o;
    o.o = rec.o; o.s = rec.s; o.x = rec.x; o.y = rec.y; o.z = rec.z;
#pragma unroll
    for (int j = 0; j < 9; j++)
        o.R[j] = rec.R[j];
    o.strength = strength;
    o.pad = 0;
    reinterpret_cast<SlabOut *>(out + 64)[g] = o;
}

// descriptor rows from the ranks' gathered blocks into the global keypoint order: row g comes from block
// map[g] >> 24, row map[g] & 0xffffff of that block (768 floats per row, one workgroup per row)
__global__ __launch_bounds__(192) void k_rows_scatter(float *__restrict__ dst, const char *__restrict__ all,
                                                      size_t blk_bytes, const uint32_t *__restrict__ map,
                                                      uint32_t n)
{
    const uint32_t g = blockIdx.x;
    if (g >= n)
        return;
    const uint32_t m = map[g];
    const float4 *src = reinterpret_cast<const float4 *>(all + (size_t)(m >> 24) * blk_bytes) +
                        (size_t)(m & 0xffffffu) * 192;
    reinterpret_cast<float4 *>(dst)[(size_t)g * 192 + threadIdx.x] = src[threadIdx.x];
}

extern "C" {

int sift3d_hip_rows_scatter(float *d_dst, const void *d_all, size_t blk_bytes, const uint32_t *d_map, uint32_t n,
                            void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    hipLaunchKernelGGL(k_rows_scatter, dim3(n), dim3(192), 0, (hipStream_t)stream, d_dst, (const char *)d_all,
                       blk_bytes, d_map, n);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

// per-(octave, level) counts of this rank: d_cnt[2 key] candidates, d_cnt[2 key + 1] kept (nkey <= 256)
int sift3d_hip_slab_count(const sift3d_hip_cand *d_cand, const int32_t *d_keep, uint32_t n, int ngl, int K,
                          int nkey, int32_t *d_cnt, void *stream)
{
    if (nkey < 1 || nkey > 256)
        return SIFT3D_FAILURE;
    HIPCHK(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * 2 * (size_t)nkey, (hipStream_t)stream));
    if (!n)
        return SIFT3D_SUCCESS;
    const uint32_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_slab_count, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, (hipStream_t)stream, d_cand,
                       d_keep, n, ngl, K, nkey, d_cnt);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

size_t sift3d_hip_slab_pack_scratch_bytes(uint32_t n)
{
    return sizeof(uint32_t) * ((size_t)(n + SLAB_PER_BLOCK - 1) / SLAB_PER_BLOCK + 1);
}

// this rank's block of the all-gather: d_vals[q] = |DoG| of candidate q; d_recs = the kept candidates as
// records in global coordinates, in order
int sift3d_hip_slab_pack(const sift3d_hip_level *d_levels, const sift3d_hip_cand *d_cand, const int32_t *d_keep,
                         const float *d_R, uint32_t n, int ngl, float *d_vals, void *d_recs, void *d_scratch,
                         void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    const uint32_t nb = (n + SLAB_PER_BLOCK - 1) / SLAB_PER_BLOCK;
    if (nb > 256 * 64)
        return SIFT3D_FAILURE;
    uint32_t *bsum = (uint32_t *)d_scratch;
    hipLaunchKernelGGL(k_slab_scan1, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_keep, n, bsum);
    hipLaunchKernelGGL(k_slab_scan2, dim3(1), dim3(256), 0, (hipStream_t)stream, bsum, nb);
    hipLaunchKernelGGL(k_slab_pack, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_levels, d_cand, d_keep, d_R, n,
                       ngl, bsum, d_vals, (SlabRec *)d_recs);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

// the global list from the gathered blocks: d_out = 64-byte header (int32 status) + tot_k records of 64 bytes
int sift3d_hip_slab_build(const void *d_all, size_t blk_bytes, size_t roff, size_t toff, int world, int nkey,
                          const uint32_t *d_tab, uint32_t tot_k, uint32_t tot_c, void *d_out, void *stream)
{
    const uint32_t nb = (tot_k + 255) / 256;
    hipLaunchKernelGGL(k_slab_build, dim3(nb ? nb : 1), dim3(256), 0, (hipStream_t)stream, (const char *)d_all,
                       blk_bytes, roff, toff, world, nkey, d_tab, tot_k, tot_c, (char *)d_out);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

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

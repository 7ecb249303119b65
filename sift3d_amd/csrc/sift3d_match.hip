// sift3d_match.hip -- brute-force nearest / second-nearest neighbour of 768-float descriptors.
//
// BASELINE config 5 ("two volumes: detect + describe both, NN match, RANSAC affine").  The
// matcher and the RANSAC fit were REMOVED from the reference fork (CHANGES.md:99-103; upstream
// description README-OLD.md:5), so there is no reference code, no oracle and no fixture for this
// stage: PARITY UNPINNED.  What is built is the textbook form of what upstream describes --
// for every descriptor of set A the nearest and second nearest descriptor of set B under the L2
// distance, accepted by Lowe's ratio test -- validated by recovering a known transform
// (tests/test_register.py).
//
// This is the one dense contraction of the project: |a - b|^2 = |a|^2 + |b|^2 - 2 a.b, an
// (nA x 768) x (768 x nB) matrix product, done on the matrix cores with
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an ordered f32 fma chain, at the
// f32 vector peak rate but with one operand register per lane, leaving the VALU to the top-2
// bookkeeping).  A workgroup owns 128 descriptors of A, walks a run of B in blocks of 128 and
// keeps, per lane and row, the two smallest distances seen; the 128 x 128 distance block is
// never stored.  B is cut into up to 16 runs per row block of A (enough workgroups to fill the
// device several times over); k_nn2_merge combines the per-run top-2.
#include "sift3d_kernels_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MT = 128, NT = 128, KC = 16;      // block of A rows, of B rows, K chunk

// squared norms of the rows
__global__ __launch_bounds__(256) void k_row_norms(const float *__restrict__ a, int n, int dim,
                                                   float *__restrict__ out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n)
        return;
    float s = 0.0f;
    for (int k = lane; k < dim; k += 64) {
        const float v = a[(size_t)row * dim + k];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        s += __shfl_xor(s, o, 64);
    if (lane == 0)
        out[row] = s;
}

struct Top2 {
    float d1, d2;
    int j1;
};

// (distance, index) pairs compare lexicographically: ties go to the smaller index, so the result
// does not depend on the order in which candidates arrive
__device__ __forceinline__ void top2_push(Top2 &t, float d, int j)
{
    const bool first = d < t.d1 || (d == t.d1 && j < t.j1);
    const float nd2 = first ? t.d1 : fminf(t.d2, d);
    t.d1 = first ? d : t.d1;
    t.j1 = first ? j : t.j1;
    t.d2 = nd2;
}

__device__ __forceinline__ void top2_merge(Top2 &t, float d1, float d2, int j1)
{
    top2_push(t, d1, j1);
    t.d2 = fminf(t.d2, d2);
}

// A: nA x dim, B: nB x dim (row-major, dim % KC == 0).  out: for each row of A the index of the
// nearest row of B, the squared distance to it and to the second nearest (+inf if nB < 2).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_nn2(const float *__restrict__ A, int nA, const float *__restrict__ B,
                                             int nB, int dim, const float *__restrict__ normA,
                                             const float *__restrict__ normB, int jb_per_split,
                                             int *__restrict__ out_j, float *__restrict__ out_d1,
                                             float *__restrict__ out_d2)
{
    // blockIdx.y: which run of jb_per_split 128-row blocks of B this workgroup scans; its top-2 go to
    // slice blockIdx.y of the outputs (merged by k_nn2_merge when there is more than one)
    const int jlo = blockIdx.y * jb_per_split * NT, jhi = min(nB, jlo + jb_per_split * NT);
    out_j += (size_t)blockIdx.y * nA;
    out_d1 += (size_t)blockIdx.y * nA;
    out_d2 += (size_t)blockIdx.y * nA;
    // k-major tiles: As[k][row], so that the lanes of an MFMA operand (row = lane & 31,
    // k = lane >> 5) read 32 consecutive floats.  Double-buffered: the global loads of chunk
    // c + 1 are in flight during the MFMAs of chunk c and go to the other buffer afterwards --
    // one barrier per chunk.
    // row stride 130: the transposing stores of a wave (lanes 2 r and 2 r + 1 write row r of k rows
    // sk = 0 and 8) land 8 * 130 = 1040 floats apart = 16 banks apart -- with stride 132 the two lanes of
    // a pair shared a bank (2-way conflict on every staging store); the operand reads stay 32
    // consecutive floats
    __shared__ float As[2][KC][MT + 2];
    __shared__ float Bs[2][KC][NT + 2];
    __shared__ float red_d1[MT][2], red_d2[MT][2];
    __shared__ int red_j[MT][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;            // 2 x 2 waves, 64 x 64 outputs each
    const int i0 = blockIdx.x * MT;
    const int lr = lane & 31, lh = lane >> 5;
    // staging role: thread t copies 8 consecutive k of one row
    const int srow = tid >> 1, sk = (tid & 1) * 8;
    const int nk = dim / KC;

    Top2 best[2][16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            best[t][r].d1 = best[t][r].d2 = __builtin_inff();
            best[t][r].j1 = 0x7fffffff;
        }
    const int ra = i0 + srow;
    const float *pa = A + (size_t)min(ra, nA - 1) * dim + sk;
    const bool a_ok = ra < nA;

    for (int j0 = jlo; j0 < jhi; j0 += NT) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[a][b][r] = 0.0f;
        const int rb = j0 + srow;
        const float *pb = B + (size_t)min(rb, nB - 1) * dim + sk;
        const bool b_ok = rb < nB;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 va0, va1, vb0, vb1;
        auto fetch = [&](int c) {
            va0 = a_ok ? ld4(pa + c * KC) : zero4;
            va1 = a_ok ? ld4(pa + c * KC + 4) : zero4;
            vb0 = b_ok ? ld4(pb + c * KC) : zero4;
            vb1 = b_ok ? ld4(pb + c * KC + 4) : zero4;
        };
        auto commit = [&](int buf) {
            As[buf][sk + 0][srow] = va0.x; As[buf][sk + 1][srow] = va0.y; As[buf][sk + 2][srow] = va0.z; As[buf][sk + 3][srow] = va0.w;
            As[buf][sk + 4][srow] = va1.x; As[buf][sk + 5][srow] = va1.y; As[buf][sk + 6][srow] = va1.z; As[buf][sk + 7][srow] = va1.w;
            Bs[buf][sk + 0][srow] = vb0.x; Bs[buf][sk + 1][srow] = vb0.y; Bs[buf][sk + 2][srow] = vb0.z; Bs[buf][sk + 3][srow] = vb0.w;
            Bs[buf][sk + 4][srow] = vb1.x; Bs[buf][sk + 5][srow] = vb1.y; Bs[buf][sk + 6][srow] = vb1.z; Bs[buf][sk + 7][srow] = vb1.w;
        };
        __syncthreads();                    // the previous block's last MFMAs have read buffer (nk - 1) & 1
        fetch(0);
        commit(0);
        __syncthreads();
        for (int c = 0; c < nk; c++) {
            const int buf = c & 1;
            if (c + 1 < nk)
                fetch(c + 1);               // in flight during the MFMAs below
#pragma unroll
            for (int kk = 0; kk < KC; kk += 2) {
                float fa[2], fb[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    fa[t] = As[buf][kk + lh][wr * 64 + t * 32 + lr];
                    fb[t] = Bs[buf][kk + lh][wc * 64 + t * 32 + lr];
                }
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
            if (c + 1 < nk)
                commit(buf ^ 1);            // (its last readers passed the barrier of chunk c - 1)
            __syncthreads();
        }
        // distances of this block: element r of lane l of tile (a, b) is
        // (row = 32 a + (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = 32 b + (l & 31))
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int col = j0 + wc * 64 + b * 32 + lr;
            const float nb = col < nB ? normB[col] : 0.0f;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    // |a - b|^2, clamped at 0 (cancellation for near-identical descriptors)
                    // (|a|^2 is re-read per block instead of living in 32 registers: the kernel
                    // then fits two waves per SIMD)
                    const int row = i0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float d = fmaxf(normA[min(row, nA - 1)] + nb - 2.0f * acc[a][b][r], 0.0f);
                    if (col < nB)
                        top2_push(best[a][r], d, col);
                }
        }
    }
    // merge over the 32 lanes that hold the same row (same lane >> 5), then over the two waves
    // that cover the two column halves
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            Top2 t = best[a][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const float d1 = __shfl_xor(t.d1, o, 64), d2 = __shfl_xor(t.d2, o, 64);
                const int j1 = __shfl_xor(t.j1, o, 64);
                top2_merge(t, d1, d2, j1);
            }
            if (lr == 0) {
                const int row = wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                red_d1[row][wc] = t.d1;
                red_d2[row][wc] = t.d2;
                red_j[row][wc] = t.j1;
            }
        }
    __syncthreads();
    if (tid < MT && i0 + tid < nA) {
        Top2 t;
        t.d1 = red_d1[tid][0]; t.d2 = red_d2[tid][0]; t.j1 = red_j[tid][0];
        top2_merge(t, red_d1[tid][1], red_d2[tid][1], red_j[tid][1]);
        out_j[i0 + tid] = t.j1;             // (0x7fffffff: none; k_nn2_merge / the launcher's final pass turn it into -1)
        out_d1[i0 + tid] = t.d1;
        out_d2[i0 + tid] = t.d2;
    }
}

// Top-2 of a row = merge of its top-2 over the runs of B (lexicographic ties: the result does not
// depend on how B was cut).
__global__ __launch_bounds__(256) void k_nn2_merge(const int *__restrict__ pj, const float *__restrict__ pd1,
                                                   const float *__restrict__ pd2, int nsplit, int nA,
                                                   int *__restrict__ out_j, float *__restrict__ out_d1,
                                                   float *__restrict__ out_d2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nA)
        return;
    Top2 t;
    t.d1 = pd1[i]; t.d2 = pd2[i]; t.j1 = pj[i];
    for (int s = 1; s < nsplit; s++)
        top2_merge(t, pd1[(size_t)s * nA + i], pd2[(size_t)s * nA + i], pj[(size_t)s * nA + i]);
    out_j[i] = t.j1 == 0x7fffffff ? -1 : t.j1;
    out_d1[i] = t.d1;
    out_d2[i] = t.d2;
}

// runs of B per row block of A: enough workgroups to fill the device several times over (a 128-row
// block of A against ALL of B is one long workgroup, and 40 000 descriptors are only 317 of them)
static int nn2_splits(int nA, int nB)
{
    const int nblkA = (nA + MT - 1) / MT, nblkB = (nB + NT - 1) / NT;
    int s = nblkA > 0 ? (2048 + nblkA - 1) / nblkA : 1;
    if (s > 16) s = 16;
    if (s > nblkB) s = nblkB;
    return s < 1 ? 1 : s;
}

extern "C" {

size_t sift3d_hip_nn2_work_floats(int nA, int nB)
{
    const size_t a = (size_t)(nA > 0 ? nA : 0), b = (size_t)(nB > 0 ? nB : 0);
    return a + b + 8 + 3 * a * (size_t)nn2_splits(nA, nB);
}

int sift3d_hip_nn2(const float *d_A, int nA, const float *d_B, int nB, int dim, int *d_j1, float *d_d1,
                   float *d_d2, float *d_work, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!d_A || !d_B || !d_j1 || !d_d1 || !d_d2 || !d_work || nA < 0 || nB < 0 || dim < KC || (dim % KC) ||
        (((uintptr_t)d_A | (uintptr_t)d_B) & 15)) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_nn2: invalid arguments");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    if (!nA)
        return SIFT3D_SUCCESS;
    float *nrmA = d_work, *nrmB = d_work + nA;
    hipLaunchKernelGGL(k_row_norms, dim3((nA + 3) / 4), dim3(256), 0, st, d_A, nA, dim, nrmA);
    if (nB)
        hipLaunchKernelGGL(k_row_norms, dim3((nB + 3) / 4), dim3(256), 0, st, d_B, nB, dim, nrmB);
    {
        const int ns = nn2_splits(nA, nB), nblkB = (nB + NT - 1) / NT;
        const int per = ns > 0 ? (nblkB + ns - 1) / ns : 0;
        // partial results: three slices of ns * nA after the norms (the index slice holds ints)
        float *part = d_work + nA + nB + 8;
        int *pj = reinterpret_cast<int *>(part);
        float *pd1 = part + (size_t)ns * nA, *pd2 = part + 2 * (size_t)ns * nA;
        hipLaunchKernelGGL(k_nn2, dim3((nA + MT - 1) / MT, ns), dim3(256), 0, st, d_A, nA, d_B, nB, dim, nrmA,
                           nrmB, per > 0 ? per : 1, pj, pd1, pd2);
        hipLaunchKernelGGL(k_nn2_merge, dim3((nA + 255) / 256), dim3(256), 0, st, pj, pd1, pd2, ns, nA, d_j1,
                           d_d1, d_d2);
    }
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

} // extern "C"

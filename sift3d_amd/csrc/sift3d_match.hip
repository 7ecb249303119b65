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

constexpr int MT = 128, NT = 128, KC = 32;      // block of A rows, of B rows, K chunk
constexpr int LDK = KC + 4;                      // LDS row stride in floats (see k_nn2)

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
    // Row-major tiles As[row][k] (as the rows lie in memory: the staging stores are 16-byte stores of what
    // the 16-byte global loads delivered, no transposition), rows LDK = KC + 4 floats apart.  A lane of an
    // MFMA operand (row = lane & 31, half h = lane >> 5) takes FOUR k at a time with one 16-byte read,
    // k = 8 g + 4 h + {0, 1, 2, 3}: the j-th of the four v_mfma_f32_32x32x2_f32 of group g then multiplies the
    // k pair (8 g + j, 8 g + 4 + j) -- any pairing will do, a dot product does not care about the order
    // of its terms, as long as A and B use the same one.  Eight consecutive lanes read rows 36 floats
    // apart: 4-float slots at 0, 4, ..., 28 modulo the 32 banks -- conflict-free; a chunk needs 16
    // 16-byte reads and 8 16-byte stores per lane where the k-major layout needed 64 + 32 4-byte ones.
    // Double-buffered: the global loads of chunk c + 1 are in flight during the 64 MFMAs of chunk c and go
    // to the other buffer afterwards -- one barrier per chunk of KC = 32 (two tiles of 2 x 18 KB per
    // workgroup, two workgroups per CU).
    __shared__ __attribute__((aligned(16))) float As[2][MT][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][NT][LDK];
    __shared__ float red_d1[MT][2], red_d2[MT][2];
    __shared__ int red_j[MT][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;            // 2 x 2 waves, 64 x 64 outputs each
    const int i0 = blockIdx.x * MT;
    const int lr = lane & 31, lh = lane >> 5;
    // staging role: thread t copies the 16-byte pieces (row (t >> 3) + 32 i, columns 4 (t & 7) ..) of both
    // tiles, i = 0..3: eight consecutive lanes cover the 128 contiguous bytes of a row's chunk
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    const int nk = dim / KC;

    Top2 best[2][16];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            best[t][r].d1 = best[t][r].d2 = __builtin_inff();
            best[t][r].j1 = 0x7fffffff;
        }
    // (row addresses are recomputed per load rather than kept in 16 registers: the kernel lives at the edge
    // of its 256)
    const uint32_t udim = (uint32_t)dim;
    for (int j0 = jlo; j0 < jhi; j0 += NT) {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[a][b][r] = 0.0f;
        float4 vs[4];                       // staging registers: the A pieces of a chunk, then its B pieces
        auto fetch = [&](const float *__restrict__ M, int r0, int nrows, int c) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int rr = r0 + srow + 32 * i;
                // rows beyond the end repeat the last row: their distances are never used (the epilogue skips
                // columns >= nB, the final store rows >= nA) -- and nothing may depend on the loaded value
                // here, or the load could not stay in flight during the MFMAs
                vs[i] = ld4(M + ((size_t)((uint32_t)min(rr, nrows - 1)) * udim + (uint32_t)(sc4 + c * KC)));
            }
        };
        auto commit = [&](float (*T)[LDK]) {
#pragma unroll
            for (int i = 0; i < 4; i++)
                *reinterpret_cast<float4 *>(&T[srow + 32 * i][sc4]) = vs[i];
        };
        // eight k of the chunk: two halves of four, each with 8-byte operand reads (16-byte reads would
        // hold 16 operand registers where 8 do, and the kernel has none to spare)
        auto group = [&](int buf, int g) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                float2 fa[2], fb[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    fa[t] = *reinterpret_cast<const float2 *>(&As[buf][wr * 64 + t * 32 + lr][8 * g + 4 * lh + 2 * hf]);
                    fb[t] = *reinterpret_cast<const float2 *>(&Bs[buf][wc * 64 + t * 32 + lr][8 * g + 4 * lh + 2 * hf]);
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float xa[2] = { j == 0 ? fa[0].x : fa[0].y, j == 0 ? fa[1].x : fa[1].y };
                    const float xb[2] = { j == 0 ? fb[0].x : fb[0].y, j == 0 ? fb[1].x : fb[1].y };
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int b = 0; b < 2; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[a], xb[b], acc[a][b], 0, 0, 0);
                }
            }
        };
        __syncthreads();                    // the previous block's last MFMAs have read buffer (nk - 1) & 1
        fetch(A, i0, nA, 0);
        commit(As[0]);
        fetch(B, j0, nB, 0);
        commit(Bs[0]);
        __syncthreads();
        static_assert(KC == 32, "four groups of eight k per chunk");
        for (int c = 0; c < nk; c++) {
            const int buf = c & 1;
            const bool more = c + 1 < nk;
            // the next chunk's pieces travel through ONE set of staging registers: A's are requested before
            // the first group of MFMAs and stored after the second, B's requested then and stored after the
            // fourth (the other buffer's last readers passed the barrier of chunk c - 1)
            if (more)
                fetch(A, i0, nA, c + 1);
            group(buf, 0);
            group(buf, 1);
            if (more) {
                commit(As[buf ^ 1]);
                fetch(B, j0, nB, c + 1);
            }
            group(buf, 2);
            group(buf, 3);
            if (more)
                commit(Bs[buf ^ 1]);
            __syncthreads();
        }
        // distances of this block: element r of lane l of tile (a, b) is
        // (row = 32 a + (r & 3) + 8 (r >> 2) + 4 (l >> 5), col = 32 b + (l & 31)).
        // |a - b|^2 = |a|^2 + (|b|^2 - 2 a.b): for a fixed row the first term is a constant, so the top-2
        // bookkeeping runs on e = |b|^2 - 2 a.b and k_nn2_merge adds |a|^2 (and clamps at 0) at the very end --
        // the row norms stay out of this loop (read here, 32 per lane and block, each load was waited for
        // before the next could be issued: as long as the block's MFMAs).
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int col = j0 + wc * 64 + b * 32 + lr;
            const float nb = col < nB ? normB[col] : 0.0f;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float e = nb - 2.0f * acc[a][b][r];
                    if (col < nB)
                        top2_push(best[a][r], e, col);
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
// depend on how B was cut), then e -> |a - b|^2 = max(|a|^2 + e, 0) (clamped: cancellation for near-identical
// descriptors).
__global__ __launch_bounds__(256) void k_nn2_merge(const int *__restrict__ pj, const float *__restrict__ pd1,
                                                   const float *__restrict__ pd2, int nsplit, int nA,
                                                   const float *__restrict__ normA, int *__restrict__ out_j,
                                                   float *__restrict__ out_d1, float *__restrict__ out_d2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nA)
        return;
    Top2 t;
    t.d1 = pd1[i]; t.d2 = pd2[i]; t.j1 = pj[i];
    for (int s = 1; s < nsplit; s++)
        top2_merge(t, pd1[(size_t)s * nA + i], pd2[(size_t)s * nA + i], pj[(size_t)s * nA + i]);
    const float na = normA[i];
    out_j[i] = t.j1 == 0x7fffffff ? -1 : t.j1;
    out_d1[i] = fmaxf(na + t.d1, 0.0f);
    out_d2[i] = fmaxf(na + t.d2, 0.0f);
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
        hipLaunchKernelGGL(k_nn2_merge, dim3((nA + 255) / 256), dim3(256), 0, st, pj, pd1, pd2, ns, nA, nrmA, d_j1,
                           d_d1, d_d2);
    }
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

} // extern "C"

// sift3d_fir_yz.hip -- the fused y + z pass of the octave-0 blurs (k_fir_yz_u1) and its C entry.
//
// A translation unit of its own because it is compiled with -fno-slp-vectorize: the SLP
// vectoriser packs the float4 arithmetic into v_pk_* instructions, which issue no faster on
// gfx950 but cost this kernel ~40 more VGPRs (a workgroup of 512 threads then no longer fits
// twice on a CU); the other FIR kernels keep the default (their register-ring sweeps spill
// without it).  Numerical contract and citations as in sift3d_kernels.hip.
#include "sift3d_kernels_common.h"

// ---- fused y + z passes, unit factor 1 -------------------------------------------------------
// dst = FIR_z(FIR_y(src)) without the y-pass result ever reaching HBM.  A workgroup owns a
// 4*TXQ(x) x TY(y) column of the volume (64 x 32 or 128 x 32) and sweeps a segment along z.  For every plane it stages
// the TY + 2*HW rows of the EXTENDED y line in LDS (coalesced 16-byte loads, edge rows built
// while staging), each thread (x-quad, y) takes the 2*HW+1 taps of its column from LDS
// (conflict-free ds_read_b128), and pushes the y-filtered value into its register ring along
// z, exactly as k_fir_sweep_u1 does with loaded rows.  Extended planes (reflected / virtual,
// wave- and block-uniform) are formed from y-filtered planes, i.e. the z edge rules act on the
// y-pass OUTPUT as in the reference (apply_Sep_FIR_filter runs the passes one after the other,
// imutil.c:1165-1188).  Per-voxel arithmetic and tap order are those of the separate passes,
// so results are bit-identical; HBM traffic drops from 16 to ~9-11 B/voxel for the pair.
template <int HW, int TY, int TXQ>
__global__ __launch_bounds__(TXQ * TY) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fir_yz_u1(FirParams P, FirTaps T, EdgeTab Ey, EdgeTab Ez)
{
    constexpr int W = 2 * HW + 1, ROWS = TY + 2 * HW;
    __shared__ float4 tile[2][ROWS][TXQ];
    const int tid = threadIdx.x;
    const int qx = tid % TXQ, ty = tid / TXQ;
    const int x = (blockIdx.x * TXQ + qx) * 4;
    const int y0 = blockIdx.y * TY;
    const int y = y0 + ty;
    const int nx = P.nx, ny = P.ny;
    const size_t plane = (size_t)nx * ny;
    const int xc = min(x, nx - 4);                    // clamped column for the loads
    const bool writer = x < nx && y < ny;
    const int nl1 = P.nz - 1;
    const int off = P.off, endz = P.n_glob - 1, endy = ny - 1;
    const int p0 = P.z_lo + blockIdx.z * P.ts;
    const int p1 = min(p0 + P.ts, P.z_hi);
    int buf = 0;

    // extended-y row i (global y index, may be outside [0, ny)) of local plane pl
    auto ext_y = [&](int pl, int i) -> float4 {
        const float *__restrict__ s = P.src + (size_t)pl * plane + xc;
        if (i < 0) {
            return ld4(s + (size_t)min(-i, endy) * nx);
        } else if (i >= endy) {
            const int m = i - endy;
            if (m > HW)
                return make_float4(0.f, 0.f, 0.f, 0.f);
            const int lo = Ey.lo[m];
            return Vec<4>::lerp(Ey.w0[m], ld4(s + (size_t)clampi(lo, 0, endy) * nx), Ey.w1[m],
                                ld4(s + (size_t)clampi(lo + 1, 0, endy) * nx));
        }
        return ld4(s + (size_t)i * nx);
    };
    // Tile rows of one plane held in registers: every thread stages row ty and, for the first
    // 2*HW rows of threads, row ty + TY.  They are fetched PD PLANES AHEAD of their use: the rows
    // of a plane are only ~1.5 16-byte loads per thread, and with one plane in flight a CU has
    // ~24 KB outstanding -- a third of what the HBM latency needs at full rate.  Slot 0 of the
    // queue is the plane the next call will ask for (exactly: `hint` follows the mirror /
    // virtual planes at the global faces); the slots behind it are the following planes, which
    // is what the sweep asks for everywhere but at those faces.  A wrong guess only costs a
    // synchronous fetch (block-uniform branch).
    static_assert(2 * HW <= TY, "two tile rows per thread");
    // (depth: deeper queues measured no faster for HW <= 5; the 15- and 17-tap instances stage
    // without prefetch -- their W-times unrolled sweep must stay inside the instruction cache)
    constexpr int PD = HW <= 5 ? 1 : (HW == 6 ? 2 : 0);
    constexpr int PQ = PD > 0 ? PD : 1;
    constexpr int NONE = -(1 << 30);
    float4 q0[PQ], q1[PQ];
    int qpl[PQ];
#pragma unroll
    for (int i = 0; i < PQ; i++) {
        q0[i] = q1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        qpl[i] = NONE;
    }
    const bool second = ty + TY < ROWS;
    // y-filtered value of this thread's column in local plane pl (block-wide call); `hint` is
    // the plane the next call will ask for (or < 0)
    auto yfilt = [&](int pl, int hint) -> float4 {
        pl = clampi(pl, 0, nl1);
        if (PD == 0) {
            // r -> (row, quad) with quad == qx because the block size is a multiple of TXQ
            for (int r = tid; r < ROWS * TXQ; r += TXQ * TY)
                tile[buf][r / TXQ][qx] = ext_y(pl, y0 - HW + r / TXQ);
            __syncthreads();
            float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int dd = -HW; dd <= HW; dd++)
                Vec<4>::mac(acc0, T.k[dd + HW], tile[buf][ty + HW - dd][qx]);
            buf ^= 1;
            return acc0;
        }
        if (qpl[0] != pl) {                     // block-uniform: not prefetched
            q0[0] = ext_y(pl, y0 - HW + ty);
            if (second)
                q1[0] = ext_y(pl, y0 - HW + ty + TY);
            qpl[0] = pl;
        }
        tile[buf][ty][qx] = q0[0];
        if (second)
            tile[buf][ty + TY][qx] = q1[0];
        __syncthreads();
        // advance the queue and top it up
        const int nxt = hint >= 0 ? clampi(hint, 0, nl1) : NONE;
#pragma unroll
        for (int i = 0; i + 1 < PQ; i++) {
            q0[i] = q0[i + 1];
            q1[i] = q1[i + 1];
            qpl[i] = qpl[i + 1];
        }
        qpl[PQ - 1] = NONE;
        if (nxt != NONE && qpl[0] != nxt) {     // the sequence jumps (faces) or starts
            q0[0] = ext_y(nxt, y0 - HW + ty);
            if (second)
                q1[0] = ext_y(nxt, y0 - HW + ty + TY);
            qpl[0] = nxt;
#pragma unroll
            for (int i = 1; i < PQ; i++)
                qpl[i] = NONE;
        }
#pragma unroll
        for (int i = 1; i < PQ; i++) {
            const int want = qpl[i - 1] == NONE || qpl[i - 1] >= nl1 ? NONE : qpl[i - 1] + 1;
            if (qpl[i] != want) {
                if (want != NONE) {
                    q0[i] = ext_y(want, y0 - HW + ty);
                    if (second)
                        q1[i] = ext_y(want, y0 - HW + ty + TY);
                }
                qpl[i] = want;
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dd = -HW; dd <= HW; dd++)
            Vec<4>::mac(acc, T.k[dd + HW], tile[buf][ty + HW - dd][qx]);
        buf ^= 1;   // the next plane is staged in the other buffer: one barrier per plane
        return acc;
    };
    // first plane that ext_z(r) will request (-1: none)
    auto first_plane = [&](int r) -> int {
        const int i = r + off;
        if (i < 0)
            return -i - off;
        if (i >= endz)
            return i - endz > HW ? -1 : Ez.lo[i - endz] - off;
        return r;
    };
    // extended-z plane i (LOCAL index r = i - off may be outside the slab at global faces):
    // one or two y-filtered planes, selected with block-uniform scalars so that yfilt has a
    // single inlined call site
    auto ext_z = [&](int r) -> float4 {
        const int i = r + off;
        int pa = r, np = 1;
        float w0 = 1.0f, w1 = 0.0f;
        if (i < 0) {
            pa = -i - off;
        } else if (i >= endz) {
            const int m = i - endz;
            if (m > HW) {
                np = 0;
            } else {
                pa = Ez.lo[m] - off;
                w0 = Ez.w0[m];
                w1 = Ez.w1[m];
                np = 2;
            }
        }
        const int nxt = first_plane(r + 1);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll 1
        for (int k = 0; k < np; k++) {
            const float4 yv = yfilt(pa + k, k + 1 < np ? pa + k + 1 : nxt);
            if (k == 0)
                a = yv;
            else
                b = yv;
        }
        return np == 2 ? Vec<4>::lerp(w0, a, w1, b) : a;
    };

    // Register window along z: the 2*HW+1 most recent extended planes.  The loop is unrolled W
    // times so that every ring position is a compile-time register (as in k_fir_sweep_u1): at
    // step j the window of output q holds plane q - HW + i in ring[(j + i) % W] -- nothing is
    // ever shifted (a shifting window cost 4*2*HW register moves per plane, a third of the VALU
    // instructions of this VALU-bound kernel).
    float4 ring[W];
#pragma unroll
    for (int i = 0; i < W; i++)
        ring[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // warm-up: planes p0 - HW .. p0 + HW - 1 into ring[0 .. 2*HW - 1] (static positions)
#pragma unroll
    for (int i = 0; i < 2 * HW; i++)
        ring[i] = ext_z(p0 - HW + i);
    float *__restrict__ d = P.dst + (size_t)y * nx + x;
#pragma unroll 1
    for (int q0 = p0; q0 < p1; q0 += W) {
#pragma unroll
        for (int j = 0; j < W; j++) {
            const int q = q0 + j;
            if (q < p1) {                              // block-uniform
                ring[(j + 2 * HW) % W] = ext_z(q + HW);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int dd = -HW; dd <= HW; dd++)
                    Vec<4>::mac(acc, T.k[dd + HW], ring[(j + HW - dd) % W]);   // E[q - d], d ascending
                if (writer)
                    st4(d + (size_t)q * plane, acc);
            }
        }
    }
}

template <int HW>
static void launch_fir_yz(const FirParams &P, const FirTaps &T, const EdgeTab &Ey, const EdgeTab &Ez,
                          int ty, hipStream_t st)
{
    const int nseg = (P.z_hi - P.z_lo + P.ts - 1) / P.ts;
    // 64(x) x 64(y) tiles on tall volumes: half the halo rows of a 32-row tile per output row (measured
    // at 512^3: 1-2 % faster at 5-7 taps, 3-4.5 % at 9-17 taps than 128 x 32); else 128(x) x 32(y) where
    // the rows fill them (512-byte row segments; 2 % faster over the octave-0 pyramid than 64 x 32),
    // 64 x 32 otherwise
    if ((P.nx & 63) == 0 && P.ny >= 128) {
        dim3 grid((P.nx / 4 + 15) / 16, (P.ny + 63) / 64, nseg);
        hipLaunchKernelGGL((k_fir_yz_u1<HW, 64, 16>), grid, dim3(1024), 0, st, P, T, Ey, Ez);
    } else if ((P.nx & 127) == 0) {
        dim3 grid((P.nx / 4 + 31) / 32, (P.ny + 31) / 32, nseg);
        hipLaunchKernelGGL((k_fir_yz_u1<HW, 32, 32>), grid, dim3(1024), 0, st, P, T, Ey, Ez);
    } else {
        dim3 grid((P.nx / 4 + 15) / 16, (P.ny + ty - 1) / ty, nseg);
        hipLaunchKernelGGL((k_fir_yz_u1<HW, 32, 16>), grid, dim3(16 * 32), 0, st, P, T, Ey, Ez);
    }
}

extern "C" {

// does sift3d_hip_fir_yz_u1 cover this configuration?  (one predicate for the kernel's own check and
// for callers that have to know BEFORE they launch -- the slab driver enqueues the halo exchange of
// the z pass's input, which is another buffer when the passes run separately)
int sift3d_hip_fir_yz_u1_covers(const float *d_src, const float *d_dst, int nx, int ny, int width, int n_glob)
{
    const int hw = width / 2;
    return !(hw < 1 || hw > 8 || (nx & 3) || ((((uintptr_t)d_src | (uintptr_t)d_dst) & 15) != 0) ||
             ny < 2 * hw + 2 || n_glob < 2 * hw + 2 || n_glob >= (1 << 22) || ny >= (1 << 22));
}

int sift3d_hip_fir_yz_u1(const float *d_src, float *d_dst, int nx, int ny, int nz, const float *taps,
                         int width, int n_glob, int off, int z_lo, int z_hi, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int hw = width / 2;
    if (!d_src || !d_dst || d_src == d_dst || nx < 4 || ny < 1 || nz < 1 || !(width & 1) || z_lo < 0 ||
        z_hi > nz || off < 0 || off + nz > n_glob) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_fir_yz_u1: invalid arguments");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    // not covered -> the caller runs the y and z passes separately
    if (!sift3d_hip_fir_yz_u1_covers(d_src, d_dst, nx, ny, width, n_glob))
        return 1;
    if (z_hi <= z_lo)
        return SIFT3D_SUCCESS;
    FirParams P;
    FirTaps T;
    memset(&T, 0, sizeof(T));
    memcpy(T.k, taps, sizeof(float) * width);
    memset(&P, 0, sizeof(P));
    P.src = d_src; P.dst = d_dst;
    P.nx = nx; P.ny = ny; P.nz = nz;
    P.axis = 2; P.hw = hw; P.uf = 1.0f; P.uhw = hw;
    P.n_glob = n_glob; P.off = off; P.z_lo = z_lo; P.z_hi = z_hi;
    // tile height: 32 rows measured best for every width
    const int ty = 32;
    {
        // z segmentation: >= 4096 waves in flight, segments of at least 32 planes
        long blocks_xy = (long)((nx / 4 + 15) / 16) * ((ny + ty - 1) / ty);
        if (blocks_xy < 1)
            blocks_xy = 1;
        const int n_out = z_hi - z_lo;
        long want = (512 + blocks_xy - 1) / blocks_xy;
        long cap = n_out / 32 > 1 ? n_out / 32 : 1;
        long nseg = want < cap ? want : cap;
        if (nseg < 1)
            nseg = 1;
        P.ts = (int)((n_out + nseg - 1) / nseg);
    }
    const EdgeTab Ey = edge_table(ny, hw), Ez = edge_table(n_glob, hw);
    switch (hw) {
    case 1: launch_fir_yz<1>(P, T, Ey, Ez, ty, st); break;
    case 2: launch_fir_yz<2>(P, T, Ey, Ez, ty, st); break;
    case 3: launch_fir_yz<3>(P, T, Ey, Ez, ty, st); break;
    case 4: launch_fir_yz<4>(P, T, Ey, Ez, ty, st); break;
    case 5: launch_fir_yz<5>(P, T, Ey, Ez, ty, st); break;
    case 6: launch_fir_yz<6>(P, T, Ey, Ez, ty, st); break;
    case 7: launch_fir_yz<7>(P, T, Ey, Ez, ty, st); break;
    default: launch_fir_yz<8>(P, T, Ey, Ez, ty, st); break;
    }
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

} // extern "C"

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
#ifndef YZ_PD78
#define YZ_PD78 0
#endif
#ifndef YZ_PD6
#define YZ_PD6 2
#endif
    constexpr int PD = HW <= 5 ? 1 : (HW == 6 ? YZ_PD6 : YZ_PD78);
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

// ---- the same pass with the tile rows staged by LDS-DMA, three planes ahead ---------------------
// k_fir_yz_u1 above is bound by latency, not by bytes or arithmetic: one workgroup per CU, one barrier
// per plane, and the rows of plane p + 1 are requested only while plane p is filtered (the 17-tap
// instance, which has no registers left for that, requests them when it needs them): 17-20 KB in flight
// per CU against the ~50 KB that 6 TB/s need at ~2 us of loaded latency; 2.0 (5 taps) to 2.8 us (17 taps)
// per plane where the arithmetic of a plane takes 0.3-1.1.  Here a plane's TY + 2 HW rows go from HBM
// straight into one of FOUR LDS tiles (global_load_lds_dwordx4: no staging registers, no ds_write), three
// requests ahead of the one being filtered: request t + 3 is issued right after the barrier that opens
// request t -- the buffer it overwrites was last read before that barrier --, each wave waits for its own
// pieces of request t with a COUNTED s_waitcnt vmcnt(N) before the barrier (N = the younger DMA pieces
// and stores of the wave, tracked per iteration), never vmcnt(0).  The DMA is inline assembly, so the
// compiler neither counts it nor drains it (cdna_hip_programming.md, 5.7); the output stores are the
// only vector-memory operations it sees.  Edges as in k_fir_yz_u1: mirrored rows and planes are source
// ADDRESSES; the virtual rows of the high y face (E[ny - 1 + m], imutil.c:846-848) are formed in LDS from
// the two staged rows they interpolate, by the workgroups of the last tile row only (one more barrier
// there); virtual planes of the high z face from two y-filtered planes, as before.  Arithmetic and tap
// order are those of the separate passes: bit-identical results.
template <int HW, int TY>
__global__ __launch_bounds__(16 * TY) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_fir_yz_dma(FirParams P, FirTaps T, EdgeTab Ey, EdgeTab Ez)
{
    constexpr int TXQ = 16, W = 2 * HW + 1, ROWS = TY + 2 * HW, ROWS4 = (ROWS + 3) & ~3, NB = 4;
    static_assert(2 * HW <= TY && (TY & 3) == 0, "at most two pieces per wave and request");
    constexpr int ND2 = (ROWS4 - TY) / 4;     // waves that stage a second piece (rows TY ..) per request
    constexpr int SEQ = 320;                  // capacity of the request list (launcher: ts <= 256)
    __shared__ float4 tile[NB][ROWS4][TXQ];
    __shared__ int seq[SEQ + 1];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int qx = tid % TXQ, ty = tid / TXQ;
    const int x = (blockIdx.x * TXQ + qx) * 4;
    const int y0 = blockIdx.y * TY;
    const int y = y0 + ty;
    const int nx = P.nx, ny = P.ny;
    const size_t plane = (size_t)nx * ny;
    const bool writer = x < nx && y < ny;
    const int nl1 = P.nz - 1;
    const int off = P.off, endz = P.n_glob - 1, endy = ny - 1;
    const int p0 = P.z_lo + blockIdx.z * P.ts;
    const int p1 = min(p0 + P.ts, P.z_hi);

    // the y-filter requests of this workgroup, in the order the sweep consumes them: local plane indices
    // (extended plane r = p0 - HW .. p1 - 1 + HW: its mirror image at the low face, the two planes a
    // virtual plane interpolates at the high face, none beyond the taps' reach)
    if (tid == 0) {
        int n = 0;
        for (int r = p0 - HW; r < p1 + HW && n + 2 <= SEQ; r++) {
            const int i = r + off;
            if (i < 0) {
                seq[n++] = clampi(-i - off, 0, nl1);
            } else if (i >= endz) {
                const int m = i - endz;
                if (m <= HW) {
                    int lo = 0;
                    for (int mm = 0; mm <= HW; mm++)
                        lo = mm == m ? Ez.lo[mm] : lo;
                    seq[n++] = clampi(lo - off, 0, nl1);
                    seq[n++] = clampi(lo + 1 - off, 0, nl1);
                }
            } else {
                seq[n++] = clampi(r, 0, nl1);
            }
        }
        seq[SEQ] = n;
    }
    __syncthreads();
    const int nreq = __builtin_amdgcn_readfirstlane(seq[SEQ]);

    // DMA pieces of this wave: piece k covers tile rows 64 k + 4 wave .. + 3 (1 KB = 4 rows of 16 quads);
    // lane -> (row, quad); the source row of tile row j is extended-y index i = y0 - HW + j
    size_t srcoff[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int j = TY * k + 4 * wave + (lane >> 4);
        const int i = y0 - HW + j;
        // (virtual and unused rows: any valid row; the real row ny - 1 where a virtual one will be formed)
        const int sr = i < 0 ? min(-i, endy) : min(i, endy);
        const int xq = min((int)(blockIdx.x * TXQ + (lane & 15)) * 4, nx - 4);
        srcoff[k] = (size_t)sr * nx + xq;
    }
    const bool two = wave < ND2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)&tile[0][0][0];
    auto stage = [&](int t) {
        // request t (clamped: beyond the list a harmless re-request keeps the count of pieces in flight)
        const int pl = __builtin_amdgcn_readfirstlane(seq[min(t, nreq - 1)]);
        const float *src = P.src + (size_t)pl * plane;
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)(lds0 + (uint32_t)((t & (NB - 1)) * (ROWS4 * TXQ * 16) + 4 * wave * (TXQ * 16))));
        unsigned keep;
        const float *g0 = src + srcoff[0];
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g0), "s"(dst) : "memory");
        if (two) {
            const float *g1 = src + srcoff[1];
            const uint32_t dst1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + TY * (TXQ * 16)));
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g1), "s"(dst1) : "memory");
        }
    };
    // wait until at most n of this wave's vector-memory operations are outstanding, then the barrier
    auto wait_barrier = [&](int n) {
        switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory"); break;
        }
    };
    // does this wave issue a store per output plane at all (else it must not count them)
    const bool wave_stores = __builtin_amdgcn_readfirstlane((int)(__ballot(writer) != 0ull)) != 0;
    // virtual rows E[endy + m], m = 0 .. HW, of this tile (the last tile row of the volume only)
    const bool yedge = y0 + TY + HW > endy;                    // block-uniform
    const int em = tid >> 4;                                   // this thread's m (tid < 16 (HW + 1))
    const int ej = endy + em - (y0 - HW);                      // its tile row
    const bool efix = yedge && em <= HW && ej < ROWS;
    int elo = 0;
    float ew0 = 0.0f, ew1 = 0.0f;
#pragma unroll
    for (int mm = 0; mm <= HW; mm++)
        if (mm == em) {
            elo = Ey.lo[mm] - (y0 - HW);
            ew0 = Ey.w0[mm];
            ew1 = Ey.w1[mm];
        }

    int t = 0;          // next request to be consumed
    int shist = 0;      // stores of the last three iterations (bits 0..2)
    const int nd = two ? 2 : 1;
    stage(0);
    stage(1);
    stage(2);
    // y-filtered value of this thread's column for the next request of the list
    auto yfilt = [&]() -> float4 {
        // younger than the pieces of request t: those of t + 1 and t + 2, and this wave's recent stores
        wait_barrier(2 * nd + __builtin_popcount(shist));
        stage(t + NB - 1);
        const int b = t & (NB - 1);
        if (yedge) {
            if (efix) {
                const float4 a = tile[b][clampi(elo, 0, ROWS - 1)][qx], c = tile[b][clampi(elo + 1, 0, ROWS - 1)][qx];
                tile[b][ej][qx] = Vec<4>::lerp(ew0, a, ew1, c);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dd = -HW; dd <= HW; dd++)
            Vec<4>::mac(acc, T.k[dd + HW], tile[b][ty + HW - dd][qx]);
        t++;
        return acc;
    };
    // extended-z plane r (local index; outside the slab at the global faces): one or two requests
    auto ext_z = [&](int r, bool stores) -> float4 {
        const int i = r + off;
        // Every plane but the high face's virtual ones (block-uniform) takes ONE request and nothing else: as a
        // path of its own it frees the plane loop of the selects and copies that merged it with the two-request
        // case (~65 of ~340 vector instructions per plane at 13 taps, and the wide instances are bound by their
        // vector instructions): the 13-tap launch in the step 0.44-0.45 -> 0.39-0.41 ms, the pyramid alone 3.55-3.59
        // -> 3.48-3.54 ms, bit-identical.  (Not at 17 taps: there the second path costs the register allocator
        // 72 dwords of scratch.)
        if (HW < 8 && i < endz) {
            const float4 yv = yfilt();
            shist = ((shist << 1) | (int)(stores && wave_stores)) & 7;
            return yv;
        }
        int np = 1;
        float w0 = 1.0f, w1 = 0.0f;
        if (i >= endz) {
            const int m = i - endz;
            np = m > HW ? 0 : 2;
#pragma unroll
            for (int mm = 0; mm <= HW; mm++)
                if (mm == m) {
                    w0 = Ez.w0[mm];
                    w1 = Ez.w1[mm];
                }
        }
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int k = 0; k < np; k++) {
            const float4 yv = yfilt();
            // (the store that follows this plane belongs to the LAST of its requests)
            shist = ((shist << 1) | (int)(stores && wave_stores && k + 1 == np)) & 7;
            // (block-uniform; the second request of a virtual plane interpolates in place: w0 * first + w1 * second)
            if (k == 0)
                a = yv;
            else
                a = Vec<4>::lerp(w0, a, w1, yv);
        }
        if (np == 0)
            shist = ((shist << 1) | (int)(stores && wave_stores)) & 7;   // (a store without a request)
        return a;
    };

    float4 ring[W];
#pragma unroll
    for (int i = 0; i < W; i++)
        ring[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 2 * HW; i++)
        ring[i] = ext_z(p0 - HW + i, false);
    float *__restrict__ d = P.dst + (size_t)y * nx + x;
#pragma unroll 1
    for (int q0 = p0; q0 < p1; q0 += W) {
#pragma unroll
        for (int j = 0; j < W; j++) {
            const int q = q0 + j;
            if (q < p1) {                              // block-uniform
                ring[(j + 2 * HW) % W] = ext_z(q + HW, true);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int dd = -HW; dd <= HW; dd++)
                    Vec<4>::mac(acc, T.k[dd + HW], ring[(j + HW - dd) % W]);   // E[q - d], d ascending
                if (writer)
                    st4(d + (size_t)q * plane, acc);
            }
        }
    }
    // (the re-requests beyond the list are still in flight: they write LDS only, and a wave's
    // vector-memory operations complete before its program ends)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    if ((P.nx & 63) == 0 && (P.ny & 63) == 0 && P.ny >= 128 && P.ts <= 256) {
        // whole 64 x 64 tiles: rows staged by LDS-DMA, three planes ahead
        // tile height (measured at 512^3, ms, 64 / 32 rows: 5 taps 0.22 / 0.24, 7 taps 0.26 / 0.25, 11 taps
        // 0.32 / 0.31, 17 taps 0.41 / 0.40): two 32-row workgroups per CU hide each other's barriers, a
        // 64-row one stages fewer halo rows per output row
        constexpr int DTY = HW <= 2 ? 64 : 32;
        dim3 grid(P.nx / 64, P.ny / DTY, nseg);
        hipLaunchKernelGGL((k_fir_yz_dma<HW, DTY>), grid, dim3(16 * DTY), 0, st, P, T, Ey, Ez);
    } else if ((P.nx & 63) == 0 && P.ny >= 128) {
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
    // (tile height of the 64 x 32 fallback; the launcher picks the tile)
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
        if ((n_out + nseg - 1) / nseg > 256)         /* (k_fir_yz_dma's request list) */
            nseg = (n_out + 255) / 256;
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

// sift3d_describe.hip -- the descriptor kernel (k_describe), the icosahedron tables it uses and
// their C entries (sift3d_hip_set_mesh, sift3d_hip_describe).
//
// A translation unit of its own because it is compiled with -fno-slp-vectorize, like
// sift3d_fir_yz.hip: packed v_pk_* arithmetic gains nothing on gfx950 and costs this kernel
// registers and ~7 % of its time.  Numerical contract (no FMA contraction) and citation style
// as in sift3d_kernels.hip.
#include "sift3d_kernels_common.h"
#include "sift3d_math.h"

#include <cstdlib>

// ---------------------------------------------------------------------------------------
// extract_descrip  (sift.c:1442-1536): one wave per keypoint.
//
// Scan    (64 voxels in parallel): window test (a cheap evaluation with margins, the reference's
//         expressions where a lane is in doubt), survivors stream-compacted into an LDS
//         queue so that the expensive phases always run on 64 window voxels.
// Phase A (one lane per voxel): gradient, Gaussian weight, rotation into keypoint space,
//         icosahedron face + barycentrics, the eight trilinear cell weights -- the reference's
//         float expressions, so every per-voxel DECISION (window, |g| threshold, face, skipped
//         corner) is the reference's, and every per-voxel VALUE bit for bit but two: the gradient
//         magnitude (v_sqrt_f32, 1 ulp) and the product mag * w_cell * bary, which the commit's
//         fused multiply-add does not round on its own (DESC_OPT below).
// Phase B (commit): a voxel adds mag*w_cell*bary_j to 24 distinct bins (8 cells x 3 face
//         vertices), which 24 lanes do as ONE plain LDS read-modify-write.  The two half-waves
//         commit two voxels per round into two PRIVATE histograms, which are added at the end.
//
// Accumulation order: the reference adds voxels in scan order (z, y, x).  Here every 64-voxel
// batch is cut into four runs of 16 consecutive voxels; half-wave 0 adds runs 0 and 2 to its
// histogram, half-wave 1 runs 1 and 3 to its own, and the two partial histograms are merged as
// h0 + h1.  The order is fixed by the keypoint alone (not by timing), so results are bitwise
// reproducible run to run; against the reference they differ by summation order only: the terms
// of a bin are >= -1e-6 * |term|, so the relative error of a bin is a few float ulps
// (BASELINE's bar: 1e-5 relative).
// LDS float atomics are not an option on gfx950: ds_add_f32 retires ~0.3 lane-adds/clk/CU
// (measured, scratch microbenchmark), 20x slower than the read-modify-write below.
// ---------------------------------------------------------------------------------------
struct FaceRec {
    float v0[3], e1[3], e2[3], t[3], q[3], e2q, idx[3];
};
static_assert(sizeof(FaceRec) == SIFT3D_HIP_FACE_FLOATS * 4, "face record layout");

// per-face constants as the kernel wants them: e1, e2, t, q, e2.q, packed bin offsets (+2 pad)
__constant__ float c_face16[20 * 16];
__constant__ int c_bin_off[12];    // LDS offset of each vertex's 64-cell block (see k_describe)
// face that contains a direction, by sign octant (bit 0/1/2 = x/y/z negative) and position
// relative to the octant's central face: see icos_guess
__constant__ int c_oct_face[32];

constexpr int DQ = 256; // compaction queue length (power of two, >= 63 + 2 * 64)
// LDS histogram: bin (cell, vertex) lives at c_bin_off[vertex] + cell, cell = ix + 4*iy + 16*iz.
// The offsets are 64*rank + {0, 8, 18, 26}[colour] for a proper 4-colouring of the
// icosahedron's vertices, which puts the 24 bins of any voxel (8 neighbouring cells x the 3
// vertices of a face) on 24 different banks of the 32 that ds_read_b32/ds_write_b32 use.
constexpr int HIST_USED = 800;   // bins + colouring gaps
constexpr int HIST_LDS = 800;
// face records in LDS: 20 floats apart (16 used), so that the 16-byte reads of lanes holding
// different faces start on different banks for 16 of the 20 faces
constexpr int FACE_STRIDE = 20;
// Phase A -> phase B records, field-major: row f holds field f of half a batch (32 voxels; the
// records go through LDS half a batch at a time: LDS capacity is what limits the waves per CU).
// Rows are 36 floats apart: the eight rows that the 24 committer lanes of a half-wave read with
// ds_read_b128 (four voxels at a time) fall on disjoint bank quads.
constexpr int RROW = 36;

#ifdef SIFT3D_AMD_DIAG
#define DESC_ABLATE_ARG , int ablate
#define DESC_ABLATE(bit) (ablate & (bit))
__device__ unsigned long long g_desc_voxels;   // window voxels committed (profiles/ model)
#else
#define DESC_ABLATE_ARG
#define DESC_ABLATE(bit) false
#endif

// Lanes of ONE wave exchanging data through LDS: the DS operations of a wave execute in issue
// order, so a read issued after a write sees it -- no s_waitcnt, no s_barrier; the fences only
// keep the compiler from moving LDS accesses across the hand-over point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// cart2bary + the acceptance test of icos_hist_bin (sift.c:276-297, 1268-1286) for one face,
// the reference's float expressions.  fr: the face's 16-float record (LDS).
__device__ __forceinline__ bool face_eval(const float4 *__restrict__ fr, float rx, float ry, float rz,
                                          float &xb, float &yb, float &zb, int &fi)
{
    const float4 A0 = fr[0], A1 = fr[1], A2 = fr[2], A3 = fr[3];
    // e1 = A0.xyz, e2 = (A0.w, A1.x, A1.y), t = (A1.z, A1.w, A2.x),
    // q = (A2.y, A2.z, A2.w), e2.q = A3.x, bin offsets = A3.y
    const float px = ry * A1.y - rz * A1.x;        // p = g x e2, sift.c:278
    const float py = rz * A0.w - rx * A1.y;
    const float pz = rx * A1.x - ry * A0.w;
    const float det = A0.x * px + A0.y * py + A0.z * pz;
    const float di = 1.0f / det;
    yb = di * (A1.z * px + A1.w * py + A2.x * pz);
    zb = di * (rx * A2.y + ry * A2.z + rz * A2.w);
    xb = 1.0f - yb - zb;
    const float kk = A3.x * di;
    fi = __float_as_int(A3.y);
    return !(fabsf(det) < 1.1920928955078125e-06f) &&            // sift.c:282
           !(xb < -1.1920928955078125e-06f || yb < -1.1920928955078125e-06f ||
             zb < -1.1920928955078125e-06f || kk < 0);            // sift.c:1277-1279
}

// A GUESS of the face a direction falls in (no decision rests on it: the caller verifies the
// guess with face_eval and falls back to the reference's full face scan).  The vertex set
// (0,+-1,+-g), (+-1,+-g,0), (+-g,0,+-1) is symmetric under sign flips, so the direction is
// reflected into the positive octant, which holds the face C = {(0,1,g), (1,g,0), (g,0,1)}
// and one third each of the three faces across C's edges; t1..t3 are the signed distances to
// the planes through the origin and C's edges.
__device__ __forceinline__ int icos_guess(float rx, float ry, float rz)
{
    const float g = 1.6180339887f, g2 = 2.6180339887f;
    const float ax = fabsf(rx), ay = fabsf(ry), az = fabsf(rz);
    const float t1 = ax + g2 * ay - g * az;      // edge (0,1,g)-(g,0,1), beyond: face with (0,-1,g)
    const float t2 = ay + g2 * az - g * ax;      // edge (g,0,1)-(1,g,0), beyond: face with (g,0,-1)
    const float t3 = az + g2 * ax - g * ay;      // edge (1,g,0)-(0,1,g), beyond: face with (-1,g,0)
    // (integer arithmetic, not nested selects: the compiler turns those into branches, and the
    // caller wants this in one basic block with the commit chain)
    const int n1 = t1 < 0.0f, n2 = t2 < 0.0f, n3 = t3 < 0.0f;
    const int cls = n1 + (1 - n1) * (2 * n2 + (1 - n2) * 3 * n3);
    return cls * 8 + (int)(rx < 0.0f) + 2 * (int)(ry < 0.0f) + 4 * (int)(rz < 0.0f);
}

// Gaussian window weights by squared voxel distance.  A keypoint sits on a voxel, and when the
// level's spacing is the same power of two u on all axes (every octave of an isotropic volume),
// the squared distance of a window voxel is EXACTLY k * u^2 in float for the integer
// k = i^2 + j^2 + l^2 (every term and partial sum is an integer times u^2 below 2^24).  The
// weight expf(-0.5f * sq / sigma^2) (sift.c:1498: an IEEE division and glibc's expf per voxel)
// then takes at most rad^2 / u^2 + 1 values per level: they are tabulated per level with exactly
// that expression and the kernel looks them up by k.  Table of level t: WL_STRIDE floats at
// t * WL_STRIDE: [0] 1.0f if the level qualifies, [1] the level's sd as float bits (2 floats),
// entries from [4].
constexpr int WL_STRIDE = 4096, WL_HEAD = 4;
constexpr int WL_COUNTERS = 16;   // uint32 work counters behind the tables (one per descriptor launch)

__global__ __launch_bounds__(256) void k_desc_wlut(const sift3d_hip_level *__restrict__ levels, int nlevels,
                                                   float *__restrict__ lut)
{
    const int t = blockIdx.x;
    if (t >= nlevels)
        return;
    if (t == 0 && threadIdx.x < WL_COUNTERS)   // the work counters of the descriptor launches that follow
        reinterpret_cast<uint32_t *>(lut + (size_t)nlevels * WL_STRIDE)[threadIdx.x] = 0u;
    const sift3d_hip_level L = levels[t];
    float *tab = lut + (size_t)t * WL_STRIDE;
    const float sigma = (float)(L.sd * 7.071067812);                  // sift.c:1453
    const float rad = (float)(2.0 * (double)sigma);                   // sift.c:1454
    const float rad2 = rad * rad, sig2 = sigma * sigma;
    int e;
    const float mant = frexpf(L.ux, &e);
    const float u2 = L.ux * L.ux;
    const bool ok = L.ux == L.uy && L.ux == L.uz && mant == 0.5f &&
                    rad2 / u2 < (float)(WL_STRIDE - WL_HEAD - 1) && u2 * (float)WL_STRIDE < 16777216.0f;
    if (threadIdx.x == 0) {
        tab[0] = ok ? 1.0f : 0.0f;
        tab[1] = 0.0f;
        tab[2] = __int_as_float((int)(__double_as_longlong(L.sd) & 0xffffffffll));
        tab[3] = __int_as_float((int)(__double_as_longlong(L.sd) >> 32));
    }
    if (!ok)
        return;
    for (int k = threadIdx.x; k < WL_STRIDE - WL_HEAD; k += 256) {
        const float sq = (float)k * u2;                               // exact
        const float arg = -0.5f * sq / sig2;
        tab[WL_HEAD + k] = arg >= -150.0f ? s3d_expf(arg) : 0.0f;     // (beyond rad^2: never read)
    }
}

constexpr int DWAVES = 4;   // keypoints (waves) per workgroup; they share the read-only tables
constexpr int DPARTS = 4;   // parts a window is summed in (fast variant; see k_describe)

// EXACT: the reference's accumulation ORDER and term arithmetic, for windows so large that a bin receives
// enough terms for any other order to drift past 1e-5 of the reference's own (float, sequential) sums --
// see "Accumulation order" above.  One histogram; the rounds of half-wave 0 (voxels 0..15 of a pass) run
// first, then those of half-wave 1 (voxels 16..31), the idle half-wave adding into the second histogram
// region, which is scratch here: every bin receives its terms in the reference's scan order (sift.c:96-108:
// z, y, x).  A term is fl(fl(mag * w_cell) * bary) added with a plain add (sift.c:1371-1373), the magnitude a
// correctly rounded sqrtf (sift.c:1331), the sum of squares of normalize_desc a sequential double sum in
// element order (sift.c:1407-1416): the histogram is the reference's bit for bit.  Twice the commit rounds:
// ~1.5x the time per window voxel.
template <bool EXACT>
__global__ __launch_bounds__(64 * DWAVES) void k_describe(const sift3d_hip_level *__restrict__ levels,
                                                 const sift3d_hip_kp *__restrict__ kps, uint32_t first,
                                                 uint32_t n,
                                                 float *__restrict__ out, float *__restrict__ out2,
                                                 const float *__restrict__ wlut,
                                                 uint32_t *__restrict__ work, float *__restrict__ part,
                                                 uint32_t *__restrict__ done DESC_ABLATE_ARG)
{
    // per wave: 2 * 3200 + 2016 + 1024 B; per workgroup 39.4 KB -> four workgroups = 16 waves per CU
    __shared__ float hist_[DWAVES][2 * HIST_LDS];   // one private histogram per half-wave
    // records of half a batch, field-major: rows 0..7 mag * trilinear weight of the eight cells,
    // 8..10 barycentric weights, 11..13 byte address of bin (base cell, face vertex j)
    __shared__ __attribute__((aligned(16))) float rec_[DWAVES][14][RROW];
    __shared__ int queue_[DWAVES][DQ];   // xx | yy<<10 | zz<<20, window-relative, in scan order
    __shared__ __attribute__((aligned(16))) float sface[20 * FACE_STRIDE]; // c_face16 (per-lane face index)
    __shared__ int soct[32];      // c_oct_face
    __shared__ uint64_t sexp[32]; // s3d_exp2_tab (per-lane index)
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float *const hist = hist_[wv];
    float(*const mw)[RROW] = &rec_[wv][0];
    float(*const bw)[RROW] = &rec_[wv][8];
    int(*const ab)[RROW] = reinterpret_cast<int(*)[RROW]>(&rec_[wv][11]);
    int *const queue = queue_[wv];
    for (int i = threadIdx.x; i < 20 * 16; i += 64 * DWAVES)
        sface[(i >> 4) * FACE_STRIDE + (i & 15)] = c_face16[i];
    if (threadIdx.x < 32) {
        soct[threadIdx.x] = c_oct_face[threadIdx.x];
        sexp[threadIdx.x] = s3d_exp2_tab[threadIdx.x];
    }
    __syncthreads();              // the only workgroup barrier: from here on the waves are independent
    // clock probe (round 5): the first wave of the launch lives as long as the kernel (the waves are
    // persistent: they loop over the work counter); it notes the shader-clock and the constant 100 MHz
    // counters when it starts and when it leaves, and stores the two differences behind the work counters --
    // the clock the chip HELD under this kernel, which bench.py prices the counter fractions with
    const bool probe = work && blockIdx.x == 0 && threadIdx.x == 0;
    long long probe_c0 = 0, probe_w0 = 0;
    if (probe) {
        probe_c0 = clock64();
        probe_w0 = wall_clock64();
    }
    // Keypoints are handed out one at a time from a counter (`work`; list order = widest windows first): a
    // wave that has finished its keypoint takes the next one.  With a fixed assignment of four keypoints to a
    // workgroup the three faster waves idled until the slowest was done -- their slots and LDS are released
    // per WORKGROUP -- which left ~15 % of the wave slots empty.  Every wave leaves the loop when the
    // counter passes the end of the list.  (work == nullptr: one keypoint per wave, by position.)
    // Round 5: a window is summed in DPARTS parts (ranges of its planes), each a work item of its own, and the
    // wave that finishes a keypoint's LAST part adds the parts' histograms in part order.  When the work counter
    // runs out every wave finishes the item it holds, so the device drains for as long as the last items take:
    // with whole windows ~1 ms of the 27 (measured: the kernel's time is 26.0 ms per list + 1.06 ms that do not
    // scale with the list; for lists of one level alone the fixed part is 0.8 x that level's window time).
    // The split is a function of the keypoint alone (never of its place in the list or of timing), so the
    // result is too.  The reference-order variant is not split: its sums have one order.
    const bool split = !EXACT && part != nullptr;
    for (bool first_pass = true;; first_pass = false) {
    uint32_t item;
    if (work) {
        uint32_t t = 0;
        if (lane == 0)
            t = atomicAdd(work, 1u);
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    } else {
        if (!first_pass)
            break;
        item = blockIdx.x * DWAVES + wv;
    }
    const uint32_t ki = first + (split ? item / DPARTS : item);
    const int pp = split ? (int)(item % DPARTS) : 0;
    if (ki >= n)
        break;
    for (int i = lane; i < 2 * HIST_LDS; i += 64)
        hist[i] = 0.0f;
    wave_sync();
    const sift3d_hip_kp K = kps[ki];
    const sift3d_hip_level L = levels[K.level];
    const uint32_t orow = K.row1 ? K.row1 - 1 : ki;   // output row (launch order may differ)
    // weight table of the keypoint's level (see k_desc_wlut): usable when the level qualified,
    // the keypoint carries the level's own scale and sits on a voxel
    const float *__restrict__ wtab = nullptr;
    const int icx = (int)K.cx, icy = (int)K.cy, icz = (int)K.cz;
    if (wlut) {
        const float *tab = wlut + (size_t)K.level * WL_STRIDE;
        const long long sdbits = ((long long)__float_as_int(tab[3]) << 32) | (unsigned)__float_as_int(tab[2]);
        if (tab[0] == 1.0f && sdbits == __double_as_longlong(K.sd) && (float)icx == K.cx &&
            (float)icy == K.cy && (float)icz == K.cz)
            wtab = tab + WL_HEAD;                    // wave-uniform
    }

    typedef const float __attribute__((address_space(1))) *gfloat_p;
    // (without a table the batch still issues its -- then unused -- weight load: any valid address)
    const gfloat_p wsrc = wtab ? (gfloat_p)wtab : (gfloat_p) reinterpret_cast<const float *>(levels);
    const int wmask = wtab ? -1 : 0;

    const float sigma = (float)(K.sd * 7.071067812);                  // sift.c:1453
    const float rad = (float)(2.0 * (double)sigma);                   // sift.c:1454
    const float half_w = (float)((double)rad / 1.4142135623730951);   // / sqrt(2), sift.c:1455
    const float desc_w = 2.0f * half_w;
    const float hist_w = desc_w / 4.0f;                               // / NHIST_PER_DIM
    const float bin_f = 1.0f / hist_w;
    const float rad2 = rad * rad;
    const float sig2 = sigma * sigma;
    const float *R = K.R; // Rt[i][j] = R[j][i]
    const float iux = 1.0f / L.ux, iuy = 1.0f / L.uy, iuz = 1.0f / L.uz;
    Box B;
    bounds_f(K.cx, rad, L.ux, L.nx, B.xs, B.xe);
    bounds_f(K.cy, rad, L.uy, L.ny, B.ys, B.ye);
    bounds_f(K.cz, rad, L.uz, L.nz_glob, B.zs, B.ze);
    // memory safety on Z-slabs: never outside the local planes (see k_orient)
    B.zs = max(B.zs, L.z_off + 1);
    B.ze = min(B.ze, L.z_off + L.nz - 2);
    // phase B roles: each half-wave commits one voxel per round; its lanes 0..23 are the
    // (trilinear cell corner, face vertex) pairs of that voxel.  Lanes 24..31 repeat lane 0's
    // work (same address, same value: harmless; giving them scratch slots of their own measured
    // MORE bank conflicts), which keeps the commit free of predication.
    const int half = lane >> 5, l5 = lane & 31;
    const int pc = l5 < 24 ? l5 / 3 : 0, pj = l5 < 24 ? l5 - 3 * pc : 0;
    const int pdx = (pc >> 2) & 1, pdy = (pc >> 1) & 1, pdz = pc & 1;
    // byte offset of this lane's cell corner, in this half-wave's histogram (EXACT: in THE histogram while
    // this half-wave's rounds run -- sequence A: half-wave 0, sequence B: half-wave 1 --, else in the scratch
    // region)
    const int corner4 = 4 * (pdx + 4 * pdy + 16 * pdz);
    const int coff_a = corner4 + (EXACT ? (half == 0 ? 0 : HIST_LDS * 4) : half * (HIST_LDS * 4));
    const int coff_b = corner4 + (half == 1 ? 0 : HIST_LDS * 4);
    (void)coff_b;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    uint32_t qhead = 0, qtail = 0; // wave-uniform
    bool pend = false;             // registers pv/ppk hold the batch starting at qhead
#ifdef SIFT3D_AMD_DIAG
    unsigned diag_voxels = 0;      // window voxels of this keypoint (one atomic per wave at the end)
#endif

    // Window voxel -> (window test, spatial bins).  Same float expressions in the scan and
    // in the batch, so both see identical values.
    auto window = [&](int x, int y, int z, float &sq, float &vbx, float &vby, float &vbz) -> bool {
        const float dx = ((float)x - K.cx) * L.ux;                 // sift.c:102-104
        const float dy = ((float)y - K.cy) * L.uy;
        const float dz = ((float)z - K.cz) * L.uz;
        sq = dx * dx + dy * dy + dz * dz;
        // vkp = Rt * vim (immacros.h:328-340)
        const float kx = R[0] * dx + R[3] * dy + R[6] * dz;
        const float ky = R[1] * dx + R[4] * dy + R[7] * dz;
        const float kz = R[2] * dx + R[5] * dy + R[8] * dz;
        vbx = (kx + half_w) * bin_f;                               // sift.c:1483-1485
        vby = (ky + half_w) * bin_f;
        vbz = (kz + half_w) * bin_f;
        // sift.c:106 (float) and sift.c:1488-1492: none of vb* < 0, none >= 4 (finite inputs)
        const float lo = fminf(fminf(vbx, vby), vbz), hi = fmaxf(fmaxf(vbx, vby), vbz);
        return !(sq > rad2) && !(lo < 0.0f) && !(hi >= 4.0f);
    };

    // One batch: the next `cnt` (<= 64) queued voxels, in scan order.
    // The six gradient samples of a batch are fetched one batch ahead (registers pv/ppk), so
    // their HBM/L2 latency overlaps the previous batch's binning and commit.
    float pv[7] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    int ppk = 0;
    // level samples are read through a global-address-space pointer (the pointer comes out of a
    // table in memory, which would otherwise make these flat loads); nx*ny < 2^31
    const gfloat_p gdata = (gfloat_p)L.data;
    const uint32_t ys32 = (uint32_t)L.nx, zs32 = (uint32_t)L.nx * (uint32_t)L.ny;
    // Sample requests of a batch, branch-free: lanes beyond cnt keep their previous (valid) voxel
    // and simply fetch it again.  The Gaussian weight (prefetch_weight) only needs the
    // coordinates; it is evaluated later, inside the commit chain of the previous batch.
    auto prefetch_loads = [&](uint32_t start, int cnt) {
        const int qv = queue[(start + lane) & (DQ - 1)];
        ppk = lane < cnt ? qv : ppk;
        const int x = B.xs + (ppk & 1023), y = B.ys + ((ppk >> 10) & 1023),
                  zl = B.zs + (ppk >> 20) - L.z_off;
        const gfloat_p p = gdata + ((uint64_t)zs32 * (uint32_t)zl + (uint32_t)(x + (int)ys32 * y));
        pv[0] = p[1]; pv[1] = *(p - 1); pv[2] = p[ys32]; pv[3] = *(p - ys32);
        pv[4] = p[zs32]; pv[5] = *(p - zs32);
        if (wtab) {
            // the Gaussian weight is a seventh (L2-resident) load, in flight with the samples
            const int i = x - icx, j = y - icy, l = B.zs + (ppk >> 20) - icz;
            pv[6] = wtab[i * i + j * j + l * l];
        }
    };
    auto prefetch_weight = [&]() {
        if (wtab)
            return;
        const int x = B.xs + (ppk & 1023), y = B.ys + ((ppk >> 10) & 1023);
        const float dx = ((float)x - K.cx) * L.ux;                 // sift.c:102-104
        const float dy = ((float)y - K.cy) * L.uy;
        const float dz = ((float)(B.zs + (ppk >> 20)) - K.cz) * L.uz;
        // (queued voxels passed the window test: the argument lies in [-2, 0])
        pv[6] = s3d_expf_in_range(-0.5f * (dx * dx + dy * dy + dz * dz) / sig2, sexp); // sift.c:1498
    };
    auto prefetch = [&](uint32_t start, int cnt) {
        prefetch_loads(start, cnt);
        prefetch_weight();
    };

    // Records of the batch that is being committed, one voxel per lane: mag * trilinear weight
    // of the eight cells, the three barycentric weights, the three bin addresses.  Empty records
    // (zero weight, bin address 0) add 0 to a valid bin.
    // An LDS write costs by the instruction and the dwords per lane, whatever the number of
    // active lanes (profiles/microbench/lds_cost.hip), so all 64 lanes write in both passes: the 14 fields
    // are traded between the half-waves (v_permlane32_swap) so that lane l holds, for voxel l & 31
    // of each half-batch, fields 0..6 (l < 32) or 7..13 (l >= 32): rp[pass][i].
    float rp[2][7] = { { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f } };
    // One commit pass: the records of half a batch (voxels 32 * pass .. 32 * pass + 31) go to
    // LDS, field-major, then round u adds voxel u of them (half-wave 0) and voxel 16 + u
    // (half-wave 1), each into its half-wave's own histogram.  The 24 lanes of a voxel own 24
    // distinct bins (8 cells x 3 face vertices), so the voxel's adds are ONE plain LDS
    // read-modify-write; the DS operations of a wave execute in issue order, so round u + 1
    // sees round u's sums.  (Reading round u + 1's bins before round u's sums are written is
    // not an option: a bin of round u + 1 is usually a bin that ANOTHER lane writes in round
    // u.)  The chain of dependent LDS round trips is the longest latency of the kernel and needs
    // almost no VALU, so each pass is issued in one basic block with half of the arithmetic of
    // the NEXT batch (see batch()), which the scheduler interleaves with it.  Records of four
    // rounds are read with three 16-byte loads, one chunk ahead.
    float *const rrow = &rec_[wv][7 * half][l5];
    auto commit_write = [&](int pass) {
#pragma unroll
        for (int i = 0; i < 7; i++)
            rrow[i * RROW] = rp[pass][i];
        wave_sync();
    };
    // The commit chain, interleaved BY HAND with the per-voxel arithmetic.  A round is
    //     address, ds_read_b32 (bin) | ... | s_waitcnt, v_add_f32, ds_write_b32
    // and the next round's read can only be issued after this round's write, so a round costs one
    // LDS round trip (>= 64 cycles, more when the pipe is busy) during which the wave has nothing
    // of the chain to do.  Left to itself the scheduler puts the whole chain in one piece and all
    // other arithmetic of the basic block before and after it (every round then exposes its full
    // latency, and only the other waves of the SIMD can fill it).  Here every round carries a
    // SLICE of independent work -- a dozen VALU instructions of phase A of this batch, or of the
    // sample requests of the next -- between the bin read and the wait for it; scheduling
    // barriers keep the slices where they are put.  Records of four rounds are read with three
    // 16-byte loads, one chunk ahead.
#define SB() __builtin_amdgcn_sched_barrier(0)
// DESC_OPT (bit mask, A/B builds only; the shipped value is the default below):
//   1 the commit adds mag*w_cell*bary to the bin with ONE fused multiply-add (the product is not rounded
//     separately: one VALU instruction less per round, the term at least as accurate)
//   2 the face GUESS with fused multiply-adds; the gradient magnitude by v_sqrt_f32 (1 ulp: a RELATIVE
//     error of 6e-8 on the term).  The barycentrics keep cart2bary's separately rounded operations and
//     the IEEE division: b0 = 1 - b1 - b2 carries the rounding noise of b1 and b2 (6e-8 ABSOLUTE), so a
//     bin fed by a few voxels with a small b0 only agrees with the reference to 1e-5 relative if that
//     noise is reproduced bit for bit (the FMA / v_rcp_f32 variant was 5 % faster and is not used)
//   4 window scan: a cheap test with margins decides, the exact one only where a lane is in doubt
#ifndef DESC_OPT
#define DESC_OPT 7
#endif
#if DESC_OPT & 1
#define COMMIT_ADD(old, m, b) (EXACT ? (old) + (m) * (b) : __builtin_fmaf(m, b, old))
#else
#define COMMIT_ADD(old, m, b) ((old) + (m) * (b))
#endif
// a value is computed in the slice that names it here (not sunk to its first use in a later one)
#define KEEP(v) asm volatile("" ::"v"(v))
#define COMMIT_BEGIN(coff)                                                                    \
    const int coff4 = (coff);                                                                 \
    int4 cb4 = *reinterpret_cast<const int4 *>(&ab[pj][hb]);                                   \
    float4 cw4 = *reinterpret_cast<const float4 *>(&mw[pc][hb]);                               \
    float4 cx4 = *reinterpret_cast<const float4 *>(&bw[pj][hb]);                               \
    int4 nb4 = cb4;                                                                           \
    float4 nw4 = cw4, nx4 = cx4;
#define ROUND(u, ...)                                                                         \
    {                                                                                         \
        if (((u) & 3) == 0 && (u) + 4 < 16) {                                                 \
            nb4 = *reinterpret_cast<const int4 *>(&ab[pj][hb + (u) + 4]);                      \
            nw4 = *reinterpret_cast<const float4 *>(&mw[pc][hb + (u) + 4]);                    \
            nx4 = *reinterpret_cast<const float4 *>(&bw[pj][hb + (u) + 4]);                    \
        }                                                                                     \
        const int mb_[4] = { cb4.x, cb4.y, cb4.z, cb4.w };                                    \
        const float mv_[4] = { cw4.x, cw4.y, cw4.z, cw4.w }, bv_[4] = { cx4.x, cx4.y, cx4.z, cx4.w }; \
        float *bin_ = reinterpret_cast<float *>(reinterpret_cast<char *>(hist) + (mb_[(u) & 3] + coff4)); \
        float old_ = 0.0f;                                                                    \
        if (!DESC_ABLATE(1))                                                                  \
            old_ = *bin_;                                                                     \
        SB();                                                                                 \
        { __VA_ARGS__ }                                                                       \
        SB();                                                                                 \
        if (!DESC_ABLATE(1))                                                                  \
            *bin_ = COMMIT_ADD(old_, mv_[(u) & 3], bv_[(u) & 3]);      /* sift.c:1371-1373 */ \
        if (((u) & 3) == 3) {                                                                 \
            cb4 = nb4;                                                                        \
            cw4 = nw4;                                                                        \
            cx4 = nx4;                                                                        \
        }                                                                                     \
        SB();                                                                                 \
    }
    // EXACT: the same 16 rounds for half-wave 1's voxels (sequence B), after half-wave 0's
#define CHAIN_B()                                                                             \
    if (EXACT) {                                                                              \
        COMMIT_BEGIN(coff_b);                                                                 \
        ROUND(0, ) ROUND(1, ) ROUND(2, ) ROUND(3, ) ROUND(4, ) ROUND(5, ) ROUND(6, ) ROUND(7, ) \
        ROUND(8, ) ROUND(9, ) ROUND(10, ) ROUND(11, ) ROUND(12, ) ROUND(13, ) ROUND(14, ) ROUND(15, ) \
    }
    // One batch: phase A of `cnt` (<= 64) voxels whose samples were fetched one batch ago (cv,
    // pk), woven into the two commit passes of the previous batch (records in rp), as are the
    // sample requests of the next batch (ncnt voxels from queue position nstart; ncnt may be 0);
    // then this batch's records take the previous one's place.
    // One commit pass: the records of half a batch (voxels 32 * pass .. 32 * pass + 31) go to
    // LDS, field-major, then round u adds voxel u of them (half-wave 0) and voxel 16 + u
    // (half-wave 1), each into its half-wave's own histogram.  The 24 lanes of a voxel own 24
    // distinct bins (8 cells x 3 face vertices), so the voxel's adds are ONE plain LDS
    // read-modify-write; the DS operations of a wave execute in issue order, so round u + 1
    // sees round u's sums.  (Reading round u + 1's bins before round u's sums are written is
    // not an option: a bin of round u + 1 is usually a bin that ANOTHER lane writes in round u.)
    auto batch = [&](int cnt, const float *cv, int pk, uint32_t nstart, int ncnt) {
#ifdef SIFT3D_AMD_DIAG
        diag_voxels += (unsigned)cnt;
#endif
        if (DESC_ABLATE(2)) { prefetch(nstart, ncnt); return; }
        const int hb = half * 16;
        // (lanes beyond cnt compute on stale -- finite -- values and are discarded below)
        int x, y, z;
        float dx, dy, dz, kx, ky, kz, vbx, vby, vbz, gx, gy, gz, rx, ry, rz, m2;
        float t1, t2, t3, px, py, pz, det, di, kk;
        float4 A0, A1, A2, A3;
        int f0;
        bool live = false, found = false;
        int fidx = 0;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f;
        // ================= pass 0: rounds of voxels 0..31  ||  window, gradient, face
        commit_write(0);
        {
            COMMIT_BEGIN(coff_a);
            ROUND(0, x = B.xs + (pk & 1023); y = B.ys + ((pk >> 10) & 1023); z = B.zs + (pk >> 20);
                     dx = ((float)x - K.cx) * L.ux; KEEP(dx);)             // sift.c:102-104
            ROUND(1, dy = ((float)y - K.cy) * L.uy; dz = ((float)z - K.cz) * L.uz; KEEP(dy); KEEP(dz);
                     // IM_GET_GRAD_ISO (sift.c:140-145, immacros.h:105-111)
                     gx = 0.5f * (cv[0] - cv[1]); gy = 0.5f * (cv[2] - cv[3]); gz = 0.5f * (cv[4] - cv[5]);)
            ROUND(2, gx *= iux; gy *= iuy; gz *= iuz;
                     gx = gx * cv[6]; gy = gy * cv[6]; gz = gz * cv[6];)   // sift.c:1498, see prefetch_weight
            // vkp = Rt * vim (immacros.h:328-340)
            ROUND(3, kx = R[0] * dx + R[3] * dy + R[6] * dz; ky = R[1] * dx + R[4] * dy + R[7] * dz; KEEP(kx); KEEP(ky);)
            ROUND(4, kz = R[2] * dx + R[5] * dy + R[8] * dz;
                     vbx = (kx + half_w) * bin_f; vby = (ky + half_w) * bin_f; vbz = (kz + half_w) * bin_f; // sift.c:1483-1485
                     KEEP(vbx); KEEP(vby); KEEP(vbz);)
            ROUND(5, rx = R[0] * gx + R[3] * gy + R[6] * gz; ry = R[1] * gx + R[4] * gy + R[7] * gz;) // sift.c:1502
            ROUND(6, rz = R[2] * gx + R[5] * gy + R[8] * gz; m2 = rx * rx + ry * ry + rz * rz;
                     live = lane < cnt && !(m2 < 1.1920928955078125e-06f);)   // sift.c:1264
            // icos_hist_bin (sift.c:1268-1286): the first face in table order whose barycentrics
            // are >= -eps wins.  A face other than the one the ray really crosses can only pass
            // if the ray misses it by ~eps, i.e. if the ray is within ~eps of an edge of its own
            // face.  So: guess the face (icos_guess), evaluate it with cart2bary's arithmetic
            // (face_eval), and accept it when it passes with all barycentrics > 2e-5 (then every
            // other face fails by a wide margin and the first match is unique); anything else --
            // ~1e-4 of the voxels -- takes the reference's scan over all 20 faces.
#if DESC_OPT & 2
            // (the guess may be computed any way: no decision rests on it)
            ROUND(7, const float g = 1.6180339887f, g2 = 2.6180339887f;
                     const float ax_ = fabsf(rx), ay_ = fabsf(ry), az_ = fabsf(rz);
                     t1 = __builtin_fmaf(g2, ay_, __builtin_fmaf(-g, az_, ax_));
                     t2 = __builtin_fmaf(g2, az_, __builtin_fmaf(-g, ax_, ay_));
                     t3 = __builtin_fmaf(g2, ax_, __builtin_fmaf(-g, ay_, az_));)
#else
            ROUND(7, const float g = 1.6180339887f, g2 = 2.6180339887f;
                     const float ax_ = fabsf(rx), ay_ = fabsf(ry), az_ = fabsf(rz);
                     t1 = ax_ + g2 * ay_ - g * az_; t2 = ay_ + g2 * az_ - g * ax_; t3 = az_ + g2 * ax_ - g * ay_;)
#endif
            // (a value read from LDS in one slice is used in a LATER one, so that no slice waits)
            ROUND(8, const int n1 = t1 < 0.0f, n2 = t2 < 0.0f, n3 = t3 < 0.0f;
                     const int cls = n1 + (1 - n1) * (2 * n2 + (1 - n2) * 3 * n3);
                     f0 = soct[cls * 8 + (int)(rx < 0.0f) + 2 * (int)(ry < 0.0f) + 4 * (int)(rz < 0.0f)];)
            ROUND(9, const float4 *fr = reinterpret_cast<const float4 *>(sface + f0 * FACE_STRIDE);
                     A0 = fr[0]; A1 = fr[1]; A2 = fr[2]; A3 = fr[3];)
            // cart2bary (sift.c:276-297): e1 = A0.xyz, e2 = (A0.w, A1.x, A1.y), t = (A1.z, A1.w, A2.x),
            // q = (A2.y, A2.z, A2.w), e2.q = A3.x, bin offsets = A3.y
            ROUND(10, px = ry * A1.y - rz * A1.x; py = rz * A0.w - rx * A1.y; pz = rx * A1.x - ry * A0.w;) // p = g x e2
            ROUND(11, det = A0.x * px + A0.y * py + A0.z * pz; di = 1.0f / det;)
            ROUND(12, b1 = di * (A1.z * px + A1.w * py + A2.x * pz);)
            ROUND(13, b2 = di * (rx * A2.y + ry * A2.z + rz * A2.w); b0 = 1.0f - b1 - b2;)
            ROUND(14, kk = A3.x * di; fidx = __float_as_int(A3.y);
                      found = !(fabsf(det) < 1.1920928955078125e-06f) &&            // sift.c:282
                              !(b0 < -1.1920928955078125e-06f || b1 < -1.1920928955078125e-06f ||
                                b2 < -1.1920928955078125e-06f || kk < 0) &&          // sift.c:1277-1279
                              fminf(b0, fminf(b1, b2)) > 2e-5f;)
            ROUND(15, )
        }
        CHAIN_B();
        if (__builtin_expect(__ballot(live && !found) != 0ull, 0)) {
            bool open = live && !found;
#pragma unroll 1
            for (int f = 0; f < 20; f++) {
                float xb, yb, zb;
                int fi;
                const bool hit = face_eval(reinterpret_cast<const float4 *>(sface + f * FACE_STRIDE), rx,
                                           ry, rz, xb, yb, zb, fi);
                if (open && hit) {
                    b0 = xb; b1 = yb; b2 = zb; fidx = fi;
                    found = true;
                    open = false;
                }
                if (__ballot(open) == 0ull)
                    break;
            }
        }
        // ================= pass 1: rounds of voxels 32..63  ||  the next batch's sample requests,
        // cell weights, this batch's records
        commit_write(1);
        {
            COMMIT_BEGIN(coff_a);
            int nx_, ny_, nzl_, qv_;
            gfloat_p np_;
            float mag, fx, fy, fz;
            int ix, iy, iz, cell4;
            float ax[2], ay[2], az[2], fld[14];
            // Sample requests of the next batch, branch-free: lanes beyond ncnt keep their previous
            // (valid) voxel and simply fetch it again
            ROUND(0, qv_ = queue[(nstart + lane) & (DQ - 1)];)
            ROUND(1, ppk = lane < ncnt ? qv_ : ppk;
                     nx_ = B.xs + (ppk & 1023); ny_ = B.ys + ((ppk >> 10) & 1023); nzl_ = B.zs + (ppk >> 20) - L.z_off;
                     np_ = gdata + ((uint64_t)zs32 * (uint32_t)nzl_ + (uint32_t)(nx_ + (int)ys32 * ny_));)
            ROUND(2, pv[0] = np_[1]; pv[1] = *(np_ - 1); pv[2] = np_[ys32]; pv[3] = *(np_ - ys32);)
            // the Gaussian weight is a seventh (L2-resident) load, in flight with the samples (without a
            // table the load is a dummy and prefetch_weight() computes the weight)
            ROUND(3, pv[4] = np_[zs32]; pv[5] = *(np_ - zs32);
                     const int i = nx_ - icx; const int j = ny_ - icy; const int l = B.zs + (ppk >> 20) - icz;
                     pv[6] = wsrc[(i * i + j * j + l * l) & wmask];)
            // trilinear cell weights (sift.c:1318-1320, 1361-1363): weight = wx * wy * wz,
            // value = mag * weight * bary.  A corner beyond the last cell is skipped by the
            // reference (sift.c:1349-1352).  The commit is free of predication -- all 24 lanes
            // of a voxel always read-modify-write -- so the 2x2x2 block of cells is shifted to stay
            // inside the grid instead: on an axis where the base cell is the last one (index 3) the
            // block covers cells {2, 3}, cell 3 keeps its weight 1 - f and cell 2 gets weight 0.
            // The 24 bins stay distinct and valid; adding mag * 0 * bary = +-0 changes nothing.
            // Voxels without a contribution: magnitude 0 (all eight weights become +-0; the stale
            // values behind them are finite), barycentrics 0 (0 * NaN would be NaN); their bin
            // addresses stay valid ones.
#if DESC_OPT & 2
            ROUND(4, mag = EXACT ? sqrtf(m2) : __builtin_amdgcn_sqrtf(m2);               // sift.c:1331 (else: 1 ulp)
                     mag = live && found ? mag : 0.0f;)
#else
            ROUND(4, mag = sqrtf(m2); mag = live && found ? mag : 0.0f;)               // sift.c:1331
#endif
            ROUND(5, fx = vbx - floorf(vbx); fy = vby - floorf(vby); fz = vbz - floorf(vbz);
                     ix = (int)vbx; iy = (int)vby; iz = (int)vbz;)
            ROUND(6, const bool lx = ix >= 3; const bool ly = iy >= 3; const bool lz = iz >= 3;
                     ax[0] = lx ? 0.0f : 1.0f - fx; ax[1] = lx ? 1.0f - fx : fx;
                     ay[0] = ly ? 0.0f : 1.0f - fy; ay[1] = ly ? 1.0f - fy : fy;
                     az[0] = lz ? 0.0f : 1.0f - fz; az[1] = lz ? 1.0f - fz : fz;)
            ROUND(7, _Pragma("unroll") for (int c = 0; c < 4; c++)
                         fld[c] = mag * (ax[(c >> 2) & 1] * ay[(c >> 1) & 1] * az[c & 1]);)
            ROUND(8, _Pragma("unroll") for (int c = 4; c < 8; c++)
                         fld[c] = mag * (ax[(c >> 2) & 1] * ay[(c >> 1) & 1] * az[c & 1]);)
            // (stale lanes may hold any coordinates: keep their cell inside the grid as well)
            ROUND(9, cell4 = 4 * (min(max(ix, 0), 2) + 4 * min(max(iy, 0), 2) + 16 * min(max(iz, 0), 2));
                     const bool ok = live && found;
                     fld[8] = ok ? b0 : 0.0f; fld[9] = ok ? b1 : 0.0f; fld[10] = ok ? b2 : 0.0f;)
            // byte addresses of the bins (base cell, face vertex j) -- the vertices addressed
            // through the UNSWAPPED idx[] of the face (quirk Q1)
            ROUND(10, _Pragma("unroll") for (int j = 0; j < 3; j++)
                          fld[11 + j] = __int_as_float(4 * ((fidx >> (10 * j)) & 1023) + cell4);)
            // swap(a, b): first result = a of lanes 0..31 | b of lanes 0..31, second = a of lanes
            // 32..63 | b of lanes 32..63
            ROUND(11, _Pragma("unroll") for (int i = 0; i < 4; i++) {
                          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(fld[i]), __float_as_uint(fld[7 + i]), false, false);
                          rp[0][i] = __uint_as_float(r[0]); rp[1][i] = __uint_as_float(r[1]); })
            ROUND(12, _Pragma("unroll") for (int i = 4; i < 7; i++) {
                          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(fld[i]), __float_as_uint(fld[7 + i]), false, false);
                          rp[0][i] = __uint_as_float(r[0]); rp[1][i] = __uint_as_float(r[1]); })
            ROUND(13, )
            ROUND(14, )
            ROUND(15, )
        }
        CHAIN_B();
        prefetch_weight();
    };
    // the commit passes of the last batch (nothing left to weave in)
    auto commit_tail = [&](int pass) {
        const int hb = half * 16;
        commit_write(pass);
        {
            COMMIT_BEGIN(coff_a);
            ROUND(0, ) ROUND(1, ) ROUND(2, ) ROUND(3, ) ROUND(4, ) ROUND(5, ) ROUND(6, ) ROUND(7, )
            ROUND(8, ) ROUND(9, ) ROUND(10, ) ROUND(11, ) ROUND(12, ) ROUND(13, ) ROUND(14, ) ROUND(15, )
        }
        CHAIN_B();
    };

    // The reference scans the whole bounding box of the sphere (sift.c:96-108).  Every voxel
    // that passes the window test lies in the sphere AND in the rotated 4x4x4 cube, so each
    // plane only needs the voxels of a (conservative: +1 voxel, +0.1 %) rectangle around the
    // plane's disc, clipped to the cube's extent along the image axes; the exact per-voxel
    // test decides.
    const float cube_x = half_w * (fabsf(R[0]) + fabsf(R[1]) + fabsf(R[2])) * 1.001f;
    const float cube_y = half_w * (fabsf(R[3]) + fabsf(R[4]) + fabsf(R[5])) * 1.001f;
    const float cube_z = half_w * (fabsf(R[6]) + fabsf(R[7]) + fabsf(R[8])) * 1.001f;
    int zs = max(B.zs, (int)floorf(K.cz - cube_z / L.uz - 1.0f));
    int ze = min(B.ze, (int)ceilf(K.cz + cube_z / L.uz + 1.0f));
    if (split) {
        // part pp: planes [zs + cut(pp), zs + cut(pp + 1)) of the window's planes -- the cuts at 0.33 / 0.5 / 0.67
        // of the range give the four parts of a sphere about the same number of voxels
        const int np_ = max(ze - zs + 1, 0);
        const int c1 = (33 * np_ + 50) / 100, c2 = (np_ + 1) / 2, c3 = (67 * np_ + 50) / 100;
        static_assert(DPARTS == 4, "three cuts");
        const int lo = pp == 0 ? 0 : pp == 1 ? c1 : pp == 2 ? c2 : c3;
        const int hi = pp == 0 ? c1 : pp == 1 ? c2 : pp == 2 ? c3 : np_;
        ze = zs + hi - 1;
        zs = zs + lo;
    }
#if DESC_OPT & 4
    // The scan's window test in two tiers.  The bin coordinates of voxel (pxs + xx, pys + yy) of a plane
    // are affine in (xx, yy): vb_i = c_i + a_i * xx + b_i * yy, sq = (dx0 + ux * xx)^2 + (dy0 + uy * yy)^2
    // + dz^2 -- 14 fused multiply-adds and conversions instead of the reference's 27 separately rounded
    // operations.  Both evaluations are within ~3e-6 of the real value (|vb| < 6, six to eight roundings
    // of 2^-24 relative each; sq: relative), so a voxel that passes the cheap test by MARGIN 1e-4 passes
    // the reference's test, one that fails it by that margin fails the reference's, and the exact
    // expressions (window()) decide only for chunks in which some lane falls inside the margin (~2 % of
    // the chunks; NaNs compare false on both sides and land there too).  Every decision is the reference's.
    const float wa0 = R[0] * L.ux * bin_f, wa1 = R[1] * L.ux * bin_f, wa2 = R[2] * L.ux * bin_f;
    const float wb0 = R[3] * L.uy * bin_f, wb1 = R[4] * L.uy * bin_f, wb2 = R[5] * L.uy * bin_f;
    const float rad2_in = rad2 * 0.9999f, rad2_out = rad2 * 1.0001f;
#endif
    for (int z = zs; z <= ze; z++) {
        const float dzp = ((float)z - K.cz) * L.uz;
        const float rz = sqrtf(fmaxf(rad2 - dzp * dzp, 0.0f)) * 1.001f;
        const float xr = fminf(rz, cube_x) / L.ux + 1.0f, yr = fminf(rz, cube_y) / L.uy + 1.0f;
        const int pxs = max(B.xs, (int)floorf(K.cx - xr)), pxe = min(B.xe, (int)ceilf(K.cx + xr));
        const int pys = max(B.ys, (int)floorf(K.cy - yr)), pye = min(B.ye, (int)ceilf(K.cy + yr));
        const int pbx = pxe - pxs + 1, pby = pye - pys + 1;
        const int ox = pxs - B.xs, oy = pys - B.ys;
#if DESC_OPT & 4
        const float dx0 = ((float)pxs - K.cx) * L.ux, dy0 = ((float)pys - K.cy) * L.uy, dz2 = dzp * dzp;
        const float wc0 = (R[0] * dx0 + R[3] * dy0 + R[6] * dzp + half_w) * bin_f;
        const float wc1 = (R[1] * dx0 + R[4] * dy0 + R[7] * dzp + half_w) * bin_f;
        const float wc2 = (R[2] * dx0 + R[5] * dy0 + R[8] * dzp + half_w) * bin_f;
        // acc: passes by margin; the return value: neither passes nor fails by margin.  Two x-adjacent voxels of a
        // row per call: what depends on the row alone (5 of a voxel's 14 fused multiply-adds) is formed once
        auto cheap2 = [&](int xx_, int yy_, bool &acc0, bool &acc1) -> bool {
            const float fx_ = (float)xx_, fy_ = (float)yy_, gx_ = (float)(xx_ + 1);
            const float t0 = __builtin_fmaf(wb0, fy_, wc0), t1 = __builtin_fmaf(wb1, fy_, wc1),
                        t2 = __builtin_fmaf(wb2, fy_, wc2);
            const float ey = __builtin_fmaf(fy_, L.uy, dy0), sy = __builtin_fmaf(ey, ey, dz2);
            const float v0 = __builtin_fmaf(wa0, fx_, t0), v1 = __builtin_fmaf(wa1, fx_, t1),
                        v2 = __builtin_fmaf(wa2, fx_, t2);
            const float u0 = __builtin_fmaf(wa0, gx_, t0), u1 = __builtin_fmaf(wa1, gx_, t1),
                        u2 = __builtin_fmaf(wa2, gx_, t2);
            const float ex = __builtin_fmaf(fx_, L.ux, dx0), fx2 = __builtin_fmaf(gx_, L.ux, dx0);
            const float sq_ = __builtin_fmaf(ex, ex, sy), sq2 = __builtin_fmaf(fx2, fx2, sy);
            const float lo = fminf(fminf(v0, v1), v2), hi = fmaxf(fmaxf(v0, v1), v2);
            const float lo2 = fminf(fminf(u0, u1), u2), hi2 = fmaxf(fmaxf(u0, u1), u2);
            acc0 = sq_ <= rad2_in && lo >= 1e-4f && hi <= 3.9999f;
            acc1 = sq2 <= rad2_in && lo2 >= 1e-4f && hi2 <= 3.9999f;
            const bool rej0 = sq_ > rad2_out || lo < -1e-4f || hi >= 4.0001f;
            const bool rej1 = sq2 > rad2_out || lo2 < -1e-4f || hi2 >= 4.0001f;
            return (!acc0 && !rej0) || (!acc1 && !rej1);
        };
#endif
        // A lane tests the voxels at positions 2 * lane and 2 * lane + 1 of a run of 128 positions of the rectangle
        // (row-major = the reference's scan order inside a plane): two neighbours in x, of ONE row -- the
        // rectangle is walked with an even width pbe >= pbx (the odd column, if any, is beyond the rectangle and
        // never accepted).  128 positions further the pair is at (yy + q128, xx + r128), one more row if xx wraps
        // (r128 and xx stay even).
        const int pbe = pbx + (pbx & 1);
        const int ppe = pbx > 0 && pby > 0 ? pbe * pby : 0;
        const int q128 = pbe > 0 ? 128 / pbe : 0, r128 = pbe > 0 ? 128 - q128 * pbe : 0;
        int yy = pbe > 0 ? (2 * lane) / pbe : 0, xx = pbe > 0 ? 2 * lane - yy * pbe : 0;
        for (int c0 = 0; c0 < ppe; c0 += 128) {
#if DESC_OPT & 4
            bool in0, in1;
            if (__builtin_expect(__ballot(cheap2(xx, yy, in0, in1)) != 0ull, 0)) {
                float sq, vbx, vby, vbz;
                in0 = window(pxs + xx, pys + yy, z, sq, vbx, vby, vbz);
                in1 = window(pxs + xx + 1, pys + yy, z, sq, vbx, vby, vbz);
            }
#else
            float sq, vbx, vby, vbz;
            bool in0 = window(pxs + xx, pys + yy, z, sq, vbx, vby, vbz);
            bool in1 = window(pxs + xx + 1, pys + yy, z, sq, vbx, vby, vbz);
#endif
            in0 = in0 && c0 + 2 * lane < ppe;
            in1 = in1 && c0 + 2 * lane < ppe && xx + 1 < pbx;
            const int pk0 = (ox + xx) | ((oy + yy) << 10) | ((z - B.zs) << 20);
            const int pk1 = pk0 + 1;
            xx += r128;
            yy += q128;
            if (xx >= pbe) {
                xx -= pbe;
                yy++;
            }
            const unsigned long long m0 = __ballot(in0), m1 = __ballot(in1);
            if ((m0 | m1) == 0ull)
                continue;
            // queue position = the number of accepted voxels at lower positions: both voxels of every lower lane,
            // and this lane's first for its second
            const uint32_t before = (uint32_t)__popcll(m0 & lt_mask) + (uint32_t)__popcll(m1 & lt_mask);
            if (in0)
                queue[(qtail + before) & (DQ - 1)] = pk0;
            if (in1)
                queue[(qtail + before + (in0 ? 1u : 0u)) & (DQ - 1)] = pk1;
            qtail += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
            wave_sync();
            // A full batch leaves the queue as soon as its samples are requested (its packed
            // coordinates travel in ppk), one batch ahead of its binning and commit
            while (qtail - qhead >= 64) {
                if (pend) {
                    float cv[7];
#pragma unroll
                    for (int k = 0; k < 7; k++)
                        cv[k] = pv[k];
                    const int cpk = ppk;
                    batch(64, cv, cpk, qhead, 64);  // the next batch's loads fly during this one
                } else {
                    prefetch(qhead, 64);
                    pend = true;
                }
                qhead += 64;
            }
        }
    }
    {
        int have = pend ? 64 : 0, rest = (int)(qtail - qhead);
        if (!have && rest) {
            prefetch(qhead, rest);
            have = rest;
            rest = 0;
        }
        while (have) {
            float cv[7];
#pragma unroll
            for (int k = 0; k < 7; k++)
                cv[k] = pv[k];
            const int cpk = ppk, cnt = have;
            batch(cnt, cv, cpk, qhead, rest);
            have = rest;
            rest = 0;
        }
    }
#ifdef SIFT3D_AMD_DIAG
    if (lane == 0)
        atomicAdd(&g_desc_voxels, (unsigned long long)diag_voxels);
#endif
    // the last batch
    commit_tail(0);
    commit_tail(1);
    wave_sync();
    // The two half-wave histograms are merged in a fixed order, then normalize_desc -> clamp ->
    // normalize_desc (sift.c:1402-1429, 1514-1526).  The reference sums the 768 squares in
    // double in element order; here every lane sums its 12-13 slots and the 64 partial sums are
    // combined by a fixed butterfly (reproducible; the double sum agrees to ~1e-16 relative).
    const float trunc = 0.2f * 128.0f / 768.0f;                               // sift.c:45
    if (!EXACT) {
        for (int i = lane; i < HIST_USED; i += 64)
            hist[i] = hist[i] + hist[HIST_LDS + i];
        wave_sync();
    }
    if (split) {
        // this part's histogram to memory; the wave that completes the keypoint adds the parts in part order.
        // The parts were written by waves of other compute units and XCDs (whose L2s are not coherent with this
        // one): every access of the hand-over is an agent-scope atomic -- the stores write through, the loads
        // read at the coherence point (`sc1` on each instruction) -- and the stores are complete (vmcnt(0))
        // before the arrival counter is bumped.  NO agent-scope fence: on gfx950 that is a write-back
        // (`buffer_wbl2`) / invalidation (`buffer_inv`) of the XCD's whole L2 -- 170 000 of them cost the kernel
        // 4.5 ms at 512^3 (measured: 31.5 against 27.0 ms).
        float *mine = part + ((size_t)(ki - first) * DPARTS + (size_t)pp) * HIST_USED;
        for (int i = lane; i < HIST_USED; i += 64)
            __hip_atomic_store(mine + i, hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t arrived = 0;
        if (lane == 0)
            arrived = __hip_atomic_fetch_add(done + (ki - first), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
        if (arrived != DPARTS - 1) {
            wave_sync();              // (the histogram is cleared for the next item behind these reads)
            continue;
        }
        const float *all = part + (size_t)(ki - first) * DPARTS * HIST_USED;
        for (int i = lane; i < HIST_USED; i += 64) {
            float v = __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 1; q < DPARTS; q++)
                v = v + __hip_atomic_load(all + (size_t)q * HIST_USED + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hist[i] = v;
        }
        wave_sync();
    }
    for (int pass = 0; pass < 2; pass++) {
        double norm = 0.0;
        if (EXACT) {
            // the reference's own order (hists, then bins: sift.c:1407-1416); every lane adds the same values
            for (int c = 0; c < 64; c++)
#pragma unroll
                for (int b = 0; b < 12; b++) {
                    const float el = hist[c_bin_off[b] + c];
                    norm += (double)el * (double)el;
                }
        } else {
            for (int i = lane; i < HIST_USED; i += 64) {
                const float el = hist[i];                                     // unused slots hold 0
                norm += (double)el * (double)el;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                norm += __shfl_xor(norm, o, 64);
        }
        norm = sqrt(norm) + 2.220446049250313e-16;                            // DBL_EPSILON
        const float inv = (float)(1.0 / norm);                                // 1.0f / norm
        wave_sync();
        for (int i = lane; i < HIST_USED; i += 64) {
            float el = hist[i] * inv;
            if (pass == 0)
                el = el < trunc ? el : trunc;                                 // sift.c:1520
            hist[i] = el;
        }
        wave_sync();
    }
    // (out: the store's page-locked host array; out2: an optional copy that stays in HBM for the matcher)
    for (int i = lane; i < 768; i += 64) {
        const float v = hist[c_bin_off[i % 12] + i / 12];
        out[(size_t)orow * 768 + i] = v;
        if (out2)
            out2[(size_t)orow * 768 + i] = v;
    }
    wave_sync();                  // (the histogram is cleared for the next keypoint behind these reads)
    }
    if (probe) {
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(work - (EXACT ? 0 : 1) + (EXACT ? 4 : 8));
        slot[0] = (unsigned long long)(clock64() - probe_c0);
        slot[1] = (unsigned long long)(wall_clock64() - probe_w0);
    }
}

extern "C" {

// host twin of face_eval's acceptance test (same float expressions; this file is compiled with
// FP contraction off for the host as well)
static bool host_face_pass(const float *f16, int f, const float *r)
{
    const float *A = f16 + f * 16;
    const float px = r[1] * A[5] - r[2] * A[4];
    const float py = r[2] * A[3] - r[0] * A[5];
    const float pz = r[0] * A[4] - r[1] * A[3];
    const float det = A[0] * px + A[1] * py + A[2] * pz;
    if (fabsf(det) < 1.1920928955078125e-06f)
        return false;
    const float di = 1.0f / det;
    const float yb = di * (A[6] * px + A[7] * py + A[8] * pz);
    const float zb = di * (r[0] * A[9] + r[1] * A[10] + r[2] * A[11]);
    const float xb = 1.0f - yb - zb;
    const float kk = A[12] * di;
    return !(xb < -1.1920928955078125e-06f || yb < -1.1920928955078125e-06f ||
             zb < -1.1920928955078125e-06f || kk < 0);
}

int sift3d_hip_set_mesh(const float *faces)
{
    int idx[60], cnt[12];
    float f16[20 * 16];
    memset(f16, 0, sizeof(f16));
    memset(cnt, 0, sizeof(cnt));
    for (int f = 0; f < 20; f++) {
        const float *r = faces + f * SIFT3D_HIP_FACE_FLOATS;
        for (int j = 0; j < 3; j++)
            idx[f * 3 + j] = (int)r[16 + j];
        memcpy(f16 + f * 16, r + 3, sizeof(float) * 13); // e1, e2, t, q, e2.q
        for (int j = 0; j < 3; j++) {
            const int id = idx[f * 3 + j];
            if (id < 0 || id >= 12 || cnt[id] >= 5) {
                snprintf(g_err, sizeof(g_err), "sift3d_hip_set_mesh: malformed face table");
                return SIFT3D_FAILURE;
            }
            cnt[id]++;
        }
    }
    for (int v = 0; v < 12; v++)
        if (cnt[v] != 5) {
            snprintf(g_err, sizeof(g_err), "sift3d_hip_set_mesh: vertex %d has %d faces", v, cnt[v]);
            return SIFT3D_FAILURE;
        }
    // Proper 4-colouring of the vertex graph (backtracking over 12 vertices), then the LDS
    // offset of each vertex's 64-cell block: blocks sorted by colour, 64 floats apart, shifted
    // by {0, 8, 18, 26} per colour -- see HIST_LDS in the kernel section.
    int colour[12], binoff[12];
    {
        bool adj[12][12];
        memset(adj, 0, sizeof(adj));
        for (int f = 0; f < 20; f++)
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++)
                    if (a != b)
                        adj[idx[f * 3 + a]][idx[f * 3 + b]] = true;
        for (int v = 0; v < 12; v++)
            colour[v] = -1;
        int v = 0;
        while (v >= 0 && v < 12) {
            int c = colour[v] + 1;
            for (; c < 4; c++) {
                bool clash = false;
                for (int u = 0; u < v; u++)
                    clash = clash || (adj[v][u] && colour[u] == c);
                if (!clash)
                    break;
            }
            if (c < 4) {
                colour[v++] = c;
            } else {
                colour[v--] = -1;
            }
        }
        if (v < 0) {
            snprintf(g_err, sizeof(g_err), "sift3d_hip_set_mesh: vertex graph is not 4-colourable");
            return SIFT3D_FAILURE;
        }
        static const int shift[4] = { 0, 8, 18, 26 };
        int rank = 0;
        for (int c = 0; c < 4; c++)
            for (int u = 0; u < 12; u++)
                if (colour[u] == c)
                    binoff[u] = 64 * rank++ + shift[c];
    }
    for (int f = 0; f < 20; f++) {
        // slot 13 of the face record: LDS bin offsets of the face's three UNSWAPPED vertex ids
        // (the bins, quirk Q1), 10 bits each
        const int packed = binoff[idx[f * 3]] | (binoff[idx[f * 3 + 1]] << 10) |
                           (binoff[idx[f * 3 + 2]] << 20);
        memcpy(f16 + f * 16 + 13, &packed, sizeof(int));
    }
    // Octant table of icos_guess: the face that holds a point well inside each of the four
    // regions of every sign octant, found with the reference's own acceptance test.
    int oct[32];
    {
        const float g = 1.6180339887f;
        const float rep[4][3] = { { 1.0f, 1.0f, 1.0f },
                                  { g / 3.0f, 0.05f, (2.0f * g + 1.0f) / 3.0f },
                                  { (2.0f * g + 1.0f) / 3.0f, g / 3.0f, 0.05f },
                                  { 0.05f, (2.0f * g + 1.0f) / 3.0f, g / 3.0f } };
        for (int c = 0; c < 4; c++)
            for (int o = 0; o < 8; o++) {
                const float r[3] = { (o & 1) ? -rep[c][0] : rep[c][0], (o & 2) ? -rep[c][1] : rep[c][1],
                                     (o & 4) ? -rep[c][2] : rep[c][2] };
                int hit = -1;
                for (int f = 0; f < 20 && hit < 0; f++)
                    if (host_face_pass(f16, f, r))
                        hit = f;
                if (hit < 0) {
                    snprintf(g_err, sizeof(g_err), "sift3d_hip_set_mesh: no face holds a probe direction");
                    return SIFT3D_FAILURE;
                }
                oct[c * 8 + o] = hit;
            }
    }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_bin_off), binoff, sizeof(binoff)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_face16), f16, sizeof(f16)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_oct_face), oct, sizeof(oct)));
    return SIFT3D_SUCCESS;
}

#ifdef SIFT3D_AMD_DIAG
// diagnostic build only: window voxels committed since the last call
__attribute__((visibility("default"))) unsigned long long sift3d_amd_diag_desc_voxels(void)
{
    unsigned long long v = 0, z = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_desc_voxels), sizeof(v));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_desc_voxels), &z, sizeof(z));
    return v;
}
#endif

size_t sift3d_hip_describe_wlut_floats(int nlevels)
{
    return nlevels > 0 ? (size_t)nlevels * WL_STRIDE + WL_COUNTERS : 0;
}

// workgroups of a descriptor launch: every keypoint's wave when there is no work counter, else no more than
// the device holds at once (four 39.6 KB workgroups per CU), each wave looping over the counter
static unsigned describe_grid(uint32_t count, bool counted)
{
    const unsigned need = (count + DWAVES - 1) / DWAVES;
    if (!counted)
        return need;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus < 1)
            cus = 256;
    }
    const unsigned cap = (unsigned)cus * 4u;
    return need < cap ? need : cap;
}

// scratch of the split windows (k_describe, DPARTS): the parts' histograms, then one arrival counter per keypoint
size_t sift3d_hip_describe_part_bytes(uint32_t n)
{
    return n ? (size_t)n * DPARTS * HIST_USED * sizeof(float) + (size_t)n * sizeof(uint32_t) : 0;
}

// records [0, n_exact) take the reference-order kernel, [n_exact, n) the fast one, its windows in DPARTS parts
// through d_part (sift3d_hip_describe_part_bytes(n - n_exact) bytes; nullptr: a temporary allocation, freed
// behind a stream synchronisation -- the entries that take no scratch)
static int describe_launch(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp, uint32_t n,
                           uint32_t n_exact, float *d_hist, float *d_hist2, const float *d_wlut,
                           void *d_part, void *stream)
{
    if (n_exact > n)
        n_exact = n;
#ifdef SIFT3D_AMD_DIAG
    // diagnostic build only (wrong results): 1 skips the commit, 2 the whole batch -- used by
    // profiles/ scripts to attribute the kernel's time to scan / per-voxel terms / commit
    static int ablate = getenv("SIFT3D_AMD_DESC_ABLATE") ? atoi(getenv("SIFT3D_AMD_DESC_ABLATE")) : 0;
#define DESC_ABLATE_PASS , ablate
#else
#define DESC_ABLATE_PASS
#endif
    // (the work counters sit behind the weight tables and were zeroed by k_desc_wlut)
    uint32_t *work = d_wlut ? reinterpret_cast<uint32_t *>(const_cast<float *>(d_wlut) + (size_t)nlevels * WL_STRIDE)
                            : nullptr;
    if (n_exact) {
        hipLaunchKernelGGL(k_describe<true>, dim3(describe_grid(n_exact, work != nullptr)), dim3(64 * DWAVES), 0,
                           (hipStream_t)stream, d_levels, d_kp, 0u, n_exact, d_hist, d_hist2, d_wlut,
                           work, (float *)nullptr, (uint32_t *)nullptr DESC_ABLATE_PASS);
        LAUNCH_CHECK();
    }
    if (n > n_exact) {
        const uint32_t nf = n - n_exact;
        void *tmp = nullptr;
        if (!d_part) {
            HIPCHK(hipMalloc(&tmp, sift3d_hip_describe_part_bytes(nf)));
            d_part = tmp;
        }
        float *parts = reinterpret_cast<float *>(d_part);
        uint32_t *done = reinterpret_cast<uint32_t *>(parts + (size_t)nf * DPARTS * HIST_USED);
        hipError_t e = hipMemsetAsync(done, 0, (size_t)nf * sizeof(uint32_t), (hipStream_t)stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_describe<false>, dim3(describe_grid(nf * DPARTS, work != nullptr)),
                               dim3(64 * DWAVES), 0, (hipStream_t)stream, d_levels, d_kp, n_exact, n, d_hist,
                               d_hist2, d_wlut, work ? work + 1 : nullptr, parts, done DESC_ABLATE_PASS);
            e = hipGetLastError();
        }
        if (tmp) {
            const hipError_t e2 = hipStreamSynchronize((hipStream_t)stream);
            (void)hipFree(tmp);
            if (e == hipSuccess)
                e = e2;
        }
        if (e != hipSuccess)
            return fail("k_describe", e, __FILE__, __LINE__);
    }
#undef DESC_ABLATE_PASS
    return SIFT3D_SUCCESS;
}

int sift3d_hip_describe(const sift3d_hip_level *d_levels, const sift3d_hip_kp *d_kp, uint32_t n,
                        float *d_hist, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    return describe_launch(d_levels, 0, d_kp, n, 0, d_hist, nullptr, nullptr, nullptr, stream);
}

// (d_part: scratch of sift3d_hip_describe_part_bytes(n - n_exact) bytes, or NULL)
int sift3d_hip_describe_parts(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                              uint32_t n, uint32_t n_exact, float *d_hist, float *d_hist2, float *d_wlut,
                              void *d_part, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    if (!d_wlut || nlevels < 1)
        return describe_launch(d_levels, 0, d_kp, n, n_exact, d_hist, d_hist2, nullptr, d_part, stream);
    hipLaunchKernelGGL(k_desc_wlut, dim3(nlevels), dim3(256), 0, (hipStream_t)stream, d_levels, nlevels,
                       d_wlut);
    return describe_launch(d_levels, nlevels, d_kp, n, n_exact, d_hist, d_hist2, d_wlut, d_part, stream);
}

int sift3d_hip_describe_ex(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                           uint32_t n, uint32_t n_exact, float *d_hist, float *d_hist2, float *d_wlut,
                           void *stream)
{
    return sift3d_hip_describe_parts(d_levels, nlevels, d_kp, n, n_exact, d_hist, d_hist2, d_wlut, nullptr,
                                     stream);
}

// shader cycles and 100 MHz ticks the first wave of the last descriptor launch (fast kernel; exact != 0: the
// reference-order kernel) was alive -- persistent waves: the kernel's duration.  Blocks on `stream`.
int sift3d_hip_describe_clock(const float *d_wlut, int nlevels, int exact, uint64_t *cycles, uint64_t *ticks,
                              void *stream)
{
    if (!d_wlut || nlevels < 1 || !cycles || !ticks)
        return SIFT3D_FAILURE;
    uint64_t v[2] = { 0, 0 };
    const uint32_t *ctr = reinterpret_cast<const uint32_t *>(d_wlut + (size_t)nlevels * WL_STRIDE);
    HIPCHK(hipMemcpyAsync(v, ctr + (exact ? 4 : 8), sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    *cycles = v[0];
    *ticks = v[1];
    return SIFT3D_SUCCESS;
}

int sift3d_hip_describe_wlut2(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                              uint32_t n, float *d_hist, float *d_hist2, float *d_wlut, void *stream)
{
    return sift3d_hip_describe_ex(d_levels, nlevels, d_kp, n, 0, d_hist, d_hist2, d_wlut, stream);
}

int sift3d_hip_describe_wlut(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                             uint32_t n, float *d_hist, float *d_wlut, void *stream)
{
    return sift3d_hip_describe_wlut2(d_levels, nlevels, d_kp, n, d_hist, nullptr, d_wlut, stream);
}

} // extern "C"

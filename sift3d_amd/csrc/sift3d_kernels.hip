// sift3d_kernels.hip -- hand-written gfx950 (CDNA4, MI355X) kernels of the SIFT3D
// detect+describe hot path and their C-ABI launchers (include/sift3d_amd.h).
//
// All stages are memory-bound stencils or per-keypoint window reductions: no MFMA.
// Design rules that matter here (cdna_hip_programming.md / MI355X_MICROARCH.md):
//   * wave = 64 lanes; coalesced 16 B/lane accesses along the unit-stride x axis
//   * the three 1-D Gaussian passes never transpose the volume in HBM: the x pass stages
//     row segments in LDS and slides a register window, the y/z passes sweep along the
//     strided axis with a register ring so every input is loaded once per thread
//   * bit-exact float32 results vs the reference CPU path: tap order d = -hw..+hw,
//     `tap * ((1-frac)*lo + frac*hi)` then `+=`, NO fused multiply-add (the file is
//     compiled with -ffp-contract=off and carries the pragma below)
//   * window reductions (orientation tensor, descriptor histogram) accumulate in the
//     reference's voxel scan order, so sums are bit-identical, not just close
//
// Reference citations are file:line under /root/reference/sift3d/.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "sift3d_kernels_common.h"
#include <cstdlib>
#include "sift3d_math.h"
#include "synth.h"

// ---------------------------------------------------------------------------------------
// error handling / plumbing
// ---------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(const char *what, hipError_t e, const char *file, int line)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    fprintf(stderr, "sift3d_amd: %s\n", g_err);
    return SIFT3D_FAILURE;
}


extern "C" {

const char *sift3d_hip_last_error(void) { return g_err; }

int sift3d_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int sift3d_amd_device_available(void) { return sift3d_hip_device_count() > 0; }

int sift3d_hip_current_device(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess)
        return -1;
    return dev;
}

int sift3d_hip_set_device(int dev)
{
    HIPCHK(hipSetDevice(dev));
    return SIFT3D_SUCCESS;
}

void *sift3d_hip_malloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 4);
    if (e != hipSuccess) {
        fail("hipMalloc", e, __FILE__, __LINE__);
        return nullptr;
    }
    return p;
}

void sift3d_hip_free(void *p)
{
    if (p)
        (void)hipFree(p);
}

void *sift3d_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault);
    if (e != hipSuccess) {
        fail("hipHostMalloc", e, __FILE__, __LINE__);
        return nullptr;
    }
    return p;
}

void *sift3d_hip_host_device_ptr(void *host)
{
    void *dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dev, host, 0);
    if (e != hipSuccess) {
        fail("hipHostGetDevicePointer", e, __FILE__, __LINE__);
        return nullptr;
    }
    return dev;
}

void sift3d_hip_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

int sift3d_hip_memcpy_h2d(void *d, const void *h, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SIFT3D_SUCCESS;
}

int sift3d_hip_memcpy_d2h(void *h, const void *d, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SIFT3D_SUCCESS;
}

int sift3d_hip_memcpy_d2d(void *d, const void *s, size_t bytes, void *stream)
{
    HIPCHK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SIFT3D_SUCCESS;
}

int sift3d_hip_memcpy2d_d2h(void *h, size_t dpitch, const void *d, size_t spitch, size_t width,
                            size_t height, void *stream)
{
    HIPCHK(hipMemcpy2DAsync(h, dpitch, d, spitch, width, height, hipMemcpyDeviceToHost,
                            (hipStream_t)stream));
    return SIFT3D_SUCCESS;
}

int sift3d_hip_stream_wait_event(void *stream, void *ev)
{
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return SIFT3D_SUCCESS;
}

int sift3d_hip_memset(void *d, int byte, size_t bytes, void *stream)
{
    HIPCHK(hipMemsetAsync(d, byte, bytes, (hipStream_t)stream));
    return SIFT3D_SUCCESS;
}

void *sift3d_hip_stream_create(void)
{
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        fail("hipStreamCreate", e, __FILE__, __LINE__);
        return nullptr;
    }
    return (void *)s;
}

// a stream whose kernels are dispatched ahead of those of ordinary streams (short, latency-bound
// work that runs beside device-filling kernels)
void *sift3d_hip_stream_create_high(void)
{
    hipStream_t s = nullptr;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi);
    if (e != hipSuccess) {
        fail("hipStreamCreateWithPriority", e, __FILE__, __LINE__);
        return nullptr;
    }
    return (void *)s;
}

void sift3d_hip_stream_destroy(void *s)
{
    if (s)
        (void)hipStreamDestroy((hipStream_t)s);
}

int sift3d_hip_stream_sync(void *s)
{
    HIPCHK(hipStreamSynchronize((hipStream_t)s));
    return SIFT3D_SUCCESS;
}

void *sift3d_hip_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
        return nullptr;
    return (void *)e;
}

void sift3d_hip_event_destroy(void *e)
{
    if (e)
        (void)hipEventDestroy((hipEvent_t)e);
}

int sift3d_hip_event_record(void *e, void *s)
{
    HIPCHK(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
    return SIFT3D_SUCCESS;
}

double sift3d_hip_event_elapsed_ms(void *a, void *b)
{
    float ms = 0.f;
    if (hipEventSynchronize((hipEvent_t)b) != hipSuccess)
        return -1.0;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess)
        return -1.0;
    return (double)ms;
}

} // extern "C"

// ---------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

// max over the workgroup (256 threads) of non-negative floats, then ONE atomic per workgroup.
// Same-address atomics serialise at the memory side (~20 ns each): one per wave cost the DoG
// kernels 1.5 ms at 512^3.  NM maxima at once; non-negative floats order like their bit patterns.
template <int NM>
__device__ __forceinline__ void block_max_atomic(const float *m, unsigned *__restrict__ out)
{
    __shared__ float red[NM][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NM; k++) {
        const float w = wave_max(m[k]);
        if (lane == 0)
            red[k][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < NM) {
        const float w = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]),
                              fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
        if (w > 0.0f)
            atomicMax(out + threadIdx.x, __float_as_uint(w));
    }
}


// ---------------------------------------------------------------------------------------
// im_max_abs / im_scale  (imutil.c:681-713)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ src, size_t n,
                                                unsigned *__restrict__ out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    float m = 0.0f;
    // four independent 16-byte loads in flight per thread and iteration
    size_t i = tid;
    for (; i + 3 * nthr < n4; i += 4 * nthr) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            v[k] = ld4(src + 4 * (i + k * nthr));
#pragma unroll
        for (int k = 0; k < 4; k++)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
    }
    for (; i < n4; i += nthr) {
        const float4 v = ld4(src + 4 * i);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (size_t j = 4 * n4 + tid; j < n; j += nthr)
        m = fmaxf(m, fabsf(src[j]));
    block_max_atomic<1>(&m, out);
}

__global__ __launch_bounds__(256) void k_scale(const float *__restrict__ src,
                                               float *__restrict__ dst, size_t n,
                                               const float *__restrict__ d_max)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    const float mx = *d_max;
    if (mx == 0.0f) { // imutil.c:706-707
        for (size_t i = tid; i < n4; i += nthr)
            st4(dst + 4 * i, ld4(src + 4 * i));
        for (size_t i = 4 * n4 + tid; i < n; i += nthr)
            dst[i] = src[i];
        return;
    }
    for (size_t i = tid; i < n4; i += nthr) {
        float4 v = ld4(src + 4 * i);
        v.x = v.x / mx; // imutil.c:711, IEEE division
        v.y = v.y / mx;
        v.z = v.z / mx;
        v.w = v.w / mx;
        st4(dst + 4 * i, v);
    }
    for (size_t i = 4 * n4 + tid; i < n; i += nthr)
        dst[i] = src[i] / mx;
}

// ---------------------------------------------------------------------------------------
// 1-D interpolating FIR  (convolve_sep_gen, imutil.c:742-861)
// ---------------------------------------------------------------------------------------
// One output sample, the literal arithmetic of the reference.  `line` points at the
// LOCAL index 0 of the 1-D line; g is the GLOBAL coordinate of the output.  Samples are
// clamped into the local buffer for memory safety only (a correct call never needs it,
// except for the weight-0 `hi` read one past the row -- SURVEY.md A.2).
template <int HWT>
__device__ __forceinline__ float fir_literal_t(const float *__restrict__ line, size_t stride,
                                               int g, int n_glob, int off, int n_loc,
                                               const float *__restrict__ taps, int hw_rt,
                                               float uf, int uhw)
{
    // HWT > 0: compile-time half width -> the tap loop is fully unrolled and its 2*(2*HWT+1)
    // loads are issued back to back (a rolled loop serialises one memory latency per tap)
    const int hw = HWT > 0 ? HWT : hw_rt;
    const int dim_end = n_glob - 1;                               // :753
    const bool interior = g >= uhw && g <= n_glob - 2 - uhw;      // :762-763, :829
    float acc = 0.0f;                                             // im_zero, :777
    float coord = (float)g;
#pragma unroll
    for (int d = -hw; d <= hw; d++) {
        const float tap = taps[d + hw];
        const float step = (float)d * uf;                         // :808 / :837
        float c;
        if (interior) {
            coord -= step;                                        // :811
            c = coord;
        } else {
            c = (float)g - step;                                  // :835,:840
            if ((int)c < 0)                                       // :843
                c = -c;
            else if ((int)c >= dim_end)                           // :846
                c = 2.0f * (float)dim_end - c - 0.1f;             // :847-848
        }
        const int lo = (int)c;                                    // trunc, :783
        const float frac = c - (float)lo;                         // :788
        const int llo = clampi(lo - off, 0, n_loc - 1);
        const int lhi = clampi(lo + 1 - off, 0, n_loc - 1);
        const float a = line[(size_t)llo * stride];
        const float b = line[(size_t)lhi * stride];
        acc += tap * ((1.0f - frac) * a + frac * b);              // :791-795
        if (interior)
            coord += step;                                        // :817
    }
    return acc;
}

__device__ __forceinline__ float fir_literal(const float *__restrict__ line, size_t stride,
                                             int g, int n_glob, int off, int n_loc,
                                             const float *__restrict__ taps, int hw,
                                             float uf, int uhw)
{
    return fir_literal_t<0>(line, stride, g, n_glob, off, n_loc, taps, hw, uf, uhw);
}

// literal kernel: one thread per output voxel, any axis, any unit factor
__global__ __launch_bounds__(256) void k_fir_literal(FirParams P, FirTaps T)
{
    const size_t plane = (size_t)P.nx * P.ny;
    const size_t total = plane * (size_t)(P.z_hi - P.z_lo);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total)
        return;
    const int z = P.z_lo + (int)(i / plane);
    const size_t r = i % plane;
    const int y = (int)(r / P.nx);
    const int x = (int)(r % P.nx);
    const size_t idx = (size_t)z * plane + r;
    // per-axis geometry by selects (no control flow): coordinate along the axis, stride,
    // local extent; n_glob / off were resolved by the launcher
    const int p = P.axis == 0 ? x : (P.axis == 1 ? y : z);
    const size_t stride = P.axis == 0 ? (size_t)1 : (P.axis == 1 ? (size_t)P.nx : plane);
    const int n_loc = P.axis == 0 ? P.nx : (P.axis == 1 ? P.ny : P.nz);
    const float *line = P.src + (idx - (size_t)p * stride);
    P.dst[idx] = fir_literal(line, stride, p + P.off, P.n_glob, P.off, n_loc, T.k, P.hw, P.uf,
                             P.uhw);
}

// Filters WIDER than the kernarg tap table (SIFT3D_HIP_MAX_TAPS; the reference accepts any sigma0 >= 0,
// sift.c:553-565 -- half width ceil(3 sigma), imutil.c:1275-1277): the literal kernel in chunks of taps.  A
// launch adds taps d in [d_lo, d_hi) to the running sum of every output voxel, which travels between the
// launches in dst (first chunk: 0); the reference adds the taps of a voxel in ascending d into one float
// accumulator (imutil.c:791-795), so the chunks reproduce its sum bit for bit.  The interior branch's
// coordinate round trip (imutil.c:811-817: coord -= step ... coord += step, a state carried from tap to tap)
// is replayed from d = -hw in every launch -- arithmetic only -- so that its state at a chunk's first tap is
// the reference's.
__global__ __launch_bounds__(256) void k_fir_literal_chunk(FirParams P, FirTaps T, int d_lo, int d_hi)
{
    const size_t plane = (size_t)P.nx * P.ny;
    const size_t total = plane * (size_t)(P.z_hi - P.z_lo);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total)
        return;
    const int z = P.z_lo + (int)(i / plane);
    const size_t r = i % plane;
    const int y = (int)(r / P.nx);
    const int x = (int)(r % P.nx);
    const size_t idx = (size_t)z * plane + r;
    const int p = P.axis == 0 ? x : (P.axis == 1 ? y : z);
    const size_t stride = P.axis == 0 ? (size_t)1 : (P.axis == 1 ? (size_t)P.nx : plane);
    const int n_loc = P.axis == 0 ? P.nx : (P.axis == 1 ? P.ny : P.nz);
    const float *__restrict__ line = P.src + (idx - (size_t)p * stride);
    const int g = p + P.off, hw = P.hw;
    const int dim_end = P.n_glob - 1;                             // :753
    const bool interior = g >= P.uhw && g <= P.n_glob - 2 - P.uhw;  // :762-763, :829
    float acc = d_lo > -hw ? P.dst[idx] : 0.0f;                   // im_zero, :777
    float coord = (float)g;
    for (int d = -hw; d < d_hi; d++) {
        const float step = (float)d * P.uf;                       // :808 / :837
        float c;
        if (interior) {
            coord -= step;                                        // :811
            c = coord;
        } else {
            c = (float)g - step;                                  // :835,:840
            if ((int)c < 0)                                       // :843
                c = -c;
            else if ((int)c >= dim_end)                           // :846
                c = 2.0f * (float)dim_end - c - 0.1f;             // :847-848
        }
        if (d >= d_lo) {
            const int lo = (int)c;                                // trunc, :783
            const float frac = c - (float)lo;                     // :788
            const int llo = clampi(lo - P.off, 0, n_loc - 1);
            const int lhi = clampi(lo + 1 - P.off, 0, n_loc - 1);
            const float a = line[(size_t)llo * stride];
            const float b = line[(size_t)lhi * stride];
            acc += T.k[d - d_lo] * ((1.0f - frac) * a + frac * b);    // :791-795
        }
        if (interior)
            coord += step;                                        // :817
    }
    P.dst[idx] = acc;
}

// ---- unit factor 1 (octave 0): edges as a staging transformation ----------------------------
// With uf == 1 every sample coordinate is an integer, and the reference's edge rules
// (imutil.c:842-850) depend only on that integer i = x - d, not on (x, d) separately:
//     i < 0          -> sample src[-i]                       (frac == 0)
//     0 <= i < end   -> sample src[i]                        (frac == 0)
//     i >= end = n-1 -> c' = 2*end - i - 0.1f, a fixed lerp of two samples near the end
//                       (including i == end itself, quirk Q4)
// and interior outputs never reach i >= end.  So the whole pass is  out[x] = sum_d k[d]*E[x-d]
// over an EXTENDED line E with reflected samples on the low side and pre-interpolated
// "virtual" samples v_m = w0_m*src[lo_m] + w1_m*src[lo_m+1] (m = i - end) on the high side.
// tap*((1-frac)*lo + frac*hi) is evaluated exactly as written -- the inner expression is
// v_m -- and for frac == 0 it equals tap*lo for finite data.  Requires n >= 2*hw + 2 (no
// double mirroring); shorter axes take the literal kernel.  The (lo_m, w0_m, w1_m) table is
// computed on the host with the reference's float expressions.
// E[i] for one line; i and the table are GLOBAL coordinates, the line pointer addresses local
// index 0 and holds global indices [off, off + n_loc).
__device__ __forceinline__ float ext_sample(const float *__restrict__ line, size_t stride, int i,
                                            int end, int off, int n_loc, int hw,
                                            const EdgeTab &E)
{
    if (i < 0) {
        return -i <= hw ? line[(size_t)clampi(-i - off, 0, n_loc - 1) * stride] : 0.0f;
    } else if (i >= end) {
        const int m = i - end;
        if (m > hw)
            return 0.0f;
        const int lo = E.lo[m];
        const float a = line[(size_t)clampi(lo - off, 0, n_loc - 1) * stride];
        const float b = line[(size_t)clampi(lo + 1 - off, 0, n_loc - 1) * stride];
        return E.w0[m] * a + E.w1[m] * b;
    }
    return line[(size_t)clampi(i - off, 0, n_loc - 1) * stride];
}

// x pass: each wave walks XROWS consecutive rows of one 512-output segment, software
// pipelined: the 16-byte global loads of row r+1 are in flight while row r is computed from
// LDS (double-buffered, wave-private, so no workgroup barrier).  The extended segment
// (+8-float halos) is staged with coalesced loads; each lane pulls two 20-float windows into
// registers (lane stride 16 B: conflict-free ds_read_b128) and emits 2 x 4 outputs with fully
// coalesced 16-byte stores, in the reference's tap order.  No edge code in the FIR itself.
constexpr int XROWS = 4;

template <int HW>
__global__ __launch_bounds__(256) void k_fir_x_u1(FirParams P, FirTaps T, EdgeTab E)
{
    constexpr int SEG = 512, HALO = 8, L = SEG + 2 * HALO, NV = (L / 4 + 63) / 64; // NV = 3
    static_assert(HW <= HALO, "halo too small");
    __shared__ __attribute__((aligned(16))) float lds[4][2][L];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nrows = P.ny * (P.z_hi - P.z_lo);
    // Blocks walk the volume from its LAST rows to its first: the producer of `src` (the
    // previous blur's z sweep) finished there, so those planes are still in the 256 MB
    // Infinity Cache; the consumer of `dst` starts at plane 0, where this kernel ends.
    const int row0 = ((gridDim.y - 1 - blockIdx.y) * 4 + wave) * XROWS;
    if (row0 >= nrows)
        return;
    const int nr = min(XROWS, nrows - row0);
    const int x0 = blockIdx.x * SEG;
    const int nx = P.nx, end = nx - 1;
    const bool vec_ok = (nx & 3) == 0 && ((((uintptr_t)P.src | (uintptr_t)P.dst) & 15) == 0);
    // rows of the plane range are contiguous in memory: row r starts at base + r*nx
    const size_t base = (size_t)P.z_lo * P.ny * nx;
    const int egi = lane < 8 ? -1 - lane : end + (lane - 8);   // edge sample this lane provides
    const int epos = egi - (x0 - HALO);
    const bool has_edge = lane < 17 && epos >= 0 && epos < L;

    // im_scale folded in (P.scale_max): the quotient v / max of imutil.c:711 is formed when a row is committed
    // to LDS -- not when it is requested: nothing may depend on a load in flight -- and the edge samples
    // are built from scaled samples, as the reference builds them from the scaled image
    const float smax = P.scale_max ? *P.scale_max : 0.0f;
    const bool scaled = smax != 0.0f;                          // (max == 0: im_scale leaves the image alone)
    float4 v[NV];
    float ve = 0.0f, ve2 = 0.0f, ew0 = 1.0f, ew1 = 0.0f;
    auto fetch = [&](int r) {
        const float *__restrict__ s = P.src + base + (size_t)(row0 + r) * nx;
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int i = lane + 64 * k;
            const int gx = x0 - HALO + 4 * i;
            if (i < L / 4) {
                if (vec_ok && gx >= 0 && gx + 3 < nx) {
                    v[k] = ld4(s + gx);
                } else {
                    v[k].x = (gx >= 0 && gx < nx) ? s[gx] : 0.0f;
                    v[k].y = (gx + 1 >= 0 && gx + 1 < nx) ? s[gx + 1] : 0.0f;
                    v[k].z = (gx + 2 >= 0 && gx + 2 < nx) ? s[gx + 2] : 0.0f;
                    v[k].w = (gx + 3 >= 0 && gx + 3 < nx) ? s[gx + 3] : 0.0f;
                }
            }
        }
        // edge samples of the extended line, one per lane: E[-1..-8] and E[end..end+8] (the two samples and
        // weights of ext_sample's cases; combined in commit)
        if (has_edge) {
            ve = ve2 = 0.0f;
            ew0 = 1.0f;
            ew1 = 0.0f;
            if (egi < 0) {
                if (-egi <= HW)
                    ve = s[clampi(-egi, 0, nx - 1)];
            } else if (egi - end <= HW) {
                const int m = egi - end, lo = E.lo[m];
                ve = s[clampi(lo, 0, nx - 1)];
                ve2 = s[clampi(lo + 1, 0, nx - 1)];
                ew0 = E.w0[m];
                ew1 = E.w1[m];
            }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const int i = lane + 64 * k;
            if (i < L / 4) {
                float4 q = v[k];
                if (scaled) {
                    q.x = q.x / smax; q.y = q.y / smax; q.z = q.z / smax; q.w = q.w / smax;   // imutil.c:711
                }
                *reinterpret_cast<float4 *>(&lds[wave][buf][4 * i]) = q;
            }
        }
        // DS writes of a wave retire in order: the edge samples overwrite the bulk values
        if (has_edge) {
            const float a = scaled ? ve / smax : ve, b = scaled ? ve2 / smax : ve2;
            // (ext_sample's cases: a lone sample is 1 * a + 0 * b = a exactly; beyond the taps' reach 0)
            lds[wave][buf][epos] = egi >= end ? ew0 * a + ew1 * b : a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };

    fetch(0);
    commit(0);
    for (int r = 0; r < nr; r++) {
        const int buf = r & 1;
        if (r + 1 < nr)
            fetch(r + 1);                     // in flight during the FIR below
        float *__restrict__ d = P.dst + base + (size_t)(row0 + r) * nx;
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int lb = grp * 256 + lane * 4;   // first output of this lane in the segment
            const int xb = x0 + lb;
            if (xb < nx) {
                float w[4 + 2 * HALO];
#pragma unroll
                for (int i = 0; i < (4 + 2 * HALO) / 4; i++) {
                    const float4 q = *reinterpret_cast<const float4 *>(&lds[wave][buf][lb + 4 * i]);
                    w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
                }
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float acc = 0.0f;
#pragma unroll
                    for (int dd = -HW; dd <= HW; dd++)
                        acc += T.k[dd + HW] * w[HALO + k - dd];   // E[x - d], d ascending
                    o[k] = acc;
                }
                if (vec_ok && xb + 4 <= nx) {
                    st4(d + xb, make_float4(o[0], o[1], o[2], o[3]));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (xb + k < nx)
                            d[xb + k] = o[k];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (r + 1 < nr)
            commit(buf ^ 1);
    }
}

// The same pass for rows of whole 512-float segments (nx % 512 == 0, 16-byte aligned: every volume of the
// BASELINE configurations), with TWO rows in flight per wave.  k_fir_x_u1 above is bound by latency x
// occupancy, not by bytes (measured: 0.228-0.236 ms at 6 waves per SIMD, 0.275-0.280 ms at 5, for the same
// 8 B/voxel; one row = 2 KB in flight per wave): here the loads of rows r + 1 AND r + 2 fly while row r is
// filtered.  All vector-memory instructions of the loop body are issued unconditionally (clamped addresses,
// values masked when they are committed to LDS; the tail re-requests the last row), so the body is
// straight-line code and the compiler's s_waitcnt vmcnt counts are exact -- a load or store behind a
// lane- or wave-dependent branch would make it wait for the younger row as well.
constexpr int XROWS_F = 8;

template <int HW, bool SCALED>
__global__ __launch_bounds__(256) void k_fir_x_u1f(FirParams P, FirTaps T, EdgeTab E)
{
    constexpr int SEG = 512, HALO = 8, L = SEG + 2 * HALO;
    static_assert(HW <= HALO, "halo too small");
    __shared__ __attribute__((aligned(16))) float lds[4][2][L];
    // where the lanes that have no third quad / no edge sample to commit put theirs: every commit is then
    // free of branches, and no load is left "maybe consumed" at the loop's back edge
    __shared__ __attribute__((aligned(16))) float lds_sink[4][64 * 4];
    // (wave-uniform by construction; said so, or the row loop's exits become lane-divergent control flow)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nrows = P.ny * (P.z_hi - P.z_lo);
    // (back to front, as k_fir_x_u1: the producer of `src` ended at the last planes)
    const int row0 = ((gridDim.y - 1 - blockIdx.y) * 4 + wave) * XROWS_F;
    if (row0 >= nrows)
        return;
    const int nr = min(XROWS_F, nrows - row0);
    const int x0 = blockIdx.x * SEG;
    const int nx = P.nx, end = nx - 1;
    const size_t base = (size_t)P.z_lo * P.ny * nx;
    // quads of the extended segment this lane stages: i = lane, lane + 64, lane + 128 (< L / 4 = 132)
    const int g0 = x0 - HALO + 4 * lane, g1 = g0 + 256, g2 = g0 + 512;
    const bool in0 = g0 >= 0, use2 = lane < 4, in2 = use2 && g2 + 3 < nx;
    const int c0 = in0 ? g0 : 0, c2 = in2 ? g2 : nx - 4;
    // edge samples of the extended line, one per lane (ext_sample's cases)
    const int egi = lane < 8 ? -1 - lane : end + (lane - 8);
    const int epos = egi - (x0 - HALO);
    const bool has_edge = lane < 17 && epos >= 0 && epos < L;
    int eo0 = 0, eo1 = 0;
    float ew0 = 0.0f, ew1 = 0.0f;                 // (beyond the taps' reach: 0)
    if (lane < 17) {
        if (egi < 0) {
            if (-egi <= HW) {
                eo0 = clampi(-egi, 0, nx - 1);
                ew0 = 1.0f;
            }
        } else if (egi - end <= HW) {
            // (selects, not E.lo[m]: a lane-dependent index into a kernel argument would go through scratch)
            const int m = egi - end;
            int lo = 0;
#pragma unroll
            for (int mm = 0; mm <= HW; mm++)
                if (mm == m) {
                    lo = E.lo[mm];
                    ew0 = E.w0[mm];
                    ew1 = E.w1[mm];
                }
            eo0 = clampi(lo, 0, nx - 1);
            eo1 = clampi(lo + 1, 0, nx - 1);
        }
    }
    // SCALED: im_scale folded in (imutil.c:698-713); a maximum of 0 leaves the image alone (imutil.c:706-707):
    // every sample is then 0 and 0 / 1 = 0 exactly
    float smax = 1.0f;
    if (SCALED) {
        smax = *P.scale_max;
        smax = smax != 0.0f ? smax : 1.0f;
    }

    struct RowRegs {
        float4 v0, v1, v2;
        float e0, e1;
    };
    auto fetch = [&](RowRegs &R, int r) {
        const float *__restrict__ s = P.src + base + (size_t)(row0 + min(r, nr - 1)) * nx;
        R.v0 = ld4(s + c0);
        R.v1 = ld4(s + g1);
        R.v2 = ld4(s + c2);
        R.e0 = s[eo0];
        R.e1 = s[eo1];
    };
    auto commit = [&](const RowRegs &R, int buf) {
        float4 q0 = R.v0, q1 = R.v1, q2 = R.v2;
        float a = R.e0, b = R.e1;
        if (SCALED) {                                           // imutil.c:711, formed when committed
            q0.x = q0.x / smax; q0.y = q0.y / smax; q0.z = q0.z / smax; q0.w = q0.w / smax;
            q1.x = q1.x / smax; q1.y = q1.y / smax; q1.z = q1.z / smax; q1.w = q1.w / smax;
            q2.x = q2.x / smax; q2.y = q2.y / smax; q2.z = q2.z / smax; q2.w = q2.w / smax;
            a = a / smax;
            b = b / smax;
        }
        // (component selects: a select between two float4 objects would go through scratch)
        q0.x = in0 ? q0.x : 0.0f; q0.y = in0 ? q0.y : 0.0f; q0.z = in0 ? q0.z : 0.0f; q0.w = in0 ? q0.w : 0.0f;
        q2.x = in2 ? q2.x : 0.0f; q2.y = in2 ? q2.y : 0.0f; q2.z = in2 ? q2.z : 0.0f; q2.w = in2 ? q2.w : 0.0f;
        float *const row = lds[wave][buf], *const sink = &lds_sink[wave][4 * lane];
        *reinterpret_cast<float4 *>(row + 4 * lane) = q0;
        *reinterpret_cast<float4 *>(row + 4 * (lane + 64)) = q1;
        *reinterpret_cast<float4 *>(use2 ? row + 4 * (lane + 128) : sink) = q2;
        // DS writes of a wave retire in order: the edge samples overwrite the bulk values
        // (ext_sample's three cases in one expression, so that nothing here branches: a lone sample has
        // weights (1, 0): 1 * a + 0 * b = a for finite data -- up to the sign of a zero, which no later
        // comparison or non-zero sum can see --, a sample beyond the taps' reach (0, 0))
        *(has_edge ? row + epos : sink) = ew0 * a + ew1 * b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    auto compute = [&](int r, int buf) {
        float *__restrict__ d = P.dst + base + (size_t)(row0 + r) * nx;
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int lb = grp * 256 + lane * 4;   // first output of this lane in the segment
            float w[4 + 2 * HALO];
#pragma unroll
            for (int i = 0; i < (4 + 2 * HALO) / 4; i++) {
                const float4 q = *reinterpret_cast<const float4 *>(&lds[wave][buf][lb + 4 * i]);
                w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w;
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float acc = 0.0f;
#pragma unroll
                for (int dd = -HW; dd <= HW; dd++)
                    acc += T.k[dd + HW] * w[HALO + k - dd];   // E[x - d], d ascending
                o[k] = acc;
            }
            st4(d + x0 + lb, make_float4(o[0], o[1], o[2], o[3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };

    // The rows of this wave.  Called with the constant XROWS_F (all but the volume's last rows) the loop is
    // unrolled completely: one basic block, in which the compiler's vmcnt counts are exact (at a loop header
    // it merges the states of the entry and the back edge and waits for the younger row too).
    auto rows = [&](const int n) __attribute__((always_inline)) {
        RowRegs A, B;
        fetch(A, 0);
        fetch(B, 1);
        commit(A, 0);
#pragma unroll
        for (int r = 0; r < XROWS_F; r += 2) {
            fetch(A, r + 2);                  // rows r + 1 (B) and r + 2 (A) in flight during the FIR below
            compute(r, 0);
            if (r + 1 >= n)
                break;
            commit(B, 1);
            fetch(B, r + 3);
            compute(r + 1, 1);
            if (r + 2 >= n)
                break;
            commit(A, 0);
        }
    };
    if (nr == XROWS_F)
        rows(XROWS_F);
    else
        rows(nr);
}

// ---- y / z pass, unit factor 1 ----------------------------------------------------------
// Each thread owns V adjacent x (one 16-byte quad for V=4) and sweeps `ts` outputs along
// the strided axis, keeping the 2*HW+1 most recent rows of the EXTENDED line in a register
// ring: every input is loaded once per thread, loads are coalesced along x, nothing is
// transposed.  Whether a ring row is plain, reflected or virtual depends only on the sweep
// coordinate, which is wave-uniform, so the edge rows cost no divergence.
struct SweepGeom {
    int ncols;          // number of V-wide columns
    int cols_inner;     // columns per contiguous run (row for the y pass, plane for z)
    size_t outer_stride;// floats between runs (plane for the y pass)
    int outer_lo;       // first run (z_lo for the y pass)
    size_t stride;      // floats between consecutive samples along the sweep axis
    int n_loc;          // local extent of the sweep axis
    int out_lo, out_hi; // local output range along the sweep axis
};

// row r (LOCAL index, may be outside [0, n_loc)) of the extended line
template <int V>
__device__ __forceinline__ typename Vec<V>::T ext_row(const float *__restrict__ s, size_t stride,
                                                      int r, int off, int end, int nl1, int hw,
                                                      const EdgeTab &E)
{
    const int i = r + off; // global, wave-uniform
    if (i < 0) {
        return Vec<V>::ld(s + (size_t)clampi(-i - off, 0, nl1) * stride);
    } else if (i >= end) {
        const int m = i - end;
        if (m > hw)
            return Vec<V>::zero();
        const int lo = E.lo[m] - off;
        return Vec<V>::lerp(E.w0[m], Vec<V>::ld(s + (size_t)clampi(lo, 0, nl1) * stride), E.w1[m],
                            Vec<V>::ld(s + (size_t)clampi(lo + 1, 0, nl1) * stride));
    }
    return Vec<V>::ld(s + (size_t)clampi(r, 0, nl1) * stride);
}

template <int HW, int V>
__global__ __launch_bounds__(256) void k_fir_sweep_u1(FirParams P, SweepGeom G, FirTaps T, EdgeTab E)
{
    typedef typename Vec<V>::T vec;
    constexpr int W = 2 * HW + 1;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= G.ncols)
        return;
    const int p0 = G.out_lo + blockIdx.y * P.ts;
    const int p1 = min(p0 + P.ts, G.out_hi);
    const size_t base = (size_t)(G.outer_lo + col / G.cols_inner) * G.outer_stride +
                        (size_t)(col % G.cols_inner) * V;
    const float *__restrict__ s = P.src + base;
    float *__restrict__ d = P.dst + base;
    const int nl1 = G.n_loc - 1;
    const int off = P.off, end = P.n_glob - 1;

    vec ring[W];
#pragma unroll
    for (int i = 0; i < 2 * HW; i++)
        ring[i] = ext_row<V>(s, G.stride, p0 - HW + i, off, end, nl1, HW, E);

#pragma unroll 1
    for (int p = p0; p < p1; p += W) {
#pragma unroll
        for (int j = 0; j < W; j++) {
            const int q = p + j;
            ring[(j + 2 * HW) % W] = ext_row<V>(s, G.stride, q + HW, off, end, nl1, HW, E);
            if (q < p1) {
                vec acc = Vec<V>::zero();
#pragma unroll
                for (int dd = -HW; dd <= HW; dd++)
                    Vec<V>::mac(acc, T.k[dd + HW], ring[(j + HW - dd) % W]);
                Vec<V>::st(d + (size_t)q * G.stride, acc);
            }
        }
    }
}

// ---- dyadic unit factors (octaves >= 1): per-tap constant (offset, frac) -------------------
// For uf = 2^-k the sample coordinate g - d*uf is exact in float, so every interior output
// uses the same per-tap integer offset and interpolation weights (computed on the host with
// the reference's float expressions).  One thread per V-wide column and output row; the
// arithmetic per tap is the literal tap*((1-frac)*lo + frac*hi).
struct DyadTaps {
    float k[SIFT3D_HIP_MAX_TAPS];
    float w0[SIFT3D_HIP_MAX_TAPS]; // 1 - frac
    float w1[SIFT3D_HIP_MAX_TAPS]; // frac
    int off[SIFT3D_HIP_MAX_TAPS];  // lo - g
};

template <int HW, int V>
__global__ __launch_bounds__(256) void k_fir_sweep_dyad(FirParams P, SweepGeom G, DyadTaps T)
{
    typedef typename Vec<V>::T vec;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= G.ncols)
        return;
    const int q = G.out_lo + blockIdx.y;
    const size_t base = (size_t)(G.outer_lo + col / G.cols_inner) * G.outer_stride +
                        (size_t)(col % G.cols_inner) * V;
    const float *__restrict__ s = P.src + base;
    float *__restrict__ d = P.dst + base;
    const int g = q + P.off;
    const int nl1 = G.n_loc - 1;
    const int W = HW > 0 ? 2 * HW + 1 : 2 * P.hw + 1;   // HW == 0: run-time width
    vec acc = Vec<V>::zero();
    if (g >= P.uhw && g <= P.n_glob - 2 - P.uhw) {
#pragma unroll
        for (int t = 0; t < W; t++) {
            const int lo = clampi(q + T.off[t], 0, nl1);
            const int hi = clampi(q + T.off[t] + 1, 0, nl1);
            const vec a = Vec<V>::ld(s + (size_t)lo * G.stride);
            const vec b = Vec<V>::ld(s + (size_t)hi * G.stride);
            const float tap = T.k[t], w0 = T.w0[t], w1 = T.w1[t];
            const float *af = reinterpret_cast<const float *>(&a);
            const float *bf = reinterpret_cast<const float *>(&b);
            float *cf = reinterpret_cast<float *>(&acc);
#pragma unroll
            for (int v = 0; v < V; v++)
                cf[v] += tap * (w0 * af[v] + w1 * bf[v]);
        }
    } else {
        float *cf = reinterpret_cast<float *>(&acc);
#pragma unroll
        for (int v = 0; v < V; v++)
            cf[v] = fir_literal_t<HW>(s + v, G.stride, g, P.n_glob, P.off, G.n_loc, T.k, P.hw, P.uf,
                                      P.uhw);
    }
    Vec<V>::st(d + (size_t)q * G.stride, acc);
}

// x pass for dyadic unit factors: one thread per output voxel
template <int HW>
__global__ __launch_bounds__(256) void k_fir_x_dyad(FirParams P, DyadTaps T)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= P.nx)
        return;
    const int row = blockIdx.y; // (y, z) pair inside the plane range
    const size_t rowoff = ((size_t)(P.z_lo + row / P.ny) * P.ny + (row % P.ny)) * P.nx;
    const float *__restrict__ s = P.src + rowoff;
    const int nl1 = P.nx - 1;
    const int W = HW > 0 ? 2 * HW + 1 : 2 * P.hw + 1;
    float acc = 0.0f;
    if (x >= P.uhw && x <= P.nx - 2 - P.uhw) {
#pragma unroll
        for (int t = 0; t < W; t++) {
            const float a = s[clampi(x + T.off[t], 0, nl1)];
            const float b = s[clampi(x + T.off[t] + 1, 0, nl1)];
            acc += T.k[t] * (T.w0[t] * a + T.w1[t] * b);
        }
    } else {
        acc = fir_literal_t<HW>(s, 1, x, P.nx, 0, P.nx, T.k, P.hw, P.uf, P.uhw);
    }
    P.dst[rowoff + x] = acc;
}

// ---- tap spacings 1/2 and 1/4 (octaves 1 and 2): register-resident source window -----------
// For uf = 2^-S the interior sample of tap d sits at g + off_d + frac_d with the COMPILE-TIME
// constants off_d = floor(-d / 2^S) and frac_d = (-d mod 2^S) / 2^S (exactly what the
// reference's float expressions give, imutil.c:783-788, since g - d*uf is exact).  Source rows
// g-R .. g+R+1 (R = ceil(HW / 2^S)) are kept in a register ring (y/z sweeps) or a register
// window filled from LDS (x), so each input is loaded once per thread, and every term is the
// literal tap*((1-frac)*lo + frac*hi) (tap*lo when frac == 0).  Outputs classified "boundary"
// by the reference (g < uhw or g > n-2-uhw) take the literal mirror arithmetic: wave-uniform
// rows in the sweeps; for the x pass a separate edge kernel overwrites the few columns.
template <int S> __host__ __device__ constexpr int dy_off(int d) { return (-d) >> S; }
template <int S> __host__ __device__ constexpr int dy_num(int d) { return (-d) - (((-d) >> S) << S); }

template <int S, int V>
__device__ __forceinline__ void dy_term(typename Vec<V>::T &acc, float k, int d,
                                        const typename Vec<V>::T &a, const typename Vec<V>::T &b)
{
    const int num = (-d) - (((-d) >> S) << S);
    if (num == 0) {
        Vec<V>::mac(acc, k, a);
    } else {
        const float w1 = (float)num / (float)(1 << S), w0 = 1.0f - w1;
        Vec<V>::mac(acc, k, Vec<V>::lerp(w0, a, w1, b));
    }
}

template <int HW, int S, int V>
__global__ __launch_bounds__(256) void k_fir_sweep_dy(FirParams P, SweepGeom G, FirTaps T)
{
    typedef typename Vec<V>::T vec;
    constexpr int R = (HW + (1 << S) - 1) >> S;   // ceil(HW / 2^S)
    constexpr int RW = 2 * R + 2;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= G.ncols)
        return;
    const int p0 = G.out_lo + blockIdx.y * P.ts;
    const int p1 = min(p0 + P.ts, G.out_hi);
    const size_t base = (size_t)(G.outer_lo + col / G.cols_inner) * G.outer_stride +
                        (size_t)(col % G.cols_inner) * V;
    const float *__restrict__ s = P.src + base;
    float *__restrict__ d = P.dst + base;
    const int nl1 = G.n_loc - 1;
    const int off = P.off, n_glob = P.n_glob, uhw = P.uhw;

    vec ring[RW];
#pragma unroll
    for (int i = 0; i < RW - 1; i++)
        ring[i] = Vec<V>::ld(s + (size_t)clampi(p0 - R + i, 0, nl1) * G.stride);

#pragma unroll 1
    for (int p = p0; p < p1; p += RW) {
#pragma unroll
        for (int j = 0; j < RW; j++) {
            const int q = p + j;
            ring[(j + RW - 1) % RW] = Vec<V>::ld(s + (size_t)clampi(q + R + 1, 0, nl1) * G.stride);
            if (q < p1) {
                const int g = q + off;
                vec acc = Vec<V>::zero();
                if (g >= uhw && g <= n_glob - 2 - uhw) {
#pragma unroll
                    for (int dd = -HW; dd <= HW; dd++) {
                        const int o = ((-dd) >> S);          // floor(-d / 2^S), compile time
                        dy_term<S, V>(acc, T.k[dd + HW], dd, ring[(j + o + R + RW) % RW],
                                      ring[(j + o + R + 1 + RW) % RW]);
                    }
                } else {
                    float *a = reinterpret_cast<float *>(&acc);
#pragma unroll
                    for (int v = 0; v < V; v++)
                        a[v] = fir_literal_t<HW>(s + v, G.stride, g, n_glob, off, G.n_loc, T.k, HW,
                                                 P.uf, uhw);
                }
                Vec<V>::st(d + (size_t)q * G.stride, acc);
            }
        }
    }
}

// RX outputs per lane: 8, or 4 for rows of at most 256 voxels (all 64 lanes busy from octave 1 of a
// 512^3 volume on)
template <int HW, int S, int RX>
__global__ __launch_bounds__(256) void k_fir_x_dy(FirParams P, FirTaps T)
{
    constexpr int SEG = 64 * RX, HALO = 8, L = SEG + 2 * HALO;
    constexpr int R = (HW + (1 << S) - 1) >> S;
    static_assert(R + 1 <= HALO, "halo too small");
    __shared__ __attribute__((aligned(16))) float lds[4][L];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nrows = P.ny * (P.z_hi - P.z_lo);
    const int nx = P.nx;
    const int nmain = (nrows + 3) / 4;             // blockIdx.y >= nmain: boundary-column blocks
    if ((int)blockIdx.y >= nmain) {
        // The uhw low and uhw + 1 high columns of every row take the reference's boundary
        // path (imutil.c:829-850): one thread per such output, the literal arithmetic.  They
        // ride in the same launch; the interior blocks below do not store those columns.
        const int nedge = 2 * P.uhw + 1;
        const size_t i = ((size_t)(blockIdx.y - nmain) * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (i >= (size_t)nrows * nedge)
            return;
        const size_t erow = i / nedge;
        const int e = (int)(i % nedge);
        const int x = e < P.uhw ? e : nx - 1 - P.uhw + (e - P.uhw);
        if (x < 0 || x >= nx || (x >= P.uhw && x <= nx - 2 - P.uhw))
            return;
        const size_t eoff = ((size_t)P.z_lo * P.ny + erow) * nx;
        P.dst[eoff + x] = fir_literal_t<HW>(P.src + eoff, 1, x, nx, 0, nx, T.k, P.hw, P.uf, P.uhw);
        return;
    }
    const int row = blockIdx.y * 4 + wave;
    const bool active = row < nrows;
    const int x0 = blockIdx.x * SEG;
    const size_t rowoff = active ? ((size_t)(P.z_lo + row / P.ny) * P.ny + (row % P.ny)) * nx : 0;
    const float *__restrict__ s = P.src + rowoff;
    float *__restrict__ d = P.dst + rowoff;
    const bool vec_ok = (nx & 3) == 0 && ((((uintptr_t)P.src | (uintptr_t)P.dst) & 15) == 0);
    if (active) {
        for (int i = lane; i < L / 4; i += 64) {
            const int gx = x0 - HALO + 4 * i;
            float4 v;
            if (vec_ok && gx >= 0 && gx + 3 < nx) {
                v = ld4(s + gx);
            } else {
                v.x = (gx >= 0 && gx < nx) ? s[gx] : 0.0f;
                v.y = (gx + 1 >= 0 && gx + 1 < nx) ? s[gx + 1] : 0.0f;
                v.z = (gx + 2 >= 0 && gx + 2 < nx) ? s[gx + 2] : 0.0f;
                v.w = (gx + 3 >= 0 && gx + 3 < nx) ? s[gx + 3] : 0.0f;
            }
            *reinterpret_cast<float4 *>(&lds[wave][4 * i]) = v;
        }
    }
    __syncthreads();
    if (!active)
        return;
    const int xb = x0 + lane * RX;
    if (xb >= nx)
        return;
    float w[RX + 2 * HALO];
#pragma unroll
    for (int i = 0; i < (RX + 2 * HALO) / 4; i++) {
        const float4 v = *reinterpret_cast<const float4 *>(&lds[wave][lane * RX + 4 * i]);
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    float o[RX];
#pragma unroll
    for (int r = 0; r < RX; r++) {
        float acc = 0.0f;
#pragma unroll
        for (int dd = -HW; dd <= HW; dd++) {
            const int of = ((-dd) >> S);
            dy_term<S, 1>(acc, T.k[dd + HW], dd, w[HALO + r + of], w[HALO + r + of + 1]);
        }
        o[r] = acc;
    }
    // boundary columns belong to the edge blocks of this launch
    const bool has_edge = xb < P.uhw || xb + RX - 1 > nx - 2 - P.uhw;
    if (has_edge) {
#pragma unroll
        for (int r = 0; r < RX; r++)
            if (xb + r < nx && xb + r >= P.uhw && xb + r <= nx - 2 - P.uhw)
                d[xb + r] = o[r];
    } else if (vec_ok && xb + RX <= nx) {
#pragma unroll
        for (int r = 0; r < RX; r += 4)
            st4(d + xb + r, make_float4(o[r], o[r + 1], o[r + 2], o[r + 3]));
    } else {
#pragma unroll
        for (int r = 0; r < RX; r++)
            if (xb + r < nx)
                d[xb + r] = o[r];
    }
}

// ---------------------------------------------------------------------------------------
// im_subtract + dogmax  (imutil.c:719-739, sift.c:821-826)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sub_absmax(const float *__restrict__ a,
                                                    const float *__restrict__ b,
                                                    float *__restrict__ dst, size_t n,
                                                    unsigned *__restrict__ out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    float m = 0.0f;
    // four independent 16-byte load pairs in flight per thread and iteration
    size_t i = tid;
    for (; i + 3 * nthr < n4; i += 4 * nthr) {
        float4 u[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u[k] = ld4(a + 4 * (i + k * nthr));
            v[k] = ld4(b + 4 * (i + k * nthr));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float4 r;
            r.x = u[k].x - v[k].x; r.y = u[k].y - v[k].y; r.z = u[k].z - v[k].z; r.w = u[k].w - v[k].w;
            st4(dst + 4 * (i + k * nthr), r);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
        }
    }
    for (; i < n4; i += nthr) {
        const float4 u = ld4(a + 4 * i), v = ld4(b + 4 * i);
        float4 r;
        r.x = u.x - v.x; r.y = u.y - v.y; r.z = u.z - v.z; r.w = u.w - v.w;
        st4(dst + 4 * i, r);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
    }
    for (size_t j = 4 * n4 + tid; j < n; j += nthr) {
        const float r = a[j] - b[j];
        dst[j] = r;
        m = fmaxf(m, fabsf(r));
    }
    if (out)                                   // kernel argument: uniform
        block_max_atomic<1>(&m, out);
}

// ---------------------------------------------------------------------------------------
// im_downsample_2x  (imutil.c:591-617)
// ---------------------------------------------------------------------------------------
// All DoG levels of one octave in one pass: NL Gaussian levels are read once (4*NL B/voxel) and
// NL-1 differences written, instead of 12 B/voxel per level pair.  Same arithmetic and the same
// order-free max as k_sub_absmax.
struct DogStack {
    const float *g[SIFT3D_HIP_MAX_DOG_STACK];
    float *d[SIFT3D_HIP_MAX_DOG_STACK - 1];
    unsigned *out; // NL-1 consecutive maxima (float bits)
};

template <int NL>
__global__ __launch_bounds__(256) void k_dog_stack(DogStack S, size_t n)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    float m[NL - 1];
#pragma unroll
    for (int k = 0; k < NL - 1; k++)
        m[k] = 0.0f;
    for (size_t i = tid; i < n4; i += nthr) {
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++)
            v[k] = ld4(S.g[k] + 4 * i);
#pragma unroll
        for (int k = 0; k < NL - 1; k++) {
            float4 r;
            r.x = v[k].x - v[k + 1].x; r.y = v[k].y - v[k + 1].y;
            r.z = v[k].z - v[k + 1].z; r.w = v[k].w - v[k + 1].w;
            st4(S.d[k] + 4 * i, r);
            m[k] = fmaxf(m[k], fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
        }
    }
    for (size_t j = 4 * n4 + tid; j < n; j += nthr) {
        float prev = S.g[0][j];
#pragma unroll
        for (int k = 0; k < NL - 1; k++) {
            const float cur = S.g[k + 1][j];
            const float r = prev - cur;
            S.d[k][j] = r;
            m[k] = fmaxf(m[k], fabsf(r));
            prev = cur;
        }
    }
    block_max_atomic<NL - 1>(m, S.out);
}

// The same maxima without the DoG levels themselves: the extrema sweep below forms the
// differences on the fly from the Gaussian levels, so the DoG pyramid is never stored
// (24 B/voxel read here instead of 24 B read + 20 B written, and 5/11 of the pyramid memory).
template <int NL>
__global__ __launch_bounds__(256) void k_dogmax_stack(DogStack S, size_t n)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n >> 2;
    float m[NL - 1];
#pragma unroll
    for (int k = 0; k < NL - 1; k++)
        m[k] = 0.0f;
    for (size_t i = tid; i < n4; i += nthr) {
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++)
            v[k] = ld4(S.g[k] + 4 * i);
#pragma unroll
        for (int k = 0; k < NL - 1; k++) {
            float4 r;
            r.x = v[k].x - v[k + 1].x; r.y = v[k].y - v[k + 1].y;
            r.z = v[k].z - v[k + 1].z; r.w = v[k].w - v[k + 1].w;
            m[k] = fmaxf(m[k], fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
        }
    }
    for (size_t j = 4 * n4 + tid; j < n; j += nthr) {
        float prev = S.g[0][j];
#pragma unroll
        for (int k = 0; k < NL - 1; k++) {
            const float cur = S.g[k + 1][j];
            m[k] = fmaxf(m[k], fabsf(prev - cur));
            prev = cur;
        }
    }
    block_max_atomic<NL - 1>(m, S.out);
}

__global__ __launch_bounds__(256) void k_downsample2(const float *__restrict__ src, int nx, int ny,
                                                     float *__restrict__ dst, int mx, int my,
                                                     int mz)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int z = blockIdx.z;
    if (x >= mx)
        return;
    dst[(size_t)x + (size_t)mx * ((size_t)y + (size_t)my * z)] =
        src[(size_t)(2 * x) + (size_t)nx * ((size_t)(2 * y) + (size_t)ny * (2 * z))];
}

// the same for rows of whole quads on both sides (mx % 4 == 0, nx >= 2 mx, 16-byte aligned): a thread reads
// two 16-byte quads and writes one; a workgroup covers 4 output rows of up to 256 voxels.  (One dword per
// thread and 1 KB per workgroup made the first link of the smaller octaves' chain -- 512^3 -> 256^3 -- a
// 65 536-workgroup launch.)
__global__ __launch_bounds__(256) void k_downsample2_q(const float *__restrict__ src, int nx, int ny,
                                                       float *__restrict__ dst, int mxq, int my, int mz)
{
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.z;
    if (qx >= mxq || y >= my)
        return;
    const float *s = src + (size_t)(8 * qx) + (size_t)nx * ((size_t)(2 * y) + (size_t)ny * (2 * z));
    const float4 a = ld4(s), b = ld4(s + 4);
    st4(dst + (size_t)(4 * qx) + (size_t)(4 * mxq) * ((size_t)y + (size_t)my * z), make_float4(a.x, a.z, b.x, b.z));
}

// ---------------------------------------------------------------------------------------
// detect_extrema  (sift.c:735-871): mask -> scan -> emit, output in scan order
// ---------------------------------------------------------------------------------------
constexpr int EX_WPB = 128; // 64-voxel words per block (32 per wave)

struct ExLevels {
    sift3d_hip_extrema_level lv[8];
};

struct ExGeom {
    int nx, ny, nz;
    int wpr;        // words per row = ceil(nx / 64)
    uint32_t nwords;// nz * ny * wpr
    uint32_t nblk;  // ceil(nwords / EX_WPB)
    double peak_thresh;
    int cuboid;     // 1: the reference's CUBOID_EXTREMA build (80 neighbours, sift.c:761-796)
};

// CMP_CUBE of the CUBOID_EXTREMA build (sift.c:761-796): strictly above (or strictly below) all
// 27 samples of the previous and next DoG level and the 26 neighbours in the current one
__device__ __forceinline__ bool cuboid_extremum(const float *__restrict__ prev,
                                                const float *__restrict__ cur,
                                                const float *__restrict__ next, size_t q, size_t ys,
                                                size_t zs, float c)
{
    bool gt = true, lt = true;
#pragma unroll
    for (int dz = -1; dz <= 1; dz++)
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const size_t r = q + dx + ys * dy + zs * dz;
                const float a = prev[r], b = next[r];
                gt = gt && c > a && c > b;
                lt = lt && c < a && c < b;
                if (dx || dy || dz) {
                    const float m = cur[r];
                    gt = gt && c > m;
                    lt = lt && c < m;
                }
            }
    return gt || lt;
}

template <bool CUBOID>
__global__ __launch_bounds__(256) void k_extrema_mask(ExLevels LV, ExGeom E,
                                                      unsigned long long *__restrict__ masks,
                                                      uint32_t *__restrict__ blk_counts)
{
    __shared__ uint32_t wc[4];
    constexpr int WPW = EX_WPB / 4; // words per wave
    const int level = blockIdx.y;
    const sift3d_hip_extrema_level L = LV.lv[level];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // thr = (float)(peak_thresh * dogmax), sift.c:829
    const float thr = (float)(E.peak_thresh * (double)(*L.d_absmax));
    const size_t ys = E.nx, zs = (size_t)E.nx * E.ny;
    const uint32_t wbase = blockIdx.x * EX_WPB + wave * WPW;
    uint32_t cnt = 0;
    if (wbase < E.nwords) {
        // (z, y, word-in-row) of the wave's first word; advanced without divisions afterwards
        const uint32_t row0 = wbase / E.wpr;
        int xw = (int)(wbase - row0 * E.wpr);
        int z = (int)(row0 / E.ny), y = (int)(row0 - (uint32_t)z * E.ny);
        const uint32_t wend = min(wbase + WPW, E.nwords);
        for (uint32_t word = wbase; word < wend; word += 4) {
            // four words per iteration: their centre samples are loaded together
            float v[4];
            size_t p[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = xw * 64 + lane;
                ok[k] = word + k < wend && z >= L.z_lo && z < L.z_hi && y >= 1 && y <= E.ny - 2 &&
                        x >= 1 && x <= E.nx - 2;
                p[k] = (size_t)x + ys * y + zs * z;
                v[k] = ok[k] ? L.cur[p[k]] : 0.0f;
                if (++xw == E.wpr) {
                    xw = 0;
                    if (++y == E.ny) {
                        y = 0;
                        ++z;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                bool hit = false;
                if (CUBOID) {
                    if (ok[k] && (v[k] > thr || v[k] < -thr))            // sift.c:842
                        hit = cuboid_extremum(L.prev, L.cur, L.next, p[k], ys, zs, v[k]);
                } else if (ok[k] && (v[k] > thr || v[k] < -thr)) {       // sift.c:842
                    const size_t q = p[k];
                    const float c = v[k];
                    const float n0 = L.prev[q], n1 = L.cur[q + 1], n2 = L.cur[q - 1],
                                n3 = L.cur[q + ys], n4 = L.cur[q - ys], n5 = L.cur[q - zs],
                                n6 = L.cur[q + zs], n7 = L.next[q];
                    hit = (c > n0 && c > n1 && c > n2 && c > n3 && c > n4 && c > n5 && c > n6 &&
                           c > n7) ||
                          (c < n0 && c < n1 && c < n2 && c < n3 && c < n4 && c < n5 && c < n6 &&
                           c < n7);                                      // sift.c:844-849
                }
                const unsigned long long m = __ballot(hit);
                if (word + k < wend) {
                    if (lane == 0)
                        masks[(size_t)level * E.nwords + word + k] = m;
                    cnt += (uint32_t)__popcll(m);
                }
            }
        }
    }
    if (lane == 0)
        wc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0)
        blk_counts[(size_t)level * E.nblk + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// ---- three keypoint levels in one z sweep (default 8-neighbour test) -----------------------
// The three keypoint levels of an octave share their DoG levels (next of level i = centre of
// level i+1), and the scattered neighbour lines of k_extrema_mask cost ~4x the centre samples.
// Here a workgroup owns a 64(x) x 16(y) column and walks z: every thread keeps three planes
// (z-1, z, z+1) of the three centre levels in registers, so each of the five DoG levels is read
// once along z; the y neighbours are two more (cache-resident) row loads, the x neighbours come
// from the adjacent lanes (DPP row shift; one scalar load at the tile's ends).  Output: the same
// 64-voxel mask words as k_extrema_mask, assembled with a DPP OR-reduction over the 16 lanes of
// a row, so the scan and emit kernels (and with them the reference's scan order) are unchanged.
struct ExSweep {
    const float *d[6];        // DoG levels s-1 .. s+3 of the three keypoint levels (k_extrema_sweep3) or
                              // the SIX Gaussian levels they are differences of (k_extrema_sweep3g)
    const float *absmax[3];
    double peak_thresh;
    int nx, ny, nz;           // local dims
    int z_lo, z_hi, ts;       // output planes [z_lo, z_hi), segment length
    int wpr;
    uint32_t nwords;
    uint32_t *masks32;        // [3][nwords] 64-bit words as uint32 pairs
    unsigned *exact;          // k_extrema_sweep3g<.., true>: the five max|DoG| of the octave are gathered here
};

// v_max3_f32 / v_min3_f32 (operands that are not NaN: the result is the exact maximum / minimum)
__device__ __forceinline__ float max3f(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float max2f(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min2f(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL> __device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);   // out-of-row lanes read 0
}

// (Stored DoG levels; the default configuration has none and runs k_extrema_sweep3g below.)
__global__ __launch_bounds__(256) void k_extrema_sweep3(ExSweep S)
{
    auto ldd4 = [&](int k, size_t o) -> float4 { return ld4(S.d[k] + o); };
    auto ldd1 = [&](int k, size_t o) -> float { return S.d[k][o]; };
    constexpr int TY = 16;
    const int qx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = (blockIdx.x * 16 + qx) * 4, y = blockIdx.y * TY + ty;
    const int nx = S.nx, ny = S.ny;
    const size_t ys = nx, zs = (size_t)nx * ny;
    const bool col = x < nx && y < ny;                 // (nx % 4 == 0: whole quads)
    const int yc = min(y, ny - 1), xc = min(x, nx - 4);
    const int yu = max(yc - 1, 0), yd = min(yc + 1, ny - 1);
    const size_t oc = (size_t)yc * ys + xc, ou = (size_t)yu * ys + xc, od = (size_t)yd * ys + xc;
    const int p0 = S.z_lo + blockIdx.z * S.ts, p1 = min(p0 + S.ts, S.z_hi);
    if (p0 >= p1)
        return;
    float thr[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
        thr[i] = (float)(S.peak_thresh * (double)(*S.absmax[i]));        // sift.c:829
    // which of the quad's four voxels may be extrema at all (sift.c:833-838: 1 .. n-2)
    bool okx[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
        okx[e] = col && y >= 1 && y <= ny - 2 && x + e >= 1 && x + e <= nx - 2;
    float4 m[3], c[3], p[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        m[i] = ldd4(i + 1, (size_t)(p0 - 1) * zs + oc);
        c[i] = ldd4(i + 1, (size_t)p0 * zs + oc);
    }
#pragma unroll 1
    for (int z = p0; z < p1; z++) {
        const size_t zo = (size_t)z * zs;
        float4 up[3], dn[3];
        float lf[3], rt[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            p[i] = ldd4(i + 1, zo + zs + oc);
            up[i] = ldd4(i + 1, zo + ou);
            dn[i] = ldd4(i + 1, zo + od);
            // x neighbours of the quad's ends: adjacent lanes of the 16-lane row, or memory at
            // the ends of the 64-voxel tile
            lf[i] = __int_as_float(dpp_i<0x111>(__float_as_int(c[i].w)));   // row_shr:1
            rt[i] = __int_as_float(dpp_i<0x101>(__float_as_int(c[i].x)));   // row_shl:1
            if (qx == 0 && col && x > 0)
                lf[i] = ldd1(i + 1, zo + oc - 1);
            if (qx == 15 && col && x + 4 < nx)
                rt[i] = ldd1(i + 1, zo + oc + 4);
        }
        const float4 d0c = ldd4(0, zo + oc), d4c = ldd4(4, zo + oc);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float4 pv = i == 0 ? d0c : c[i - 1], nv = i == 2 ? d4c : c[i + 1];
            const float cv[4] = { c[i].x, c[i].y, c[i].z, c[i].w };
            const float pr[4] = { pv.x, pv.y, pv.z, pv.w }, ne[4] = { nv.x, nv.y, nv.z, nv.w };
            const float uu[4] = { up[i].x, up[i].y, up[i].z, up[i].w };
            const float dd[4] = { dn[i].x, dn[i].y, dn[i].z, dn[i].w };
            const float zm[4] = { m[i].x, m[i].y, m[i].z, m[i].w };
            const float zp[4] = { p[i].x, p[i].y, p[i].z, p[i].w };
            int nib = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v = cv[e];
                const float xm = e > 0 ? cv[e - 1] : lf[i], xp = e < 3 ? cv[e + 1] : rt[i];
                const bool hit =
                    okx[e] && (v > thr[i] || v < -thr[i]) &&                           // sift.c:842
                    ((v > pr[e] && v > xp && v > xm && v > dd[e] && v > uu[e] && v > zm[e] &&
                      v > zp[e] && v > ne[e]) ||
                     (v < pr[e] && v < xp && v < xm && v < dd[e] && v < uu[e] && v < zm[e] &&
                      v < zp[e] && v < ne[e]));                                        // sift.c:844-849
                nib |= hit ? (1 << e) : 0;
            }
            // 64-bit word of the row: voxel 4*qx + e -> bit 4*qx + e; OR over the 16 lanes
            int lo = qx < 8 ? nib << (4 * qx) : 0, hi = qx >= 8 ? nib << (4 * (qx - 8)) : 0;
            lo |= dpp_i<0x111>(lo); hi |= dpp_i<0x111>(hi);
            lo |= dpp_i<0x112>(lo); hi |= dpp_i<0x112>(hi);
            lo |= dpp_i<0x114>(lo); hi |= dpp_i<0x114>(hi);
            lo |= dpp_i<0x118>(lo); hi |= dpp_i<0x118>(hi);
            if (qx == 15 && y < ny) {
                const size_t w = (size_t)i * S.nwords + ((size_t)z * ny + y) * S.wpr + blockIdx.x;
                *reinterpret_cast<uint2 *>(S.masks32 + 2 * w) = make_uint2((unsigned)lo, (unsigned)hi);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            m[i] = c[i];
            c[i] = p[i];
        }
    }
}

// ---- the same sweep straight from the SIX Gaussian levels, every sample loaded once ----------
// Forming the differences inside k_extrema_sweep3's loads would ask the memory system for ~26 KB
// per wave and plane (the y neighbours and both Gaussian levels of every difference loaded again by
// every thread that needs them): 5x the bytes of the levels, and the L2 -> L1 path, not HBM, then
// sets the time (round 2 started that way).  Here a thread loads
// exactly its own quad of each Gaussian level once per plane (G1..G4 one plane ahead, G0 and G5 at
// the centre plane), keeps what the next step needs in registers, and the workgroup trades the
// centre-plane differences through an LDS tile (64 x 16 voxels + one halo row above and below,
// loaded by 32 of the 256 threads) for the y neighbours.  One barrier per plane (the tile is
// double-buffered).  Arithmetic, order of the tests and output are those of k_extrema_sweep3.
// EST: S.absmax[] hold LOWER BOUNDS of the three maxima (k_dogmax_sub's maxima over a sub-lattice), so the
// masks are a SUPERSET of the reference's; the sweep gathers the exact maxima of all five DoG levels over
// its centre planes on the way (it forms every difference of those planes anyway) and
// k_extrema_refilter then applies the reference's threshold (sift.c:829, 842) to the marked voxels.  The
// octave's Gaussian levels are read once instead of twice.
template <int TXQ, bool EST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_extrema_sweep3g(ExSweep S)
{
    constexpr int TY = 256 / TXQ;          // tile: 4 * TXQ voxels along x, TY rows
    __shared__ float4 tile[2][3][TY + 2][TXQ];
    const int qx = threadIdx.x % TXQ, ty = threadIdx.x / TXQ;
    const int q16 = qx & 15;               // position in the 16-lane row = the 64-voxel mask word
    // Workgroups go to the eight XCDs round robin in launch order, and an XCD's L2 is its own: tiles that share
    // halo rows (y neighbours) should meet in ONE L2.  XCD k takes the k-th eighth of the tiles in (x, y, z
    // segment) order -- at 512^3 exactly one z segment --, in that order.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy * gridDim.z;
        const unsigned L = bx + gx * (by + gy * bz), xcd = L & 7u, j = L >> 3;
        const unsigned q = T >> 3, r = T & 7u;
        const unsigned t = xcd * q + (xcd < r ? xcd : r) + j;
        bx = (int)(t % gx);
        by = (int)((t / gx) % gy);
        bz = (int)(t / (gx * gy));
    }
    const int x = (bx * TXQ + qx) * 4, y0 = by * TY, y = y0 + ty;
    const int nx = S.nx, ny = S.ny;
    const size_t ys = nx, zs = (size_t)nx * ny;
    const bool col = x < nx && y < ny;                 // (nx % 4 == 0: whole quads)
    const int yc = min(y, ny - 1), xc = min(x, nx - 4);
    const size_t oc = (size_t)yc * ys + xc;
    // halo rows of the tile (rows y0 - 1 and y0 + TY, clamped like the y neighbours of the
    // reference loop's border voxels, which are never extrema): threads 0..31
    const bool halo = threadIdx.x < 2 * TXQ;
    const int hr = threadIdx.x / TXQ;                  // 0: row above, 1: row below (halo threads)
    const int yh = hr == 0 ? max(y0 - 1, 0) : min(y0 + TY, ny - 1);
    const size_t oh = (size_t)yh * ys + xc;
    const int p0 = S.z_lo + bz * S.ts, p1 = min(p0 + S.ts, S.z_hi);
    if (p0 >= p1)
        return;
    float thr[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
        thr[i] = (float)(S.peak_thresh * (double)(*S.absmax[i]));        // sift.c:829
    bool okx[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
        okx[e] = col && y >= 1 && y <= ny - 2 && x + e >= 1 && x + e <= nx - 2;
    auto sub4 = [](const float4 &a, const float4 &b) {
        return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);   // im_subtract, imutil.c:719-739
    };
    float mx[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
    auto amax4 = [](float mm, const float4 &v) {
        float r;
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(mm), "v"(v.x), "v"(v.y));
        asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(r), "v"(v.z), "v"(v.w));
        return r;
    };
    // differences 1..3 at planes z-1 (m), z (c), z+1 (p); Gaussian levels 1 and 4 at plane z
    float4 m[3], c[3], p[3], g1c, g4c;
    {
        float4 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            a[k] = ld4(S.d[k + 1] + (size_t)(p0 - 1) * zs + oc);
            b[k] = ld4(S.d[k + 1] + (size_t)p0 * zs + oc);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            m[i] = sub4(a[i], a[i + 1]);
            c[i] = sub4(b[i], b[i + 1]);
        }
        g1c = b[0];
        g4c = b[3];
        // (the halo rows' differences go straight into the tile the plane will use: the buffer of the NEXT
        // plane was last read two planes ago, behind a barrier; nothing of them is carried in registers)
        if (halo) {
            float4 h[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                h[k] = ld4(S.d[k + 1] + (size_t)p0 * zs + oh);
#pragma unroll
            for (int i = 0; i < 3; i++)
                tile[0][i][hr * (TY + 1)][qx] = sub4(h[i], h[i + 1]);
        }
    }
    // The x neighbours of a row segment's two end quads come from memory (every other one from the adjacent
    // lane): Gaussian levels 1..4 at x - 1 (lane qx == 0) or x + 4 (lane qx == TXQ - 1) of the centre plane.
    // They are requested ONE PLANE AHEAD, like every other sample of the sweep: requested where they are
    // needed, each of the six differences cost the wave a full memory round trip per plane -- s_waitcnt
    // vmcnt(0) six times, draining the plane's 16-byte loads with it -- and every wave of a 256-voxel row
    // segment holds both end lanes (measured: 6.8 us per plane and workgroup, 2.7 TB/s).
    const bool edge = col && ((qx == 0 && x > 0) || (qx == TXQ - 1 && x + 4 < nx));
    const size_t oe = oc + (qx == 0 ? (size_t)-1 : (size_t)4);
    float ed[3] = { 0.f, 0.f, 0.f };          // centre plane's differences (beyond the volume: 0, as before)
    if (edge) {
        float e0[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            e0[k] = S.d[k + 1][(size_t)p0 * zs + oe];
#pragma unroll
        for (int i = 0; i < 3; i++)
            ed[i] = e0[i] - e0[i + 1];
    }
    int buf = 0;
#pragma unroll 1
    for (int z = p0; z < p1; z++) {
        const size_t zo = (size_t)z * zs;
        // centre-plane differences into the tile (known since the previous step)
#pragma unroll
        for (int i = 0; i < 3; i++)
            tile[buf][i][ty + 1][qx] = c[i];
        // this step's loads: every Gaussian level once
        float4 n[4], hn[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            n[k] = ld4(S.d[k + 1] + zo + zs + oc);
        const float4 g0 = ld4(S.d[0] + zo + oc), g5 = ld4(S.d[5] + zo + oc);
        if (halo) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                hn[k] = ld4(S.d[k + 1] + zo + zs + oh);
        }
        float en[4] = { 0.f, 0.f, 0.f, 0.f };     // the end quads' outer neighbours of the NEXT centre plane
        if (edge) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                en[k] = S.d[k + 1][zo + zs + oe];
        }
        float lf[3], rt[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            // the neighbours come from the adjacent lanes of the WAVE (DPP wave shift; a wave holds
            // 64 / TXQ whole row segments), memory (ec, requested a plane ago) only at the two ends of a
            // row segment
            lf[i] = __int_as_float(dpp_i<0x138>(__float_as_int(c[i].w)));   // wave_shr:1
            rt[i] = __int_as_float(dpp_i<0x130>(__float_as_int(c[i].x)));   // wave_shl:1
            lf[i] = qx == 0 ? ed[i] : lf[i];
            rt[i] = qx == TXQ - 1 ? ed[i] : rt[i];
        }
        __syncthreads();
        float4 up[3], dn[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            up[i] = tile[buf][i][ty][qx];
            dn[i] = tile[buf][i][ty + 2][qx];
        }
        buf ^= 1;
#pragma unroll
        for (int i = 0; i < 3; i++)
            p[i] = sub4(n[i], n[i + 1]);
        const float4 d0c = sub4(g0, g1c), d4c = sub4(g4c, g5);
        if (EST && col) {
            // (clamped duplicates of the last column / row would not matter to a maximum either)
            mx[0] = amax4(mx[0], d0c);
            mx[1] = amax4(mx[1], c[0]);
            mx[2] = amax4(mx[2], c[1]);
            mx[3] = amax4(mx[3], c[2]);
            mx[4] = amax4(mx[4], d4c);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float4 pv = i == 0 ? d0c : c[i - 1], nv = i == 2 ? d4c : c[i + 1];
            const float cv[4] = { c[i].x, c[i].y, c[i].z, c[i].w };
            const float pr[4] = { pv.x, pv.y, pv.z, pv.w }, ne[4] = { nv.x, nv.y, nv.z, nv.w };
            const float uu[4] = { up[i].x, up[i].y, up[i].z, up[i].w };
            const float dd[4] = { dn[i].x, dn[i].y, dn[i].z, dn[i].w };
            const float zm[4] = { m[i].x, m[i].y, m[i].z, m[i].w };
            const float zp[4] = { p[i].x, p[i].y, p[i].z, p[i].w };
            int nib = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float v = cv[e];
                const float xm = e > 0 ? cv[e - 1] : lf[i], xp = e < 3 ? cv[e + 1] : rt[i];
                // "greater than each of the eight" = greater than their maximum (sift.c:844-849; the differences
                // of finite samples are never NaN): 3 x v_max3 + v_max and the same for the minimum instead of
                // sixteen compares and as many scalar ANDs -- the sweep's arithmetic, not its loads, is what a
                // plane costs beyond the copy rate.  (No short circuit: in a wave some lane nearly always passes
                // the threshold, so branches only cost.)
                const float hi8 = max2f(max3f(max3f(max3f(uu[e], dd[e], zm[e]), zp[e], pr[e]), ne[e], xm), xp);
                const float lo8 = min2f(min3f(min3f(min3f(uu[e], dd[e], zm[e]), zp[e], pr[e]), ne[e], xm), xp);
                const bool hit = okx[e] & (fabsf(v) > thr[i]) &                       // sift.c:842
                                 ((v > hi8) | (v < lo8));
                nib |= hit ? (1 << e) : 0;
            }
            // 64-bit word of the row: voxel 4*qx + e -> bit 4*qx + e; OR over the 16 lanes
            int lo = q16 < 8 ? nib << (4 * q16) : 0, hi = q16 >= 8 ? nib << (4 * (q16 - 8)) : 0;
            lo |= dpp_i<0x111>(lo); hi |= dpp_i<0x111>(hi);
            lo |= dpp_i<0x112>(lo); hi |= dpp_i<0x112>(hi);
            lo |= dpp_i<0x114>(lo); hi |= dpp_i<0x114>(hi);
            lo |= dpp_i<0x118>(lo); hi |= dpp_i<0x118>(hi);
            const int wcol = bx * (TXQ / 16) + (qx >> 4);     // 64-voxel word of the row
            if (q16 == 15 && wcol < S.wpr && y < ny) {
                const size_t w = (size_t)i * S.nwords + ((size_t)z * ny + y) * S.wpr + wcol;
                *reinterpret_cast<uint2 *>(S.masks32 + 2 * w) = make_uint2((unsigned)lo, (unsigned)hi);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            m[i] = c[i];
            c[i] = p[i];
        }
        g1c = n[0];
        g4c = n[3];
#pragma unroll
        for (int i = 0; i < 3; i++)
            ed[i] = en[i] - en[i + 1];
        if (halo) {
#pragma unroll
            for (int i = 0; i < 3; i++)
                tile[buf][i][hr * (TY + 1)][qx] = sub4(hn[i], hn[i + 1]);   // (buf: the next plane's)
        }
    }
    if (EST) {
        __syncthreads();                   // (block_max_atomic has a shared array of its own; the tile is done)
        block_max_atomic<5>(mx, S.exact);
    }
}

// max|DoG| of an octave's levels over the sub-lattice z = 1, 6, 11, ..., y = 0, 3, 6, ...: LOWER bounds of the
// maxima (k_extrema_sweep3g<.., true> wants nothing more of them), one fifteenth of the octave's bytes.  (Strides
// 5 and 3: a lattice point within (2, 1) voxels of every voxel, and no common factor with power-of-two
// structure in the data.)
constexpr int SUB_Z = 5, SUB_Y = 3;
template <int NL>
__global__ __launch_bounds__(256) void k_dogmax_sub(DogStack S, int nx, int ny, int nz)
{
    const uint32_t q = (uint32_t)nx >> 2, rpp = ((uint32_t)ny + SUB_Y - 1) / SUB_Y;
    const uint32_t npl = nz >= 2 ? ((uint32_t)nz - 2) / SUB_Z + 1 : 1;      // planes 1, 6, ... (plane 0 if nz < 2)
    const uint64_t items = (uint64_t)q * rpp * npl;
    const uint64_t nthr = (uint64_t)gridDim.x * blockDim.x;
    float m[NL - 1];
#pragma unroll
    for (int k = 0; k < NL - 1; k++)
        m[k] = 0.0f;
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += nthr) {
        const uint32_t r = (uint32_t)(it / q), qq = (uint32_t)(it - (uint64_t)r * q);
        const uint32_t pl = r / rpp, row = r - pl * rpp;
        const uint32_t z = nz >= 2 ? 1 + SUB_Z * pl : 0, y = SUB_Y * row;
        const size_t off = ((size_t)z * ny + y) * nx + 4 * qq;
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++)
            v[k] = ld4(S.g[k] + off);
#pragma unroll
        for (int k = 0; k < NL - 1; k++) {
            float4 r4;
            r4.x = v[k].x - v[k + 1].x; r4.y = v[k].y - v[k + 1].y;
            r4.z = v[k].z - v[k + 1].z; r4.w = v[k].w - v[k + 1].w;
            m[k] = fmaxf(m[k], fmaxf(fmaxf(fabsf(r4.x), fabsf(r4.y)), fmaxf(fabsf(r4.z), fabsf(r4.w))));
        }
    }
    block_max_atomic<NL - 1>(m, S.out);
}

// The masks of k_extrema_sweep3g<.., true> hold every extremum above a LOWER bound of the threshold; with
// the exact maxima known, the reference's test (sift.c:829, 842) is applied to the marked voxels: one
// thread per 64-voxel mask word, nearly all of them zero.
__global__ __launch_bounds__(256) void k_extrema_refilter(ExSweep S)
{
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= (uint64_t)3 * S.nwords)
        return;
    unsigned long long *word = reinterpret_cast<unsigned long long *>(S.masks32) + w;
    unsigned long long bits = *word;
    if (!bits)
        return;
    const int i = (int)(w / S.nwords);
    const uint32_t r = (uint32_t)(w - (uint64_t)i * S.nwords);
    const uint32_t rowi = r / (uint32_t)S.wpr, wc = r - rowi * (uint32_t)S.wpr;   // rowi = z * ny + y
    const size_t base = (size_t)rowi * S.nx + 64u * wc;
    const float thr = (float)(S.peak_thresh * (double)__uint_as_float(S.exact[1 + i]));    // sift.c:829
    const float *ga = S.d[i + 1], *gb = S.d[i + 2];
    unsigned long long keep = bits;
    while (bits) {
        const int b = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const float v = ga[base + b] - gb[base + b];                               // im_subtract
        if (!((v > thr) | (v < -thr)))                                             // sift.c:842
            keep &= ~(1ull << b);
    }
    *word = keep;
}

// candidates per block of EX_WPB mask words (what k_extrema_mask counts itself)
__global__ __launch_bounds__(256) void k_extrema_count(const unsigned long long *__restrict__ masks,
                                                       uint32_t nwords, uint32_t nblk,
                                                       uint32_t *__restrict__ blk_counts)
{
    __shared__ uint32_t wc[4];
    const int level = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t cnt = 0;
    for (uint32_t w = blockIdx.x * EX_WPB + threadIdx.x; w < min((blockIdx.x + 1) * (uint32_t)EX_WPB, nwords);
         w += 256)
        cnt += (uint32_t)__popcll(masks[(size_t)level * nwords + w]);
    cnt = wave_sum_u32(cnt);
    if (lane == 0)
        wc[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0)
        blk_counts[(size_t)level * nblk + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}

// exclusive scan of the block counts (all levels of the launch), continuing from *d_count.  One workgroup
// walks the array in chunks of 8192 entries: a thread loads eight consecutive entries (two 16-byte loads,
// coalesced), scans them, the thread sums are scanned by wave shifts and the sixteen wave totals by the
// first wave -- two barriers per chunk (49 152 entries at 512^3: 6 chunks; the chunked Hillis-Steele scan
// this replaces took 480 barriers and 85 us there).
__device__ __forceinline__ uint32_t ex_scan_range(uint32_t *__restrict__ blk, uint32_t n, uint32_t carry,
                                                  uint32_t *wtot, uint32_t &wsum)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (uint32_t base = 0; base < n; base += 8192) {
        const uint32_t i0 = base + 8u * (uint32_t)t;
        uint32_t v[8];
        if (i0 + 8 <= n && (((uintptr_t)(blk + i0)) & 15) == 0) {
            const uint4 a = *reinterpret_cast<const uint4 *>(blk + i0), b = *reinterpret_cast<const uint4 *>(blk + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                v[k] = i0 + k < n ? blk[i0 + k] : 0;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            sum += v[k];
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            inc += lane >= o ? u : 0;
        }
        if (lane == 63)
            wtot[wave] = inc;
        __syncthreads();
        if (wave == 0) {
            uint32_t w = lane < 16 ? wtot[lane] : 0, winc = w;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const uint32_t u = __shfl_up(winc, o, 64);
                winc += lane >= o ? u : 0;
            }
            if (lane < 16)
                wtot[lane] = winc - w;                 // exclusive
            if (lane == 15)
                wsum = winc;
        }
        __syncthreads();
        uint32_t run = carry + wtot[wave] + inc - sum;
        carry += wsum;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t x = v[k];
            v[k] = run;
            run += x;
        }
        if (i0 + 8 <= n && (((uintptr_t)(blk + i0)) & 15) == 0) {
            *reinterpret_cast<uint4 *>(blk + i0) = make_uint4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint4 *>(blk + i0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (i0 + k < n)
                    blk[i0 + k] = v[k];
        }
        __syncthreads();                   // wtot / wsum are rewritten by the next chunk
    }
    return carry;
}

__global__ __launch_bounds__(1024) void k_extrema_scan(uint32_t *__restrict__ blk, uint32_t n,
                                                       uint32_t *__restrict__ d_count)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t wsum;
    const uint32_t carry = ex_scan_range(blk, n, *d_count, wtot, wsum);
    if (threadIdx.x == 0)
        *d_count = carry;
}

// Scan + emission of ALL octaves of a detect call in two launches (round 5; before: a scan and an emission
// launch per octave, fourteen dependent short launches at 512^3): the octaves' block-count arrays are scanned
// one after the other by the one workgroup (the running total carries over: octave order), and the emission
// grid covers every octave's blocks -- a workgroup finds its octave in a table of first-block numbers.
constexpr int EX_MAX_OCT = SIFT3D_HIP_EXTREMA_MAX_OCT;
struct ExOct {
    const float *g[4];                    // Gaussian levels 1..4: keypoint DoG level i = g[i] - g[i + 1]
    const unsigned long long *masks;      // [3][nwords]
    uint32_t *blk;                        // [3][nblk] block counts -> offsets
    int nx, ny, wpr;
    uint32_t nwords, nblk;
    int tag0;
    uint32_t blk_first;                   // first workgroup (x) of this octave in the emission grid
};
struct ExMulti {
    int n;
    ExOct o[EX_MAX_OCT];
};

__global__ __launch_bounds__(1024) void k_extrema_scan_multi(ExMulti M, uint32_t *__restrict__ d_count)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t wsum;
    uint32_t carry = *d_count;
    for (int i = 0; i < M.n; i++)
        carry = ex_scan_range(M.o[i].blk, M.o[i].nblk * 3u, carry, wtot, wsum);
    if (threadIdx.x == 0)
        *d_count = carry;
}

// FROM_G: `cur` and `next` of a level hold the two Gaussian levels whose difference is the DoG
// level (the DoG pyramid is not stored)
// one emission workgroup: block `bx` of a level (masks / blk_off: that level's own arrays)
template <bool FROM_G>
__device__ __forceinline__ void ex_emit_block(const float *__restrict__ cur, const float *__restrict__ next,
                                              int tag, int nx, int ny, int wpr, uint32_t nwords, uint32_t bx,
                                              const unsigned long long *__restrict__ masks,
                                              const uint32_t *__restrict__ blk_off,
                                              sift3d_hip_cand *__restrict__ out, uint32_t cap)
{
    __shared__ uint32_t pre[EX_WPB + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t w0 = bx * EX_WPB;
    // pre[i] = candidates in the block's words before word i: the first EX_WPB / 64 waves hold one word per
    // lane, scan their counts by wave shifts and add the totals of the waves before them
    static_assert(EX_WPB % 64 == 0 && EX_WPB <= 256, "one word per thread of the first waves");
    __shared__ uint32_t wtot[EX_WPB / 64];
    {
        const uint32_t word = w0 + threadIdx.x;
        uint32_t inc = threadIdx.x < EX_WPB && word < nwords ? (uint32_t)__popcll(masks[word]) : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(inc, o, 64);
            inc += lane >= o ? v : 0;
        }
        if (threadIdx.x < EX_WPB) {
            pre[threadIdx.x + 1] = inc;
            if (lane == 63)
                wtot[wave] = inc;
        }
        if (threadIdx.x == 0)
            pre[0] = 0;
        __syncthreads();
        if (threadIdx.x >= 64 && threadIdx.x < EX_WPB) {
            uint32_t add = 0;
            for (int u = 0; u < wave; u++)
                add += wtot[u];
            pre[threadIdx.x + 1] += add;
        }
        __syncthreads();
    }
    if (pre[EX_WPB] == 0)
        return;
    const uint32_t base = blk_off[bx];
    const size_t ys = nx, zs = (size_t)nx * ny;
    // the wave's EX_WPB / 4 (<= 64) mask words: one load, lane i holds word i
    static_assert(EX_WPB / 4 <= 64, "one word per lane");
    unsigned long long mine = 0ull;
    if (lane < EX_WPB / 4 && w0 + wave * (EX_WPB / 4) + lane < nwords)
        mine = masks[w0 + wave * (EX_WPB / 4) + lane];
    // only the words that hold a candidate (a few per cent of them) are visited
    unsigned long long todo = __ballot(mine != 0ull);
    while (todo) {
        const int w = __ffsll((long long)todo) - 1;          // wave-uniform
        todo &= todo - 1ull;
        const int wi = wave * (EX_WPB / 4) + w;
        const uint32_t word = w0 + wi;
        const unsigned long long m =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine >> 32), w) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)mine, w);
        if (!((m >> lane) & 1ull))
            continue;
        const uint32_t pos = base + pre[wi] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (pos >= cap)
            continue;
        const uint32_t row = word / wpr;
        const int x = (int)(word % wpr) * 64 + lane;
        const size_t p = (size_t)x + ys * (row % ny) + zs * (row / ny);
        sift3d_hip_cand c;
        c.idx = (uint32_t)p;
        c.tag = tag;
        c.val = fabsf(FROM_G ? cur[p] - next[p] : cur[p]);     // sift.c:864
        out[pos] = c;
    }
}

template <bool FROM_G>
__global__ __launch_bounds__(256) void k_extrema_emit(ExLevels LV, ExGeom E,
                                                      const unsigned long long *__restrict__ masks,
                                                      const uint32_t *__restrict__ blk_off,
                                                      sift3d_hip_cand *__restrict__ out,
                                                      uint32_t cap)
{
    const int level = blockIdx.y;
    const sift3d_hip_extrema_level L = LV.lv[level];
    ex_emit_block<FROM_G>(L.cur, L.next, L.tag, E.nx, E.ny, E.wpr, E.nwords, blockIdx.x,
                          masks + (size_t)level * E.nwords, blk_off + (size_t)level * E.nblk, out, cap);
}

__global__ __launch_bounds__(256) void k_extrema_emit_multi(ExMulti M, sift3d_hip_cand *__restrict__ out,
                                                            uint32_t cap)
{
    int i = 0;                            // (wave-uniform: blockIdx only)
    while (i + 1 < M.n && blockIdx.x >= M.o[i + 1].blk_first)
        i++;
    const ExOct &O = M.o[i];
    const int level = blockIdx.y;
    ex_emit_block<true>(O.g[level], O.g[level + 1], O.tag0 + level, O.nx, O.ny, O.wpr, O.nwords,
                        blockIdx.x - O.blk_first, O.masks + (size_t)level * O.nwords,
                        O.blk + (size_t)level * O.nblk, out, cap);
}

// ---------------------------------------------------------------------------------------
// window geometry shared by orientation and descriptor (IM_LOOP_SPHERE_START, sift.c:86-107)
// ---------------------------------------------------------------------------------------
// IM_GET_GRAD_ISO (sift.c:140-145, immacros.h:105-111); z is a LOCAL plane index
__device__ __forceinline__ void grad_iso(const sift3d_hip_level &L, int x, int y, int zl, float &gx,
                                         float &gy, float &gz)
{
    const size_t ys = L.nx, zs = (size_t)L.nx * L.ny;
    const float *p = L.data + (size_t)x + ys * y + zs * zl;
    gx = 0.5f * (p[1] - *(p - 1));
    gy = 0.5f * (p[ys] - *(p - ys));
    gz = 0.5f * (p[zs] - *(p - zs));
    gx *= 1.0f / L.ux;
    gy *= 1.0f / L.uy;
    gz *= 1.0f / L.uz;
}

// ---------------------------------------------------------------------------------------
// assign_eig_ori + assign_orientation_thresh  (sift.c:926-1102): one wave per candidate.
//
// The window is walked in the reference's scan order (z, y, x) in chunks of 64 voxels.
// Lanes compute their voxel's nine terms in parallel; the terms are then added in voxel
// order by nine accumulator lanes (six double structure-tensor sums, three float gradient
// sums), which makes every sum bit-identical to the serial CPU loop.
// ---------------------------------------------------------------------------------------
// Lanes of ONE wave exchanging data through LDS: the DS operations of a wave execute in issue order, so a read
// issued after a write sees it -- no s_waitcnt, no s_barrier; the fences only keep the compiler from moving
// LDS accesses across the hand-over point (as wave_sync() in sift3d_describe.hip).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void orient_serial(const sift3d_hip_level *__restrict__ levels,
                                              const sift3d_hip_cand *__restrict__ cand, uint32_t ci,
                                              double corner_thresh, float *__restrict__ Rout,
                                              int32_t *__restrict__ keep)
{
    // (ONE wave per workgroup -- k_orient, k_orient_fix: the hand-overs through LDS below are between lanes of
    // that wave, whose DS operations execute in issue order: wave_lds_sync() only stops the compiler from
    // moving LDS accesses across them.  A workgroup barrier here would also drain the wave's vector-memory
    // queue -- s_waitcnt vmcnt(0) -- i.e. wait for the samples just requested for the NEXT batch.)
    // rows padded by 16 bytes: the accumulator lanes' 16-byte reads fall on different banks
    __shared__ __attribute__((aligned(16))) double td[6][66];
    __shared__ __attribute__((aligned(16))) float tf[3][68];
    const int lane = threadIdx.x;
    const sift3d_hip_cand C = cand[ci];
    const sift3d_hip_level L = levels[C.tag];
    const size_t plane = (size_t)L.nx * L.ny;
    const int kz_loc = (int)(C.idx / plane);
    const int rem = (int)(C.idx % plane);
    const int ky = rem / L.nx, kx = rem % L.nx, kz = kz_loc + L.z_off;
    // vcenter = {key->xd, key->yd, key->zd} as float (sift.c:1124)
    const float cx = (float)kx, cy = (float)ky, cz = (float)kz;
    const double sigma = 1.5 * L.sd;            // ori_sig_fctr, sift.c:1125
    const double rad = sigma * 3.0;             // ori_rad_fctr, sift.c:936
    const double rad2 = rad * rad;
    const double sig2 = sigma * sigma;
    Box B;
    bounds_d(cx, rad, L.ux, L.nx, B.xs, B.xe);
    bounds_d(cy, rad, L.uy, L.ny, B.ys, B.ye);
    bounds_d(cz, rad, L.uz, L.nz_glob, B.zs, B.ze);
    // memory safety on Z-slabs: never outside the local planes (the gradient reads z -+ 1).  A
    // caller whose halo is thinner than the window gets wrong sums, not a fault; the slab driver
    // sizes its halos from sigma0 / units and refuses configurations that do not fit.
    B.zs = max(B.zs, L.z_off + 1);
    B.ze = min(B.ze, L.z_off + L.nz - 2);
    double dacc = 0.0; // lanes 0..5: A00 A01 A02 A11 A12 A22
    float facc = 0.0f; // lanes 6..8: vd_win x y z
    // accumulator lane -> its row of terms and the bytes per voxel in it
    const char *arow = lane < 6 ? reinterpret_cast<const char *>(td[lane])
                                : reinterpret_cast<const char *>(tf[lane < 9 ? lane - 6 : 0]);
    const int astride = lane < 6 ? 8 : 4;
    __shared__ int queue[256];   // in-sphere voxels, window relative, in scan order
    uint32_t qhead = 0, qtail = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // Gaussian window weights.  The centre is a voxel, and when the level's spacing is the same
    // power of two u on all axes (every octave of an isotropic volume) the squared distance of a
    // window voxel is EXACTLY k * u^2 in float for the integer k = i^2 + j^2 + l^2 (each term and
    // each partial sum is an integer times u^2 below 2^24).  So the weight, expf of a double
    // division per voxel in the reference (sift.c:972), takes at most rad^2 / u^2 + 1 values,
    // which are tabulated once per candidate with the reference's expression and looked up by k.
    constexpr int WLUT = 192;
    __shared__ float wlut[WLUT];
    bool use_lut = false;
    {
        int e;
        const float mant = frexpf(L.ux, &e);
        const float u2 = L.ux * L.ux;
        const double kmax = rad2 / (double)u2;
        if (L.ux == L.uy && L.ux == L.uz && mant == 0.5f && kmax < (double)(WLUT - 1) &&
            u2 * (float)WLUT < 16777216.0f) {                     // wave-uniform
            use_lut = true;
            for (int k = lane; k < WLUT; k += 64) {
                const float sq = (float)k * u2;                   // exact
                wlut[k] = s3d_expf((float)(-0.5 * (double)sq / sig2));   // sift.c:972
            }
        }
    }
    wave_lds_sync();

    // 64 queued (in-sphere) voxels: their nine terms in parallel, then added in voxel order by
    // the nine accumulator lanes.  Lanes beyond `cnt` contribute exact zeros (a no-op).
    // The six gradient samples of a batch are REQUESTED one batch ahead (round 5): a batch's serial chain is
    // ~0.3 us of dependent adds, its samples come from L2 / HBM in ~2 us -- requested where they were needed,
    // every batch of the longest window exposed that latency in full, and k_orient_fix lasts as long as its
    // longest window.  request(): the samples and the weight of the batch at queue position `from` into
    // registers; batch(): terms and sums from the registers of an earlier request.
    struct Req {
        float s[6], w;
    };
    auto request = [&](uint32_t from, int cnt, Req &q) {
        q.w = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++)
            q.s[k] = 0.f;
        if (lane < cnt) {
            const int pk = queue[(from + lane) & 255];
            const int x = B.xs + (pk & 1023), y = B.ys + ((pk >> 10) & 1023), z = B.zs + (pk >> 20);
            if (use_lut) {
                const int i = x - kx, j = y - ky, l = z - kz;
                q.w = wlut[min(i * i + j * j + l * l, WLUT - 1)];   // (in-sphere: k <= rad^2 / u^2)
            } else {
                const float dx = ((float)x - cx) * L.ux;          // sift.c:102-104
                const float dy = ((float)y - cy) * L.uy;
                const float dz = ((float)z - cz) * L.uz;
                const float sq = dx * dx + dy * dy + dz * dz;     // sift.c:105
                q.w = s3d_expf((float)(-0.5 * (double)sq / sig2));  // sift.c:972
            }
            const size_t ys = L.nx, zs = (size_t)L.nx * L.ny;
            const float *p = L.data + (size_t)x + ys * y + zs * (z - L.z_off);
            q.s[0] = p[1]; q.s[1] = *(p - 1); q.s[2] = p[ys]; q.s[3] = *(p - ys);
            q.s[4] = p[zs]; q.s[5] = *(p - zs);
        }
    };
    auto batch = [&](int cnt, const Req &q) {
        const bool in = lane < cnt;
        // IM_GET_GRAD_ISO (sift.c:140-145, immacros.h:105-111), as grad_iso()
        float gx = 0.5f * (q.s[0] - q.s[1]), gy = 0.5f * (q.s[2] - q.s[3]), gz = 0.5f * (q.s[4] - q.s[5]);
        gx *= 1.0f / L.ux;
        gy *= 1.0f / L.uy;
        gz *= 1.0f / L.uz;
        const float w = q.w;
        // sift.c:978-987
        td[0][lane] = in ? (double)gx * (double)gx * (double)w : 0.0;
        td[1][lane] = in ? (double)gx * (double)gy * (double)w : 0.0;
        td[2][lane] = in ? (double)gx * (double)gz * (double)w : 0.0;
        td[3][lane] = in ? (double)gy * (double)gy * (double)w : 0.0;
        td[4][lane] = in ? (double)gy * (double)gz * (double)w : 0.0;
        td[5][lane] = in ? (double)gz * (double)gz * (double)w : 0.0;
        tf[0][lane] = in ? gx * w : 0.0f;
        tf[1][lane] = in ? gy * w : 0.0f;
        tf[2][lane] = in ? gz * w : 0.0f;
        wave_lds_sync();
        // The nine accumulator lanes run both serial sums (the double and the float chain
        // interleave and hide each other's latency; only lanes 0..5 / 6..8 hold meaningful rows).
        // The other lanes are masked off, and one pair of 16-byte reads serves both kinds of
        // row (4 voxels of a double row are 32 bytes, of a float row the first 16 of them): an
        // LDS read costs by the instruction and by the bytes it moves.
        // (round 5: the reads of HALF a batch are issued back to back, then its 32 dependent adds run -- left to
        // the compiler every group of four adds waited for its own pair of reads, an LDS round trip sixteen times
        // per batch, and the longest window's batches are what k_orient_fix lasts)
        if (lane < 9) {
#pragma unroll
            for (int h = 0; h < 64; h += 32) {
                double2 u0[8], u1[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    u0[k] = *reinterpret_cast<const double2 *>(arow + (size_t)(h + 4 * k) * astride);
                    u1[k] = *reinterpret_cast<const double2 *>(arow + (size_t)(h + 4 * k) * astride + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // (a float row's four voxels are the four dwords of u0)
                    dacc += u0[k].x; facc += __int_as_float(__double2loint(u0[k].x));
                    dacc += u0[k].y; facc += __int_as_float(__double2hiint(u0[k].x));
                    dacc += u1[k].x; facc += __int_as_float(__double2loint(u0[k].y));
                    dacc += u1[k].y; facc += __int_as_float(__double2hiint(u0[k].y));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wave_lds_sync();
    };

    // Only the (conservative: +0.1 %, against float error of at most 1e-5 relative) bounding
    // rectangle of each plane's disc is scanned; the exact per-voxel test (sift.c:106) and the scan
    // order are unchanged.
    const float rad2f = (float)rad2;
    Req preq;                 // the batch whose samples are in flight (or have arrived)
    bool pend = false;
    request(0, 0, preq);
    for (int z = B.zs; z <= B.ze; z++) {
        const float dz = ((float)z - cz) * L.uz;
        const float rz = sqrtf(fmaxf(rad2f - dz * dz, 0.0f)) * 1.001f;
        const float xr = rz / L.ux, yr = rz / L.uy;
        const int pxs = max(B.xs, (int)floorf(cx - xr)), pxe = min(B.xe, (int)ceilf(cx + xr));
        const int pys = max(B.ys, (int)floorf(cy - yr)), pye = min(B.ye, (int)ceilf(cy + yr));
        const int pbx = pxe - pxs + 1, pby = pye - pys + 1;
        const int ppl = pbx > 0 && pby > 0 ? pbx * pby : 0;
        const int ox = pxs - B.xs, oy = pys - B.ys;
        // lane -> (row, column) of the rectangle, then 64 further per chunk.  The quotients come
        // from a float reciprocal: (i + 0.5) / pbx is at least 0.5 / pbx away from an integer, far
        // more than the rounding error for i <= 64 and pbx <= 1024, so the floor is exact.
        const float rpbx = 1.0f / (float)max(pbx, 1);
        int yy = (int)(((float)lane + 0.5f) * rpbx), xx = lane - yy * pbx;
        const int q64 = (int)(64.5f * rpbx), r64 = 64 - q64 * pbx;
        for (int c0 = 0; c0 < ppl; c0 += 64) {
            bool in = false;
            int pk = 0;
            if (c0 + lane < ppl) {
                const float dx = ((float)(pxs + xx) - cx) * L.ux;
                const float dy = ((float)(pys + yy) - cy) * L.uy;
                const float sq = dx * dx + dy * dy + dz * dz;
                in = !((double)sq > rad2);                        // sift.c:106 (double)
                pk = (ox + xx) | ((oy + yy) << 10) | ((z - B.zs) << 20);
            }
            xx += r64;
            yy += q64;
            if (xx >= pbx) {
                xx -= pbx;
                yy++;
            }
            const unsigned long long m = __ballot(in);
            if (m == 0ull)
                continue;
            if (in)
                queue[(qtail + (uint32_t)__popcll(m & lt_mask)) & 255] = pk;
            qtail += (uint32_t)__popcll(m);
            wave_lds_sync();
            if (qtail - qhead >= 64) {
                // the new batch's samples are requested, THEN the batch before it is summed
                Req nreq;
                request(qhead, 64, nreq);
                if (pend)
                    batch(64, preq);
                preq = nreq;
                pend = true;
                qhead += 64;
            }
        }
    }
    {
        const int rest = (int)(qtail - qhead);
        Req nreq;
        if (rest)
            request(qhead, rest, nreq);
        if (pend)
            batch(64, preq);
        if (rest)
            batch(rest, nreq);
    }
    // gather the nine sums on every lane (uniform epilogue, no divergence)
    double A[9];
    A[0] = __shfl(dacc, 0, 64); A[1] = __shfl(dacc, 1, 64); A[2] = __shfl(dacc, 2, 64);
    A[4] = __shfl(dacc, 3, 64); A[5] = __shfl(dacc, 4, 64); A[8] = __shfl(dacc, 5, 64);
    A[3] = A[1]; A[6] = A[2]; A[7] = A[5];
    const float wx = __shfl(facc, 6, 64), wy = __shfl(facc, 7, 64), wz = __shfl(facc, 8, 64);

    int kept = 1;
    float R[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    if (wx * wx + wy * wy + wz * wz < (float)1E-10) {             // sift.c:997
        kept = 0;
    } else {
        double Q[9], Lm[3];
        s3d_eigen3(A, Q, Lm);                                     // eigen_Mat_rm, imutil.c:984
        if (fabs(Lm[0] / Lm[1]) > 0.90 || fabs(Lm[1] / Lm[2]) > 0.90) { // sift.c:1011-1015
            kept = 0;
        } else {
            double corner = 1.7976931348623157e308;               // DBL_MAX, sift.c:1018
            float v[2][3];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int e = 2 - i;
                float vx = (float)Q[0 * 3 + e], vy = (float)Q[1 * 3 + e], vz = (float)Q[2 * 3 + e];
                const double d = (double)(wx * vx + wy * vy + wz * vz);           // sift.c:1029
                const double cos_ang =
                    d / (double)(sqrtf(vx * vx + vy * vy + vz * vz) *
                                 sqrtf(wx * wx + wy * wy + wz * wz));             // sift.c:1032
                const double ac = fabs(cos_ang);
                corner = corner < ac ? corner : ac;                               // sift.c:1036
                const float sgn = d > 0.0 ? 1.0f : -1.0f;
                vx = vx * sgn; vy = vy * sgn; vz = vz * sgn;
                R[0 * 3 + i] = vx; R[1 * 3 + i] = vy; R[2 * 3 + i] = vz;
                v[i][0] = vx; v[i][1] = vy; v[i][2] = vz;
            }
            R[0 * 3 + 2] = v[0][1] * v[1][2] - v[0][2] * v[1][1];                 // sift.c:1054
            R[1 * 3 + 2] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
            R[2 * 3 + 2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
            if (corner < corner_thresh)                                           // sift.c:1100
                kept = 0;
        }
    }
    if (lane < 9)
        Rout[(size_t)ci * 9 + lane] = R[lane];
    if (lane == 0)
        keep[ci] = kept;
}

#ifdef SIFT3D_AMD_DIAG
__device__ unsigned long long g_orient_undecided;   // candidates re-run by k_orient_fix (profiles/)
#endif

// every candidate with the reference's serial sums (the original path: sift3d_hip_orient, or sift3d_hip_orient_tab without a table)
__global__ __launch_bounds__(64) void k_orient(const sift3d_hip_level *__restrict__ levels,
                                               const sift3d_hip_cand *__restrict__ cand, uint32_t n,
                                               double corner_thresh, float *__restrict__ Rout,
                                               int32_t *__restrict__ keep)
{
    if (blockIdx.x >= n)
        return;
    // candidates arrive in (o, s, z, y, x) order and the window grows with s: walking the list
    // backwards starts the widest windows first (longest-job-first, short kernel tail)
    orient_serial(levels, cand, n - 1 - blockIdx.x, corner_thresh, Rout, keep);
}

// the candidates k_orient_decide left undecided (its list: count, then indices), with the serial
// sums: one wave per window, so the kernel lasts about as long as the longest of them
__global__ __launch_bounds__(64) void k_orient_fix(const sift3d_hip_level *__restrict__ levels,
                                                   const sift3d_hip_cand *__restrict__ cand, uint32_t n,
                                                   double corner_thresh, float *__restrict__ Rout,
                                                   int32_t *__restrict__ keep,
                                                   const uint32_t *__restrict__ undecided)
{
    const uint32_t cnt = min(undecided[0], n);
#ifdef SIFT3D_AMD_DIAG
    if (threadIdx.x == 0 && blockIdx.x == 0)
        atomicAdd(&g_orient_undecided, (unsigned long long)cnt);
#endif
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {      // wave-uniform
        orient_serial(levels, cand, undecided[1 + i], corner_thresh, Rout, keep);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// assign_eig_ori + assign_orientation_thresh with PARALLEL sums and decisions by margin.
//
// The reference adds the window's terms in scan order (sift.c:978-990): six double sums (the
// structure tensor A) and three float sums (the window gradient vd_win).  Reproducing those bits
// needs a serial chain per candidate (orient_serial above: nine lanes work, 55 wait).  But the
// sums only feed (a) three threshold decisions (sift.c:997, 1011-1015, 1100) and (b) the float
// casts of two eigenvectors (sift.c:1025).  So: every lane keeps private double sums of its own
// voxels, a fixed butterfly adds them (reproducible), and each decision is taken only when it
// holds for EVERY value the serial sums can have; otherwise the candidate is marked undecided
// (keep = 2) and k_orient_fix runs it through orient_serial.  What "can have" means:
//   float sums   a serial float sum of n terms differs from the exact sum by at most
//                (n - 1) 2^-24 sum|t_i| (first order; the parallel double sum of the exact
//                products is exact to ~2^-53 relative): e_k = (n + 2) 2^-24 sum|g_k w|, with
//                sum|g_k w| <= sqrt(A_kk sum w) (Cauchy-Schwarz), |e| = the 2-norm of e;
//   tensor       a serial double sum of n terms lies within n 2^-53 sum|terms| of the exact one
//                (one rounding per term, one per add), the parallel one within ~110 2^-53 of it;
//                sum|g_i g_j w| <= (A_ii + A_jj) / 2, so the Frobenius norm of the difference is
//                <= 1.6 n 2^-53 trace A: E = (2 n + 256) 2^-53 trace A (the 256: the parallel
//                sum's and Jacobi's own backward error); eigenvalues move by <= E (Weyl),
//                eigenvector entries by <= 2 E / gap (Davis-Kahan), gap = distance to the
//                nearest other eigenvalue;
//   R            a kept candidate's R is written only if all six eigenvector entries round to the
//                same float over [q - d, q + d], d = 2.5 E / gap + 4e-16, and the sign of
//                vd_win . v cannot flip (|cos| >= margin) -- so R is the serial path's R bit for bit.
// Result: keypoint lists and R identical to the serial kernel (tests: both modes, all fixtures);
// ~1-2 % of the candidates take the second kernel.
// ---------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, the same on every lane, in a fixed order
__device__ __forceinline__ double wave_sum_d(double v)
{
    v += dpp_d<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);       // row_half_mirror
    v += dpp_d<0x140>(v);       // row_mirror: every lane of a row holds the row's sum
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                                       __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16),
                                       __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32),
                                       __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48),
                                       __builtin_amdgcn_readlane(__double2loint(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// Window table of a level: the voxels inside the orientation sphere around a centre voxel, as
// QUADS of up to four x-consecutive voxels, in the reference's scan order (l, j, i ascending),
// with their Gaussian weights.  A keypoint candidate sits ON a voxel, so (float)x - cx is the
// exact integer i and the reference's per-voxel expressions (sift.c:102-106, 972) depend on the
// level alone: every candidate of a level walks the same list (minus what its clipped box cuts
// off) -- no per-plane rectangles, no sphere test, few empty lanes; and a lane that owns four
// consecutive voxels fetches their 24 gradient samples with five 16-byte loads and one 8-byte
// load instead of 24 4-byte gathers (the L1 address path, not arithmetic, bounds this kernel).
// Table t lives at t * ORI_TAB_STRIDE bytes:
//   header   u32 quads (ORI_TAB_NONE: no table -- sphere too large), i32 imin, imax, jmin, jmax, lmin,
//            lmax (extent of the sphere), u32 voxels, at byte 32: double sum of the weights, words 10,
//            11: first and one-past-last candidate of the level (k_orient_groups), word 12: largest
//            i of a quad slot (a row's last quad may reach up to three voxels beyond the sphere),
//            word 13: valid mark, bytes 64..111: the level parameters the table was built for
//   meta     8 bytes per quad from byte ORI_TAB_HEAD: (i0 + 512) | (j + 512) << 10 | (l + 512) << 20
//            | (len - 1) << 30, and i0 + nx * (j + ny * l) (offset in floats from the centre voxel)
//   weights  16 bytes per quad from byte ORI_TAB_HEAD + 8 * ORI_TAB_CAP (0 beyond len)
constexpr uint32_t ORI_TAB_CAP = 8192, ORI_TAB_NONE = 0xffffffffu;
constexpr int ORI_TAB_ROWS = 4096;               // rows (j, l) of the search box
constexpr size_t ORI_TAB_HEAD = 128, ORI_TAB_STRIDE = ORI_TAB_HEAD + (size_t)ORI_TAB_CAP * 24;
constexpr int ORI_CPW = 4;                       // candidates (waves) per workgroup of k_orient_sums
constexpr int ORI_PLAN_MAX = 62;                 // levels with a launch plan (4 words each after the count)

__global__ __launch_bounds__(256) void k_orient_table(const sift3d_hip_level *__restrict__ levels, int lv_lo,
                                                      int lv_hi, unsigned char *__restrict__ tabs)
{
    const int t = lv_lo + (int)blockIdx.x;
    if (t >= lv_hi)
        return;
    const sift3d_hip_level L = levels[t];
    unsigned char *tab = tabs + (size_t)t * ORI_TAB_STRIDE;
    // A table depends on the level's scale, units and row / plane strides only: when the scratch still
    // holds the table of exactly these (the usual case: one detector, one image size), keep it.  The
    // scratch is zeroed when it is allocated, so the signature of a fresh buffer never matches.
    {
        uint32_t *head = reinterpret_cast<uint32_t *>(tab);
        const double *sig = reinterpret_cast<const double *>(tab + 64);
        const bool same = head[13] == 0x53494654u && sig[0] == L.sd && sig[1] == (double)L.ux &&
                          sig[2] == (double)L.uy && sig[3] == (double)L.uz && sig[4] == (double)L.nx &&
                          sig[5] == (double)L.ny;                     // block-uniform
        if (same) {
            if (threadIdx.x == 0)
                head[10] = head[11] = 0;
            return;
        }
    }
    uint2 *meta = reinterpret_cast<uint2 *>(tab + ORI_TAB_HEAD);
    float4 *wts = reinterpret_cast<float4 *>(tab + ORI_TAB_HEAD + (size_t)ORI_TAB_CAP * 8);
    const double sigma = 1.5 * L.sd;            // ori_sig_fctr, sift.c:1125
    const double rad = sigma * 3.0;             // ori_rad_fctr, sift.c:936
    const double rad2 = rad * rad, sig2 = sigma * sigma;
    // half extents of the search box in voxels (+2: safely beyond the sphere)
    const double ex = rad / (double)L.ux + 2.0, ey = rad / (double)L.uy + 2.0, ez = rad / (double)L.uz + 2.0;
    bool fits = ex < 500.0 && ey < 500.0 && ez < 500.0 && ex > 0.0 && ey > 0.0 && ez > 0.0;
    const int mx = fits ? (int)ex : 0, my = fits ? (int)ey : 0, mz = fits ? (int)ez : 0;
    const int wy = 2 * my + 1, wz = 2 * mz + 1, rows = wy * wz;
    fits = fits && rows <= ORI_TAB_ROWS;
    // per row: first in-sphere i and the number of in-sphere voxels (an interval: sq grows with |i|),
    // then the position of the row's first quad
    __shared__ short rfirst[ORI_TAB_ROWS], rlen[ORI_TAB_ROWS];
    __shared__ uint32_t rpos[ORI_TAB_ROWS];
    __shared__ int ext[6], qmax;
    __shared__ uint32_t tot[2];
    __shared__ double wpart[256];
    if (threadIdx.x == 0) {
        ext[0] = ext[2] = ext[4] = 1 << 20;
        ext[1] = ext[3] = ext[5] = qmax = -(1 << 20);
    }
    auto in_sphere = [&](int i, int j, int l, float &sq) -> bool {
        const float dx = (float)i * L.ux, dy = (float)j * L.uy, dz = (float)l * L.uz;   // sift.c:102-104
        sq = dx * dx + dy * dy + dz * dz;                                               // sift.c:105
        return !((double)sq > rad2);                                                    // sift.c:106
    };
    __syncthreads();
    for (int r = threadIdx.x; fits && r < rows; r += 256) {
        const int j = r % wy - my, l = r / wy - mz;
        int first = 0, len = 0;
        for (int i = -mx; i <= mx; i++) {
            float sq;
            if (in_sphere(i, j, l, sq)) {
                if (!len)
                    first = i;
                len++;
            }
        }
        rfirst[r] = (short)first;
        rlen[r] = (short)len;
        if (len) {
            atomicMin(&ext[0], first); atomicMax(&ext[1], first + len - 1);
            atomicMax(&qmax, first + 4 * ((len + 3) / 4) - 1);
            atomicMin(&ext[2], j); atomicMax(&ext[3], j);
            atomicMin(&ext[4], l); atomicMax(&ext[5], l);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t q = 0, v = 0;
        for (int r = 0; fits && r < rows; r++) {
            rpos[r] = q;
            q += (uint32_t)(rlen[r] + 3) / 4;
            v += (uint32_t)rlen[r];
        }
        tot[0] = q;
        tot[1] = v;
    }
    __syncthreads();
    const bool ok = fits && tot[0] <= ORI_TAB_CAP;
    double wacc = 0.0;
    for (int r = threadIdx.x; ok && r < rows; r += 256) {
        const int j = r % wy - my, l = r / wy - mz, first = rfirst[r], len = rlen[r];
        for (int q = 0; 4 * q < len; q++) {
            const int i0 = first + 4 * q, n4 = min(4, len - 4 * q);
            float w[4];
            for (int k = 0; k < 4; k++) {
                float sq;
                in_sphere(i0 + k, j, l, sq);
                w[k] = k < n4 ? s3d_expf((float)(-0.5 * (double)sq / sig2)) : 0.0f;         // sift.c:972
                wacc += (double)w[k];
            }
            meta[rpos[r] + q] = make_uint2((uint32_t)(i0 + 512) | ((uint32_t)(j + 512) << 10) |
                                               ((uint32_t)(l + 512) << 20) | ((uint32_t)(n4 - 1) << 30),
                                           (uint32_t)(i0 + L.nx * (j + L.ny * l)));
            wts[rpos[r] + q] = make_float4(w[0], w[1], w[2], w[3]);
        }
    }
    wpart[threadIdx.x] = wacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double wtot = 0.0;
        for (int k = 0; k < 256; k++)
            wtot += wpart[k];
        uint32_t *head = reinterpret_cast<uint32_t *>(tab);
        head[0] = ok ? tot[0] : ORI_TAB_NONE;
        for (int k = 0; k < 6; k++)
            head[1 + k] = (uint32_t)ext[k];
        head[7] = tot[1];
        *reinterpret_cast<double *>(tab + 32) = wtot;
        head[10] = head[11] = 0;
        head[12] = (uint32_t)qmax;
        double *sig = reinterpret_cast<double *>(tab + 64);
        sig[0] = L.sd; sig[1] = (double)L.ux; sig[2] = (double)L.uy; sig[3] = (double)L.uz;
        sig[4] = (double)L.nx; sig[5] = (double)L.ny;
        __threadfence();
        head[13] = 0x53494654u;                                       // table valid
    }
}

// Launch plan of k_orient_sums.  The candidates arrive sorted by (level, z, y, x).  Workgroups go
// to the eight XCDs in rotation (workgroup b runs on XCD b % 8), and every XCD has its own L2: if
// consecutive candidates went to consecutive workgroups, all eight XCDs would walk the whole
// volume and each would fetch it into its own L2 (measured: 13 GB of L2 fills for 1.8 GB of
// level data -- the fabric, not arithmetic, then bounds the kernel).  So each level's candidates
// are cut into eight contiguous runs (= eight Z slabs of the level) and XCD k takes run k:
// level g owns the workgroups [P, P + 8 ceil(q / CPW)), q = ceil(m / 8), and wave w of workgroup
// P + 8 r + k handles candidate S + k q + CPW r + w (a workgroup is CPW independent waves: 160 000
// one-wave workgroups cost more to dispatch than their windows take to sum).  k_orient_groups finds S and m (first / last candidate of a level),
// k_orient_plan lays the levels out, widest windows (highest level index) first.
__global__ __launch_bounds__(256) void k_orient_groups(const sift3d_hip_cand *__restrict__ cand, uint32_t n,
                                                       unsigned char *__restrict__ tabs, int nlevels)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const int tag = cand[i].tag;
    if (tag < 0 || tag >= nlevels)
        return;
    uint32_t *head = reinterpret_cast<uint32_t *>(tabs + (size_t)tag * ORI_TAB_STRIDE);
    if (i == 0 || cand[i - 1].tag != tag)
        head[10] = i;
    if (i == n - 1 || cand[i + 1].tag != tag)
        head[11] = i + 1;
}

__global__ __launch_bounds__(64) void k_orient_plan(unsigned char *__restrict__ tabs, int lv_lo, int lv_hi,
                                                    uint32_t *__restrict__ plan)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    uint32_t P = 0, G = 0;
    for (int t = lv_hi - 1; t >= lv_lo && G < ORI_PLAN_MAX; t--) {
        const uint32_t *head = reinterpret_cast<const uint32_t *>(tabs + (size_t)t * ORI_TAB_STRIDE);
        const uint32_t S = head[10], m = head[11] - head[10];
        if (!m)
            continue;
        const uint32_t q = (m + 7) / 8;
        plan[1 + 4 * G + 0] = P;
        plan[1 + 4 * G + 1] = S;
        plan[1 + 4 * G + 2] = m;
        plan[1 + 4 * G + 3] = q;
        P += 8 * ((q + ORI_CPW - 1) / ORI_CPW);
        G++;
    }
    plan[0] = G;
}

// OWAVES waves share one candidate (chunk c of its quad list goes to wave c % OWAVES).  Measured with
// 4: no faster than 1 -- neither the sample loads nor the arithmetic of the loop set this kernel's
// time (ablations in profiles/), the per-candidate epilogue did, which is why the decisions now run
// one candidate per LANE in k_orient_decide.
constexpr int OWAVES = 1;
constexpr int ORI_SUMS = 10;   // doubles per candidate: A00 A01 A02 A11 A12 A22, sum g w (x, y, z), voxels
__global__ __launch_bounds__(64 * ORI_CPW) void k_orient_sums(const sift3d_hip_level *__restrict__ levels,
                                                    const sift3d_hip_cand *__restrict__ cand, uint32_t n,
                                                    const unsigned char *__restrict__ tabs,
                                                    const uint32_t *__restrict__ plan,
                                                    double *__restrict__ sums
#ifdef SIFT3D_AMD_DIAG
                                                    , int ablate
#endif
                                                    )
{
    // which candidate: see k_orient_plan (wave-uniform, scalar loads)
    uint32_t ci = 0xffffffffu;
    {
        const uint32_t G = plan[0], b = blockIdx.x;
        const uint32_t wave = threadIdx.x >> 6;
        for (uint32_t g = 0; g < G; g++) {
            const uint32_t P = plan[1 + 4 * g], S = plan[2 + 4 * g], m = plan[3 + 4 * g], q = plan[4 + 4 * g];
            const uint32_t nwg = 8 * ((q + ORI_CPW - 1) / ORI_CPW);
            if (b >= P && b < P + nwg) {
                const uint32_t j = b - P, k = j % 8, r = (j / 8) * ORI_CPW + wave;
                if (r < q && k * q + r < m)
                    ci = S + k * q + r;
                break;
            }
        }
    }
    if (ci >= n)
        return;
    const int lane = threadIdx.x & 63, wv = 0;      // (the waves of a workgroup are independent)
    const sift3d_hip_cand C = cand[ci];
    const sift3d_hip_level L = levels[C.tag];
    const size_t plane = (size_t)L.nx * L.ny;
    const int kz_loc = (int)(C.idx / plane);
    const int rem = (int)(C.idx % plane);
    const int ky = rem / L.nx, kx = rem % L.nx, kz = kz_loc + L.z_off;
    const float cx = (float)kx, cy = (float)ky, cz = (float)kz;   // sift.c:1124
    const double rad = 1.5 * L.sd * 3.0;        // ori_sig_fctr * ori_rad_fctr, sift.c:1125, 936
    Box B;
    bounds_d(cx, rad, L.ux, L.nx, B.xs, B.xe);
    bounds_d(cy, rad, L.uy, L.ny, B.ys, B.ye);
    bounds_d(cz, rad, L.uz, L.nz_glob, B.zs, B.ze);
    B.zs = max(B.zs, L.z_off + 1);              // memory safety on Z-slabs, see orient_serial
    B.ze = min(B.ze, L.z_off + L.nz - 2);
    const unsigned char *tab = tabs + (size_t)C.tag * ORI_TAB_STRIDE;
    const uint32_t *head = reinterpret_cast<const uint32_t *>(tab);
    const uint32_t count = head[0];
    if (count == ORI_TAB_NONE || count == 0) {  // no table for this level: the serial path decides
        if (lane == 0)
            sums[(size_t)ci * ORI_SUMS + 9] = -1.0;
        return;
    }
    // the whole sphere inside the (clipped) window box (sift.c:93-99), and every quad slot (weight 0
    // beyond the sphere) at least one voxel inside the row?  Then no voxel needs a test and every
    // load is safe (the box ends one voxel inside the volume).
    const bool interior = kx + (int)head[1] >= B.xs && kx + (int)head[2] <= B.xe && ky + (int)head[3] >= B.ys &&
                          ky + (int)head[4] <= B.ye && kz + (int)head[5] >= B.zs && kz + (int)head[6] <= B.ze &&
                          kx + (int)head[12] <= L.nx - 2;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned 16-byte load
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    typedef const u2v __attribute__((address_space(1))) *gmeta_p;
    typedef const f4v __attribute__((address_space(1))) *gwts_p;
    typedef const f4u __attribute__((address_space(1))) *gf4_p;
    typedef const f2u __attribute__((address_space(1))) *gf2_p;
    typedef const float __attribute__((address_space(1))) *gfloat_p;
    const gmeta_p meta = (gmeta_p) reinterpret_cast<const u2v *>(tab + ORI_TAB_HEAD);
    const gwts_p wts = (gwts_p) reinterpret_cast<const f4v *>(tab + ORI_TAB_HEAD + (size_t)ORI_TAB_CAP * 8);
    const int ys32 = L.nx, zs32 = L.nx * L.ny;                      // (nx * ny < 2^31)
    const gfloat_p centre = (gfloat_p)L.data + ((uint64_t)(uint32_t)zs32 * (uint32_t)kz_loc + (uint32_t)rem);
    // IM_GET_GRAD_ISO (sift.c:140-145, immacros.h:105-111): g = 0.5f * (v+ - v-), then g *= 1.0f / u.
    // 0.5f * d is exact, so (0.5f * d) * iu == d * (0.5f * iu) bit for bit.
    const float hux = 0.5f * (1.0f / L.ux), huy = 0.5f * (1.0f / L.uy), huz = 0.5f * (1.0f / L.uz);

    double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;   // tensor (sift.c:978-984)
    double vx = 0, vy = 0, vz = 0;                                  // sum g w (sift.c:987), exact products
    double nvox;
    auto add_voxel = [&](float dxv, float dyv, float dzv, float w) {
        const float gx = dxv * hux, gy = dyv * huy, gz = dzv * huz;
        const double dgx = (double)gx, dgy = (double)gy, dgz = (double)gz, dw = (double)w;
        const double wx_ = dgx * dw, wy_ = dgy * dw, wz_ = dgz * dw;     // exact (24 x 24 bits)
        a00 = __builtin_fma(wx_, dgx, a00);
        a01 = __builtin_fma(wx_, dgy, a01);
        a02 = __builtin_fma(wx_, dgz, a02);
        a11 = __builtin_fma(wy_, dgy, a11);
        a12 = __builtin_fma(wy_, dgz, a12);
        a22 = __builtin_fma(wz_, dgz, a22);
        vx += wx_; vy += wy_; vz += wz_;
    };
    // the four voxels at p .. p + 3 with weights w: 24 samples in six loads
    struct Quad {
        f4u ra, yp, ym, zp, zm;
        f2u rb;
    };
    auto load_quad = [&](gfloat_p p, Quad &q) {
        q.ra = *(gf4_p)(p - 1);
        q.rb = *(gf2_p)(p + 3);
        q.yp = *(gf4_p)(p + ys32);
        q.ym = *(gf4_p)(p - ys32);
        q.zp = *(gf4_p)(p + zs32);
        q.zm = *(gf4_p)(p - zs32);
    };
    auto sum_quad = [&](const Quad &q, f4v w) {
        add_voxel(q.ra.z - q.ra.x, q.yp.x - q.ym.x, q.zp.x - q.zm.x, w.x);
        add_voxel(q.ra.w - q.ra.y, q.yp.y - q.ym.y, q.zp.y - q.zm.y, w.y);
        add_voxel(q.rb.x - q.ra.z, q.yp.z - q.ym.z, q.zp.z - q.zm.z, w.z);
        add_voxel(q.rb.y - q.ra.w, q.yp.w - q.ym.w, q.zp.w - q.zm.w, w.w);
    };
    auto add_quad = [&](gfloat_p p, f4v w) {
        Quad q;
#ifdef SIFT3D_AMD_DIAG
        if (ablate & 1) {            // no sample loads (wrong results): what the arithmetic costs
            q.ra = q.yp = q.ym = q.zp = q.zm = f4u{ w.x, w.y, w.z, w.w };
            q.rb = f2u{ w.x, w.y };
            sum_quad(q, w);
            return;
        }
        if (ablate & 2) {            // loads only (wrong results)
            load_quad(p, q);
            vx += (double)(q.ra.x + q.rb.x + q.yp.x + q.ym.x + q.zp.x + q.zm.x + w.x);
            return;
        }
#endif
        load_quad(p, q);
        sum_quad(q, w);
    };
    const f4v zero4 = { 0.f, 0.f, 0.f, 0.f };
    if (interior) {
        // One chunk of 64 quads per iteration, the next chunk's quad offset requested one iteration ahead
        // (the sample addresses come out of it: without the lookahead every chunk would expose two dependent
        // memory round trips).  What bounds the kernel are the sample loads themselves (1.47 of its 1.49 ms
        // with the arithmetic compiled out, 0.74 ms for the arithmetic alone).  Measured 1.89 / 1.96 / 1.98 / 2.05 ms for 1 / 2 / 3 / 4 chunks per
        // iteration: more chunks in flight per wave cost registers, i.e. waves (5 per SIMD at 92 VGPRs;
        // forcing 6-8 waves per SIMD spills: 2.03 / 2.28 / 3.2 ms; a lean launch for the unclipped windows
        // alone fits 6 waves and gains nothing).  Idle lanes repeat the last quad with weight 0.
        const uint32_t last = count - 1;
        auto slot = [&](uint32_t t0) -> uint32_t { return min(t0 + (uint32_t)lane, last); };
        // (Written as plain loads the compiler rotates the loop and uses an entry right after requesting
        // it.  The OFFSET of the next chunk's quad -- all the sample addresses need -- is therefore requested
        // by an instruction the compiler cannot move, and awaited at the end of the iteration; the loads it
        // knows nothing about only make its own s_waitcnt counts conservative.  The weights do not feed an
        // address: they travel with the samples.  `on` starts as a copy of the current offset, so a read
        // before the wait could only repeat a valid address.)
        typedef const uint32_t __attribute__((address_space(1))) *gu32_p;
        const gu32_p moff = (gu32_p) reinterpret_cast<const uint32_t *>(tab + ORI_TAB_HEAD) + 1;   // meta[i].y
        uint32_t oc = moff[2 * slot(0)];
        for (uint32_t t0 = 0; t0 < count; t0 += 64) {
            uint32_t on = oc;
            const gu32_p pn = moff + 2 * slot(t0 + 64);
            asm volatile("global_load_dword %0, %1, off" : "+v"(on) : "v"(pn));
            const f4v wc = wts[slot(t0)];
            Quad q;
            load_quad(centre + (int)oc, q);
            sum_quad(q, t0 + (uint32_t)lane < count ? wc : zero4);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(on));
            oc = on;
        }
        nvox = (double)head[7];
    } else {
        const uint32_t xr = (uint32_t)(B.xe - B.xs), yr = (uint32_t)(B.ye - B.ys), zr = (uint32_t)(B.ze - B.zs);
        const bool box_ok = B.xe >= B.xs && B.ye >= B.ys && B.ze >= B.zs;
        uint32_t cnt = 0;
        for (uint32_t t0 = 64 * wv; t0 < count && box_ok; t0 += 64 * OWAVES) {
            const uint32_t t = t0 + (uint32_t)lane;
            const uint32_t tt = min(t, count - 1);
            const u2v m = meta[tt];
            const f4v w4 = wts[tt];
            const int x0 = kx + (int)(m.x & 1023u) - 512, y = ky + (int)((m.x >> 10) & 1023u) - 512,
                      z = kz + (int)((m.x >> 20) & 1023u) - 512;
            const bool row_in = t < count && (uint32_t)(y - B.ys) <= yr && (uint32_t)(z - B.zs) <= zr;
            float w[4] = { w4.x, w4.y, w4.z, w4.w };
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool in = row_in && (uint32_t)(x0 + k - B.xs) <= xr && w[k] != 0.0f;   // (weights are > 0 inside)
                w[k] = in ? w[k] : 0.0f;
                cnt += in ? 1u : 0u;
                any = any || in;
            }
            // the vector loads touch x0 - 1 .. x0 + 4 of five rows (rows y -+ 1, planes z -+ 1 exist for a
            // row inside the box): safe when all four slots lie at least one voxel inside the row
            const bool safe = !any || (x0 >= 1 && x0 + 3 <= L.nx - 2);
            const gfloat_p p = centre + (any ? (int)m.y : 0);
            if (__ballot(!safe) == 0ull) {
                if (any) {
                    const f4v wv = { w[0], w[1], w[2], w[3] };
                    add_quad(p, wv);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (w[k] != 0.0f) {
                        const gfloat_p q = p + k;
                        add_voxel(q[1] - *(q - 1), q[ys32] - *(q - ys32), q[zs32] - *(q - zs32), w[k]);
                    }
                }
            }
        }
        nvox = wave_sum_d((double)cnt);
    }
    a00 = wave_sum_d(a00); a01 = wave_sum_d(a01); a02 = wave_sum_d(a02);
    a11 = wave_sum_d(a11); a12 = wave_sum_d(a12); a22 = wave_sum_d(a22);
    vx = wave_sum_d(vx); vy = wave_sum_d(vy); vz = wave_sum_d(vz);
    // the sums of this candidate (k_orient_decide takes it from here, one candidate per lane)
    if (lane < ORI_SUMS) {
        const double v = lane == 0 ? a00 : lane == 1 ? a01 : lane == 2 ? a02 : lane == 3 ? a11 : lane == 4 ? a12
                       : lane == 5 ? a22 : lane == 6 ? vx : lane == 7 ? vy : lane == 8 ? vz : nvox;
        sums[(size_t)ci * ORI_SUMS + lane] = v;
    }
}

// The decisions of assign_eig_ori / assign_orientation_thresh on the parallel sums, one candidate
// per lane (the 3x3 eigen-decomposition and the margins cost a few thousand instructions: done by a
// whole wave per candidate they took longer than the window sums themselves).
__global__ __launch_bounds__(64) void k_orient_decide(const sift3d_hip_cand *__restrict__ cand, uint32_t n,
                                                      double corner_thresh, float *__restrict__ Rout,
                                                      int32_t *__restrict__ keep,
                                                      const unsigned char *__restrict__ tabs,
                                                      const double *__restrict__ sums,
                                                      uint32_t *__restrict__ undecided)
{
    const uint32_t ci = blockIdx.x * 64 + threadIdx.x;
    if (ci >= n)
        return;
    const double *sm = sums + (size_t)ci * ORI_SUMS;
    const double a00 = sm[0], a01 = sm[1], a02 = sm[2], a11 = sm[3], a12 = sm[4], a22 = sm[5];
    const double vx = sm[6], vy = sm[7], vz = sm[8], nvox = sm[9];
    if (nvox < 0.0) {                           // no table for this level: the serial path decides
        keep[ci] = 2;
        undecided[1 + atomicAdd(&undecided[0], 1u)] = ci;
        return;
    }
    const double wsum = *reinterpret_cast<const double *>(tabs + (size_t)cand[ci].tag * ORI_TAB_STRIDE + 32);
    // ---- decisions ----
    int kept = 1;          // 0 rejected, 1 kept, 2 undecided
    float R[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const double u24 = 5.9604644775390625e-08, u53 = 1.1102230246251565e-16;
    // sum|g_k w| <= sqrt(sum g_k^2 w * sum w) (Cauchy-Schwarz; sum w over the whole sphere is an
    // upper bound for a clipped window too)
    const double nf = nvox + 2.0;
    const double ex = nf * u24 * sqrt(a00 * wsum), ey = nf * u24 * sqrt(a11 * wsum), ez = nf * u24 * sqrt(a22 * wsum);
    const double enorm = sqrt(ex * ex + ey * ey + ez * ez) * 1.0001 + 1e-300;
    const double vnorm = sqrt(vx * vx + vy * vy + vz * vz);
    const double T1 = (double)(float)1E-10;                       // sift.c:997 (float compare)
    {
        const double lo = fmax(vnorm - enorm, 0.0), hi = vnorm + enorm;
        if (hi * hi * (1.0 + 1e-5) < T1)
            kept = 0;                                             // certainly below
        else if (!(lo * lo * (1.0 - 1e-5) > T1))
            kept = 2;
    }
    if (kept == 1) {
        double A[9], Q[9], Lm[3];
        A[0] = a00; A[1] = a01; A[2] = a02; A[4] = a11; A[5] = a12; A[8] = a22;
        A[3] = a01; A[6] = a02; A[7] = a12;
        s3d_eigen3(A, Q, Lm);                                     // eigen_Mat_rm, imutil.c:984
        const double E = (2.0 * nvox + 256.0) * u53 * (a00 + a11 + a22);
        // sift.c:1011-1015: reject if |L0 / L1| > 0.9 or |L1 / L2| > 0.9
        const double r01 = fabs(Lm[0]) - 0.90 * fabs(Lm[1]), r12 = fabs(Lm[1]) - 0.90 * fabs(Lm[2]);
        const double t2 = 4.0 * E + 1e-14 * fabs(Lm[2]);
        if (r01 > t2 || r12 > t2) {
            kept = 0;
        } else if (!(r01 < -t2 && r12 < -t2)) {
            kept = 2;
        } else {
            const float wx = (float)vx, wy = (float)vy, wz = (float)vz;    // ~ the serial float sums
            const double vlo = fmax(vnorm - enorm, 1e-300);
            const double margin = 2.02 * enorm / vlo + 1e-5;
            double corner = 1.7976931348623157e308;               // DBL_MAX, sift.c:1018
            float v[2][3];
            bool exact = true;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int e = 2 - i;
                const double gap = i == 0 ? Lm[2] - Lm[1] : fmin(Lm[2] - Lm[1], Lm[1] - Lm[0]);
                const double dq = 2.5 * E / fmax(gap, 1e-300) + 4e-16;
                float vf[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double q = Q[k * 3 + e];
                    vf[k] = (float)q;
                    exact = exact && (float)(q - dq) == (float)(q + dq);
                }
                float vx_ = vf[0], vy_ = vf[1], vz_ = vf[2];
                const double d = (double)(wx * vx_ + wy * vy_ + wz * vz_);         // sift.c:1029
                const double cos_ang =
                    d / (double)(sqrtf(vx_ * vx_ + vy_ * vy_ + vz_ * vz_) *
                                 sqrtf(wx * wx + wy * wy + wz * wz));             // sift.c:1032
                const double ac = fabs(cos_ang);
                corner = corner < ac ? corner : ac;                               // sift.c:1036
                const float sgn = d > 0.0 ? 1.0f : -1.0f;
                vx_ = vx_ * sgn; vy_ = vy_ * sgn; vz_ = vz_ * sgn;
                R[0 * 3 + i] = vx_; R[1 * 3 + i] = vy_; R[2 * 3 + i] = vz_;
                v[i][0] = vx_; v[i][1] = vy_; v[i][2] = vz_;
            }
            R[0 * 3 + 2] = v[0][1] * v[1][2] - v[0][2] * v[1][1];                 // sift.c:1054
            R[1 * 3 + 2] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
            R[2 * 3 + 2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
            // sift.c:1100: reject if corner < corner_thresh.  Kept only when the serial value is
            // certainly >= the threshold AND far enough from 0 for the signs above (margin)
            if (corner < corner_thresh - margin)
                kept = 0;
            else if (!(corner >= corner_thresh + margin && corner > margin && exact))
                kept = 2;
        }
    }
    // (Rout / keep may be page-locked host memory: only what the host will read is written)
    if (kept == 1) {
#pragma unroll
        for (int k = 0; k < 9; k++)
            Rout[(size_t)ci * 9 + k] = R[k];
    }
    keep[ci] = kept;
    // the list of the undecided (its order varies from run to run; every entry is computed on its
    // own, so the results do not)
    if (kept == 2)
        undecided[1 + atomicAdd(&undecided[0], 1u)] = ci;
}

// ---------------------------------------------------------------------------------------
// synthetic lattice volume (twin of synth.c:sift3d_amd_synth_lattice_voxel)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t v)
{
    v += 0x9E3779B97F4A7C15ull;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return v ^ (v >> 31);
}

__device__ __forceinline__ float unit_f(uint64_t h, int k)
{
    return (float)((h >> (16 * k)) & 0xFFFF) * (1.0f / 65536.0f);
}

__global__ __launch_bounds__(256) void k_synth_lattice(float *__restrict__ dst, int nx, int ny,
                                                       int nz, int z_off, uint64_t seed)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, zl = blockIdx.z, z = zl + z_off;
    if (x >= nx)
        return;
    const int cell = SIFT3D_AMD_SYNTH_CELL;
    const uint64_t hv = mix64(seed ^ mix64(((uint64_t)(uint32_t)x) | ((uint64_t)(uint32_t)y << 21) |
                                           ((uint64_t)(uint32_t)z << 42)));
    float v = 0.05f * unit_f(hv, 0);
    const int gx = x / cell, gy = y / cell, gz = z / cell;
    for (int iz = gz - 1; iz <= gz + 1; iz++)
        for (int iy = gy - 1; iy <= gy + 1; iy++)
            for (int ix = gx - 1; ix <= gx + 1; ix++) {
                if (ix < 0 || iy < 0 || iz < 0)
                    continue;
                const uint64_t h1 = mix64(seed + 0x51ED270B1ull +
                                          mix64(((uint64_t)ix) | ((uint64_t)iy << 21) |
                                                ((uint64_t)iz << 42)));
                const uint64_t h2 = mix64(h1);
                const float cx = ((float)ix + unit_f(h1, 0)) * (float)cell;
                const float cy = ((float)iy + unit_f(h1, 1)) * (float)cell;
                const float cz = ((float)iz + unit_f(h1, 2)) * (float)cell;
                const float sg = 1.5f + 2.5f * unit_f(h1, 3);
                const float a = 2.0f * unit_f(h2, 0) - 1.0f;
                const float dx = (float)x - cx, dy = (float)y - cy, dz = (float)z - cz;
                const float q = dx * dx + 1.3f * dy * dy + 0.7f * dz * dz;
                if (q > 18.0f * sg * sg)
                    continue;
                v += a * s3d_expf(-q / (2.0f * sg * sg));   // == the host twin's libm expf, bit for bit
            }
    dst[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * zl)] = v;
}

// device evaluation of the shared math, for tests
__global__ void k_test_expf(const float *in, float *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = s3d_expf(in[i]);
}

__global__ void k_test_eigen3(const double *A, double *Q, double *L, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        s3d_eigen3(A + 9 * i, Q + 9 * i, L + 3 * i);
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static int grid_for(size_t n, int per_thread);
// streaming reductions: few, long-running workgroups (one atomic each at the end)
static int grid_reduce(size_t n)
{
    const int b = grid_for(n, 4);
    return b < 256 * 4 ? b : 256 * 4;
}

static int grid_for(size_t n, int per_thread)
{
    size_t b = (n / per_thread + 255) / 256;
    if (b < 1)
        b = 1;
    if (b > 256 * 16)
        b = 256 * 16; // ~16 blocks per CU, grid-stride beyond that
    return (int)b;
}

template <int HW>
static void launch_fir_x_u1(const FirParams &P, const FirTaps &T, const EdgeTab &E, hipStream_t st)
{
    const int nrows = P.ny * (P.z_hi - P.z_lo);
    if ((P.nx & 511) == 0 && ((((uintptr_t)P.src | (uintptr_t)P.dst) & 15) == 0)) {
        // rows of whole segments: two rows in flight per wave
        dim3 grid(P.nx / 512, (nrows + 4 * XROWS_F - 1) / (4 * XROWS_F));
        if (P.scale_max)
            hipLaunchKernelGGL((k_fir_x_u1f<HW, true>), grid, dim3(256), 0, st, P, T, E);
        else
            hipLaunchKernelGGL((k_fir_x_u1f<HW, false>), grid, dim3(256), 0, st, P, T, E);
        return;
    }
    dim3 grid((P.nx + 511) / 512, (nrows + 4 * XROWS - 1) / (4 * XROWS));
    hipLaunchKernelGGL(k_fir_x_u1<HW>, grid, dim3(256), 0, st, P, T, E);
}

template <int HW>
static void launch_fir_sweep_u1(const FirParams &P, const SweepGeom &G, const FirTaps &T,
                                const EdgeTab &E, int V, hipStream_t st)
{
    const int nseg = (G.out_hi - G.out_lo + P.ts - 1) / P.ts;
    dim3 grid((G.ncols + 255) / 256, nseg);
    if (V == 4)
        hipLaunchKernelGGL((k_fir_sweep_u1<HW, 4>), grid, dim3(256), 0, st, P, G, T, E);
    else
        hipLaunchKernelGGL((k_fir_sweep_u1<HW, 1>), grid, dim3(256), 0, st, P, G, T, E);
}

// per-tap (offset, 1-frac, frac) of a dyadic unit factor: the reference's float expressions
// (imutil.c:783-788) evaluated at an index where g -+ hw*uf is exact
static void dyad_table(DyadTaps &D, const float *taps, int width, float uf)
{
    const int hw = width / 2;
    memset(&D, 0, sizeof(D));
    for (int d = -hw; d <= hw; d++) {
        const int g0 = 1 << 10;
        const float c = (float)g0 - (float)d * uf;
        const int lo = (int)c;
        const float frac = c - (float)lo;
        D.k[d + hw] = taps[d + hw];
        D.w0[d + hw] = 1.0f - frac;
        D.w1[d + hw] = frac;
        D.off[d + hw] = lo - g0;
    }
}

template <int HW, int S>
static void launch_fir_dy(const FirParams &P, const SweepGeom &G, const FirTaps &T, int V,
                          hipStream_t st)
{
    if (P.axis == 0) {
        const int nrows = P.ny * (P.z_hi - P.z_lo);
        const bool narrow = P.nx <= 256;
        const unsigned gx = narrow ? (P.nx + 255) / 256 : (P.nx + 511) / 512;
        const size_t nedge = (size_t)nrows * (2 * P.uhw + 1);
        const unsigned eblocks = (unsigned)((nedge + (size_t)256 * gx - 1) / ((size_t)256 * gx));
        dim3 grid(gx, (nrows + 3) / 4 + eblocks);   // interior blocks, then boundary-column blocks
        if (narrow)
            hipLaunchKernelGGL((k_fir_x_dy<HW, S, 4>), grid, dim3(256), 0, st, P, T);
        else
            hipLaunchKernelGGL((k_fir_x_dy<HW, S, 8>), grid, dim3(256), 0, st, P, T);
    } else {
        const int nseg = (G.out_hi - G.out_lo + P.ts - 1) / P.ts;
        dim3 grid((G.ncols + 255) / 256, nseg);
        hipLaunchKernelGGL((k_fir_sweep_dy<HW, S, 4>), grid, dim3(256), 0, st, P, G, T);
    }
}

template <int HW>
static void launch_fir_dyad_hw(const FirParams &P, const SweepGeom &G, const DyadTaps &dt, int V,
                               hipStream_t st)
{
    if (P.axis == 0) {
        dim3 grid((P.nx + 255) / 256, P.ny * (P.z_hi - P.z_lo));
        hipLaunchKernelGGL(k_fir_x_dyad<HW>, grid, dim3(256), 0, st, P, dt);
    } else {
        dim3 grid((G.ncols + 255) / 256, G.out_hi - G.out_lo);
        if (V == 4)
            hipLaunchKernelGGL((k_fir_sweep_dyad<HW, 4>), grid, dim3(256), 0, st, P, G, dt);
        else
            hipLaunchKernelGGL((k_fir_sweep_dyad<HW, 1>), grid, dim3(256), 0, st, P, G, dt);
    }
}

static void launch_fir_dyad_generic(const FirParams &P, const SweepGeom &G, const DyadTaps &dt,
                                    int V, hipStream_t st)
{
    switch (P.hw) {
    case 1: launch_fir_dyad_hw<1>(P, G, dt, V, st); break;
    case 2: launch_fir_dyad_hw<2>(P, G, dt, V, st); break;
    case 3: launch_fir_dyad_hw<3>(P, G, dt, V, st); break;
    case 4: launch_fir_dyad_hw<4>(P, G, dt, V, st); break;
    case 5: launch_fir_dyad_hw<5>(P, G, dt, V, st); break;
    case 6: launch_fir_dyad_hw<6>(P, G, dt, V, st); break;
    case 7: launch_fir_dyad_hw<7>(P, G, dt, V, st); break;
    case 8: launch_fir_dyad_hw<8>(P, G, dt, V, st); break;
    default: launch_fir_dyad_hw<0>(P, G, dt, V, st); break;
    }
}

template <int S>
static bool launch_fir_dy_hw(const FirParams &P, const SweepGeom &G, const FirTaps &T, int V,
                             hipStream_t st)
{
    switch (P.hw) {
    case 1: launch_fir_dy<1, S>(P, G, T, V, st); return true;
    case 2: launch_fir_dy<2, S>(P, G, T, V, st); return true;
    case 3: launch_fir_dy<3, S>(P, G, T, V, st); return true;
    case 4: launch_fir_dy<4, S>(P, G, T, V, st); return true;
    case 5: launch_fir_dy<5, S>(P, G, T, V, st); return true;
    case 6: launch_fir_dy<6, S>(P, G, T, V, st); return true;
    case 7: launch_fir_dy<7, S>(P, G, T, V, st); return true;
    case 8: launch_fir_dy<8, S>(P, G, T, V, st); return true;
    default: return false;
    }
}

template <int NL>
static void launch_dog_stack(const DogStack &S, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL((k_dog_stack<NL>), dim3(grid_reduce(n)), dim3(256), 0, st, S, n);
}

static bool is_dyadic(float uf, int *shift)
{
    int e;
    const float m = frexpf(uf, &e);
    if (m != 0.5f || e > 1)
        return false;
    *shift = 1 - e;
    return true;
}

extern "C" {

int sift3d_hip_absmax(const float *d_src, size_t n, float *d_max, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    hipLaunchKernelGGL(k_absmax, dim3(grid_reduce(n)), dim3(256), 0, (hipStream_t)stream, d_src, n,
                       reinterpret_cast<unsigned *>(d_max));
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_scale(const float *d_src, float *d_dst, size_t n, const float *d_max, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 16)), dim3(256), 0, (hipStream_t)stream, d_src,
                       d_dst, n, d_max);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

static int fir_impl(const sift3d_hip_fir_args *a, const float *d_scale_max, void *stream);

int sift3d_hip_fir(const sift3d_hip_fir_args *a, void *stream) { return fir_impl(a, nullptr, stream); }

// The x pass of a unit-spaced blur on src / *d_max (im_scale, imutil.c:698-713, folded into the pass: the
// scaled image is never stored).  1: not covered (the caller scales first, then calls sift3d_hip_fir).
// (one predicate for the entry's own check and for callers that must know BEFORE they launch: the slab
// driver's ranks have to agree on the path whatever happens to one of them)
int sift3d_hip_fir_x_scaled_covers(const sift3d_hip_fir_args *a)
{
    if (!a)
        return 0;
    const int hw = a->width / 2;
    return !(a->axis != 0 || a->variant == 1 || a->unit_factor != 1.0f || hw < 1 || hw > 8 ||
             a->nx < 2 * hw + 2 || a->nx >= (1 << 22));
}

int sift3d_hip_fir_x_scaled(const sift3d_hip_fir_args *a, const float *d_max, void *stream)
{
    if (!a || !d_max)
        return SIFT3D_FAILURE;
    if (!sift3d_hip_fir_x_scaled_covers(a))
        return 1;
    return fir_impl(a, d_max, stream);
}

static int fir_impl(const sift3d_hip_fir_args *a, const float *d_scale_max, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!a || !a->src || !a->dst || a->nx < 1 || a->ny < 1 || a->nz < 1 || a->axis < 0 ||
        a->axis > 2 || a->width < 1 || !(a->width & 1) || a->width > (1 << 20) ||
        a->z_lo < 0 || a->z_hi > a->nz || a->src == a->dst) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_fir: invalid arguments");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    if (a->z_hi <= a->z_lo)
        return SIFT3D_SUCCESS;
    FirParams P;
    FirTaps T;
    memset(&T, 0, sizeof(T));
    memcpy(T.k, a->taps, sizeof(float) * (a->width < SIFT3D_HIP_MAX_TAPS ? a->width : SIFT3D_HIP_MAX_TAPS));
    P.src = a->src; P.dst = a->dst;
    P.nx = a->nx; P.ny = a->ny; P.nz = a->nz;
    P.axis = a->axis;
    P.hw = a->width / 2;
    P.uf = a->unit_factor;
    P.uhw = (int)ceilf((float)P.hw * P.uf);           // imutil.c:756-757
    const int dims[3] = { a->nx, a->ny, a->nz };
    P.n_glob = a->axis == 2 ? a->n_glob : dims[a->axis];
    P.off = a->axis == 2 ? a->off : 0;
    P.z_lo = a->z_lo; P.z_hi = a->z_hi;
    P.ts = 64;
    P.scale_max = d_scale_max;
    if (a->axis == 2 && (P.off < 0 || P.off + a->nz > P.n_glob)) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_fir: slab outside the global axis");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    const size_t plane = (size_t)a->nx * a->ny;
    if (a->width > SIFT3D_HIP_MAX_TAPS) {
        // wider than the tap tables of the fast kernels: the literal kernel, SIFT3D_HIP_MAX_TAPS taps per launch
        const size_t total = plane * (size_t)(a->z_hi - a->z_lo);
        for (int d_lo = -P.hw; d_lo <= P.hw; d_lo += SIFT3D_HIP_MAX_TAPS) {
            const int d_hi = d_lo + SIFT3D_HIP_MAX_TAPS < P.hw + 1 ? d_lo + SIFT3D_HIP_MAX_TAPS : P.hw + 1;
            memcpy(T.k, a->taps + (d_lo + P.hw), sizeof(float) * (size_t)(d_hi - d_lo));
            hipLaunchKernelGGL(k_fir_literal_chunk, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, T,
                               d_lo, d_hi);
        }
        LAUNCH_CHECK();
        return SIFT3D_SUCCESS;
    }
    int shift = 0;
    const bool dyadic = is_dyadic(P.uf, &shift) && shift <= 12 && P.n_glob < (1 << (23 - shift));
    const bool aligned = (((uintptr_t)a->src | (uintptr_t)a->dst) & 15) == 0;

    // geometry of the strided sweeps (y and z passes)
    SweepGeom G;
    int V = 1;
    if (a->axis == 1) {
        V = (aligned && (a->nx & 3) == 0) ? 4 : 1;
        G.cols_inner = a->nx / V;
        G.ncols = G.cols_inner * (a->z_hi - a->z_lo);
        G.outer_stride = plane;
        G.outer_lo = a->z_lo;
        G.stride = a->nx;
        G.n_loc = a->ny;
        G.out_lo = 0; G.out_hi = a->ny;
    } else if (a->axis == 2) {
        V = (aligned && (plane & 3) == 0) ? 4 : 1;
        G.cols_inner = (int)(plane / V);
        G.ncols = G.cols_inner;
        G.outer_stride = 0;
        G.outer_lo = 0;
        G.stride = plane;
        G.n_loc = a->nz;
        G.out_lo = a->z_lo; G.out_hi = a->z_hi;
    }

    if (a->axis != 0) {
        // sweep segmentation: aim at >= 8 waves per SIMD (8192 waves) without letting the ring
        // warm-up (2*hw extra rows per segment) dominate
        const int n_out = G.out_hi - G.out_lo;
        long want = (8192L * 64 + G.ncols - 1) / (G.ncols > 0 ? G.ncols : 1);
        long cap = n_out / 16 > 1 ? n_out / 16 : 1;
        long nseg = want < cap ? want : cap;
        if (nseg < 1)
            nseg = 1;
        P.ts = (int)((n_out + nseg - 1) / nseg);
        if (P.ts < 1)
            P.ts = 1;
    }
    if (a->variant != 1 && P.uf == 1.0f && P.hw >= 1 && P.hw <= 8 &&
        P.n_glob >= 2 * P.hw + 2 && P.n_glob < (1 << 22)) {
        // unit-spaced taps (octave 0): extended-line register-window kernels
        const EdgeTab E = edge_table(P.n_glob, P.hw);
        if (a->axis == 0) {
            switch (P.hw) {
            case 1: launch_fir_x_u1<1>(P, T, E, st); break;
            case 2: launch_fir_x_u1<2>(P, T, E, st); break;
            case 3: launch_fir_x_u1<3>(P, T, E, st); break;
            case 4: launch_fir_x_u1<4>(P, T, E, st); break;
            case 5: launch_fir_x_u1<5>(P, T, E, st); break;
            case 6: launch_fir_x_u1<6>(P, T, E, st); break;
            case 7: launch_fir_x_u1<7>(P, T, E, st); break;
            default: launch_fir_x_u1<8>(P, T, E, st); break;
            }
        } else {
            switch (P.hw) {
            case 1: launch_fir_sweep_u1<1>(P, G, T, E, V, st); break;
            case 2: launch_fir_sweep_u1<2>(P, G, T, E, V, st); break;
            case 3: launch_fir_sweep_u1<3>(P, G, T, E, V, st); break;
            case 4: launch_fir_sweep_u1<4>(P, G, T, E, V, st); break;
            case 5: launch_fir_sweep_u1<5>(P, G, T, E, V, st); break;
            case 6: launch_fir_sweep_u1<6>(P, G, T, E, V, st); break;
            case 7: launch_fir_sweep_u1<7>(P, G, T, E, V, st); break;
            default: launch_fir_sweep_u1<8>(P, G, T, E, V, st); break;
            }
        }
    } else if (a->variant != 1 && dyadic && (shift == 1 || shift == 2) && P.hw <= 8 &&
               (a->axis == 0 || V == 4) &&
               (shift == 1 ? launch_fir_dy_hw<1>(P, G, T, V, st) : launch_fir_dy_hw<2>(P, G, T, V, st))) {
        // octaves 1 and 2: compile-time tap spacing, register-resident source window
    } else if (a->variant != 1 && dyadic && P.hw < 1024) {
        DyadTaps dt;
        dyad_table(dt, a->taps, a->width, P.uf);
        launch_fir_dyad_generic(P, G, dt, V, st);
    } else {
        const size_t total = plane * (size_t)(a->z_hi - a->z_lo);
        hipLaunchKernelGGL(k_fir_literal, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, P, T);
    }
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_subtract_absmax(const float *d_a, const float *d_b, float *d_dst, size_t n,
                               float *d_absmax, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    hipLaunchKernelGGL(k_sub_absmax, dim3(grid_reduce(n)), dim3(256), 0, (hipStream_t)stream, d_a,
                       d_b, d_dst, n, reinterpret_cast<unsigned *>(d_absmax));
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_dog_stack(const float *const *d_g, float *const *d_d, int n_gauss, size_t n,
                         float *d_absmax, void *stream)
{
    if (n_gauss < 2 || n_gauss > SIFT3D_HIP_MAX_DOG_STACK)
        return 1; // not covered: the caller subtracts level pairs
    if (!n)
        return SIFT3D_SUCCESS;
    DogStack S;
    memset(&S, 0, sizeof(S));
    for (int k = 0; k < n_gauss; k++) {
        S.g[k] = d_g[k];
        if (((uintptr_t)d_g[k] & 15) || (k < n_gauss - 1 && ((uintptr_t)d_d[k] & 15)))
            return 1;
        if (k < n_gauss - 1)
            S.d[k] = d_d[k];
    }
    S.out = reinterpret_cast<unsigned *>(d_absmax);
    hipStream_t st = (hipStream_t)stream;
    switch (n_gauss) {
    case 2: launch_dog_stack<2>(S, n, st); break;
    case 3: launch_dog_stack<3>(S, n, st); break;
    case 4: launch_dog_stack<4>(S, n, st); break;
    case 5: launch_dog_stack<5>(S, n, st); break;
    case 6: launch_dog_stack<6>(S, n, st); break;
    case 7: launch_dog_stack<7>(S, n, st); break;
    default: launch_dog_stack<8>(S, n, st); break;
    }
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_downsample2(const float *d_src, int nx, int ny, float *d_dst, int mx, int my, int mz,
                           void *stream)
{
    if (mx < 1 || my < 1 || mz < 1)
        return SIFT3D_SUCCESS;
    if ((mx & 3) == 0 && (nx & 3) == 0 && nx >= 2 * mx && ((((uintptr_t)d_src | (uintptr_t)d_dst) & 15) == 0))
        hipLaunchKernelGGL(k_downsample2_q, dim3((mx / 4 + 63) / 64, (my + 3) / 4, mz), dim3(256), 0,
                           (hipStream_t)stream, d_src, nx, ny, d_dst, mx / 4, my, mz);
    else
        hipLaunchKernelGGL(k_downsample2, dim3((mx + 255) / 256, my, mz), dim3(256), 0,
                           (hipStream_t)stream, d_src, nx, ny, d_dst, mx, my, mz);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

// The sweeps write EVERY mask word of the planes they test (z_lo <= z < z_hi), zeros included: only the
// words of the planes outside that range (the first and the last plane of a volume, the halo planes of a
// slab) have to be cleared -- not the whole mask (50 MB at 512^3: a 0.28 ms fill per step).
__global__ __launch_bounds__(256) void k_zero_mask_planes(unsigned long long *__restrict__ masks, uint32_t nwords,
                                                          uint32_t wpp, int z_lo, int z_hi, int nz)
{
    const uint32_t nout = (uint32_t)(z_lo + (nz - z_hi)) * wpp;       // words per level outside the range
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nout; i += gridDim.x * 256) {
        const uint32_t pl = i / wpp, w = i - pl * wpp;
        const uint32_t z = pl < (uint32_t)z_lo ? pl : (uint32_t)z_hi + (pl - (uint32_t)z_lo);
        masks[(size_t)blockIdx.y * nwords + (size_t)z * wpp + w] = 0ull;
    }
}

static int zero_mask_planes(unsigned long long *masks, uint32_t nwords, int wpr, int ny, int nz, int z_lo,
                            int z_hi, hipStream_t st)
{
    const uint32_t wpp = (uint32_t)ny * (uint32_t)wpr;
    const long nout = (long)(z_lo + (nz - z_hi)) * wpp;
    if (z_lo < 0 || z_hi > nz || z_hi < z_lo)
        return SIFT3D_FAILURE;
    if (nout > 0) {
        const long nb = (nout + 255) / 256;
        hipLaunchKernelGGL(k_zero_mask_planes, dim3((unsigned)(nb < 512 ? nb : 512), 3), dim3(256), 0, st, masks,
                           nwords, wpp, z_lo, z_hi, nz);
        LAUNCH_CHECK();
    }
    return SIFT3D_SUCCESS;
}

static ExGeom ex_geom(int nx, int ny, int nz, double peak, int cuboid = 0)
{
    ExGeom E;
    E.cuboid = cuboid;
    E.nx = nx; E.ny = ny; E.nz = nz;
    E.wpr = (nx + 63) / 64;
    E.nwords = (uint32_t)((size_t)nz * ny * E.wpr);
    E.nblk = (E.nwords + EX_WPB - 1) / EX_WPB;
    E.peak_thresh = peak;
    return E;
}

size_t sift3d_hip_extrema_work_bytes(int nx, int ny, int nz, int nlevels)
{
    const ExGeom E = ex_geom(nx, ny, nz, 0.0);
    return (size_t)nlevels * ((size_t)E.nwords * 8 + (size_t)E.nblk * 4) + 256;
}

int sift3d_hip_extrema(const sift3d_hip_extrema_level *levels, int nlevels, int nx, int ny, int nz,
                       double peak_thresh, sift3d_hip_cand *d_out, uint32_t cap, uint32_t *d_count,
                       void *d_work, size_t work_bytes, void *stream)
{
    return sift3d_hip_extrema_mode(levels, nlevels, nx, ny, nz, peak_thresh, 0, d_out, cap, d_count,
                                   d_work, work_bytes, stream);
}

int sift3d_hip_extrema_mode(const sift3d_hip_extrema_level *levels, int nlevels, int nx, int ny,
                            int nz, double peak_thresh, int cuboid, sift3d_hip_cand *d_out,
                            uint32_t cap, uint32_t *d_count, void *d_work, size_t work_bytes,
                            void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (nlevels < 1 || nlevels > 8 || (size_t)nx * ny * nz >= (1ull << 32) ||
        work_bytes < sift3d_hip_extrema_work_bytes(nx, ny, nz, nlevels)) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_extrema: invalid arguments");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    const ExGeom E = ex_geom(nx, ny, nz, peak_thresh, cuboid ? 1 : 0);
    ExLevels LV;
    memset(&LV, 0, sizeof(LV));
    for (int i = 0; i < nlevels; i++)
        LV.lv[i] = levels[i];
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(d_work);
    uint32_t *blk = reinterpret_cast<uint32_t *>(masks + (size_t)nlevels * E.nwords);
    // default configuration (three keypoint levels sharing their DoG levels, whole quads): one z
    // sweep over the five DoG levels instead of three scattered-neighbour passes
#ifdef SIFT3D_AMD_DIAG
    static const bool no_sweep = getenv("SIFT3D_AMD_NO_EXSWEEP") != nullptr;   // A/B of the sweep kernel
#else
    const bool no_sweep = false;
#endif
    bool sweep = !E.cuboid && !no_sweep && nlevels == 3 && (nx & 3) == 0 && nz >= 3;
    if (sweep) {
        const float *ptrs[5] = { levels[0].prev, levels[0].cur, levels[1].cur, levels[2].cur, levels[2].next };
        sweep = levels[0].next == levels[1].cur && levels[1].prev == levels[0].cur &&
                levels[1].next == levels[2].cur && levels[2].prev == levels[1].cur &&
                levels[0].z_lo == levels[1].z_lo && levels[1].z_lo == levels[2].z_lo &&
                levels[0].z_hi == levels[1].z_hi && levels[1].z_hi == levels[2].z_hi &&
                levels[0].z_lo >= 1 && levels[0].z_hi <= nz - 1;
        for (int i = 0; i < 5; i++)
            sweep = sweep && (((uintptr_t)ptrs[i]) & 15) == 0;
        if (sweep) {
            ExSweep S;
            memset(&S, 0, sizeof(S));
            for (int i = 0; i < 5; i++)
                S.d[i] = ptrs[i];
            for (int i = 0; i < 3; i++)
                S.absmax[i] = levels[i].d_absmax;
            S.peak_thresh = peak_thresh;
            S.nx = nx; S.ny = ny; S.nz = nz;
            S.z_lo = levels[0].z_lo; S.z_hi = levels[0].z_hi;
            S.wpr = E.wpr; S.nwords = E.nwords;
            S.masks32 = reinterpret_cast<uint32_t *>(masks);
            const int n_out = S.z_hi - S.z_lo;
            if (n_out > 0) {
                if (zero_mask_planes(masks, E.nwords, E.wpr, ny, nz, S.z_lo, S.z_hi, st))
                    return SIFT3D_FAILURE;
            } else {
                HIPCHK(hipMemsetAsync(masks, 0, (size_t)3 * E.nwords * 8, st));
            }
            if (n_out > 0) {
                const long bxy = (long)((nx + 63) / 64) * ((ny + 15) / 16);
                long nseg = (2048 + bxy - 1) / bxy;
                const long cap_seg = n_out / 16 > 1 ? n_out / 16 : 1;
                nseg = nseg < cap_seg ? nseg : cap_seg;
                S.ts = (int)((n_out + nseg - 1) / nseg);
                dim3 grid((nx + 63) / 64, (ny + 15) / 16, (n_out + S.ts - 1) / S.ts);
                hipLaunchKernelGGL(k_extrema_sweep3, grid, dim3(256), 0, st, S);
            }
            hipLaunchKernelGGL(k_extrema_count, dim3(E.nblk, 3), dim3(256), 0, st, masks, E.nwords,
                               E.nblk, blk);
        }
    }
    if (sweep)
        ;
    else if (E.cuboid)
        hipLaunchKernelGGL(k_extrema_mask<true>, dim3(E.nblk, nlevels), dim3(256), 0, st, LV, E, masks,
                           blk);
    else
        hipLaunchKernelGGL(k_extrema_mask<false>, dim3(E.nblk, nlevels), dim3(256), 0, st, LV, E, masks,
                           blk);
    hipLaunchKernelGGL(k_extrema_scan, dim3(1), dim3(1024), 0, st, blk, E.nblk * (uint32_t)nlevels,
                       d_count);
    hipLaunchKernelGGL(k_extrema_emit<false>, dim3(E.nblk, nlevels), dim3(256), 0, st, LV, E, masks, blk,
                       d_out, cap);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_dogmax_stack(const float *const *d_g, int n_gauss, size_t n, float *d_absmax, void *stream)
{
    if (n_gauss < 2 || n_gauss > SIFT3D_HIP_MAX_DOG_STACK)
        return 1; // not covered
    if (!n)
        return SIFT3D_SUCCESS;
    DogStack S;
    memset(&S, 0, sizeof(S));
    for (int k = 0; k < n_gauss; k++) {
        S.g[k] = d_g[k];
        if ((uintptr_t)d_g[k] & 15)
            return 1;
    }
    S.out = reinterpret_cast<unsigned *>(d_absmax);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_reduce(n)), block(256);
    switch (n_gauss) {
    case 2: hipLaunchKernelGGL((k_dogmax_stack<2>), grid, block, 0, st, S, n); break;
    case 3: hipLaunchKernelGGL((k_dogmax_stack<3>), grid, block, 0, st, S, n); break;
    case 4: hipLaunchKernelGGL((k_dogmax_stack<4>), grid, block, 0, st, S, n); break;
    case 5: hipLaunchKernelGGL((k_dogmax_stack<5>), grid, block, 0, st, S, n); break;
    case 6: hipLaunchKernelGGL((k_dogmax_stack<6>), grid, block, 0, st, S, n); break;
    case 7: hipLaunchKernelGGL((k_dogmax_stack<7>), grid, block, 0, st, S, n); break;
    default: hipLaunchKernelGGL((k_dogmax_stack<8>), grid, block, 0, st, S, n); break;
    }
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_extrema_gauss6(const float *const *d_g, const float *d_absmax, int nx, int ny, int nz,
                              int z_lo, int z_hi, int tag0, double peak_thresh, sift3d_hip_cand *d_out,
                              uint32_t cap, uint32_t *d_count, void *d_work, size_t work_bytes,
                              void *stream)
{
    return sift3d_hip_extrema_gauss6_phase(d_g, d_absmax, nx, ny, nz, z_lo, z_hi, tag0, peak_thresh, d_out,
                                           cap, d_count, d_work, work_bytes, stream, 0);
}

// Lower bounds of an octave's five max|DoG| from a sub-lattice of its six Gaussian levels (one fifteenth of the
// bytes), atomically maxed into d_est[0..4] (zeroed by the caller): what sift3d_hip_extrema_gauss6_est_phase
// thresholds its sweep with.  1: not covered.
int sift3d_hip_dogmax_sub(const float *const *d_g, int nx, int ny, int nz, float *d_est, void *stream)
{
    if ((nx & 3) || nx < 4 || ny < 1 || nz < 1)
        return 1;
    DogStack S;
    memset(&S, 0, sizeof(S));
    for (int k = 0; k < 6; k++) {
        S.g[k] = d_g[k];
        if ((uintptr_t)d_g[k] & 15)
            return 1;
    }
    S.out = reinterpret_cast<unsigned *>(d_est);
    const size_t items = (size_t)(nx / 4) * ((ny + SUB_Y - 1) / SUB_Y) * (nz >= 2 ? (nz - 2) / SUB_Z + 1 : 1);
    hipLaunchKernelGGL((k_dogmax_sub<6>), dim3(grid_reduce(4 * items)), dim3(256), 0, (hipStream_t)stream, S, nx,
                       ny, nz);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

static int extrema_gauss6_impl(const float *const *d_g, const float *d_absmax, const float *d_est,
                               float *d_exact, int nx, int ny, int nz, int z_lo, int z_hi, int tag0,
                               double peak_thresh, sift3d_hip_cand *d_out, uint32_t cap, uint32_t *d_count,
                               void *d_work, size_t work_bytes, void *stream, int phase);

// phase 1: the sweep (masks + per-block counts in d_work; independent of every other octave);
// phase 2: scan + emission, which appends to d_out at *d_count (so: in octave order); 0: both
int sift3d_hip_extrema_gauss6_phase(const float *const *d_g, const float *d_absmax, int nx, int ny, int nz,
                                    int z_lo, int z_hi, int tag0, double peak_thresh,
                                    sift3d_hip_cand *d_out, uint32_t cap, uint32_t *d_count, void *d_work,
                                    size_t work_bytes, void *stream, int phase)
{
    return extrema_gauss6_impl(d_g, d_absmax, nullptr, nullptr, nx, ny, nz, z_lo, z_hi, tag0, peak_thresh, d_out,
                               cap, d_count, d_work, work_bytes, stream, phase);
}

// The same stage WITHOUT a separate pass for the maxima: d_est[0..4] are lower bounds of the octave's
// max|DoG| (sift3d_hip_dogmax_sub), the sweep marks every extremum above peak_thresh * bound and gathers the
// exact maxima into d_exact[0..4] (zeroed by the caller before phase 1; the two planes the sweep has no
// centre on are added by two one-plane launches), and the reference's threshold is then applied to the
// marked voxels.  Whole volumes only (z_lo = 1, z_hi = nz - 1: the maxima are those of the planes swept).
int sift3d_hip_extrema_gauss6_est_phase(const float *const *d_g, const float *d_est, float *d_exact, int nx,
                                        int ny, int nz, int tag0, double peak_thresh, sift3d_hip_cand *d_out,
                                        uint32_t cap, uint32_t *d_count, void *d_work, size_t work_bytes,
                                        void *stream, int phase)
{
    if (!d_est || !d_exact)
        return SIFT3D_FAILURE;
    return extrema_gauss6_impl(d_g, d_exact, d_est, d_exact, nx, ny, nz, 1, nz - 1, tag0, peak_thresh, d_out, cap,
                               d_count, d_work, work_bytes, stream, phase);
}

static int extrema_gauss6_impl(const float *const *d_g, const float *d_absmax, const float *d_est,
                               float *d_exact, int nx, int ny, int nz, int z_lo, int z_hi, int tag0,
                               double peak_thresh, sift3d_hip_cand *d_out, uint32_t cap, uint32_t *d_count,
                               void *d_work, size_t work_bytes, void *stream, int phase)
{
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)nx * ny * nz >= (1ull << 32) || work_bytes < sift3d_hip_extrema_work_bytes(nx, ny, nz, 3)) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_extrema_gauss6: invalid arguments");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    // covered: whole quads, aligned levels, at least one interior plane
    if ((nx & 3) || nz < 3 || z_lo < 1 || z_hi > nz - 1)
        return 1;
    for (int i = 0; i < 6; i++)
        if ((uintptr_t)d_g[i] & 15)
            return 1;
    const ExGeom E = ex_geom(nx, ny, nz, peak_thresh, 0);
    unsigned long long *masks = reinterpret_cast<unsigned long long *>(d_work);
    uint32_t *blk = reinterpret_cast<uint32_t *>(masks + (size_t)3 * E.nwords);
    ExSweep S;
    memset(&S, 0, sizeof(S));
    for (int i = 0; i < 6; i++)
        S.d[i] = d_g[i];
    for (int i = 0; i < 3; i++)
        S.absmax[i] = (d_est ? d_est : d_absmax) + 1 + i;       // DoG levels 1..3 are the keypoint levels
    S.exact = reinterpret_cast<unsigned *>(d_exact);
    S.peak_thresh = peak_thresh;
    S.nx = nx; S.ny = ny; S.nz = nz;
    S.z_lo = z_lo; S.z_hi = z_hi;
    S.wpr = E.wpr; S.nwords = E.nwords;
    S.masks32 = reinterpret_cast<uint32_t *>(masks);
    const int n_out = z_hi - z_lo;
    if (phase != 2) {
        if (n_out > 0) {
            if (zero_mask_planes(masks, E.nwords, E.wpr, ny, nz, z_lo, z_hi, st))
                return SIFT3D_FAILURE;
        } else {
            HIPCHK(hipMemsetAsync(masks, 0, (size_t)3 * E.nwords * 8, st));
        }
    }
    if (n_out > 0 && phase != 2) {
        // tile width: a whole wave per row where the rows are long enough -- 1 KB row segments
        // (measured at 512^3, the sweep alone: 0.58 ms against 0.66 / 0.80 with 512 / 256-byte segments,
        // although the 4-row tiles re-read more halo rows)
        const int txq = nx >= 256 ? 64 : nx >= 128 ? 32 : 16, tyy = 256 / txq;
        const long bxy = (long)((nx + 4 * txq - 1) / (4 * txq)) * ((ny + tyy - 1) / tyy);
        long nseg = (2048 + bxy - 1) / bxy;
        const long cap_seg = n_out / 16 > 1 ? n_out / 16 : 1;
        nseg = nseg < cap_seg ? nseg : cap_seg;
        S.ts = (int)((n_out + nseg - 1) / nseg);
        dim3 grid((nx + 4 * txq - 1) / (4 * txq), (ny + tyy - 1) / tyy, (n_out + S.ts - 1) / S.ts);
        if (d_est) {
            if (txq == 32)
                hipLaunchKernelGGL((k_extrema_sweep3g<32, true>), grid, dim3(256), 0, st, S);
            else if (txq == 64)
                hipLaunchKernelGGL((k_extrema_sweep3g<64, true>), grid, dim3(256), 0, st, S);
            else
                hipLaunchKernelGGL((k_extrema_sweep3g<16, true>), grid, dim3(256), 0, st, S);
        } else if (txq == 32)
            hipLaunchKernelGGL((k_extrema_sweep3g<32>), grid, dim3(256), 0, st, S);
        else if (txq == 64)
            hipLaunchKernelGGL((k_extrema_sweep3g<64>), grid, dim3(256), 0, st, S);
        else
            hipLaunchKernelGGL((k_extrema_sweep3g<16>), grid, dim3(256), 0, st, S);
    }
    if (d_est && phase != 2) {
        // the first and the last plane, then the reference's threshold on the marked voxels
        const size_t plane = (size_t)nx * ny;
        const float *g0[6], *g1[6];
        for (int i = 0; i < 6; i++) {
            g0[i] = d_g[i];
            g1[i] = d_g[i] + (size_t)(nz - 1) * plane;
        }
        if (sift3d_hip_dogmax_stack(g0, 6, plane, d_exact, stream) != SIFT3D_SUCCESS ||
            sift3d_hip_dogmax_stack(g1, 6, plane, d_exact, stream) != SIFT3D_SUCCESS)
            return SIFT3D_FAILURE;
        hipLaunchKernelGGL(k_extrema_refilter, dim3((unsigned)(((size_t)3 * E.nwords + 255) / 256)), dim3(256), 0,
                           st, S);
    }
    if (phase != 2)
        hipLaunchKernelGGL(k_extrema_count, dim3(E.nblk, 3), dim3(256), 0, st, masks, E.nwords, E.nblk, blk);
    if (phase == 1) {
        LAUNCH_CHECK();
        return SIFT3D_SUCCESS;
    }
    hipLaunchKernelGGL(k_extrema_scan, dim3(1), dim3(1024), 0, st, blk, E.nblk * 3u, d_count);
    ExLevels LV;
    memset(&LV, 0, sizeof(LV));
    for (int i = 0; i < 3; i++) {
        LV.lv[i].cur = d_g[i + 1];            // DoG level i + 1 = G[i + 1] - G[i + 2]
        LV.lv[i].next = d_g[i + 2];
        LV.lv[i].tag = tag0 + i;
    }
    hipLaunchKernelGGL(k_extrema_emit<true>, dim3(E.nblk, 3), dim3(256), 0, st, LV, E, masks, blk, d_out,
                       cap);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

// Phase 2 of sift3d_hip_extrema_gauss6_[est_]phase for ALL octaves of a call at once: one scan launch over the
// octaves' block counts in order, one emission launch over every octave's blocks; appends at *d_count in
// (octave, level, z, y, x) order -- the reference's (sift.c:835-868).  1: more octaves than one launch takes
// (the caller then issues phase 2 per octave).
int sift3d_hip_extrema_gauss6_finish(const sift3d_hip_extrema_oct *octs, int n_oct, double peak_thresh,
                                     sift3d_hip_cand *d_out, uint32_t cap, uint32_t *d_count, void *stream)
{
    if (n_oct < 1 || n_oct > EX_MAX_OCT)
        return 1;
    ExMulti M;
    memset(&M, 0, sizeof(M));
    M.n = n_oct;
    uint32_t nb = 0;
    for (int i = 0; i < n_oct; i++) {
        const sift3d_hip_extrema_oct &q = octs[i];
        if ((size_t)q.nx * q.ny * q.nz >= (1ull << 32) ||
            q.work_bytes < sift3d_hip_extrema_work_bytes(q.nx, q.ny, q.nz, 3)) {
            snprintf(g_err, sizeof(g_err), "sift3d_hip_extrema_gauss6_finish: invalid arguments");
            fprintf(stderr, "sift3d_amd: %s\n", g_err);
            return SIFT3D_FAILURE;
        }
        const ExGeom E = ex_geom(q.nx, q.ny, q.nz, peak_thresh, 0);
        ExOct &O = M.o[i];
        for (int k = 0; k < 4; k++)
            O.g[k] = q.d_g[k + 1];
        unsigned long long *masks = reinterpret_cast<unsigned long long *>(q.d_work);
        O.masks = masks;
        O.blk = reinterpret_cast<uint32_t *>(masks + (size_t)3 * E.nwords);
        O.nx = q.nx; O.ny = q.ny; O.wpr = E.wpr;
        O.nwords = E.nwords; O.nblk = E.nblk;
        O.tag0 = q.tag0;
        O.blk_first = nb;
        nb += E.nblk;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_extrema_scan_multi, dim3(1), dim3(1024), 0, st, M, d_count);
    hipLaunchKernelGGL(k_extrema_emit_multi, dim3(nb, 3), dim3(256), 0, st, M, d_out, cap);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

#ifdef SIFT3D_AMD_DIAG
// diagnostic build only: candidates k_orient_fix re-ran since the last call
__attribute__((visibility("default"))) unsigned long long sift3d_amd_diag_orient_undecided(void)
{
    unsigned long long v = 0, z = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_orient_undecided), sizeof(v));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_orient_undecided), &z, sizeof(z));
    return v;
}
#endif

int sift3d_hip_orient(const sift3d_hip_level *d_levels, const sift3d_hip_cand *d_cand, uint32_t n,
                      double corner_thresh, float *d_R, int32_t *d_keep, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    hipLaunchKernelGGL(k_orient, dim3(n), dim3(64), 0, (hipStream_t)stream, d_levels, d_cand, n,
                       corner_thresh, d_R, d_keep);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

// tables + two launch plans (rounded up to 256 bytes), then ORI_SUMS doubles per candidate
constexpr size_t ORI_PLAN_BYTES = 4 * (1 + 4 * (size_t)ORI_PLAN_MAX);
static size_t orient_tab_head_bytes(int nlevels)
{
    return (((size_t)nlevels * ORI_TAB_STRIDE + 2 * ORI_PLAN_BYTES) + 255) & ~(size_t)255;
}

// ... then ORI_SUMS doubles per candidate, then two lists of the undecided (count + indices each)
size_t sift3d_hip_orient_tab_bytes(int nlevels, uint32_t max_cand)
{
    return nlevels > 0 ? orient_tab_head_bytes(nlevels) + sizeof(double) * ORI_SUMS * (size_t)max_cand +
                             2 * sizeof(uint32_t) * ((size_t)max_cand + 1)
                       : 0;
}

int sift3d_hip_orient_tab_part(const sift3d_hip_level *d_levels, int nlevels, int lv_lo, int lv_hi,
                               const sift3d_hip_cand *d_cand, uint32_t first, uint32_t n, double corner_thresh,
                               float *d_R, int32_t *d_keep, void *d_tab, uint32_t max_cand, int slot, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    if (lv_lo < 0 || lv_hi > nlevels || lv_lo >= lv_hi || slot < 0 || slot > 1) {
        snprintf(g_err, sizeof(g_err), "sift3d_hip_orient_tab_part: invalid arguments");
        fprintf(stderr, "sift3d_amd: %s\n", g_err);
        return SIFT3D_FAILURE;
    }
    // everything below sees the part as a list of its own: candidate i of the part is candidate first + i
    const sift3d_hip_cand *cand = d_cand + first;
    float *R = d_R + (size_t)9 * first;
    int32_t *keep = d_keep + first;
    if (!d_tab || nlevels > ORI_PLAN_MAX || (uint64_t)first + n > max_cand)
        return sift3d_hip_orient(d_levels, cand, n, corner_thresh, R, keep, stream);
    // window tables of the part's levels, parallel sums with decisions by margin, then the undecided
    // candidates with the serial sums
    hipStream_t st = (hipStream_t)stream;
    unsigned char *tabs = (unsigned char *)d_tab;
    uint32_t *plan = reinterpret_cast<uint32_t *>(tabs + (size_t)nlevels * ORI_TAB_STRIDE + (size_t)slot * ORI_PLAN_BYTES);
    hipLaunchKernelGGL(k_orient_table, dim3(lv_hi - lv_lo), dim3(256), 0, st, d_levels, lv_lo, lv_hi, tabs);
    hipLaunchKernelGGL(k_orient_groups, dim3((n + 255) / 256), dim3(256), 0, st, cand, n, tabs, nlevels);
    hipLaunchKernelGGL(k_orient_plan, dim3(1), dim3(64), 0, st, tabs, lv_lo, lv_hi, plan);
    // (every level's share of the grid is rounded up to a multiple of 8 workgroups)
    double *d_sums = reinterpret_cast<double *>(tabs + orient_tab_head_bytes(nlevels)) + (size_t)ORI_SUMS * first;
    uint32_t *d_und = reinterpret_cast<uint32_t *>(reinterpret_cast<double *>(tabs + orient_tab_head_bytes(nlevels)) +
                                                   (size_t)ORI_SUMS * max_cand) +
                      (size_t)slot * ((size_t)max_cand + 1);
    HIPCHK(hipMemsetAsync(d_und, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_orient_sums, dim3((n + ORI_CPW - 1) / ORI_CPW + 16 * (uint32_t)(lv_hi - lv_lo)),
                       dim3(64 * ORI_CPW), 0, st, d_levels, cand, n, (const unsigned char *)tabs,
                       (const uint32_t *)plan, d_sums
#ifdef SIFT3D_AMD_DIAG
                       , getenv("SIFT3D_AMD_ORI_ABLATE") ? atoi(getenv("SIFT3D_AMD_ORI_ABLATE")) : 0
#endif
                       );
    hipLaunchKernelGGL(k_orient_decide, dim3((n + 63) / 64), dim3(64), 0, st, cand, n, corner_thresh, R, keep,
                       (const unsigned char *)tabs, d_sums, d_und);
    hipLaunchKernelGGL(k_orient_fix, dim3(n < 8192u ? n : 8192u), dim3(64), 0, st, d_levels, cand, n, corner_thresh,
                       R, keep, d_und);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_orient_tab(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_cand *d_cand,
                          uint32_t n, double corner_thresh, float *d_R, int32_t *d_keep, void *d_tab,
                          uint32_t max_cand, void *stream)
{
    if (!n)
        return SIFT3D_SUCCESS;
    if (!d_tab || nlevels < 1 || nlevels > ORI_PLAN_MAX || n > max_cand)
        return sift3d_hip_orient(d_levels, d_cand, n, corner_thresh, d_R, d_keep, stream);
    return sift3d_hip_orient_tab_part(d_levels, nlevels, 0, nlevels, d_cand, 0, n, corner_thresh, d_R, d_keep, d_tab,
                                      max_cand, 0, stream);
}

int sift3d_hip_synth_lattice(float *d_dst, int nx, int ny, int nz, int z_off, uint64_t seed,
                             void *stream)
{
    hipLaunchKernelGGL(k_synth_lattice, dim3((nx + 255) / 256, ny, nz), dim3(256), 0,
                       (hipStream_t)stream, d_dst, nx, ny, nz, z_off, seed);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

void sift3d_amd_host_expf(const float *in, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++)
        out[i] = s3d_expf(in[i]);
}

void sift3d_amd_host_eigen3(const double *A9, double *Q9, double *L3) { s3d_eigen3(A9, Q9, L3); }

int sift3d_hip_test_expf(const float *d_in, float *d_out, size_t n, void *stream)
{
    hipLaunchKernelGGL(k_test_expf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_in, d_out, n);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

int sift3d_hip_test_eigen3(const double *d_A9, double *d_Q9, double *d_L3, size_t n, void *stream)
{
    hipLaunchKernelGGL(k_test_eigen3, dim3((unsigned)((n + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, d_A9, d_Q9, d_L3, n);
    LAUNCH_CHECK();
    return SIFT3D_SUCCESS;
}

} // extern "C"

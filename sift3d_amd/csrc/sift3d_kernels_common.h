// sift3d_kernels_common.h -- declarations shared by the translation units of the device code
// (sift3d_kernels.hip; sift3d_fir_yz.hip, which is compiled with other optimiser settings).
#ifndef SIFT3D_KERNELS_COMMON_H
#define SIFT3D_KERNELS_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

// Part of the numerical contract: no fused multiply-add anywhere (see sift3d_kernels.hip).
#pragma clang fp contract(off)

#include "../../include/sift3d_amd.h"

// ---- error plumbing (defined in sift3d_kernels.hip) ----------------------------------------
extern thread_local char g_err[512];
int fail(const char *what, hipError_t e, const char *file, int line);

#define HIPCHK(call)                                                  \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess)                                         \
            return fail(#call, e_, __FILE__, __LINE__);               \
    } while (0)

#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

// ---- small device helpers ------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- 1-D interpolating FIR (convolve_sep_gen, imutil.c:742-861): shared types -----------------
struct FirTaps {
    float k[SIFT3D_HIP_MAX_TAPS];
};

struct FirParams {
    const float *src;
    float *dst;
    int nx, ny, nz;   // local dims
    int axis;
    int hw;           // half width
    float uf;         // unit factor
    int uhw;          // (int)ceilf(hw*uf), imutil.c:756-757
    int n_glob, off;  // along the filtered axis
    int z_lo, z_hi;   // output planes
    int ts;           // sweep segment length (sweep kernels)
    const float *scale_max;   // k_fir_x_u1 only: every sample is divided by *scale_max first (im_scale,
                              // imutil.c:698-713, folded into the first pass of the pyramid); or null
};

// High-edge samples of the extended line of a unit-spaced pass (see "unit factor 1" in
// sift3d_kernels.hip)
struct EdgeTab {
    int lo[9];
    float w0[9], w1[9];
};

template <int V> struct Vec;
template <> struct Vec<4> {
    typedef float4 T;
    static __device__ __forceinline__ T ld(const float *p) { return ld4(p); }
    static __device__ __forceinline__ void st(float *p, T v) { st4(p, v); }
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void mac(T &acc, float k, const T &v)
    {
        acc.x += k * v.x; acc.y += k * v.y; acc.z += k * v.z; acc.w += k * v.w;
    }
    static __device__ __forceinline__ T lerp(float w0, const T &a, float w1, const T &b)
    {
        return make_float4(w0 * a.x + w1 * b.x, w0 * a.y + w1 * b.y, w0 * a.z + w1 * b.z,
                           w0 * a.w + w1 * b.w);
    }
};
template <> struct Vec<1> {
    typedef float T;
    static __device__ __forceinline__ T ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.0f; }
    static __device__ __forceinline__ void mac(T &acc, float k, const T &v) { acc += k * v; }
    static __device__ __forceinline__ T lerp(float w0, const T &a, float w1, const T &b)
    {
        return w0 * a + w1 * b;
    }
};

// High-edge table of the extended line (imutil.c:846-848 + :783-788), reference float ops.
static inline EdgeTab edge_table(int n_glob, int hw)
{
    EdgeTab E;
    memset(&E, 0, sizeof(E));
    const int dim_end = n_glob - 1;
    for (int m = 0; m <= hw && m < 9; m++) {
        float c = (float)(dim_end + m);            // (float)x - d, an exact integer
        c = 2.0f * (float)dim_end - c - 0.1f;      // conv_eps mirror
        const int lo = (int)c;
        const float frac = c - (float)lo;
        E.lo[m] = lo;
        E.w0[m] = 1.0f - frac;
        E.w1[m] = frac;
    }
    return E;
}

// ---- window geometry shared by orientation and descriptor (IM_LOOP_SPHERE_START, sift.c:86-107)
struct Box {
    int xs, xe, ys, ye, zs, ze; // inclusive, global z
};

// rad is double in assign_eig_ori (sift.c:936) and float in extract_descrip (sift.c:1454);
// the macro's expressions are promoted accordingly before floorf/ceilf.
__device__ __forceinline__ void bounds_d(float c, double rad, float u, int n, int &s, int &e)
{
    const float lo = floorf((float)((double)c - rad / (double)u));
    const float hi = ceilf((float)((double)c + rad / (double)u));
    s = (int)(lo > 1.0f ? lo : 1.0f);
    e = (int)(hi < (float)(n - 2) ? hi : (float)(n - 2));
}

__device__ __forceinline__ void bounds_f(float c, float rad, float u, int n, int &s, int &e)
{
    const float lo = floorf(c - rad / u);
    const float hi = ceilf(c + rad / u);
    s = (int)(lo > 1.0f ? lo : 1.0f);
    e = (int)(hi < (float)(n - 2) ? hi : (float)(n - 2));
}

#endif

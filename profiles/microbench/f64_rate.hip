// Issue rate of double-precision vector instructions on gfx950 (cycles per wave64 instruction per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip && ./f64_rate
// 8 independent accumulator chains per lane, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// OP: 0 v_fma_f64, 1 v_mul_f64, 2 v_add_f64, 3 v_cvt_f64_f32, 4 v_cvt_f32_f64, 5 v_fma_f32, 6 v_add_f32
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed)
{
    double a[8];
    float f[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; }
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"((float)m), "v"((float)c));
                if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"((float)c));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char *name, double *d_o)
{
    for (int bpc = 1; bpc <= 8; bpc *= 2) {            // blocks of 4 waves per CU: 1, 2, 4, 8 waves per SIMD
        const int iters = 4000;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<OP>), dim3(256 * bpc), dim3(256), 0, 0, d_o, iters, 1.0);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double instr_per_simd = (double)bpc * iters * 32;
        printf("%-16s %d waves/SIMD: %8.3f ms -> %6.2f cycles / instruction / SIMD (2.4 GHz)\n", name, bpc, ms,
               ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
}

int main()
{
    double *d_o; CHK(hipMalloc(&d_o, 256 * 8 * 256 * 8));
    run<5>("v_fma_f32", d_o);
    run<6>("v_add_f32", d_o);
    run<0>("v_fma_f64", d_o);
    run<1>("v_mul_f64", d_o);
    run<2>("v_add_f64", d_o);
    run<3>("v_cvt_f64_f32", d_o);
    run<4>("v_cvt_f32_f64", d_o);
    return 0;
}

// Sustained rate of v_mfma_f32_32x32x2_f32 on gfx950: what a kernel made of nothing but independent MFMAs
// reaches, against the 157.3 TFLOP/s the data sheet clock (2.4 GHz) gives -- the ceiling k_nn2 is priced
// against (bench.py --register).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++)
            acc[i][r] = seed + i + r;
    const float a = seed + threadIdx.x, b = seed - threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++)
            s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC> void run(float *d_o)
{
    for (int bpc = 1; bpc <= 2; bpc++) {                // workgroups of 4 waves per CU: 1, 2 waves per SIMD
        const int iters = 4000;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<NACC>), dim3(256 * bpc), dim3(256), 0, 0, d_o, iters, 1.0f);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double n_mfma = 256.0 * bpc * 4 * iters * 8 * NACC;     // wave-level MFMAs
        const double flops = n_mfma * 32 * 32 * 2 * 2;
        printf("%d accumulators, %d waves/SIMD: %8.3f ms -> %7.1f TFLOP/s, %6.1f cycles per MFMA per SIMD at 2.4 GHz\n", NACC,
               bpc, ms, flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
    }
}

int main()
{
    float *d_o; CHK(hipMalloc(&d_o, 256 * 2 * 256 * 4));
    run<1>(d_o);
    run<2>(d_o);
    run<4>(d_o);
    return 0;
}

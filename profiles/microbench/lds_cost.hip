// LDS pipe cost of one instruction as a function of width, active lanes and address pattern (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// W: bytes per lane (4, 8, 16); PAT: 0 consecutive, 1 three addresses by (lane % 3), 2 one row per lane (stride 528 B),
// 3 all lanes same address;  NACT: active lanes (lanes with (lane & 31) < NACT32 if HALF else lane < NACT)
template <int W, int PAT, int NACT, bool HALF, bool WRITE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) char buf[16384 + 64 * 528 + 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (int)sizeof(buf) / 4; i += 256)
        reinterpret_cast<float *>(buf)[i] = (float)i;
    __syncthreads();
    const bool on = HALF ? (lane & 31) < NACT : lane < NACT;
    int base;
    if (PAT == 0) base = lane * (W < 4 ? 4 : W);
    else if (PAT == 1) base = (lane % 3) * 1040;
    else if (PAT == 2) base = lane * 528;
    else base = 0;
    base += wave * 64;
    float acc = 0.f;
    if (on) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int a = base + ((it * 16 + u) & 7) * 16 * (PAT == 0 ? 64 : 1);
                if (W == 2) {
                    if (WRITE) asm volatile("ds_write2_b32 %0, %1, %1 offset1:36" :: "v"(a), "v"(acc) : "memory");
                    else { f2v v; asm volatile("ds_read2_b32 %0, %1 offset1:36" : "=v"(v) : "v"(a)); acc += v.x; }
                } else if (W == 1) {
                    float v; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v) : "v"(a), "v"(acc)); acc += v;
                } else if (WRITE) {
                    if (W == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(acc) : "memory");
                    if (W == 8) { f2v v = {acc, acc}; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }
                    if (W == 16) { f4v v = {acc, acc, acc, acc}; asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory"); }
                } else {
                    if (W == 4) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); acc += v; }
                    if (W == 8) { f2v v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); acc += v.x; }
                    if (W == 16) { f4v v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); acc += v.x; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int W, int PAT, int NACT, bool HALF, bool WRITE>
void run(const char *name, float *d_o)
{
    const int nblk = 256 * 4, iters = 2000;    // 4 blocks of 4 waves per CU
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<W, PAT, NACT, HALF, WRITE>), dim3(nblk), dim3(256), 0, 0, d_o, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double instr_per_cu = 16.0 * iters * 16;     // 16 waves x iters x 16
    printf("%-44s %.3f ms  -> %.2f cycles / instruction / CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main()
{
    float *d_o; CHK(hipMalloc(&d_o, 256 * 4 * 256 * 4));
    run<16, 0, 64, false, false>("read b128 64 lanes consecutive", d_o);
    run<16, 1, 64, false, false>("read b128 64 lanes, 3 addresses", d_o);
    run<16, 1, 24, true, false>("read b128 48 lanes (24 per half), 3 addresses", d_o);
    run<16, 3, 64, false, false>("read b128 64 lanes, 1 address", d_o);
    run<16, 2, 9, false, false>("read b128 9 lanes, own rows", d_o);
    run<16, 2, 36, false, false>("read b128 36 lanes, own rows", d_o);
    run<16, 2, 64, false, false>("read b128 64 lanes, own rows", d_o);
    run<8, 0, 64, false, false>("read b64 64 lanes consecutive", d_o);
    run<8, 2, 9, false, false>("read b64 9 lanes own rows", d_o);
    run<4, 0, 64, false, false>("read b32 64 lanes consecutive", d_o);
    run<4, 0, 24, true, false>("read b32 48 lanes consecutive", d_o);
    run<4, 0, 32, false, false>("read b32 32 lanes consecutive", d_o);
    run<4, 3, 64, false, false>("read b32 64 lanes 1 address", d_o);
    run<4, 0, 64, false, true>("write b32 64 lanes consecutive", d_o);
    run<4, 0, 24, true, true>("write b32 48 lanes consecutive", d_o);
    run<8, 0, 64, false, true>("write b64 64 lanes consecutive", d_o);
    run<16, 0, 64, false, true>("write b128 64 lanes consecutive", d_o);
    run<16, 0, 32, false, true>("write b128 32 lanes consecutive", d_o);
    run<2, 0, 64, false, true>("write2_b32 64 lanes", d_o);
    run<2, 0, 32, false, true>("write2_b32 32 lanes", d_o);
    run<2, 0, 64, false, false>("read2_b32 64 lanes", d_o);
    run<1, 0, 64, false, false>("bpermute_b32 64 lanes", d_o);
    run<4, 2, 24, true, false>("read b32 48 lanes scattered rows", d_o);
    run<4, 2, 24, true, true>("write b32 48 lanes scattered rows", d_o);
    return 0;
}

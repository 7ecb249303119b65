// LDS float atomics and the read-modify-write chain on gfx950: what one histogram update costs.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip && ./lds_atomic
// (a) ds_add_f32 / ds_add_rtn_f32 / ds_add_u32 issued back to back (no dependence), by address pattern;
// (b) the dependent chain of k_describe's commit: ds_read_b32 -> v_add_f32 -> ds_write_b32, the next
//     read issued only after the write, at 1..5 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// OP: 0 ds_add_f32, 1 ds_add_rtn_f32, 2 ds_add_u32, 3 ds_write_b32 (reference), 4 ds_max_f32, 5 ds_add_f64
// PAT: 0 lanes on 64 consecutive words, 1 lane pairs share a word (2-way same address), 2 all lanes one word,
//      3 the 24-bin pattern of a voxel: lanes l and l+32 hit the same word (two voxels, same bins),
//      4 four lanes per word, 5 eight lanes per word
template <int OP, int PAT, int NACT>
__global__ __launch_bounds__(256) void ka(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256)
        buf[i] = 0.0f;
    __syncthreads();
    int w;
    if (PAT == 0) w = lane;
    else if (PAT == 1) w = lane >> 1;
    else if (PAT == 2) w = 0;
    else if (PAT == 3) w = lane & 31;
    else if (PAT == 4) w = lane >> 2;
    else w = lane >> 3;
    const int base = (wave * 1024 + w) * (OP == 5 ? 8 : 4);
    float acc = 1.0f + lane;
    double dacc = 1.0 + lane;
    unsigned iacc = lane;
    if (lane < NACT) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int a = base + (u & 7) * 256 * (OP == 5 ? 2 : 1);
                if (OP == 0) asm volatile("ds_add_f32 %0, %1" :: "v"(a), "v"(acc) : "memory");
                if (OP == 1) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(acc) : "memory"); acc += r * 1e-30f; }
                if (OP == 2) asm volatile("ds_add_u32 %0, %1" :: "v"(a), "v"(iacc) : "memory");
                if (OP == 3) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(acc) : "memory");
                if (OP == 4) asm volatile("ds_max_f32 %0, %1" :: "v"(a), "v"(acc) : "memory");
                if (OP == 5) asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(dacc) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x] + acc + (float)iacc + (float)dacc;
}

template <int OP, int PAT, int NACT>
void runa(const char *name, float *d_o)
{
    const int nblk = 256 * 4, iters = 500;    // 4 blocks of 4 waves per CU
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((ka<OP, PAT, NACT>), dim3(nblk), dim3(256), 0, 0, d_o, iters);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double instr_per_cu = 16.0 * iters * 16;
    printf("%-58s %8.3f ms -> %7.2f cycles / instruction / CU (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

// The commit chain: `rounds` dependent read-modify-writes of 48 lanes on a private histogram.
// CH independent chains per wave (each on its own histogram), interleaved.
template <int CH>
__global__ __launch_bounds__(256) void kchain(float *out, int rounds, const int *perm)
{
    __shared__ float hist[4][CH][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * CH * 1024; i += 256)
        (&hist[0][0][0])[i] = 0.0f;
    __syncthreads();
    const int half = lane >> 5, l5 = lane & 31;
    int addr[CH];
#pragma unroll
    for (int c = 0; c < CH; c++)
        addr[c] = 4 * (((wave * CH + c) * 1024) + half * 512 + (l5 < 24 ? l5 : 0));   // (hist is the only LDS object: offset 0)
    const float v = 1.0f + lane;
    int off = perm[lane & 7] * 128;   // round-dependent bin offset (a multiple of 32 words keeps banks distinct)
    for (int r = 0; r < rounds; r++) {
        float x[CH];
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("ds_read_b32 %0, %1" : "=v"(x[c]) : "v"(addr[c] + off) : "memory");
#pragma unroll
        for (int c = 0; c < CH; c++) {
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(CH - 1) : "memory");   // read c has returned (in order)
            x[c] += v;
            asm volatile("ds_write_b32 %0, %1" :: "v"(addr[c] + off), "v"(x[c]) : "memory");
        }
        off = (off + 128) & 1023;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = hist[wave][0][lane];
}

template <int CH>
void runchain(int blocks_per_cu, float *d_o, const int *d_perm)
{
    const int nblk = 256 * blocks_per_cu, rounds = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((kchain<CH>), dim3(nblk), dim3(256), 0, 0, d_o, rounds, d_perm);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("RMW chain: %d chain(s)/wave, %2d waves/CU: %8.3f ms -> %6.1f cycles per round per wave, %6.2f cycles per "
           "(read+write) pair per CU\n", CH, 4 * blocks_per_cu, ms, cyc / rounds, cyc / rounds / (4.0 * blocks_per_cu * CH));
}

int main()
{
    float *d_o; CHK(hipMalloc(&d_o, 256 * 8 * 256 * 4));
    int perm[8] = {0, 1, 2, 3, 4, 5, 6, 7}, *d_perm; CHK(hipMalloc(&d_perm, sizeof(perm)));
    CHK(hipMemcpy(d_perm, perm, sizeof(perm), hipMemcpyHostToDevice));
    runa<3, 0, 64>("ds_write_b32 64 lanes consecutive (reference)", d_o);
    runa<0, 0, 64>("ds_add_f32 64 lanes, 64 distinct words", d_o);
    runa<0, 0, 48>("ds_add_f32 48 lanes, distinct words", d_o);
    runa<0, 0, 32>("ds_add_f32 32 lanes, distinct words", d_o);
    runa<0, 0, 16>("ds_add_f32 16 lanes, distinct words", d_o);
    runa<0, 0, 1>("ds_add_f32 1 lane", d_o);
    runa<0, 3, 64>("ds_add_f32 64 lanes, lanes l and l+32 share a word", d_o);
    runa<0, 1, 64>("ds_add_f32 64 lanes, lane pairs share a word", d_o);
    runa<0, 4, 64>("ds_add_f32 64 lanes, 4 lanes per word", d_o);
    runa<0, 5, 64>("ds_add_f32 64 lanes, 8 lanes per word", d_o);
    runa<0, 2, 64>("ds_add_f32 64 lanes, one word", d_o);
    runa<1, 0, 64>("ds_add_rtn_f32 64 lanes distinct", d_o);
    runa<2, 0, 64>("ds_add_u32 64 lanes distinct", d_o);
    runa<2, 3, 64>("ds_add_u32 64 lanes, l and l+32 share a word", d_o);
    runa<2, 2, 64>("ds_add_u32 64 lanes, one word", d_o);
    runa<4, 0, 64>("ds_max_f32 64 lanes distinct", d_o);
    runa<5, 0, 64>("ds_add_f64 64 lanes distinct", d_o);
    for (int b = 1; b <= 5; b++) runchain<1>(b, d_o, d_perm);
    for (int b = 1; b <= 4; b++) runchain<2>(b, d_o, d_perm);
    for (int b = 1; b <= 2; b++) runchain<4>(b, d_o, d_perm);
    return 0;
}

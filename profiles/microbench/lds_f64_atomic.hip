// lds_f64_atomic.hip -- what a histogram commit by ds_add_f64 costs on gfx950 at k_describe's REAL
// bin addresses (round 5; the question DESIGN 3.3 left on paper).
//   python3 profiles/microbench/desc_trace.py /tmp/desc_trace.bin        (window voxels of 48 real windows)
//   hipcc --offload-arch=gfx950 -O3 -o lds_f64_atomic lds_f64_atomic.hip && ./lds_f64_atomic /tmp/desc_trace.bin
//
// Every wave takes one 64-voxel batch of the trace (consecutive window voxels in scan order) and commits
// it `iters` times into its private histogram, 16 waves per CU, all instructions issued back to back:
//   F64  : one f64 histogram (768 bins + gaps); 8 voxels per 3 instructions -- lane = (voxel of the group,
//          cell corner), instruction j adds the term of face vertex j: all 64 lanes useful, no dependence
//          between instructions (fire and forget).  Lanes of one instruction DO share words whenever two
//          of the eight voxels share a (cell, vertex) bin: the hardware's same-address serialisation is
//          what is measured.
//   RMW  : the shipped scheme: two f32 histograms, 2 voxels per round, lanes 0..23 of each half-wave
//          ds_read_b32 -> v_add_f32 -> ds_write_b32, the next round's read behind this round's write
//          (dependent, as in the kernel) or issued back to back (the pipe cost alone).
// Reported: shader cycles per 64 window voxels per CU (s_memtime delta of the kernel / batches per CU) and
// the same from wall time at the measured clock; the clock the chip held (s_memtime against the constant
// 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__constant__ int c_boff[12];        // word offset of each vertex's 64-cell block

struct Rec { unsigned char cell, v0, v1, v2; };

// MAP 0: lane = 8 * voxel + corner, 1: lane = voxel + 8 * corner
// SYN  0: the trace, 1: all eight voxels of a group the same bins, 2: voxel pairs share, 3: all distinct cells
template <int MAP, int SYN>
__global__ __launch_bounds__(256) void k_f64(const Rec *__restrict__ trace, int nbatch, int iters,
                                             double *__restrict__ out, long long *__restrict__ clk)
{
    __shared__ double hist_[4][832];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *hist = hist_[wave];
    for (int i = lane; i < 832; i += 64)
        hist[i] = 0.0;
    __syncthreads();
    const int gw = blockIdx.x * 4 + wave;
    const Rec *b = trace + (size_t)((gw * 7919) % nbatch) * 64;
    const int v = MAP == 0 ? lane >> 3 : lane & 7, c = MAP == 0 ? lane & 7 : lane >> 3;
    const int corner = ((c >> 2) & 1) + 4 * ((c >> 1) & 1) + 16 * (c & 1);
    int addr[8][3];
#pragma unroll
    for (int g = 0; g < 8; g++) {
        Rec r = b[8 * g + v];
        if (SYN == 1) r = b[8 * g];
        if (SYN == 2) r = b[8 * g + (v & ~1)];
        if (SYN == 3) { r.cell = (unsigned char)((v & 1) * 2 + (v & 2) * 4 + (v & 4) * 8); r.v0 = 0; r.v1 = 1; r.v2 = 2; }
        const int cell = r.cell + corner;
        addr[g][0] = (int)(size_t)(hist + c_boff[r.v0] + cell);
        addr[g][1] = (int)(size_t)(hist + c_boff[r.v1] + cell);
        addr[g][2] = (int)(size_t)(hist + c_boff[r.v2] + cell);
    }
    const double val = 1.0 + lane;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 8; g++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                asm volatile("ds_add_f64 %0, %1" :: "v"(addr[g][j]), "v"(val) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    __syncthreads();
    double s = 0;
    for (int i = lane; i < 832; i += 64)
        s += hist[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) {
        clk[2 * gw] = t1 - t0;
        clk[2 * gw + 1] = w1 - w0;
    }
}

// the shipped commit: DEP 1 = dependent rounds (read, wait, add, write; next read after the write), 0 = back to back
template <int DEP>
__global__ __launch_bounds__(256) void k_rmw(const Rec *__restrict__ trace, int nbatch, int iters,
                                             float *__restrict__ out, long long *__restrict__ clk)
{
    __shared__ float hist_[4][2 * 832];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *hist = hist_[wave];
    for (int i = lane; i < 2 * 832; i += 64)
        hist[i] = 0.0f;
    __syncthreads();
    const int gw = blockIdx.x * 4 + wave;
    const Rec *b = trace + (size_t)((gw * 7919) % nbatch) * 64;
    const int half = lane >> 5, l5 = lane & 31;
    const int pc = l5 < 24 ? l5 / 3 : 0, pj = l5 < 24 ? l5 - 3 * pc : 0;
    const int corner = ((pc >> 2) & 1) + 4 * ((pc >> 1) & 1) + 16 * (pc & 1);
    int addr[32];
#pragma unroll
    for (int u = 0; u < 32; u++) {
        // round u: half-wave 0 voxel (u / 16) * 32 + u % 16, half-wave 1 that + 16 (the kernel's order)
        const Rec r = b[(u >> 4) * 32 + (u & 15) + 16 * half];
        const int vert = pj == 0 ? r.v0 : pj == 1 ? r.v1 : r.v2;
        addr[u] = (int)(size_t)(hist + half * 832 + c_boff[vert] + r.cell + corner);
    }
    const float val = 1.0f + lane;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            float x;
            asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr[u]) : "memory");
            if (DEP) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                x += val;
            } else {
                x = val;      // (no dependence on the read: the pipe cost of the pair alone)
            }
            asm volatile("ds_write_b32 %0, %1" :: "v"(addr[u]), "v"(x) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    __syncthreads();
    float s = 0;
    for (int i = lane; i < 2 * 832; i += 64)
        s += hist[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) {
        clk[2 * gw] = t1 - t0;
        clk[2 * gw + 1] = w1 - w0;
    }
}

static int g_cus = 256;

template <typename F>
static void report(const char *name, F launch, long long *d_clk, int nblk, int iters)
{
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<long long> clk(2 * nblk * 4);
    CHK(hipMemcpy(clk.data(), d_clk, clk.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sc = 0, wc = 0;
    for (int i = 0; i < nblk * 4; i++) { sc += (double)clk[2 * i]; wc += (double)clk[2 * i + 1]; }
    sc /= nblk * 4; wc /= nblk * 4;                       // mean over waves: shader cycles, 100 MHz ticks
    const double ghz = sc / (wc / 100e6) / 1e9;
    const double waves_per_cu = (double)nblk * 4 / g_cus;
    // every wave commits `iters` batches of 64 voxels in `sc` cycles, waves_per_cu of them share a CU
    const double cyc_per_batch_cu = sc / iters / waves_per_cu;
    printf("%-64s %8.3f ms  %7.1f cycles / 64 voxels / CU  (%5.2f per voxel)   clock %.3f GHz\n", name, ms,
           cyc_per_batch_cu, cyc_per_batch_cu / 64.0, ghz);
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "desc_trace.bin";
    FILE *f = fopen(path, "rb");
    if (!f) { printf("cannot open %s (run desc_trace.py first)\n", path); return 1; }
    int hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 1;
    std::vector<Rec> recs(hdr[0]);
    if (fread(recs.data(), 4, hdr[0], f) != (size_t)hdr[0]) return 1;
    fclose(f);
    const int nbatch = hdr[0] / 64;
    // how many DISTINCT words the 64 lanes of an F64 instruction touch on the trace
    {
        double distinct = 0; long n = 0;
        for (int b = 0; b + 8 <= hdr[0]; b += 8)
            for (int j = 0; j < 3; j++) {
                bool seen[12 * 64]; memset(seen, 0, sizeof(seen)); int d = 0;
                for (int v = 0; v < 8; v++)
                    for (int c = 0; c < 8; c++) {
                        const Rec &r = recs[b + v];
                        const int vert = j == 0 ? r.v0 : j == 1 ? r.v1 : r.v2;
                        const int w = vert * 64 + r.cell + ((c >> 2) & 1) + 4 * ((c >> 1) & 1) + 16 * (c & 1);
                        if (!seen[w]) { seen[w] = true; d++; }
                    }
                distinct += d; n++;
            }
        printf("trace: %d window voxels; an F64 instruction (8 voxels x 8 cells, one vertex) touches %.1f distinct "
               "words of 64 on average\n", hdr[0], distinct / n);
    }
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0)); g_cus = prop.multiProcessorCount;
    Rec *d_tr; CHK(hipMalloc(&d_tr, recs.size() * 4));
    CHK(hipMemcpy(d_tr, recs.data(), recs.size() * 4, hipMemcpyHostToDevice));
    const int nblk = g_cus * 4, iters = 2000;     // 4 blocks of 4 waves per CU = 16 waves per CU
    double *d_o; CHK(hipMalloc(&d_o, (size_t)nblk * 256 * 8));
    long long *d_clk; CHK(hipMalloc(&d_clk, (size_t)nblk * 4 * 2 * 8));
    // the kernel's layout: 64 * rank + {0, 8, 18, 26}[colour]; a 4-colouring of the icosahedron by id
    static const int colour[12] = { 0, 1, 2, 3, 1, 0, 3, 2, 2, 3, 0, 1 };
    static const int shift[4] = { 0, 8, 18, 26 };
    for (int layout = 0; layout < 3; layout++) {
        int boff[12], rank = 0;
        for (int c = 0; c < 4; c++)
            for (int u = 0; u < 12; u++)
                if (colour[u] == c)
                    boff[u] = layout == 0 ? 64 * rank++ + shift[c] : layout == 1 ? 64 * u : 68 * u;
        CHK(hipMemcpyToSymbol(HIP_SYMBOL(c_boff), boff, sizeof(boff)));
        printf("-- vertex block offsets: %s\n", layout == 0 ? "64 * rank + {0, 8, 18, 26}[colour] (the kernel's)"
                                               : layout == 1 ? "64 * vertex" : "68 * vertex");
#define F64(M, S, name) report(name, [&] { hipLaunchKernelGGL((k_f64<M, S>), dim3(nblk), dim3(256), 0, 0, d_tr, nbatch, iters, d_o, d_clk); }, d_clk, nblk, iters)
        F64(0, 0, "ds_add_f64, real windows, lane = 8 * voxel + corner");
        F64(1, 0, "ds_add_f64, real windows, lane = voxel + 8 * corner");
        if (layout == 0) {
            F64(0, 3, "ds_add_f64, 64 distinct words per instruction");
            F64(0, 2, "ds_add_f64, voxel pairs share their bins (2 lanes per word)");
            F64(0, 1, "ds_add_f64, all 8 voxels of a group share their bins (8 per word)");
            F64(1, 1, "ds_add_f64, 8 per word, lane = voxel + 8 * corner");
        }
        report("f32 read-modify-write, real windows, dependent rounds (shipped)",
               [&] { hipLaunchKernelGGL((k_rmw<1>), dim3(nblk), dim3(256), 0, 0, d_tr, nbatch, iters / 4, (float *)d_o, d_clk); },
               d_clk, nblk, iters / 4);
        report("f32 read + write pairs back to back (pipe cost alone)",
               [&] { hipLaunchKernelGGL((k_rmw<0>), dim3(nblk), dim3(256), 0, 0, d_tr, nbatch, iters, (float *)d_o, d_clk); },
               d_clk, nblk, iters);
    }
    return 0;
}

/* synth.c -- deterministic synthetic test volumes (host side, plain C).
 *
 * Two generators:
 *
 *  sift3d_amd_synth_survey()  The "noise + random anisotropic blobs" volume whose
 *      spec is SURVEY.md section 8(d).  It is inherently sequential (one xorshift64
 *      stream, blobs added in draw order) and is used for the small parity /
 *      golden configurations (64^3 ... 256^3).
 *
 *  sift3d_amd_synth_lattice() An order-independent variant (hash noise + one
 *      jittered blob per lattice cell) whose value at a voxel depends only on the
 *      voxel coordinates and the seed.  It is what bench.py uses for the large
 *      configurations; a device twin of the same formula lives in
 *      sift3d_kernels.hip (values may differ in the last ulp because host and
 *      device exp() differ -- it is bench data, not a parity input).
 *
 * Neither generator exists in the reference (it ships no data and no tests,
 * SURVEY.md section 4); they are owned by this repository.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <omp.h>

#include "synth.h"

/* (an explicit team: libgomp's default is one thread per CPU it sees -- 256 on a GPU box whose quota is 16 cores) */
static int synth_threads(void)
{
    const int t = omp_get_num_procs();
    return t > 16 ? 16 : t < 1 ? 1 : t;
}

static inline uint64_t xs64(uint64_t *s)
{
    uint64_t v = *s;
    v ^= v << 13;
    v ^= v >> 7;
    v ^= v << 17;
    *s = v;
    return v;
}

static inline double xs64_unit(uint64_t *s)
{
    return (double)(xs64(s) >> 11) * (1.0 / 9007199254740992.0);
}

void sift3d_amd_synth_survey(float *vol, int nx, int ny, int nz, int nblob,
                             uint64_t seed)
{
    uint64_t st = seed ? seed : SIFT3D_AMD_SYNTH_DEFAULT_SEED;
    const size_t n = (size_t)nx * ny * nz;
    size_t i;
    int b;

    for (i = 0; i < n; i++)
        vol[i] = (float)(0.05 * xs64_unit(&st));

    for (b = 0; b < nblob; b++) {
        const double cx = xs64_unit(&st) * nx;
        const double cy = xs64_unit(&st) * ny;
        const double cz = xs64_unit(&st) * nz;
        const double sg = 1.5 + 4.0 * xs64_unit(&st);
        const double a = 2.0 * xs64_unit(&st) - 1.0;
        const int r = (int)(3.0 * sg) + 1;
        const int ix = (int)cx, iy = (int)cy, iz = (int)cz;
        int x, y, z;

        for (z = iz - r; z <= iz + r; z++) {
            if (z < 0 || z >= nz)
                continue;
            for (y = iy - r; y <= iy + r; y++) {
                if (y < 0 || y >= ny)
                    continue;
                for (x = ix - r; x <= ix + r; x++) {
                    double dx, dy, dz, q;
                    if (x < 0 || x >= nx)
                        continue;
                    dx = x - cx;
                    dy = y - cy;
                    dz = z - cz;
                    q = dx * dx + 1.3 * dy * dy + 0.7 * dz * dz;
                    vol[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * z)] +=
                        (float)(a * exp(-q / (2.0 * sg * sg)));
                }
            }
        }
    }
}

/* splitmix64 finaliser: the per-voxel / per-cell hash of the lattice generator */
static inline uint64_t mix64(uint64_t v)
{
    v += 0x9E3779B97F4A7C15ull;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return v ^ (v >> 31);
}

static inline float unit_f(uint64_t h, int k)
{
    /* 16 bits per draw, four draws per hash */
    return (float)((h >> (16 * k)) & 0xFFFF) * (1.0f / 65536.0f);
}

float sift3d_amd_synth_lattice_voxel(int x, int y, int z, uint64_t seed)
{
    const int cell = SIFT3D_AMD_SYNTH_CELL;
    const uint64_t hv = mix64(seed ^ mix64(((uint64_t)(uint32_t)x) |
                                           ((uint64_t)(uint32_t)y << 21) |
                                           ((uint64_t)(uint32_t)z << 42)));
    float v = 0.05f * unit_f(hv, 0);
    const int gx = x / cell, gy = y / cell, gz = z / cell;
    int ix, iy, iz;

    for (iz = gz - 1; iz <= gz + 1; iz++)
        for (iy = gy - 1; iy <= gy + 1; iy++)
            for (ix = gx - 1; ix <= gx + 1; ix++) {
                uint64_t h1, h2;
                float cx, cy, cz, sg, a, dx, dy, dz, q;
                if (ix < 0 || iy < 0 || iz < 0)
                    continue;
                h1 = mix64(seed + 0x51ED270B1ull +
                           mix64(((uint64_t)ix) | ((uint64_t)iy << 21) |
                                 ((uint64_t)iz << 42)));
                h2 = mix64(h1);
                cx = ((float)ix + unit_f(h1, 0)) * (float)cell;
                cy = ((float)iy + unit_f(h1, 1)) * (float)cell;
                cz = ((float)iz + unit_f(h1, 2)) * (float)cell;
                sg = 1.5f + 2.5f * unit_f(h1, 3);
                a = 2.0f * unit_f(h2, 0) - 1.0f;
                dx = (float)x - cx;
                dy = (float)y - cy;
                dz = (float)z - cz;
                q = dx * dx + 1.3f * dy * dy + 0.7f * dz * dz;
                if (q > 18.0f * sg * sg)
                    continue;
                v += a * expf(-q / (2.0f * sg * sg));
            }
    return v;
}

void sift3d_amd_synth_lattice(float *vol, int nx, int ny, int nz, uint64_t seed)
{
    int z;
#pragma omp parallel for schedule(static) num_threads(synth_threads())
    for (z = 0; z < nz; z++) {
        int x, y;
        for (y = 0; y < ny; y++)
            for (x = 0; x < nx; x++)
                vol[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * z)] =
                    sift3d_amd_synth_lattice_voxel(x, y, z, seed);
    }
}

/* synth.h -- deterministic synthetic volumes (see synth.c). */
#ifndef SIFT3D_AMD_SYNTH_H
#define SIFT3D_AMD_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT3D_AMD_SYNTH_DEFAULT_SEED 88172645463325252ull
/* lattice pitch: one blob per CELL^3 voxels (~ the survey generator's density
 * of 200 blobs per 64^3) */
#define SIFT3D_AMD_SYNTH_CELL 11

#ifndef SIFT3D_AMD_API
#define SIFT3D_AMD_API __attribute__((visibility("default")))
#endif

/* SURVEY.md section 8(d) generator: sequential noise + nblob blobs. */
SIFT3D_AMD_API void sift3d_amd_synth_survey(float *vol, int nx, int ny, int nz,
                                            int nblob, uint64_t seed);

/* Order-independent generator (bench data). */
SIFT3D_AMD_API void sift3d_amd_synth_lattice(float *vol, int nx, int ny, int nz,
                                             uint64_t seed);
SIFT3D_AMD_API float sift3d_amd_synth_lattice_voxel(int x, int y, int z,
                                                    uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif

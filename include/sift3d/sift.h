/* sift.h -- keypoint detection and description, the drop-in C API.
 *
 * Replaces the installed header of the same name of fatimp/SIFT3D v2.0
 * (reference: sift3d/sift.h:24-208).  The two hot entry points run on one
 * MI355X through hand-written gfx950 kernels (see include/sift3d_amd.h for the
 * device-level C ABI underneath); everything else is host bookkeeping with the
 * reference's semantics.
 */
#ifndef SIFT3D_AMD_SIFT_H
#define SIFT3D_AMD_SIFT_H

#include "imtypes.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- detector ------------------------------------------------------------ */

/* New detector with the reference defaults: peak 0.1, corner 0.4, 3 keypoint
 * levels/octave, sigma_n 1.15, sigma0 1.6 (reference: sift3d/sift.c:31-35).
 * reference: sift3d/sift.h:24-25, sift3d/sift.c:1841-1854 */
SIFT3D_EXPORT sift3d_detector *
sift3d_make_detector();

/* reference: sift3d/sift.h:30-31, sift3d/sift.c:1856-1859 */
SIFT3D_EXPORT void
sift3d_free_detector(sift3d_detector *);

/* Relative DoG peak threshold, must be in (0, 1].
 * reference: sift3d/sift.h:40-41, sift3d/sift.c:499-509 */
SIFT3D_EXPORT int
sift3d_detector_set_peak_thresh(sift3d_detector *const, const double);

/* Corner score threshold, must be in [0, 1].
 * reference: sift3d/sift.h:49-50, sift3d/sift.c:512-523 */
SIFT3D_EXPORT int
sift3d_detector_set_corner_thresh(sift3d_detector *const, const double);

/* Levels per octave in which keypoints are searched; reallocates the pyramids
 * and rebuilds the filter bank when an image is already set.
 * reference: sift3d/sift.h:57-58, sift3d/sift.c:527-533 */
SIFT3D_EXPORT int
sift3d_detector_set_num_kp_levels(sift3d_detector *const, const unsigned int);

/* Nominal scale of the input, >= 0 and <= sigma0 * 2^(-1/levels).
 * reference: sift3d/sift.h:67-68, sift3d/sift.c:537-549 */
SIFT3D_EXPORT int
sift3d_detector_set_sigma_n(sift3d_detector *const, const double);

/* Scale of level 0 of octave 0, >= 0.
 * reference: sift3d/sift.h:77-78, sift3d/sift.c:553-565 */
SIFT3D_EXPORT int
sift3d_detector_set_sigma0(sift3d_detector *const, const double);

/* ---- hot path ------------------------------------------------------------ */

/* Scale the image to max|v| = 1, build the Gaussian and DoG pyramids, find DoG
 * extrema and assign orientations.  The image is copied (to HBM); the detector
 * keeps the pyramids for sift3d_extract_descriptors().  Blocks until the
 * keypoints are in `store` (host memory).  Fails (-1) when nc != 1 or any
 * dimension is below 8.
 * reference: sift3d/sift.h:92-95, sift3d/sift.c:1217-1249 */
SIFT3D_EXPORT int
sift3d_detect_keypoints(sift3d_detector *const detector,
                        const sift3d_image *const image,
                        sift3d_keypoint_store *const store);

/* Descriptors for the keypoints of `kp_store` from the pyramid of the last
 * successful detect on this detector.  Fails (-1) on an empty store, on
 * out-of-bounds keypoints or without a prior detect.
 * reference: sift3d/sift.h:108-111, sift3d/sift.c:1615-1635 */
SIFT3D_EXPORT int
sift3d_extract_descriptors(sift3d_detector *const detector,
                           const sift3d_keypoint_store *const kp_store,
                           sift3d_descriptor_store *const desc_store);

/* ---- keypoint store ------------------------------------------------------- */

/* reference: sift3d/sift.h:121-122, sift3d/sift.c:1861-1866 */
SIFT3D_EXPORT sift3d_keypoint_store *
sift3d_make_keypoint_store();

/* reference: sift3d/sift.h:127-128, sift3d/sift.c:1868-1871 */
SIFT3D_EXPORT void
sift3d_free_keypoint_store(sift3d_keypoint_store *);

/* N x 3 DOUBLE matrix of coordinates in octave-0 voxels (x,y,z times 2^o).
 * reference: sift3d/sift.h:139-141, sift3d/sift.c:1644-1671 */
SIFT3D_EXPORT int
sift3d_keypoint_store_to_mat_rm(const sift3d_keypoint_store *const,
                                sift3d_mat_rm *const);

/* CSV (.csv) or gzip CSV (.gz): strength,x,y,z,o,sd,R00..R22 per row, "%f".
 * reference: sift3d/sift.h:151-153, sift3d/sift.c:1741-1803 */
SIFT3D_EXPORT int
sift3d_keypoint_store_save(const char *path,
                           const sift3d_keypoint_store *const);

/* Sort by descending strength; keep at most `limit` entries when limit != 0.
 * reference: sift3d/sift.h:164-165, sift3d/sift.c:1832-1837,1885-1900 */
SIFT3D_EXPORT void
sift3d_keypoint_store_sort_by_strength(sift3d_keypoint_store *const, int limit);

/* ---- descriptor store ------------------------------------------------------ */

/* reference: sift3d/sift.h:175-176, sift3d/sift.c:1873-1878 */
SIFT3D_EXPORT sift3d_descriptor_store *
sift3d_make_descriptor_store();

/* reference: sift3d/sift.h:181-182, sift3d/sift.c:1880-1883 */
SIFT3D_EXPORT void
sift3d_free_descriptor_store(sift3d_descriptor_store *);

/* CSV / gzip CSV of the N x 771 matrix below.
 * reference: sift3d/sift.h:192-194, sift3d/sift.c:1807-1830 */
SIFT3D_EXPORT int
sift3d_descriptor_store_save(const char *path,
                             const sift3d_descriptor_store *const);

/* N x 771 FLOAT matrix: x, y, z (octave-0 voxels) then 64 spatial cells x 12
 * icosahedron-vertex bins, column 3 + 12*(cx + 4*cy + 16*cz) + bin.
 * reference: sift3d/sift.h:206-208, sift3d/sift.c:1683-1726 */
SIFT3D_EXPORT int
sift3d_descriptor_store_to_mat_rm(const sift3d_descriptor_store *const,
                                  sift3d_mat_rm *const);

#ifdef __cplusplus
}
#endif
#endif

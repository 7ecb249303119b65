/* imutil.h -- images and row-major matrices of the drop-in C API.
 *
 * Replaces the installed header of the same name of fatimp/SIFT3D v2.0
 * (reference: sift3d/imutil.h:39-110; implementations sift3d/imutil.c:1639-1710).
 * Names, argument meaning, ownership and error behaviour are the reference's.
 */
#ifndef SIFT3D_AMD_IMUTIL_H
#define SIFT3D_AMD_IMUTIL_H

#include "imtypes.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: sift3d/imutil.h:17,23 -- extra codes of sift3d_read_image's internals */
#define SIFT3D_UNSUPPORTED_FILE_TYPE 2
#define SIFT3D_WRAPPER_NOT_COMPILED 3

/* ---- images ------------------------------------------------------------- */

/* New zero-filled nx*ny*nz image with units (1,1,1); nc must be 1 for the
 * detector.  Caller frees with sift3d_free_image(); NULL on failure.
 * reference: sift3d/imutil.h:39-40, sift3d/imutil.c:1644-1655 */
SIFT3D_EXPORT sift3d_image *
sift3d_make_image(const int nx, const int ny, const int nz, const int nc);

/* reference: sift3d/imutil.h:45-46, sift3d/imutil.c:1639-1642 */
SIFT3D_EXPORT void
sift3d_free_image(sift3d_image *);

/* Read a single-file NIfTI-1 image (.nii, .nii.gz) with the built-in reader: the
 * subset of nifticlib that the reference's wrapper uses (sift3d/nifti.c:52-167:
 * dimensions, pixdim spacing, every scalar datatype, scl_slope / scl_inter).
 * Analyze pairs (.hdr/.img) and DICOM directories are not read: NULL with a message
 * on stderr, like a reference build without nifticlib (sift3d/nifti.c:16-31).
 * reference: sift3d/imutil.h:54-55, sift3d/imutil.c:1657-1670 */
SIFT3D_EXPORT sift3d_image *
sift3d_read_image(const char *path);

/* Host raster of the image, x fastest: index = x + nx*(y + ny*z).  The caller
 * fills it after sift3d_make_image().
 * reference: sift3d/imutil.h:64-65, sift3d/imutil.c:1672-1674 */
SIFT3D_EXPORT float *
sift3d_image_data(const sift3d_image *);

/* ---- matrices ----------------------------------------------------------- */

/* New empty (0x0, float) matrix; the to_mat_rm converters resize it.
 * reference: sift3d/imutil.h:77-78, sift3d/imutil.c:1676-1687 */
SIFT3D_EXPORT sift3d_mat_rm *
sift3d_make_mat_rm();

/* reference: sift3d/imutil.h:83-84, sift3d/imutil.c:1689-1692 */
SIFT3D_EXPORT void
sift3d_free_mat_rm(sift3d_mat_rm *);

/* reference: sift3d/imutil.h:92-93, sift3d/imutil.c:1694-1696 */
SIFT3D_EXPORT void *
sift3d_mat_rm_data(sift3d_mat_rm *);

/* NB argument order: columns first, then rows; either may be NULL.
 * reference: sift3d/imutil.h:103-104, sift3d/imutil.c:1698-1706 */
SIFT3D_EXPORT void
sift3d_mat_rm_dimensions(const sift3d_mat_rm *, int *num_cols, int *num_rows);

/* reference: sift3d/imutil.h:109-110, sift3d/imutil.c:1708-1710 */
SIFT3D_EXPORT sift3d_mat_type
sift3d_mat_rm_type(const sift3d_mat_rm *);

#ifdef __cplusplus
}
#endif
#endif

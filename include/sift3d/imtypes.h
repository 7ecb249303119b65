/* imtypes.h -- opaque handle types and return codes of the drop-in C API.
 *
 * Replaces the installed header of the same name of fatimp/SIFT3D v2.0
 * (reference: sift3d/imtypes.h:14-46).  Only forward typedefs are public, exactly
 * as in the reference, so object layouts are private to this implementation
 * (they hold device-memory handles instead of host rasters).
 */
#ifndef SIFT3D_AMD_IMTYPES_H
#define SIFT3D_AMD_IMTYPES_H

#ifdef __cplusplus
extern "C" {
#endif

/* reference: sift3d/imtypes.h:14 */
#define SIFT3D_EXPORT __attribute__((visibility("default")))

/* reference: sift3d/imtypes.h:20,25 -- every int-returning call yields one of these */
#define SIFT3D_SUCCESS 0
#define SIFT3D_FAILURE -1

/* reference: sift3d/imtypes.h:28-29 */
#define SIFT3D_TRUE 1
#define SIFT3D_FALSE 0

/* reference: sift3d/imtypes.h:31-35 */
typedef struct _sift3d_detector sift3d_detector;
typedef struct _sift3d_keypoint_store sift3d_keypoint_store;
typedef struct _sift3d_descriptor_store sift3d_descriptor_store;
typedef struct _sift3d_image sift3d_image;
typedef struct _sift3d_mat_rm sift3d_mat_rm;

/* Element type of a sift3d_mat_rm (reference: sift3d/imtypes.h:40-44) */
typedef enum {
    SIFT3D_DOUBLE,
    SIFT3D_FLOAT,
    SIFT3D_INT
} sift3d_mat_type;

#ifdef __cplusplus
}
#endif
#endif

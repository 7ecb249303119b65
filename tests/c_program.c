/* A program written against the reference's public headers only (doc/Manual.md:25-78,
 * cli/kpSift3D.c:96-146): make image -> detect -> sort -> describe -> matrices.  Compiled by
 * tests/test_c_program.py against include/sift3d and libsift3d_amd.so.
 * usage: c_program nx ny nz seed   -> prints "rc_detect rc_describe nkp cols checksum" */
#include <sift3d/imtypes.h>
#include <sift3d/imutil.h>
#include <sift3d/sift.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 32, ny = argc > 2 ? atoi(argv[2]) : 32,
              nz = argc > 3 ? atoi(argv[3]) : 32;
    unsigned long long s = argc > 4 ? strtoull(argv[4], NULL, 10) : 1ull;
    sift3d_image *image = sift3d_make_image(nx, ny, nz, 1);
    sift3d_detector *detector = sift3d_make_detector();
    sift3d_keypoint_store *kps = sift3d_make_keypoint_store();
    sift3d_descriptor_store *descs = sift3d_make_descriptor_store();
    sift3d_mat_rm *mat = sift3d_make_mat_rm();
    float *data;
    double checksum = 0.0;
    int rc1, rc2 = -2, cols = 0, rows = 0, i;
    size_t n;
    if (!image || !detector || !kps || !descs || !mat)
        return 2;
    data = sift3d_image_data(image);
    n = (size_t)nx * ny * nz;
    for (size_t k = 0; k < n; k++) {                 /* xorshift noise + a few bumps */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        data[k] = (float)((s >> 11) * (1.0 / 9007199254740992.0));
    }
    for (i = 0; i < 40; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        data[(size_t)(s % n)] += 25.0f;
    }
    if (sift3d_detector_set_peak_thresh(detector, 0.05) != SIFT3D_SUCCESS)
        return 3;
    rc1 = sift3d_detect_keypoints(detector, image, kps);
    if (rc1 == SIFT3D_SUCCESS) {
        sift3d_keypoint_store_sort_by_strength(kps, 100);        /* void, sift.h:152-155 */
        rc2 = sift3d_extract_descriptors(detector, kps, descs);
        if (rc2 == SIFT3D_SUCCESS && sift3d_descriptor_store_to_mat_rm(descs, mat) == SIFT3D_SUCCESS) {
            const float *m = (const float *)sift3d_mat_rm_data(mat);
            sift3d_mat_rm_dimensions(mat, &cols, &rows);
            if (sift3d_mat_rm_type(mat) != SIFT3D_FLOAT)
                return 5;
            for (size_t k = 0; k < (size_t)cols * rows; k++)
                checksum += (double)m[k] * (double)(1 + k % 7);
        }
    }
    printf("%d %d %d %d %.9e\n", rc1, rc2, rows, cols, checksum);
    sift3d_free_image(image);
    sift3d_free_detector(detector);
    sift3d_free_descriptor_store(descs);
    sift3d_free_keypoint_store(kps);
    sift3d_free_mat_rm(mat);
    return 0;
}

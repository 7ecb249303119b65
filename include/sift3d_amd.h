/* sift3d_amd.h -- MI355X-side C ABI underneath the drop-in API (include/sift3d/).
 *
 * Two groups of entry points, all `extern "C"`, plain pointers and sizes:
 *
 *  sift3d_amd_*  extensions of the reference API that only make sense with a
 *                device: hand the detector a volume that is already resident in
 *                HBM, query stage timings, generate synthetic volumes.
 *  sift3d_hip_*  the stage kernels themselves, operating on DEVICE pointers and a
 *                HIP stream.  The C host code of the drop-in library calls these;
 *                so does the Z-slab multi-GPU driver (sift3d_amd/sharded.py), which
 *                is why every stage takes local-slab geometry (global length,
 *                offset of the local buffer, plane range to produce).
 *
 * Each stage cites the reference function (file:line under /root/reference/sift3d/)
 * whose results it reproduces.  Volumes are float32, x fastest:
 * index = x + nx*(y + ny*z) (reference: imutil.c:520-533).
 */
#ifndef SIFT3D_AMD_H
#define SIFT3D_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "sift3d/imtypes.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SIFT3D_AMD_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Extensions of the drop-in API                                            */
/* ------------------------------------------------------------------------ */

/* As sift3d_detect_keypoints (reference: sift.c:1217-1249) for a single-channel
 * nx*ny*nz float32 volume that already lives in device memory (`d_volume`), with
 * voxel spacing (ux,uy,uz) > 0.  The volume is not modified.
 * Synchronisation contract: the work is issued on the detector's own stream, which is NOT
 * ordered after other streams -- the caller makes sure the producer of d_volume has finished
 * (e.g. torch.cuda.synchronize(), or an event wait) before calling; on return the results are
 * complete (the call synchronises its stream).  A detector belongs to the HIP device that was
 * current at its first detect call. */
SIFT3D_AMD_API int
sift3d_amd_detect_keypoints_device(sift3d_detector *det, const float *d_volume,
                                   int nx, int ny, int nz, double ux, double uy,
                                   double uz, sift3d_keypoint_store *store);

/* Voxel spacing of an image made by sift3d_make_image (the reference sets units
 * only through its NIfTI reader, nifti.c:52-167). */
SIFT3D_AMD_API int
sift3d_amd_image_set_units(sift3d_image *im, double ux, double uy, double uz);

/* Extremum neighbourhood of the following detect calls: 0 (default) = the default build's
 * 8-neighbour test, non-zero = the reference's compile-time CUBOID_EXTREMA variant
 * (sift.c:24, 761-796), here a run-time option. */
SIFT3D_AMD_API int
sift3d_amd_detector_set_cuboid_extrema(sift3d_detector *det, int on);

/* The dogmax scan (sift.c:821-826) of octave 0 as a pass of its own over the octave's Gaussian levels
 * (non-zero) or gathered by the extrema sweep from lower bounds (0, the default: see
 * sift3d_hip_extrema_gauss6_est_phase).  Same candidates either way; an A/B switch for tests and profiles. */
SIFT3D_AMD_API int
sift3d_amd_detector_set_dogmax_pass(sift3d_detector *det, int on);
/* Orientation window sums: 0 (default) = parallel sums with decisions by margin and a serial re-run of the
 * undecided candidates; non-zero = the reference's serial sums (sift.c:978-990) for every candidate.  Same
 * keypoints and R bit for bit either way; an A/B switch for tests and profiles. */
SIFT3D_AMD_API int
sift3d_amd_detector_set_serial_orientation(sift3d_detector *det, int on);
/* Descriptor accumulation (sift3d_extract_descriptors): 0 (default) = automatic: keypoints whose window
 * holds more than ~1.9e5 voxels -- sigma0 * 2^(s/K) above ~2.9 voxels, never with the default parameters --
 * are computed in the reference's accumulation order (their histograms are the reference's bit for bit),
 * the others by the fast two-histogram commit (within 1e-5 relative, elementwise); 1 = every keypoint in
 * the reference's order (bit-exact descriptors, ~1.5x the time); -1 = never. */
SIFT3D_AMD_API int
sift3d_amd_detector_set_exact_descriptors(sift3d_detector *det, int mode);
/* max|DoG| of every DoG level of the last detect call, out[octave * levels + level] (the values
 * detect_extrema scales peak_thresh with, sift.c:821-829); returns their number, -1 on failure. */
SIFT3D_AMD_API int
sift3d_amd_detector_dogmax(const sift3d_detector *det, float *out, int cap);

/* Dimensions (nx, ny, nz, nc) and voxel spacing of an image, e.g. one returned by
 * sift3d_read_image (the reference keeps both private).  Either output may be NULL. */
SIFT3D_AMD_API int
sift3d_amd_image_info(const sift3d_image *im, int *dims4, double *units3);

/* Wall-clock seconds of the stages of the last detect/describe on `det`:
 * [0] upload+scale  [1] Gaussian pyramid  [2] DoG  [3] extrema  [4] orientation
 * [5] describe  [6] pyramid kernels only, device time from HIP events
 * [7] whole detect, device time  [8] whole describe, device time
 * [9] the LAST fused y+z FIR launch of octave 0 alone (HIP events on its stream; 0 when that blur did
 *     not take the fused kernel) -- in the pipeline it shares the device with the octave streams.
 * [10 .. 10+B-1]    the x-pass launch of blur s = 0 .. B-1 of octave 0 (B = SIFT3D_AMD_TIMED_BLURS),
 * [10+B .. 10+2B-1] its fused y+z launch: HIP events around each launch on the stream it runs on (0 for a
 *     blur that did not take the fused kernel).  Every octave-0 pyramid launch of the step is timed, so the
 *     LONGEST in-step launch can be named (bench.py's roofline.kernel).
 * Since round 5 the stages overlap: [1] ends with the last of the pyramid's three chains, [2] and [3] start on
 * the main stream when octave 0 is complete -- their sum can exceed [7]. */
#define SIFT3D_AMD_TIMED_BLURS 8
/* [10+2B] the device span of the whole detect call (first to last stage event): [7] minus this is what the host
 * adds (enqueueing, its synchronisations, the candidate -> keypoint compaction); [10+2B+1] the compaction alone;
 * [10+2B+2], [10+2B+3] first stage event -> end of the orientation kernels of octave 0's candidates / of the
 * other octaves' (the default schedule orients octave 0 while the smaller octaves are still swept; 0 otherwise) */
#define SIFT3D_AMD_NUM_TIMINGS (10 + 2 * SIFT3D_AMD_TIMED_BLURS + 4)
SIFT3D_AMD_API const double *
sift3d_amd_timings(const sift3d_detector *det);

/* max|v| + the Gaussian pyramid alone of a float32 volume in device memory (the first part of
 * sift3d_amd_detect_keypoints_device: sift.c:645-649, 662-711), blocking; timings()[1] is its device time. */
SIFT3D_AMD_API int
sift3d_amd_build_pyramid_device(sift3d_detector *det, const float *d_volume, int nx, int ny, int nz,
                                double ux, double uy, double uz);

/* Shader cycles and seconds (from the device's constant 100 MHz counter) of the fast descriptor kernel of the
 * last sift3d_extract_descriptors on `det`, measured by the kernel itself (sift3d_hip_describe_clock). */
SIFT3D_AMD_API int
sift3d_amd_describe_clock(const sift3d_detector *det, double *cycles, double *seconds);

/* Number of DoG extrema before orientation filtering in the last detect. */
SIFT3D_AMD_API int
sift3d_amd_num_candidates(const sift3d_detector *det);

/* Copy level (o, s) of the Gaussian (which=0) or DoG (which=1) pyramid, or the
 * scaled input (which=2), of the last detect to host memory.  dims receives
 * nx,ny,nz.  `out` may be NULL to query dims only. */
SIFT3D_AMD_API int
sift3d_amd_copy_level(const sift3d_detector *det, int which, int o, int s,
                      float *out, int *dims);

/* Raw views of the stores (the reference keeps these private; the parity tests
 * need R, sd, strength without the "%f" CSV rounding of *_save). */
SIFT3D_AMD_API int
sift3d_amd_keypoint_store_size(const sift3d_keypoint_store *);
SIFT3D_AMD_API int
sift3d_amd_keypoint_store_get(const sift3d_keypoint_store *, int i, int *o, int *s,
                              double *xyz_sd /*4*/, float *strength, float *R /*9*/);
SIFT3D_AMD_API int
sift3d_amd_keypoint_store_set(sift3d_keypoint_store *, int n, const int *os /*2n*/,
                              const double *xyz_sd /*4n*/, const float *strength,
                              const float *R /*9n*/);
SIFT3D_AMD_API int
sift3d_amd_descriptor_store_size(const sift3d_descriptor_store *);
/* Fill a descriptor store from host arrays (n records of {x, y, z, sd} and 768 floats; the image
 * dimensions the reference keeps in the store): lets the writers and converters be checked
 * against reference-written files without a device. */
SIFT3D_AMD_API int
sift3d_amd_descriptor_store_set(sift3d_descriptor_store *, int n, const double *xyz_sd /*4n*/,
                                const float *hist /*768n*/, int nx, int ny, int nz);

/* init_Gauss_filter (imutil.c:1267-1319) on the host: normalised taps for `sigma`.
 * Returns the width (taps are written when width <= max_taps) or -1. */
SIFT3D_AMD_API int sift3d_amd_gauss_filter(double sigma, float *taps, int max_taps);

/* Uploads the icosahedron tables the descriptor kernel needs (done implicitly by the first
 * detect); for callers that drive the sift3d_hip_* stages themselves. */
SIFT3D_AMD_API int sift3d_amd_init(void);

/* 1 when a usable HIP device is present. */
SIFT3D_AMD_API int sift3d_amd_device_available(void);
SIFT3D_AMD_API const char *sift3d_amd_version(void);

/* ------------------------------------------------------------------------ */
/* Registration: descriptor matching + RANSAC affine (BASELINE config 5)     */
/* ------------------------------------------------------------------------ */
/* Removed from the reference fork (CHANGES.md:99-103; upstream: README-OLD.md:5) -- no reference
 * code, no oracle: PARITY UNPINNED.  See sift3d_amd/csrc/sift3d_register.c. */

/* Nearest / second-nearest neighbour of each of the nA rows of d_A (nA x dim floats, row-major)
 * among the nB rows of d_B under the squared L2 distance, on the matrix cores
 * (v_mfma_f32_32x32x2_f32).  d_j1[i] = index of the nearest row (-1: nB == 0), d_d1 / d_d2 =
 * squared distances to the nearest and second nearest (+inf when absent).  dim % 16 == 0.
 * d_work: sift3d_hip_nn2_work_floats(nA, nB) floats of device scratch. */
SIFT3D_AMD_API size_t sift3d_hip_nn2_work_floats(int nA, int nB);
SIFT3D_AMD_API int
sift3d_hip_nn2(const float *d_A, int nA, const float *d_B, int nB, int dim, int *d_j1, float *d_d1,
               float *d_d2, float *d_work, void *stream);

/* match_ab[i] (i < size of a) = index in b of the descriptor matched to descriptor i of a, or -1:
 * nearest neighbour accepted when (nearest distance) / (second nearest) < nn_thresh (e.g. 0.8)
 * in BOTH directions and mutual. */
SIFT3D_AMD_API int
sift3d_amd_nn_match(const sift3d_descriptor_store *a, const sift3d_descriptor_store *b,
                    double nn_thresh, int *match_ab);
/* A matcher object: its own stream and scratch (grown on demand, reused from call to call: no
 * allocation per match).  sift3d_amd_matcher_match is sift3d_amd_nn_match on it;
 * sift3d_amd_matcher_seconds returns the device time of the two nearest-neighbour searches of the last
 * match (HIP events on the matcher's stream).  Descriptor stores marked with
 * sift3d_amd_descriptor_store_keep_device(store, 1) keep a copy of their histograms in HBM (written by
 * sift3d_extract_descriptors beside the host array), which the matcher reads in place; other stores
 * are uploaded per call. */
typedef struct sift3d_amd_matcher sift3d_amd_matcher;
SIFT3D_AMD_API sift3d_amd_matcher *sift3d_amd_make_matcher(void);
SIFT3D_AMD_API void sift3d_amd_free_matcher(sift3d_amd_matcher *);
SIFT3D_AMD_API int sift3d_amd_matcher_match(sift3d_amd_matcher *, const sift3d_descriptor_store *a,
                                            const sift3d_descriptor_store *b, double nn_thresh,
                                            int *match_ab);
SIFT3D_AMD_API double sift3d_amd_matcher_seconds(const sift3d_amd_matcher *);
SIFT3D_AMD_API int sift3d_amd_descriptor_store_keep_device(sift3d_descriptor_store *, int on);
SIFT3D_AMD_API int
sift3d_amd_descriptor_store_xyz(const sift3d_descriptor_store *, int i, double *xyz /*3*/);
SIFT3D_AMD_API int
sift3d_amd_descriptor_store_xyz_all(const sift3d_descriptor_store *, double *xyz /*3 per descriptor*/);

/* RANSAC fit of the affine map dst = A [src; 1] (tform: 3 x 4 doubles, row-major) to n point
 * pairs (n x 3 doubles each): num_iter minimal samples of 4 pairs, inliers = residual <=
 * err_thresh, least-squares refit on the best consensus set.  inlier (n bytes, may be NULL)
 * receives 0 / 1.  Deterministic for a given seed. */
SIFT3D_AMD_API int
sift3d_amd_ransac_affine(const double *src, const double *dst, int n, double err_thresh, int num_iter,
                         uint64_t seed, double *tform, unsigned char *inlier, int *num_inliers);

/* ------------------------------------------------------------------------ */
/* Multi-GPU: one process per GPU, the volume cut into Z-slabs               */
/* ------------------------------------------------------------------------ */

/* The three exchanges of the slab driver (sift3d_amd/csrc/sift3d_sharded.c).  All buffers are
 * DEVICE pointers; every call is enqueued on `stream` (stream-ordered, like RCCL) or completes
 * before returning.  A NULL send/recv pair of `halo` means "no neighbour on that side".  Return 0
 * on success. */
typedef struct {
    int rank, world;
    void *ctx;
    /* nearest neighbours: send send_lo to rank-1 / send_hi to rank+1, receive recv_lo from
     * rank-1 / recv_hi from rank+1, `bytes` each */
    int (*halo)(void *ctx, const void *d_send_lo, void *d_recv_lo, const void *d_send_hi,
                void *d_recv_hi, size_t bytes, void *stream);
    int (*allreduce_max)(void *ctx, float *d_buf, int n, void *stream);          /* in place */
    int (*allgather)(void *ctx, const void *d_send, void *d_recv, size_t bytes_per_rank,
                     void *stream);                     /* d_recv: world * bytes, rank order */
} sift3d_amd_transport;

/* RCCL over xGMI (librccl is loaded at run time).  Rank 0 makes the 128-byte unique id, the
 * application distributes it (any out-of-band channel), every rank builds its transport on its own
 * current HIP device. */
SIFT3D_AMD_API int sift3d_amd_rccl_unique_id(void *id128);
SIFT3D_AMD_API int sift3d_amd_rccl_transport(sift3d_amd_transport *out, int world, int rank,
                                             const void *id128);
SIFT3D_AMD_API void sift3d_amd_rccl_transport_free(sift3d_amd_transport *t);

/* Rehearsal / test transport: the `world` ranks are THREADS of one process (one slab driver each, on the
 * same device), exchanging device-to-device.  Stream-ordered with the completion semantics of
 * ncclSend / ncclRecv / ncclAllGather / ncclAllReduce and no host-side stream synchronisation
 * (sift3d_amd/csrc/sift3d_thread_transport.c): (pointer, event) pairs travel through a host mailbox, the
 * copies run on the receiver's stream behind the sender's `ready` event, the sender's stream waits for the
 * receiver's `consumed` event.  An error in the driver's event edges between its streams therefore shows
 * on one GPU as a wrong result, as it would over RCCL on eight.  _abort wakes every rank waiting in an
 * exchange (their calls fail) -- for a caller whose rank has given up. */
typedef struct sift3d_amd_thread_group sift3d_amd_thread_group;
SIFT3D_AMD_API sift3d_amd_thread_group *sift3d_amd_thread_group_create(int world);
SIFT3D_AMD_API void sift3d_amd_thread_group_free(sift3d_amd_thread_group *);
SIFT3D_AMD_API void sift3d_amd_thread_group_abort(sift3d_amd_thread_group *);
SIFT3D_AMD_API int sift3d_amd_thread_transport(sift3d_amd_transport *out, sift3d_amd_thread_group *,
                                               int rank);
SIFT3D_AMD_API void sift3d_amd_thread_transport_free(sift3d_amd_transport *t);

/* sift3d_detect_keypoints + sift3d_extract_descriptors (sift.c:1217-1249, 1615-1635) on ONE
 * nx*ny*nz volume cut into `world` Z-slabs; results equal the single-GPU ones bit for bit.
 * `params` supplies thresholds, scales, the number of keypoint levels per octave and the extrema
 * neighbourhood (NULL: defaults); every configuration the drop-in API accepts is supported.  This rank's raw planes [z0, z1) go to the device
 * buffer sift3d_amd_sharded_input() (x fastest, (z1 - z0) * ny * nx floats).  detect fills `kp`
 * with the GLOBAL keypoint list on every rank; describe computes the descriptors of the
 * keypoints this rank owns (their positions in `kp` go to own_idx, capacity kp's size).
 * Failures: detect and the descriptor gather are collective.  A failure that every rank sees (bad arguments,
 * a transport error, an exchange buffer that cannot be allocated) returns SIFT3D_FAILURE at once.  A rank whose
 * LOCAL work fails (a launch, an allocation) keeps issuing every exchange of the step in the common order, its
 * status word travels behind the blocks of both all-gathers, and EVERY rank returns SIFT3D_FAILURE at the same
 * point with the transport in step for the next call (the reference: every stage returns -1 to its caller,
 * immacros.h:27-32). */
typedef struct sift3d_amd_sharded sift3d_amd_sharded;
SIFT3D_AMD_API sift3d_amd_sharded *
sift3d_amd_sharded_create(int nx, int ny, int nz, const sift3d_amd_transport *t,
                          const sift3d_detector *params, double ux, double uy, double uz);
SIFT3D_AMD_API void sift3d_amd_sharded_free(sift3d_amd_sharded *);
SIFT3D_AMD_API int sift3d_amd_sharded_own_planes(const sift3d_amd_sharded *, int *z0, int *z1);
SIFT3D_AMD_API float *sift3d_amd_sharded_input(sift3d_amd_sharded *);
SIFT3D_AMD_API int sift3d_amd_sharded_synth(sift3d_amd_sharded *, uint64_t seed);
SIFT3D_AMD_API int sift3d_amd_sharded_detect(sift3d_amd_sharded *, sift3d_keypoint_store *kp);
SIFT3D_AMD_API int sift3d_amd_sharded_describe(sift3d_amd_sharded *, const sift3d_keypoint_store *kp,
                                               sift3d_descriptor_store *desc, int *own_idx,
                                               int *n_own);
/* The descriptors of ALL keypoints of `kp`, in its (global) order, from the rows every rank computed
 * (sift3d_amd_sharded_describe: desc_own / own_idx / n_own of THIS rank): one all-gather of the ranks' row
 * blocks, device to device.  root < 0: `all` is filled on every rank; else on rank `root` only (the others
 * take part in the exchange and leave `all` alone).  Collective; the status word behind every block makes a
 * rank-local failure return SIFT3D_FAILURE on every rank -- a rank whose sift3d_amd_sharded_describe failed
 * still calls this, with n_own = -1. */
SIFT3D_AMD_API int
sift3d_amd_sharded_gather_descriptors(sift3d_amd_sharded *, const sift3d_keypoint_store *kp,
                                      const sift3d_descriptor_store *desc_own, const int *own_idx, int n_own,
                                      sift3d_descriptor_store *all, int root);
/* test hook: the next detect (where = 1, 2, 3: before the pyramid, after the extrema, between the two
 * all-gathers) or descriptor gather (4) of this rank fails LOCALLY; 0 clears */
SIFT3D_AMD_API int sift3d_amd_sharded_inject_failure(sift3d_amd_sharded *, int where);
SIFT3D_AMD_API int sift3d_amd_sharded_num_candidates(const sift3d_amd_sharded *);
/* eight doubles of the last step: [0] Gaussian pyramid (device s, halo exchanges of the blurs inside)
 * [1] detect wall  [2] describe wall  [3] DoG maxima + extrema (device s, incl. the all-reduce)
 * [4] wait for the window halos + orientation (device s)  [5] gathers + global keypoint list (host s)
 * [6] input scaling (device s, incl. the all-reduce)  [7] unused */
SIFT3D_AMD_API const double *sift3d_amd_sharded_timings(const sift3d_amd_sharded *);
SIFT3D_AMD_API int sift3d_amd_sharded_info(const sift3d_amd_sharded *, int *num_octaves, int *o_shard,
                                           int *halo);

/* ------------------------------------------------------------------------ */
/* Device plumbing (so that the C host code needs no HIP headers)           */
/* ------------------------------------------------------------------------ */
SIFT3D_AMD_API int sift3d_hip_device_count(void);
SIFT3D_AMD_API int sift3d_hip_set_device(int dev);
SIFT3D_AMD_API int sift3d_hip_current_device(void);   /* -1 on error */
SIFT3D_AMD_API void *sift3d_hip_malloc(size_t bytes);
SIFT3D_AMD_API void sift3d_hip_free(void *d_ptr);
SIFT3D_AMD_API void *sift3d_hip_host_alloc(size_t bytes); /* pinned */
SIFT3D_AMD_API void sift3d_hip_host_free(void *h_ptr);
/* device-side address of a sift3d_hip_host_alloc block (kernels may write results into it) */
SIFT3D_AMD_API void *sift3d_hip_host_device_ptr(void *h_ptr);
SIFT3D_AMD_API int sift3d_hip_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
SIFT3D_AMD_API int sift3d_hip_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
SIFT3D_AMD_API int sift3d_hip_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
SIFT3D_AMD_API int sift3d_hip_memcpy2d_d2h(void *h_dst, size_t dst_pitch, const void *d_src,
                                           size_t src_pitch, size_t width, size_t height, void *stream);
SIFT3D_AMD_API int sift3d_hip_stream_wait_event(void *stream, void *ev);
/* descriptor rows (768 floats) out of the ranks' gathered blocks into the global order: row g = row
 * (d_map[g] & 0xffffff) of block (d_map[g] >> 24) */
SIFT3D_AMD_API int sift3d_hip_rows_scatter(float *d_dst, const void *d_all, size_t blk_bytes,
                                           const uint32_t *d_map, uint32_t n, void *stream);
/* d_dst[i] = max over r of d_rows[r * n + i] (the reduction step of the thread transport's all-reduce) */
SIFT3D_AMD_API int sift3d_hip_max_rows(float *d_dst, const float *d_rows, int nrows, int n, void *stream);
SIFT3D_AMD_API int sift3d_hip_memset(void *d_dst, int byte, size_t bytes, void *stream);
SIFT3D_AMD_API void *sift3d_hip_stream_create(void);
SIFT3D_AMD_API void *sift3d_hip_stream_create_high(void);   /* highest dispatch priority */
SIFT3D_AMD_API void sift3d_hip_stream_destroy(void *stream);
SIFT3D_AMD_API int sift3d_hip_stream_sync(void *stream);
/* HIP events, for device-side stage timing */
SIFT3D_AMD_API void *sift3d_hip_event_create(void);
SIFT3D_AMD_API void sift3d_hip_event_destroy(void *ev);
SIFT3D_AMD_API int sift3d_hip_event_record(void *ev, void *stream);
SIFT3D_AMD_API double sift3d_hip_event_elapsed_ms(void *ev_start, void *ev_stop); /* syncs on stop */

/* ------------------------------------------------------------------------ */
/* Stage kernels                                                            */
/* ------------------------------------------------------------------------ */

/* im_max_abs (imutil.c:681-695): *d_max = max(*d_max, max|src[i]|).  d_max is a
 * device float the caller zeroes first (max is order independent => exact). */
SIFT3D_AMD_API int
sift3d_hip_absmax(const float *d_src, size_t n, float *d_max, void *stream);

/* im_scale (imutil.c:699-713): dst = src / *d_max, a plain copy when *d_max == 0. */
SIFT3D_AMD_API int
sift3d_hip_scale(const float *d_src, float *d_dst, size_t n, const float *d_max,
                 void *stream);

#define SIFT3D_HIP_MAX_TAPS 65

/* One 1-D pass of convolve_sep_gen (imutil.c:742-861) along `axis`, applied directly
 * on the x-fastest volume (no im_permute copies, imutil.c:907-958).
 *
 *  unit_factor = (float)(unit / units[axis])             (imutil.c:754-755)
 *  z_lo, z_hi    local plane range [z_lo, z_hi) of outputs to produce
 *  n_glob, off   axis 2 only: the buffers hold planes [off, off + nz) of a global
 *                axis of n_glob planes; mirror rules use global coordinates and
 *                interior samples must be present locally (halo).  Unsharded:
 *                n_glob = nz, off = 0.
 *  variant       0 = pick the fastest specialised kernel, 1 = force the literal
 *                one-thread-per-voxel kernel (used by tests to A/B the fast paths) */
typedef struct {
    const float *src;
    float *dst;
    int nx, ny, nz;
    int axis;
    int width;
    const float *taps; /* host pointer */
    float unit_factor;
    int n_glob, off;
    int z_lo, z_hi;
    int variant;
} sift3d_hip_fir_args;

SIFT3D_AMD_API int
sift3d_hip_fir(const sift3d_hip_fir_args *args, void *stream);
/* The x pass of a unit-spaced blur applied to src / *d_max: im_scale (imutil.c:698-713; *d_max = max|src|
 * from sift3d_hip_absmax, 0: no scaling) folded into the first pass of the pyramid, so that the scaled image
 * is never stored.  Same result as sift3d_hip_scale followed by sift3d_hip_fir.  1: not covered (axis other
 * than x, tap spacing other than 1, more than 17 taps, the literal variant). */
SIFT3D_AMD_API int
sift3d_hip_fir_x_scaled(const sift3d_hip_fir_args *args, const float *d_max, void *stream);
SIFT3D_AMD_API int sift3d_hip_fir_x_scaled_covers(const sift3d_hip_fir_args *args);   /* 1: covered */

/* The y and z passes of one blur fused into one launch when both have tap spacing 1 (octave 0):
 * dst = FIR_z(FIR_y(src)), bit-identical to two sift3d_hip_fir calls, without the intermediate
 * volume touching HBM.  Slab arguments as for axis 2 of sift3d_hip_fir.  Returns SIFT3D_SUCCESS,
 * SIFT3D_FAILURE, or 1 when the configuration is not covered (width > 17, nx % 4 != 0, unaligned
 * buffers, an axis shorter than width + 1): the caller then issues the two passes itself. */
SIFT3D_AMD_API int
sift3d_hip_fir_yz_u1(const float *d_src, float *d_dst, int nx, int ny, int nz, const float *taps,
                     int width, int n_glob, int off, int z_lo, int z_hi, void *stream);
/* 1 when sift3d_hip_fir_yz_u1 covers the configuration (it returns 1 = "not covered" otherwise); for
 * callers that must know before they launch. */
SIFT3D_AMD_API int sift3d_hip_fir_yz_u1_covers(const float *d_src, const float *d_dst, int nx, int ny,
                                               int width, int n_glob);

/* im_subtract (imutil.c:719-739) fused with the dogmax scan of detect_extrema
 * (sift.c:821-826): dst = a - b and *d_absmax = max(*d_absmax, max|dst|).
 * d_absmax may be NULL. */
SIFT3D_AMD_API int
sift3d_hip_subtract_absmax(const float *d_a, const float *d_b, float *d_dst, size_t n,
                           float *d_absmax, void *stream);

/* build_dog (sift.c:713-732) for one octave in one pass: d_d[k] = d_g[k] - d_g[k+1] for
 * k < n_gauss-1 and d_absmax[k] = max(d_absmax[k], max|d_d[k]|).  Every Gaussian level is read
 * once.  Returns 1 (nothing done) when n_gauss > SIFT3D_HIP_MAX_DOG_STACK or a pointer is not
 * 16-byte aligned: the caller then uses sift3d_hip_subtract_absmax per level pair. */
#define SIFT3D_HIP_MAX_DOG_STACK 8
SIFT3D_AMD_API int sift3d_hip_dog_stack(const float *const *d_g, float *const *d_d, int n_gauss,
                                        size_t n, float *d_absmax, void *stream);

/* im_downsample_2x (imutil.c:591-617): dst(x,y,z) = src(2x,2y,2z) for the mx*my*mz
 * output box; src rows are nx long, planes nx*ny. */
SIFT3D_AMD_API int
sift3d_hip_downsample2(const float *d_src, int nx, int ny, float *d_dst, int mx, int my,
                       int mz, void *stream);

/* One DoG level for the extrema search */
typedef struct {
    const float *prev, *cur, *next; /* D[o,s-1], D[o,s], D[o,s+1]: same local dims */
    const float *d_absmax;          /* device float: max|D[o,s]| over the GLOBAL level */
    int z_lo, z_hi;                 /* local planes to test; planes z-1 and z+1 must exist */
    int tag;                        /* copied into every record (level id) */
} sift3d_hip_extrema_level;

typedef struct {
    uint32_t idx;   /* local linear index x + nx*(y + ny*z) */
    int32_t tag;
    float val;      /* |D| at the voxel = keypoint strength (sift.c:864) */
} sift3d_hip_cand;

/* detect_extrema (sift.c:735-871, default 8-neighbour build) for `nlevels` levels of
 * one octave (all nx*ny*nz).  Records are APPENDED to d_out starting at *d_count in
 * the reference's scan order (level, z, y, x); *d_count (device uint32) is advanced
 * by the number found, even past `cap` (records beyond cap are dropped, the caller
 * compares the final count with cap).  d_work: scratch of
 * sift3d_hip_extrema_work_bytes() bytes. */
SIFT3D_AMD_API size_t
sift3d_hip_extrema_work_bytes(int nx, int ny, int nz, int nlevels);
SIFT3D_AMD_API int
sift3d_hip_extrema(const sift3d_hip_extrema_level *levels, int nlevels, int nx, int ny,
                   int nz, double peak_thresh, sift3d_hip_cand *d_out, uint32_t cap,
                   uint32_t *d_count, void *d_work, size_t work_bytes, void *stream);

/* The same with the neighbourhood selectable: cuboid = 0 is the default build's 8-neighbour test
 * (sift.c:797-810), cuboid = 1 the reference's compile-time CUBOID_EXTREMA variant
 * (sift.c:24, 761-796: 27 + 26 + 27 samples). */
SIFT3D_AMD_API int
sift3d_hip_extrema_mode(const sift3d_hip_extrema_level *levels, int nlevels, int nx, int ny, int nz,
                        double peak_thresh, int cuboid, sift3d_hip_cand *d_out, uint32_t cap,
                        uint32_t *d_count, void *d_work, size_t work_bytes, void *stream);

/* build_dog + detect_extrema WITHOUT a stored DoG pyramid, default configuration (three keypoint
 * levels per octave = six Gaussian levels, 8-neighbour test):
 *   sift3d_hip_dogmax_stack    d_absmax[k] = max(d_absmax[k], max|d_g[k] - d_g[k+1]|), k < n_gauss-1
 *                              (the dogmax scan, sift.c:821-826, on differences formed on the fly);
 *   sift3d_hip_extrema_gauss6  detect_extrema for DoG levels 1..3 of the octave, the differences
 *                              (im_subtract, imutil.c:719-739) again formed when loaded; records as
 *                              sift3d_hip_extrema, tags tag0, tag0+1, tag0+2; d_absmax = the five
 *                              maxima of the octave (global over slabs).
 * Both return 1 (nothing done) when the configuration is not covered (nx % 4 != 0, unaligned
 * levels, more than SIFT3D_HIP_MAX_DOG_STACK levels): the caller then stores the DoG levels
 * (sift3d_hip_dog_stack) and calls sift3d_hip_extrema_mode. */
SIFT3D_AMD_API int sift3d_hip_dogmax_stack(const float *const *d_g, int n_gauss, size_t n,
                                           float *d_absmax, void *stream);
SIFT3D_AMD_API int
sift3d_hip_extrema_gauss6(const float *const *d_g, const float *d_absmax, int nx, int ny, int nz,
                          int z_lo, int z_hi, int tag0, double peak_thresh, sift3d_hip_cand *d_out,
                          uint32_t cap, uint32_t *d_count, void *d_work, size_t work_bytes,
                          void *stream);
/* The same in two phases, so that the sweeps of different octaves can run side by side on different
 * streams: phase 1 = the sweep (bit masks + block counts into d_work; touches nothing else), phase 2 =
 * scan + emission appending at *d_count -- to be issued in octave order once the sweeps are done --,
 * phase 0 = both.  d_work must keep its contents between the phases. */
SIFT3D_AMD_API int
sift3d_hip_extrema_gauss6_phase(const float *const *d_g, const float *d_absmax, int nx, int ny, int nz,
                                int z_lo, int z_hi, int tag0, double peak_thresh, sift3d_hip_cand *d_out,
                                uint32_t cap, uint32_t *d_count, void *d_work, size_t work_bytes,
                                void *stream, int phase);
/* The stage with ONE pass over the octave's Gaussian levels (the dogmax scan of sift.c:821-826 needs no
 * pass of its own).  sift3d_hip_dogmax_sub: maxima of the five |DoG| levels over the sub-lattice
 * z = 1, 6, 11, ..., y = 0, 3, 6, ... (one fifteenth of the bytes), atomically maxed into d_est[0..4] (zeroed by the
 * caller) -- LOWER bounds of the reference's maxima.  sift3d_hip_extrema_gauss6_est_phase: phases as above
 * on the whole volume (planes 1 .. nz - 2); the sweep marks every extremum above peak_thresh * d_est[level]
 * -- a superset of the reference's candidates --, gathers the EXACT maxima into d_exact[0..4] (zeroed by the
 * caller before phase 1) and the reference's threshold (sift.c:829, 842) is then applied to the marked
 * voxels: candidates and maxima are those of sift3d_hip_dogmax_stack + sift3d_hip_extrema_gauss6_phase.
 * Both return 1 when the configuration is not covered. */
SIFT3D_AMD_API int sift3d_hip_dogmax_sub(const float *const *d_g, int nx, int ny, int nz, float *d_est,
                                         void *stream);
/* Phase 2 (scan + emission) of the two entries above/below for ALL octaves of a call in two launches instead
 * of two per octave: octs[i] = what phase 1 of octave i was given.  Appends to d_out at *d_count in (octave,
 * level, z, y, x) order (sift.c:835-868).  Returns 1 when n_oct exceeds what one launch takes. */
typedef struct {
    const float *const *d_g;   /* the octave's six Gaussian levels */
    int nx, ny, nz;
    int tag0;
    void *d_work;
    size_t work_bytes;
} sift3d_hip_extrema_oct;
#define SIFT3D_HIP_EXTREMA_MAX_OCT 12   /* octaves one sift3d_hip_extrema_gauss6_finish call takes */
SIFT3D_AMD_API int
sift3d_hip_extrema_gauss6_finish(const sift3d_hip_extrema_oct *octs, int n_oct, double peak_thresh,
                                 sift3d_hip_cand *d_out, uint32_t cap, uint32_t *d_count, void *stream);
SIFT3D_AMD_API int
sift3d_hip_extrema_gauss6_est_phase(const float *const *d_g, const float *d_est, float *d_exact, int nx,
                                    int ny, int nz, int tag0, double peak_thresh, sift3d_hip_cand *d_out,
                                    uint32_t cap, uint32_t *d_count, void *d_work, size_t work_bytes,
                                    void *stream, int phase);

/* Geometry of one Gaussian level, as the window kernels see it (a table of these
 * lives in device memory, indexed by the `tag`/`level` of a record). */
typedef struct {
    const float *data;
    int nx, ny, nz;  /* local buffer dims */
    int z_off;       /* global z of local plane 0 */
    int nz_glob;     /* global number of planes (window clipping, sift.c:97-99) */
    float ux, uy, uz;/* (float) units of the level (sift.c:88-90) */
    int octave;
    double sd;       /* level scale (sift.c:860) */
} sift3d_hip_level;

/* assign_eig_ori + assign_orientation_thresh (sift.c:926-1102) for n candidates.
 * d_R: 9 floats per candidate (row-major), d_keep: 1 = kept, 0 = rejected. */
SIFT3D_AMD_API int
sift3d_hip_orient(const sift3d_hip_level *d_levels, const sift3d_hip_cand *d_cand,
                  uint32_t n, double corner_thresh, float *d_R, int32_t *d_keep,
                  void *stream);
/* The same with PARALLEL window sums.  Per level a table of the window's voxel offsets and weights is
 * built (a candidate sits on a voxel, so the reference's per-voxel window expressions depend on the
 * level alone); every lane keeps private double sums of its voxels, a fixed butterfly adds them;
 * every decision (sift.c:997, 1011-1015, 1100) and the float casts of the eigenvectors are accepted
 * only when they hold for every value the reference's serial sums can have, and the remaining
 * candidates (a few per cent) are re-run with the serial sums: keypoint lists and R are those of
 * sift3d_hip_orient bit for bit.  d_tab: device scratch of sift3d_hip_orient_tab_bytes(nlevels,
 * max_cand) bytes (tables, launch plan, ten double sums per candidate, list of the undecided).  The
 * caller ZEROES it once after allocating it: a level's table is kept from call to call while the
 * level's scale, units and strides stay the same (a validity mark + the parameters sit in its
 * header).  nlevels = entries of d_levels; a call with n > max_cand takes the serial path. */
SIFT3D_AMD_API size_t sift3d_hip_orient_tab_bytes(int nlevels, uint32_t max_cand);
SIFT3D_AMD_API int
sift3d_hip_orient_tab(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_cand *d_cand,
                      uint32_t n, double corner_thresh, float *d_R, int32_t *d_keep, void *d_tab,
                      uint32_t max_cand, void *stream);
/* (d_tab = NULL: the serial sums for every candidate, = sift3d_hip_orient) */
/* A PART of the candidate list on its own: candidates first .. first + n - 1 (d_cand, d_R, d_keep are the
 * arrays of the WHOLE list), all of levels lv_lo .. lv_hi - 1.  Two parts with disjoint level ranges and
 * different slots (0 or 1: launch plan and undecided list of their own) may run at the same time on two
 * streams over one d_tab: the detector starts octave 0's candidates while the smaller octaves' extrema are
 * still being found.  Results: those of one call over the whole list. */
SIFT3D_AMD_API int
sift3d_hip_orient_tab_part(const sift3d_hip_level *d_levels, int nlevels, int lv_lo, int lv_hi,
                           const sift3d_hip_cand *d_cand, uint32_t first, uint32_t n, double corner_thresh,
                           float *d_R, int32_t *d_keep, void *d_tab, uint32_t max_cand, int slot, void *stream);

typedef struct {
    float R[9];
    float cx, cy, cz; /* (float) keypoint voxel coordinates in its level, global z */
    int32_t level;
    uint32_t row1;    /* 0: the histogram goes to row i of d_hist (i = position in d_kp);
                       * r + 1: to row r -- lets the caller launch the widest windows first
                       * (longest-job-first keeps the kernel's tail short) */
    double sd;
} sift3d_hip_kp;

/* extract_descrip (sift.c:1442-1536): 768 floats per keypoint into d_hist. */
SIFT3D_AMD_API int
sift3d_hip_describe(const sift3d_hip_level *d_levels, const sift3d_hip_kp *d_kp,
                    uint32_t n, float *d_hist, void *stream);

/* The same with the Gaussian window weights tabulated per level (they are looked up by the
 * integer squared voxel distance instead of one division + expf per window voxel; exact for
 * levels whose spacing is one power of two on all axes, the others fall back).  d_wlut: device
 * scratch of sift3d_hip_describe_wlut_floats(nlevels) floats, rebuilt by every call; nlevels =
 * number of entries of d_levels. */
SIFT3D_AMD_API size_t sift3d_hip_describe_wlut_floats(int nlevels);
SIFT3D_AMD_API int
sift3d_hip_describe_wlut(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                         uint32_t n, float *d_hist, float *d_wlut, void *stream);
/* The full entry: as sift3d_hip_describe_ex with the scratch of the split windows.  The fast kernel sums a window
 * in four parts (ranges of its planes: work items of a quarter of the size shorten the drain of the persistent
 * kernel from ~1 ms to ~0.3 at 512^3) and the wave that finishes a keypoint's last part adds the parts' histograms
 * in part order -- a function of the keypoint alone, so every entry of this family gives the same bits for the
 * same keypoint.  d_part: device scratch of sift3d_hip_describe_part_bytes(n - n_exact) bytes (3.2 KB per part,
 * 12.8 KB per keypoint) or NULL; the entries without the argument allocate it for the call and free it behind a
 * stream synchronisation. */
SIFT3D_AMD_API size_t sift3d_hip_describe_part_bytes(uint32_t n);
SIFT3D_AMD_API int
sift3d_hip_describe_parts(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                          uint32_t n, uint32_t n_exact, float *d_hist, float *d_hist2, float *d_wlut,
                          void *d_part, void *stream);
/* Clock probe of the last descriptor launch through d_wlut (the fast kernel; exact != 0: the reference-order
 * one): shader cycles and ticks of the constant 100 MHz counter during which the launch's first -- persistent --
 * wave was alive.  cycles / (ticks / 1e8) = the clock the device held under the kernel.  Blocks on `stream`. */
SIFT3D_AMD_API int
sift3d_hip_describe_clock(const float *d_wlut, int nlevels, int exact, uint64_t *cycles, uint64_t *ticks,
                          void *stream);
/* The same with a second destination: d_hist2 (device memory, may be NULL) receives a copy of every
 * histogram -- the matcher's input stays in HBM (sift3d_amd_descriptor_store_keep_device). */
SIFT3D_AMD_API int
sift3d_hip_describe_wlut2(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp,
                          uint32_t n, float *d_hist, float *d_hist2, float *d_wlut, void *stream);
/* The same with the first n_exact records computed in the reference's accumulation ORDER and term
 * arithmetic (one histogram, voxels added in scan order, rounded products, sequential double norm): those
 * histograms are the reference's bit for bit, at ~1.5x the time per window voxel.  For windows so wide that
 * a bin receives enough terms for any other summation order to drift past 1e-5 relative of the reference's
 * own sequential float sums (sift.c:1371-1373); see sift3d_amd_detector_set_exact_descriptors. */
SIFT3D_AMD_API int
sift3d_hip_describe_ex(const sift3d_hip_level *d_levels, int nlevels, const sift3d_hip_kp *d_kp, uint32_t n,
                       uint32_t n_exact, float *d_hist, float *d_hist2, float *d_wlut, void *stream);

/* Device stages of the slab driver's keypoint exchange (sift3d_slab.hip): per-(octave, level) counts of a
 * rank's candidates; the rank's block of the all-gather (|DoG| of every candidate, then the kept candidates as
 * 56-byte records {o, s, x, y, z (global), R[9]} in order); the global list built from the gathered blocks
 * (64-byte records {.., strength, pad} behind a 64-byte header whose first int32 is the ranks' status). */
SIFT3D_AMD_API int sift3d_hip_slab_count(const sift3d_hip_cand *d_cand, const int32_t *d_keep, uint32_t n,
                                         int ngl, int K, int nkey, int32_t *d_cnt, void *stream);
SIFT3D_AMD_API size_t sift3d_hip_slab_pack_scratch_bytes(uint32_t n);
SIFT3D_AMD_API int sift3d_hip_slab_pack(const sift3d_hip_level *d_levels, const sift3d_hip_cand *d_cand,
                                        const int32_t *d_keep, const float *d_R, uint32_t n, int ngl,
                                        float *d_vals, void *d_recs, void *d_scratch, void *stream);
SIFT3D_AMD_API int sift3d_hip_slab_build(const void *d_all, size_t blk_bytes, size_t roff, size_t toff,
                                         int world, int nkey, const uint32_t *d_tab, uint32_t tot_k,
                                         uint32_t tot_c, void *d_out, void *stream);

/* Icosahedron face table for the descriptor kernel (init_geometry, sift.c:148-259;
 * per-face constants of cart2bary, sift.c:276-297).  20 records of
 * {v0[3], e1[3], e2[3], t[3], q[3], e2.q, idx[3] (as float)} = 19 floats each. */
#define SIFT3D_HIP_FACE_FLOATS 19
SIFT3D_AMD_API int
sift3d_hip_set_mesh(const float *faces /* 20 * SIFT3D_HIP_FACE_FLOATS */);

/* Order-independent synthetic volume (sift3d_amd/csrc/synth.c) generated on the
 * device: planes [z_off, z_off + nz) of a volume with nx*ny rows. */
SIFT3D_AMD_API int
sift3d_hip_synth_lattice(float *d_dst, int nx, int ny, int nz, int z_off, uint64_t seed,
                         void *stream);

/* Host evaluations of the device math (tests compare them with libm / LAPACK). */
SIFT3D_AMD_API void sift3d_amd_host_expf(const float *in, float *out, size_t n);
SIFT3D_AMD_API void sift3d_amd_host_eigen3(const double *A9, double *Q9, double *L3);
/* The same two routines evaluated ON the device (one thread per element). */
SIFT3D_AMD_API int sift3d_hip_test_expf(const float *d_in, float *d_out, size_t n, void *stream);
SIFT3D_AMD_API int sift3d_hip_test_eigen3(const double *d_A9, double *d_Q9, double *d_L3,
                                          size_t n, void *stream);

SIFT3D_AMD_API const char *sift3d_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif

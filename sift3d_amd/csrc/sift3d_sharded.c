/* sift3d_sharded.c -- Z-slab multi-GPU detect + describe, host side in C (included at the end of
 * sift3d_host.c: it uses the private store layouts).
 *
 * One process per GPU; the volume is cut along z into one slab per rank.  Everything that is
 * local in z runs unchanged on the slab (x / y FIR passes, dogmax, extrema, down-sampling on
 * 2^k-aligned slab boundaries); exactly three kinds of exchange cross ranks, all through a small
 * transport vtable (sift3d_amd_transport: RCCL send/recv, all-reduce, all-gather over xGMI in
 * production -- sift3d_amd_rccl_transport below -- or anything a test provides):
 *
 *   halo      nearest-neighbour exchange of z planes: the input of every z FIR pass
 *             (ceil(hw * unit_factor) + 1 planes per side), and after the pyramid the planes the
 *             orientation / descriptor windows of this rank's keypoints reach into;
 *   max       all-reduce(max) of max|input| (im_scale, imutil.c:699-713) and of the per-level
 *             max|DoG| (sift.c:821-826) -- order-free, hence exact;
 *   gather    all-gather of the per-level candidate counts, the candidates' |DoG| values and the
 *             ORIENTED keypoints (rejected candidates never leave their rank).
 *
 * Ranks are concatenated in slab order inside each (octave, level), which reproduces the
 * reference's global (o, s, z, y, x) order and its stale-strength quirk (sift.c:372-384), so the
 * result equals the single-GPU result bit for bit (tests/test_gpu_sharded.py).  Mirror rules
 * apply at the GLOBAL faces only: the stage kernels take the global length and the slab offset.
 * Octaves whose slabs would be thinner than the window halo are all-gathered and computed on
 * every rank (< 1 % of the voxels); their window work is still split by z.
 *
 * The z-halo exchange of a blur overlaps with the z pass of the slab's interior planes: the
 * exchange is enqueued on a second stream as soon as the x pass has finished, the interior planes
 * (which need no halo) are filtered meanwhile, the 2 * reach boundary planes afterwards.
 *
 * Scope: everything the drop-in API accepts -- any number of keypoint levels per octave
 * (sift.c:527-533), the cuboid extrema neighbourhood (sift.c:24, 761-796), any volume size (the
 * reference halves dimensions with integer division, imutil.c:1545-1547, so rows that are not
 * whole quads are a normal case).  Octaves the DoG-free extrema sweep does not cover store their
 * DoG levels and take sift3d_hip_extrema_mode, exactly as the single-GPU path does
 * (sift3d_host.c: detect_on_device).  tests/sharded_py.py is the same orchestration in Python on a
 * pluggable compute backend: test infrastructure for the gloo runs on the CPU oracle. */

#include <dlfcn.h>

#define SH_MAX_K 8                  /* keypoint levels per octave (as sift3d_host.c) */
#define SH_MAX_NGL (SH_MAX_K + 3)
#define SH_MAX_NDL (SH_MAX_K + 2)
#define SH_MAX_OCT 32
#define SH_MAX_WORLD 64

typedef struct {
    float *t;           /* device planes [off, off + nloc) of the level */
    int off, nloc;      /* local buffer */
    int z0, z1;         /* owned (or, on replicated octaves, work-split) planes, global */
    int nz_glob;
} sh_level;

typedef struct {          /* an oriented keypoint as exchanged between ranks */
    int32_t o, s, x, y, z;
    float R[9];
} sh_gkp;

struct sift3d_amd_sharded {
    sift3d_amd_transport T;
    int rank, world;
    double peak_thresh, corner_thresh, sigma_n, sigma0;
    double units[3];
    int nx, ny, nz;
    /* geometry */
    int num_octaves, o_shard, halo;
    int dims[SH_MAX_OCT][3];
    int b0[SH_MAX_WORLD + 1];
    int bounds[SH_MAX_OCT][SH_MAX_WORLD + 1];
    int K, ngl, ndl;                 /* keypoint / Gaussian / DoG levels per octave */
    int cuboid;
    int exact_desc;                  /* sift3d_amd_detector_set_exact_descriptors of `params` */
    int dog_free[SH_MAX_OCT];        /* octave takes the DoG-free sweep (no stored DoG levels) */
    int win_reach[SH_MAX_NGL];
    filter_t filt[SH_MAX_NGL];
    /* device state */
    void *stream, *comm_stream, *ev_x, *ev_halo, *ev0, *ev1, *ev2, *ev3, *ev4;
    void *oct_stream, *ev_fork, *ev_join;   /* extrema sweeps of octaves >= 1 beside octave 0's */
    /* The pyramid's three dependency chains, as in the single-GPU path (detect_on_device): [0] octave 0,
     * [1] the first levels of the octaves >= 1 (their down-sampling source included), [2] their last
     * levels, which nothing but their own DoG waits for.  A chain has its stream, its scratch volumes and
     * its pair of events around the z-halo exchange (which always travels on comm_stream). */
    struct sh_chain {
        void *stream;
        float *tmp_a, *tmp_b;
        void *ev_x, *ev_halo;
    } chain[3];
    void *side_stream, *ev_oct[SH_MAX_OCT], *ev_join2, *ev_tr;
    float *d_tmp2_a, *d_tmp2_b, *d_tmp3_a, *d_tmp3_b;
    void *d_work2;                          /* their work areas, kept until the ordered emission */
    size_t work2_off[SH_MAX_OCT], work2_sz[SH_MAX_OCT], work2_bytes;
    sh_level G[SH_MAX_OCT][SH_MAX_NGL];
    float *D[SH_MAX_OCT][SH_MAX_NDL];   /* stored DoG levels (geometry of G[o][*]), where needed */
    sh_level tmp_a, tmp_b, im;       /* scratch levels of octave 0 size (re-described per octave) */
    float *d_tmp_a, *d_tmp_b, *d_im, *d_raw, *d_stage;
    size_t tmp_elems, stage_elems;
    int in_z0, in_z1;
    float *d_scalars;                /* [0] input max, [1] candidate count, [8 + 5 o + k] dogmax */
    sift3d_hip_level h_levels[SH_MAX_OCT * SH_MAX_NGL], *d_levels;
    void *d_work;
    size_t work_bytes;
    float *d_wlut;
    void *d_dpart;         /* scratch of the descriptor kernel's split windows */
    size_t dpart_bytes;
    void *d_otab;          /* orientation window tables + candidate sums (sift3d_hip_orient_tab) */
    size_t otab_bytes;
    sift3d_hip_cand *d_cand;
    float *d_R;
    int32_t *d_keep;
    uint32_t cand_cap;
    void *d_xchg, *h_xchg;           /* the ranks' gathered blocks (device); the final list (pinned host) */
    size_t xchg_bytes, hxchg_cap;
    void *d_cnt, *h_cnt, *d_tab, *h_tab, *d_out, *d_pack;   /* counts, segment tables, list, scan scratch */
    size_t cnt_cap, hcnt_cap, tab_cap, htab_cap, out_cap, pack_cap;
    void *d_gath;                    /* descriptor gather staging (sift3d_amd_sharded_gather_descriptors) */
    size_t gath_cap;
    int failed, inject;              /* local failure mark of the running call; test hook */
    sift3d_hip_kp *h_kp;
    uint32_t kp_cap;
    int ncand;
    /* [0] pyramid (device s)  [1] detect wall  [2] describe wall  [3] DoG maxima + extrema (device s,
     * incl. the all-reduce)  [4] wait for the window halos + orientation (device s)  [5] gathers +
     * global list (host s)  [6] input scaling (device s, incl. the all-reduce) */
    double t[8];
    double t_gather0;
};

static int sh_sharded(const sift3d_amd_sharded *S, int o) { return o < S->o_shard; }

static double sh_scale(const sift3d_amd_sharded *S, int o, int s)
{
    return S->sigma0 * pow(2.0, o + (double)s / S->K);           /* imutil.c:1578-1579 */
}

static size_t sh_plane(const sift3d_amd_sharded *S, int o) { return (size_t)S->dims[o][0] * S->dims[o][1]; }

/* ---- geometry ------------------------------------------------------------------------------ */
static int sh_geometry(sift3d_amd_sharded *S)
{
    int mn = S->nx < S->ny ? S->nx : S->ny, last, o, r, d[3], align, min_slab;
    mn = mn < S->nz ? mn : S->nz;
    last = (int)log2((double)mn) - 3;                              /* sift.c:442-444 */
    if (last < 0) {
        ERR("resize_SIFT3D: input image is too small: must have at least 8 voxels in each "
            "dimension \n");
        return SIFT3D_FAILURE;
    }
    S->num_octaves = last + 1;
    if (S->num_octaves > SH_MAX_OCT)
        return SIFT3D_FAILURE;
    d[0] = S->nx; d[1] = S->ny; d[2] = S->nz;
    for (o = 0; o < S->num_octaves; o++) {
        memcpy(S->dims[o], d, sizeof(d));
        d[0] /= 2; d[1] /= 2; d[2] /= 2;                           /* imutil.c:1545-1547 */
    }
    min_slab = S->halo + 8;
    S->o_shard = 0;
    if (S->world > 1)
        while (S->o_shard < S->num_octaves && S->dims[S->o_shard][2] / S->world >= min_slab)
            S->o_shard++;
    /* slab bounds at octave 0: multiples of 2^o_shard, so that every slab boundary is even in
     * every sharded octave and im_downsample_2x never needs a neighbour's plane */
    align = 1 << S->o_shard;
    S->b0[0] = 0;
    for (r = 1; r < S->world; r++)
        S->b0[r] = (int)nearbyint((double)S->nz * r / S->world / align) * align;
    S->b0[S->world] = S->nz;
    for (o = 0; o < S->num_octaves; o++) {
        const int nzo = S->dims[o][2];
        for (r = 0; r < S->world; r++)
            S->bounds[o][r] = sh_sharded(S, o) ? ((S->b0[r] >> o) < nzo ? (S->b0[r] >> o) : nzo)
                                                 : (int)((long)nzo * r / S->world);
        S->bounds[o][S->world] = nzo;
    }
    return SIFT3D_SUCCESS;
}

static void sh_free_level(sh_level *L)
{
    sift3d_hip_free(L->t);
    L->t = NULL;
}

void sift3d_amd_sharded_free(sift3d_amd_sharded *S)
{
    int o, s;
    if (!S)
        return;
    if (S->stream)
        sift3d_hip_stream_sync(S->stream);
    if (S->comm_stream)
        sift3d_hip_stream_sync(S->comm_stream);
    if (S->oct_stream)
        sift3d_hip_stream_sync(S->oct_stream);
    if (S->side_stream)
        sift3d_hip_stream_sync(S->side_stream);
    for (o = 0; o < SH_MAX_OCT; o++) {
        for (s = 0; s < SH_MAX_NGL; s++)
            sh_free_level(&S->G[o][s]);
        for (s = 0; s < SH_MAX_NDL; s++)
            sift3d_hip_free(S->D[o][s]);
    }
    for (s = 0; s < SH_MAX_NGL; s++)
        free(S->filt[s].taps);
    sift3d_hip_free(S->d_tmp_a); sift3d_hip_free(S->d_tmp_b); sift3d_hip_free(S->d_im);
    sift3d_hip_free(S->d_tmp2_a); sift3d_hip_free(S->d_tmp2_b); sift3d_hip_free(S->d_tmp3_a);
    sift3d_hip_free(S->d_tmp3_b);
    sift3d_hip_free(S->d_raw); sift3d_hip_free(S->d_stage); sift3d_hip_free(S->d_scalars);
    sift3d_hip_free(S->d_levels); sift3d_hip_free(S->d_work); sift3d_hip_free(S->d_work2);
    sift3d_hip_free(S->d_cand);
    sift3d_hip_free(S->d_R); sift3d_hip_free(S->d_keep); sift3d_hip_free(S->d_xchg);
    sift3d_hip_free(S->d_wlut); sift3d_hip_free(S->d_otab); sift3d_hip_free(S->d_dpart);
    sift3d_hip_host_free(S->h_xchg); sift3d_hip_host_free(S->h_kp);
    sift3d_hip_free(S->d_cnt); sift3d_hip_free(S->d_tab); sift3d_hip_free(S->d_out); sift3d_hip_free(S->d_pack);
    sift3d_hip_free(S->d_gath);
    sift3d_hip_host_free(S->h_cnt); sift3d_hip_host_free(S->h_tab);
    sift3d_hip_event_destroy(S->ev_x); sift3d_hip_event_destroy(S->ev_halo);
    sift3d_hip_event_destroy(S->ev0); sift3d_hip_event_destroy(S->ev1);
    sift3d_hip_event_destroy(S->ev2); sift3d_hip_event_destroy(S->ev3); sift3d_hip_event_destroy(S->ev4);
    sift3d_hip_event_destroy(S->ev_fork); sift3d_hip_event_destroy(S->ev_join);
    sift3d_hip_event_destroy(S->ev_join2); sift3d_hip_event_destroy(S->ev_tr);
    sift3d_hip_event_destroy(S->chain[1].ev_x); sift3d_hip_event_destroy(S->chain[1].ev_halo);
    sift3d_hip_event_destroy(S->chain[2].ev_x); sift3d_hip_event_destroy(S->chain[2].ev_halo);
    {
        int oo;
        for (oo = 0; oo < SH_MAX_OCT; oo++)
            sift3d_hip_event_destroy(S->ev_oct[oo]);
    }
    sift3d_hip_stream_destroy(S->side_stream);
    sift3d_hip_stream_destroy(S->oct_stream);
    sift3d_hip_stream_destroy(S->comm_stream);
    sift3d_hip_stream_destroy(S->stream);
    free(S);
}

sift3d_amd_sharded *sift3d_amd_sharded_create(int nx, int ny, int nz, const sift3d_amd_transport *T,
                                              const sift3d_detector *params, double ux, double uy,
                                              double uz)
{
    sift3d_amd_sharded *S;
    int o, s, i, hw_max = 0, blur_reach;
    size_t work = 0, n0;
    if (!T || T->world < 1 || T->world > SH_MAX_WORLD || T->rank < 0 || T->rank >= T->world ||
        nx < 8 || ny < 8 || nz < 8 || !(ux > 0) || !(uy > 0) || !(uz > 0))
        return NULL;
    if (T->world > 1 && (!T->halo || !T->allreduce_max || !T->allgather))
        return NULL;
    if (!sift3d_amd_device_available()) {
        ERR("sift3d_amd: no HIP device is available; this library has no CPU path \n");
        return NULL;
    }
    if (params && (params->num_kp_levels < 1 || params->num_kp_levels > SH_MAX_K)) {
        ERR("sift3d_amd_sharded: 1 to %d keypoint levels per octave are supported \n", SH_MAX_K);
        return NULL;
    }
    S = (sift3d_amd_sharded *)calloc(1, sizeof(*S));
    if (!S)
        return NULL;
    S->K = params ? params->num_kp_levels : 3;
    S->ngl = S->K + 3;                                             /* sift.c:434-437 */
    S->ndl = S->K + 2;
    S->cuboid = params ? params->cuboid_extrema : 0;
    S->exact_desc = params ? params->exact_desc : 0;
    S->T = *T;
    S->rank = T->rank; S->world = T->world;
    S->peak_thresh = params ? params->peak_thresh : peak_thresh_default;
    S->corner_thresh = params ? params->corner_thresh : corner_thresh_default;
    S->sigma_n = params ? params->sigma_n : sigma_n_default;
    S->sigma0 = params ? params->sigma0 : sigma0_default;
    S->units[0] = ux; S->units[1] = uy; S->units[2] = uz;
    S->nx = nx; S->ny = ny; S->nz = nz;
    if (S->sigma0 * pow(2.0, -1.0 / S->K) < S->sigma_n) {          /* imutil.c:1582-1588 */
        ERR("set_scales_Pyramid: sigma_n too large for these settings. \n");
        goto fail;
    }
    /* filter bank (make_gss, imutil.c:1360-1409) */
    for (i = 0; i < S->ngl; i++) {
        const double s_cur = i == 0 ? S->sigma_n : S->sigma0 * pow(2.0, (double)(i - 2) / S->K);
        const double s_next = S->sigma0 * pow(2.0, (double)(i - 1) / S->K);
        if (gauss_filter(&S->filt[i], sqrt(s_next * s_next - s_cur * s_cur)))
            goto fail;
        if (S->filt[i].width / 2 > hw_max)
            hw_max = S->filt[i].width / 2;
    }
    /* halo planes a slab needs from its neighbours, in LEVEL planes (the same in every octave):
     * descriptor window of Gaussian level s: 14.1422 * sigma0 * 2^((s-1)/K) / uz + 2
     * (sift.c:1453-1454); z pass: ceil(hw * unit_factor) + 1 (imutil.c:756-757) */
    for (s = 0; s < S->ngl; s++)
        S->win_reach[s] = (s >= 1 && s <= S->K)
                              ? (int)ceil(14.1422 * S->sigma0 * pow(2.0, (double)(s - 1) / S->K) / uz) + 2
                              : 1;
    blur_reach = (int)ceil((double)hw_max * (double)(float)(1.0 / uz)) + 1;
    S->halo = blur_reach;
    for (s = 0; s < S->ngl; s++)
        if (S->win_reach[s] > S->halo)
            S->halo = S->win_reach[s];
    if (S->halo > 500) {
        ERR("sift3d_amd_sharded: sigma0 / units give a %d-plane window \n", S->halo);
        goto fail;
    }
    if (sh_geometry(S))
        goto fail;
    /* the DoG-free sweep covers the default level count and 8-neighbour test on rows of whole quads
     * (sift3d_hip_dogmax_stack / sift3d_hip_extrema_gauss6); other octaves store their DoG levels */
    for (o = 0; o < S->num_octaves; o++)
        S->dog_free[o] = !S->cuboid && S->ngl == 6 && (S->dims[o][0] & 3) == 0;
    if (!(S->stream = sift3d_hip_stream_create()) || !(S->comm_stream = sift3d_hip_stream_create()) ||
        !(S->ev_x = sift3d_hip_event_create()) || !(S->ev_halo = sift3d_hip_event_create()) ||
        !(S->ev0 = sift3d_hip_event_create()) || !(S->ev1 = sift3d_hip_event_create()) ||
        !(S->ev2 = sift3d_hip_event_create()) || !(S->ev3 = sift3d_hip_event_create()) ||
        !(S->ev4 = sift3d_hip_event_create()) ||
        !(S->oct_stream = sift3d_hip_stream_create_high()) || !(S->ev_fork = sift3d_hip_event_create()) ||
        !(S->ev_join = sift3d_hip_event_create()) || !(S->side_stream = sift3d_hip_stream_create_high()) ||
        !(S->ev_join2 = sift3d_hip_event_create()) || !(S->ev_tr = sift3d_hip_event_create()) ||
        !(S->chain[1].ev_x = sift3d_hip_event_create()) || !(S->chain[1].ev_halo = sift3d_hip_event_create()) ||
        !(S->chain[2].ev_x = sift3d_hip_event_create()) || !(S->chain[2].ev_halo = sift3d_hip_event_create()) ||
        upload_mesh())
        goto fail;
    for (o = 0; o < S->num_octaves; o++)
        if (!(S->ev_oct[o] = sift3d_hip_event_create()))
            goto fail;
    /* levels */
    for (o = 0; o < S->num_octaves; o++) {
        const int nzo = S->dims[o][2];
        const int z0 = S->bounds[o][S->rank], z1 = S->bounds[o][S->rank + 1];
        const int off = sh_sharded(S, o) ? (z0 - S->halo > 0 ? z0 - S->halo : 0) : 0;
        const int hi = sh_sharded(S, o) ? (z1 + S->halo < nzo ? z1 + S->halo : nzo) : nzo;
        const size_t w = sift3d_hip_extrema_work_bytes(S->dims[o][0], S->dims[o][1], hi - off, S->K);
        work = w > work ? w : work;
        if (o >= 1) {
            S->work2_off[o] = S->work2_bytes;
            S->work2_sz[o] = w;
            S->work2_bytes += (w + 255) & ~(size_t)255;
        }
        /* (all levels of an octave share one geometry -- the fused sweeps take them as a stack; only
         * the keypoint levels 1..K USE the full window halo, the others a blur's or the extrema
         * test's reach) */
        for (s = 0; s < S->ngl; s++) {
            sh_level *L = &S->G[o][s];
            L->off = off; L->nloc = hi - off; L->z0 = z0; L->z1 = z1; L->nz_glob = nzo;
            if (!(L->t = (float *)sift3d_hip_malloc(sh_plane(S, o) * (size_t)L->nloc * sizeof(float))))
                goto fail;
        }
        if (!S->dog_free[o])
            for (s = 0; s < S->ndl; s++)
                if (!(S->D[o][s] = (float *)sift3d_hip_malloc(sh_plane(S, o) * (size_t)(hi - off) * sizeof(float))))
                    goto fail;
    }
    n0 = sh_plane(S, 0) * (size_t)S->G[0][0].nloc;
    S->tmp_elems = n0;
    S->in_z0 = sh_sharded(S, 0) ? S->bounds[0][S->rank] : 0;
    S->in_z1 = sh_sharded(S, 0) ? S->bounds[0][S->rank + 1] : nz;
    S->d_tmp_a = (float *)sift3d_hip_malloc(n0 * sizeof(float));
    S->d_tmp_b = (float *)sift3d_hip_malloc(n0 * sizeof(float));
    S->d_im = (float *)sift3d_hip_malloc(n0 * sizeof(float));
    {
        const size_t n1 = S->num_octaves > 1 ? sh_plane(S, 1) * (size_t)S->G[1][0].nloc : 4;
        S->d_tmp2_a = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        S->d_tmp2_b = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        S->d_tmp3_a = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        S->d_tmp3_b = (float *)sift3d_hip_malloc(n1 * sizeof(float));
        if (!S->d_tmp2_a || !S->d_tmp2_b || !S->d_tmp3_a || !S->d_tmp3_b)
            goto fail;
    }
    S->d_raw = (float *)sift3d_hip_malloc(sh_plane(S, 0) * (size_t)(S->in_z1 - S->in_z0) * sizeof(float));
    /* staging of the sharded -> replicated transition: this rank's down-sampled slab, then the
     * world slabs gathered (all padded to the largest) */
    {
        size_t stage = 16;
        if (S->world > 1 && S->o_shard >= 1 && S->o_shard < S->num_octaves) {
            const int t = S->o_shard, mzg = S->dims[t][2];
            int r, mxn = 1;
            for (r = 0; r < S->world; r++) {
                int lo = S->b0[r] >> t, hi = r + 1 < S->world ? S->b0[r + 1] >> t : mzg;
                lo = lo < mzg ? lo : mzg;
                hi = hi < mzg ? hi : mzg;
                if (hi - lo > mxn)
                    mxn = hi - lo;
            }
            stage = (size_t)mxn * sh_plane(S, t) * (size_t)(S->world + 1);
        }
        S->stage_elems = stage;
        S->d_stage = (float *)sift3d_hip_malloc(stage * sizeof(float));
    }
    S->d_scalars = (float *)sift3d_hip_malloc(sizeof(float) * (8 + SH_MAX_NDL * SH_MAX_OCT));
    S->d_levels = (sift3d_hip_level *)sift3d_hip_malloc(sizeof(sift3d_hip_level) * SH_MAX_OCT * SH_MAX_NGL);
    S->d_work = sift3d_hip_malloc(work);
    S->work_bytes = work;
    if (S->work2_bytes && !(S->d_work2 = sift3d_hip_malloc(S->work2_bytes)))
        goto fail;
    S->d_wlut = (float *)sift3d_hip_malloc(sizeof(float) *
                                           sift3d_hip_describe_wlut_floats(S->num_octaves * S->ngl));
    if (!S->d_wlut || !S->d_tmp_a || !S->d_tmp_b || !S->d_im || !S->d_raw || !S->d_stage || !S->d_scalars ||
        !S->d_levels || !S->d_work)
        goto fail;
    S->chain[0].stream = S->stream;      S->chain[0].tmp_a = S->d_tmp_a;  S->chain[0].tmp_b = S->d_tmp_b;
    S->chain[0].ev_x = S->ev_x;          S->chain[0].ev_halo = S->ev_halo;
    S->chain[1].stream = S->oct_stream;  S->chain[1].tmp_a = S->d_tmp2_a; S->chain[1].tmp_b = S->d_tmp2_b;
    S->chain[2].stream = S->side_stream; S->chain[2].tmp_a = S->d_tmp3_a; S->chain[2].tmp_b = S->d_tmp3_b;
    /* level table (window kernels) */
    for (o = 0; o < S->num_octaves; o++)
        for (s = 0; s < S->ngl; s++) {
            sift3d_hip_level *L = &S->h_levels[o * S->ngl + s];
            L->data = S->G[o][s].t;
            L->nx = S->dims[o][0]; L->ny = S->dims[o][1]; L->nz = S->G[o][s].nloc;
            L->z_off = S->G[o][s].off;
            L->nz_glob = S->dims[o][2];
            L->ux = (float)(ux * ldexp(1.0, o));
            L->uy = (float)(uy * ldexp(1.0, o));
            L->uz = (float)(uz * ldexp(1.0, o));
            L->octave = o;
            L->sd = sh_scale(S, o, s - 1);
        }
    if (sift3d_hip_memcpy_h2d(S->d_levels, S->h_levels,
                              sizeof(sift3d_hip_level) * (size_t)S->num_octaves * S->ngl, S->stream) ||
        sift3d_hip_stream_sync(S->stream))
        goto fail;
    return S;
fail:
    sift3d_amd_sharded_free(S);
    return NULL;
}

int sift3d_amd_sharded_own_planes(const sift3d_amd_sharded *S, int *z0, int *z1)
{
    if (!S)
        return SIFT3D_FAILURE;
    if (z0) *z0 = S->in_z0;
    if (z1) *z1 = S->in_z1;
    return SIFT3D_SUCCESS;
}

float *sift3d_amd_sharded_input(sift3d_amd_sharded *S) { return S ? S->d_raw : NULL; }

int sift3d_amd_sharded_synth(sift3d_amd_sharded *S, uint64_t seed)
{
    if (!S)
        return SIFT3D_FAILURE;
    if (sift3d_hip_synth_lattice(S->d_raw, S->nx, S->ny, S->in_z1 - S->in_z0, S->in_z0, seed, S->stream))
        return SIFT3D_FAILURE;
    return sift3d_hip_stream_sync(S->stream);
}

int sift3d_amd_sharded_num_candidates(const sift3d_amd_sharded *S) { return S ? S->ncand : -1; }
const double *sift3d_amd_sharded_timings(const sift3d_amd_sharded *S) { return S ? S->t : NULL; }
int sift3d_amd_sharded_info(const sift3d_amd_sharded *S, int *num_octaves, int *o_shard, int *halo)
{
    if (!S)
        return SIFT3D_FAILURE;
    if (num_octaves) *num_octaves = S->num_octaves;
    if (o_shard) *o_shard = S->o_shard;
    if (halo) *halo = S->halo;
    return SIFT3D_SUCCESS;
}

/* ---- exchanges ------------------------------------------------------------------------------ */
/* Fill halo planes on both sides of the owned range of a level buffer (plane stride `plane`,
 * geometry of `L`) from the slab neighbours, on `stream`: the planes at distance (d0, h] from the
 * slab boundary (d0 = 0: all h of them). */
static int sh_halo_range(sift3d_amd_sharded *S, float *t, const sh_level *L, size_t plane, int d0, int h,
                         void *stream)
{
    const int a = L->z0 - L->off, b = L->z1 - L->off;
    const int lo = S->rank > 0 && L->z0 > 0, hi = S->rank < S->world - 1 && L->z1 < L->nz_glob;
    if (S->world == 1 || h <= d0 || (!lo && !hi))
        return SIFT3D_SUCCESS;
    if (h > S->halo || h > b - a || d0 < 0) {
        ERR("sift3d_amd_sharded: a %d-plane halo does not fit slabs of %d planes \n", h, b - a);
        return SIFT3D_FAILURE;
    }
    return S->T.halo(S->T.ctx, lo ? t + (size_t)(a + d0) * plane : NULL, lo ? t + (size_t)(a - h) * plane : NULL,
                     hi ? t + (size_t)(b - h) * plane : NULL, hi ? t + (size_t)(b + d0) * plane : NULL,
                     (size_t)(h - d0) * plane * sizeof(float), stream);
}

static int sh_halo(sift3d_amd_sharded *S, float *t, const sh_level *L, size_t plane, int h, void *stream)
{
    return sh_halo_range(S, t, L, plane, 0, h, stream);
}

static int sh_fir(sift3d_amd_sharded *S, const float *src, float *dst, int o, int nloc, int axis,
                  const filter_t *f, float uf, int off, int z_lo, int z_hi, void *stream)
{
    sift3d_hip_fir_args a;
    memset(&a, 0, sizeof(a));
    a.src = src; a.dst = dst;
    a.nx = S->dims[o][0]; a.ny = S->dims[o][1]; a.nz = nloc;
    a.axis = axis; a.width = f->width; a.taps = f->taps;
    a.unit_factor = uf;
    a.n_glob = S->dims[o][2]; a.off = off;
    a.z_lo = z_lo; a.z_hi = z_hi;
    return sift3d_hip_fir(&a, stream);
}

/* ---- failures between collectives ------------------------------------------------------------- */
/* detect and describe are collective.  A rank whose LOCAL work fails (a launch, an allocation, an injected
 * test failure) must not leave the others waiting in the next exchange: it marks itself failed, skips its
 * remaining local work and still issues every exchange in the common order (with whatever its buffers hold);
 * its status word travels behind the first all-gather of the step (the counts) and behind the second (the
 * records), and EVERY rank returns SIFT3D_FAILURE at the same point, with the transport in step for the next
 * call -- the reference's "every stage returns -1 to its caller" (immacros.h:27-32), across ranks.  A
 * failure of the transport itself is everybody's by nature and returns at once. */
#define SH_DO(call)                                        \
    do {                                                   \
        if (!S->failed && (call))                          \
            S->failed = __LINE__;                          \
    } while (0)
#define SH_COMM(call)                                      \
    do {                                                   \
        if (call)                                          \
            return SIFT3D_FAILURE;                         \
    } while (0)

/* test hook: the next detect (where = 1: before the pyramid, 2: after the extrema, 3: between the two
 * all-gathers) or describe-gather (4) of THIS rank fails locally */
int sift3d_amd_sharded_inject_failure(sift3d_amd_sharded *S, int where)
{
    if (!S || where < 0 || where > 4)
        return SIFT3D_FAILURE;
    S->inject = where;
    return SIFT3D_SUCCESS;
}

/* apply_Sep_FIR_filter (imutil.c:1127-1206) on a level: src -> dst (same geometry).  Sharded
 * octaves: x (and y) on the owned planes, halo exchange of the z pass's input overlapped with the
 * z pass of the interior planes. */
/* d_scale_max (the first blur of the pyramid only, else NULL): the blur of src / *d_scale_max -- im_scale
 * folded into the x pass (sift3d_hip_fir_x_scaled); returns 2 without doing anything when that x pass does
 * not cover the configuration (the same answer on every rank) */
static int sh_blur(sift3d_amd_sharded *S, const struct sh_chain *C, int o, const float *src, const sh_level *Lg,
                   float *dst, const filter_t *f, const float *d_scale_max)
{
    const size_t plane = sh_plane(S, o);
    const int nx = S->dims[o][0], ny = S->dims[o][1], nzo = S->dims[o][2];
    const float ufx = (float)(1.0 / (S->units[0] * ldexp(1.0, o)));
    const float ufy = (float)(1.0 / (S->units[1] * ldexp(1.0, o)));
    const float ufz = (float)(1.0 / (S->units[2] * ldexp(1.0, o)));
    const int hw = f->width / 2;
    const int a = sh_sharded(S, o) ? Lg->z0 - Lg->off : 0;
    const int b = sh_sharded(S, o) ? Lg->z1 - Lg->off : Lg->nloc;
    const int reach = (int)ceilf((float)hw * ufz) + 1;
    const int exch = S->world > 1 && sh_sharded(S, o);
    float *zin;
    int fused, ia, ib;
    if (d_scale_max) {
        sift3d_hip_fir_args fa;
        int rc;
        memset(&fa, 0, sizeof(fa));
        fa.src = src; fa.dst = C->tmp_a;
        fa.nx = nx; fa.ny = ny; fa.nz = Lg->nloc;
        fa.axis = 0; fa.width = f->width; fa.taps = f->taps;
        fa.unit_factor = ufx;
        fa.n_glob = nzo; fa.off = 0;
        fa.z_lo = a; fa.z_hi = b;
        /* (whether this x pass covers the configuration does not depend on the rank) */
        rc = sift3d_hip_fir_x_scaled_covers(&fa) ? 0 : 1;
        if (rc == 1)
            return 2;
        SH_DO(sift3d_hip_fir_x_scaled(&fa, d_scale_max, C->stream) != SIFT3D_SUCCESS);
    } else {
        SH_DO(sh_fir(S, src, C->tmp_a, o, Lg->nloc, 0, f, ufx, 0, a, b, C->stream));
    }
    fused = ufy == 1.0f && ufz == 1.0f && sift3d_hip_fir_yz_u1_covers(C->tmp_a, dst, nx, ny, f->width, nzo);
    zin = C->tmp_a;
    if (!fused) {
        SH_DO(sh_fir(S, C->tmp_a, C->tmp_b, o, Lg->nloc, 1, f, ufy, 0, a, b, C->stream));
        zin = C->tmp_b;
    }
    /* interior planes need no halo: [ia, ib) */
    ia = exch && Lg->z0 > 0 ? a + reach : a;
    ib = exch && Lg->z1 < nzo ? b - reach : b;
    if (ib < ia)
        ia = ib = a;                                     /* thin slab: everything after the exchange */
    if (exch) {
        SH_COMM(sift3d_hip_event_record(C->ev_x, C->stream) ||
                sift3d_hip_stream_wait_event(S->comm_stream, C->ev_x) ||
                sh_halo(S, zin, Lg, plane, reach, S->comm_stream) ||
                sift3d_hip_event_record(C->ev_halo, S->comm_stream));
    }
#define SH_ZPASS(lo_, hi_)                                                                           \
    do {                                                                                             \
        if ((hi_) > (lo_)) {                                                                         \
            if (fused)                                                                               \
                SH_DO(sift3d_hip_fir_yz_u1(zin, dst, nx, ny, Lg->nloc, f->taps, f->width, nzo,       \
                                           Lg->off, (lo_), (hi_), C->stream) != SIFT3D_SUCCESS);     \
            else                                                                                     \
                SH_DO(sh_fir(S, zin, dst, o, Lg->nloc, 2, f, ufz, Lg->off, (lo_), (hi_), C->stream)); \
        }                                                                                            \
    } while (0)
    SH_ZPASS(ia, ib);
    if (exch) {
        SH_COMM(sift3d_hip_stream_wait_event(C->stream, C->ev_halo));
        SH_ZPASS(a, ia);
        SH_ZPASS(ib, b);
    }
#undef SH_ZPASS
    return SIFT3D_SUCCESS;
}

/* a device buffer of at least `bytes` (grown with a quarter of slack) */
static int sh_ensure_dev(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap)
        return SIFT3D_SUCCESS;
    sift3d_hip_free(*p);
    *cap = 0;
    if (!(*p = sift3d_hip_malloc(bytes + bytes / 4)))
        return SIFT3D_FAILURE;
    *cap = bytes + bytes / 4;
    return SIFT3D_SUCCESS;
}

static int sh_ensure_pinned(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap)
        return SIFT3D_SUCCESS;
    sift3d_hip_host_free(*p);
    *cap = 0;
    if (!(*p = sift3d_hip_host_alloc(bytes + bytes / 4)))
        return SIFT3D_FAILURE;
    *cap = bytes + bytes / 4;
    return SIFT3D_SUCCESS;
}

/* all-gather of one device block per rank, on the main stream (one rank: a copy) */
static int sh_allgather_dev(sift3d_amd_sharded *S, const void *d_mine, void *d_all, size_t bytes)
{
    if (S->world == 1)
        return sift3d_hip_memcpy_d2d(d_all, d_mine, bytes, S->stream);
    return S->T.allgather(S->T.ctx, d_mine, d_all, bytes, S->stream);
}

static int sh_ensure_cand(sift3d_amd_sharded *S, uint32_t cap)
{
    if (cap <= S->cand_cap)
        return SIFT3D_SUCCESS;
    sift3d_hip_free(S->d_cand); sift3d_hip_free(S->d_R); sift3d_hip_free(S->d_keep);
    S->cand_cap = 0;
    S->d_cand = (sift3d_hip_cand *)sift3d_hip_malloc(sizeof(sift3d_hip_cand) * (size_t)cap);
    S->d_R = (float *)sift3d_hip_malloc(sizeof(float) * 9 * (size_t)cap);
    S->d_keep = (int32_t *)sift3d_hip_malloc(sizeof(int32_t) * (size_t)cap);
    if (!S->d_cand || !S->d_R || !S->d_keep)
        return SIFT3D_FAILURE;
    S->cand_cap = cap;
    return SIFT3D_SUCCESS;
}

/* ---- detect -------------------------------------------------------------------------------- */
typedef struct {          /* a record of the global list as sift3d_hip_slab_build delivers it */
    int32_t o, s, x, y, z;
    float R[9];
    float strength;
    int32_t pad;
} sh_out;

int sift3d_amd_sharded_detect(sift3d_amd_sharded *S, sift3d_keypoint_store *kp)
{
    const double t_start = now_s();
    uint32_t count = 0;
    int o, s, r, k, attempt, nkey, all_free = 1;
    if (!S || !kp)
        return SIFT3D_FAILURE;
    nkey = S->num_octaves * S->K;
    if (nkey > 256)
        return SIFT3D_FAILURE;
    for (o = 0; o < S->num_octaves; o++)
        all_free = all_free && S->dog_free[o];
    S->failed = S->inject == 1 ? -1 : 0;

    /* set_im_SIFT3D: scale by the GLOBAL max|v| (sift.c:645-649) */
    sift3d_hip_event_record(S->ev4, S->stream);
    {
        const size_t n_in = sh_plane(S, 0) * (size_t)(S->in_z1 - S->in_z0);
        SH_DO(sift3d_hip_memset(S->d_scalars, 0, sizeof(float) * (8 + SH_MAX_NDL * SH_MAX_OCT), S->stream) ||
              sift3d_hip_absmax(S->d_raw, n_in, S->d_scalars, S->stream));
        if (S->world > 1)
            SH_COMM(S->T.allreduce_max(S->T.ctx, S->d_scalars, 1, S->stream));
        /* (im_scale itself: folded into the first blur below, or run there) */
    }
    /* build_gpyr, sift.c:662-711 -- on three dependency chains, as the single-GPU path builds it
     * (detect_on_device): octave o + 1 starts from level K of octave o (sift.c:696-704) and the levels after
     * K feed nothing but their octave's DoG, so once level K of octave 0 exists the smaller octaves --
     * short launches that cannot fill the device -- are built on a second stream BESIDE the last levels of
     * octave 0, their own last levels on a third, and all are joined before the DoG stage.  The z-halo
     * exchanges of all chains travel on the one communication stream in the order the host issues them
     * (the same on every rank); the collective of the sharded -> replicated transition stays on the main
     * stream, where every other collective of a step is issued. */
    sift3d_hip_event_record(S->ev0, S->stream);
    {
        const int forked = S->num_octaves > 1 && S->K + 1 < S->ngl;
        const struct sh_chain *C0 = &S->chain[0];
        for (o = 0; o < S->num_octaves; o++) {
            const struct sh_chain *Ca = (o > 0 && forked) ? &S->chain[1] : C0;     /* levels 1 .. K */
            const struct sh_chain *Cb = (o > 0 && forked) ? &S->chain[2] : C0;     /* levels K + 1 .. */
            if (o == 0) {
                /* the first blur reads the raw slab and scales as it stages (im_scale folded into the x
                 * pass: the scaled image is not stored); where that x pass does not apply, scale first */
                const sh_level *L0 = &S->G[0][0];
                const int a0 = sh_sharded(S, 0) ? L0->z0 - L0->off : 0;
                const size_t n_in = sh_plane(S, 0) * (size_t)(S->in_z1 - S->in_z0);
                /* (d_raw holds the owned planes only: plane index a0 of the level buffer is its plane 0) */
                int rc = sh_blur(S, C0, 0, S->d_raw - (size_t)a0 * sh_plane(S, 0), L0, L0->t, &S->filt[0],
                                 S->d_scalars);
                if (rc == 2) {
                    SH_DO(sift3d_hip_scale(S->d_raw, S->d_im + (size_t)a0 * sh_plane(S, 0), n_in, S->d_scalars,
                                           S->stream));
                    SH_COMM(sh_blur(S, C0, 0, S->d_im, L0, L0->t, &S->filt[0], NULL));
                } else if (rc != SIFT3D_SUCCESS) {
                    return SIFT3D_FAILURE;
                }
            }
            for (s = 1; s <= S->K && s < S->ngl; s++)
                SH_COMM(sh_blur(S, Ca, o, S->G[o][s - 1].t, &S->G[o][s], S->G[o][s].t, &S->filt[s], NULL));
            if (forked) {
                /* level K exists: the next octave (chain 1) and this octave's last levels (chain 2, or the
                 * main stream for octave 0) go their own ways */
                if (o == 0)
                    SH_COMM(sift3d_hip_event_record(S->ev_fork, S->stream) ||
                            sift3d_hip_stream_wait_event(S->oct_stream, S->ev_fork));
                else
                    SH_COMM(sift3d_hip_event_record(S->ev_oct[o], Ca->stream) ||
                            sift3d_hip_stream_wait_event(Cb->stream, S->ev_oct[o]));
            }
            if (o != S->num_octaves - 1) {
                /* level max(s_end - 2, first_level) = Gaussian index K, sift.c:696-704; on the stream that
                 * builds the next octave */
                void *ds = forked ? S->oct_stream : S->stream;
                const sh_level *src = &S->G[o][S->K];
                sh_level *dst = &S->G[o + 1][0];
                const int mx = S->dims[o + 1][0], my = S->dims[o + 1][1], mzg = S->dims[o + 1][2];
                const size_t pl_s = sh_plane(S, o), pl_d = sh_plane(S, o + 1);
                if (sh_sharded(S, o) && !sh_sharded(S, o + 1)) {
                    /* transition to the replicated octaves: down-sample the owned planes into a
                     * staging slab, all-gather the slabs (padded to the largest) */
                    int zb[SH_MAX_WORLD + 1], mxn = 1;
                    for (r = 0; r < S->world; r++) {
                        zb[r] = S->b0[r] >> (o + 1);
                        if (zb[r] > mzg) zb[r] = mzg;
                    }
                    zb[S->world] = mzg;
                    for (r = 0; r < S->world; r++)
                        if (zb[r + 1] - zb[r] > mxn) mxn = zb[r + 1] - zb[r];
                    {
                        const int z0 = zb[S->rank], z1 = zb[S->rank + 1];
                        float *mine = S->d_stage, *all;
                        const size_t slab = (size_t)mxn * pl_d;
                        if (slab * (size_t)(S->world + 1) > S->stage_elems)
                            return SIFT3D_FAILURE;       /* (geometry: the same on every rank) */
                        all = mine + slab;
                        SH_DO(sift3d_hip_memset(mine, 0, slab * sizeof(float), ds));
                        if (z1 > z0)
                            SH_DO(sift3d_hip_downsample2(src->t + (size_t)(2 * z0 - src->off) * pl_s,
                                                         S->dims[o][0], S->dims[o][1], mine, mx, my, z1 - z0, ds));
                        /* the collective on the main stream (behind what that stream holds), the next
                         * octave behind the collective */
                        if (ds != S->stream)
                            SH_COMM(sift3d_hip_event_record(S->ev_tr, ds) ||
                                    sift3d_hip_stream_wait_event(S->stream, S->ev_tr));
                        SH_COMM(S->T.allgather(S->T.ctx, mine, all, slab * sizeof(float), S->stream));
                        for (r = 0; r < S->world; r++)
                            if (zb[r + 1] > zb[r])
                                SH_DO(sift3d_hip_memcpy_d2d(dst->t + (size_t)zb[r] * pl_d, all + (size_t)r * slab,
                                                            (size_t)(zb[r + 1] - zb[r]) * pl_d * sizeof(float),
                                                            S->stream));
                        if (ds != S->stream)
                            SH_COMM(sift3d_hip_event_record(S->ev_tr, S->stream) ||
                                    sift3d_hip_stream_wait_event(ds, S->ev_tr));
                    }
                } else {
                    const int z0 = sh_sharded(S, o + 1) ? dst->z0 : 0, z1 = sh_sharded(S, o + 1) ? dst->z1 : mzg;
                    if (z1 > z0)
                        SH_DO(sift3d_hip_downsample2(src->t + (size_t)(2 * z0 - src->off) * pl_s, S->dims[o][0],
                                                     S->dims[o][1], dst->t + (size_t)(z0 - dst->off) * pl_d, mx,
                                                     my, z1 - z0, ds));
                }
            }
            for (s = S->K + 1; s < S->ngl; s++)
                SH_COMM(sh_blur(S, Cb, o, S->G[o][s - 1].t, &S->G[o][s], S->G[o][s].t, &S->filt[s], NULL));
        }
        if (forked)
            SH_COMM(sift3d_hip_event_record(S->ev_join, S->oct_stream) ||
                    sift3d_hip_stream_wait_event(S->stream, S->ev_join) ||
                    sift3d_hip_event_record(S->ev_join2, S->side_stream) ||
                    sift3d_hip_stream_wait_event(S->stream, S->ev_join2));
    }
    sift3d_hip_event_record(S->ev1, S->stream);

    /* The nearest halo plane of every level of the sharded octaves: the extrema test looks one plane
     * past the slab, and stored DoG levels are formed on it */
    for (o = 0; o < S->o_shard; o++)
        for (s = 0; s < S->ngl; s++)
            SH_COMM(sh_halo_range(S, S->G[o][s].t, &S->G[o][s], sh_plane(S, o), 0, 1, S->stream));

    /* dogmax (sift.c:821-826) of every octave on the owned planes (every plane is owned by some
     * rank), one all-reduce for all of them.  Octaves of the DoG-free sweep store no DoG level; the
     * others form theirs (build_dog, sift.c:713-732) on the owned planes and one plane around them
     * -- the neighbours' planes are theirs to count, but a maximum does not mind a plane twice. */
    for (o = 0; o < S->num_octaves; o++) {
        const sh_level *L = &S->G[o][0];
        const size_t plane = sh_plane(S, o);
        const int lo = sh_sharded(S, o) ? L->z0 - L->off : 0;
        const int hi = sh_sharded(S, o) ? L->z1 - L->off : L->nloc;
        const float *g[SH_MAX_NGL];
        if (S->dog_free[o]) {
            for (s = 0; s < S->ngl; s++)
                g[s] = S->G[o][s].t + (size_t)lo * plane;
            if (hi > lo)
                SH_DO(sift3d_hip_dogmax_stack(g, S->ngl, (size_t)(hi - lo) * plane,
                                              S->d_scalars + 8 + S->ndl * o, S->stream) != SIFT3D_SUCCESS);
        } else {
            const int lo2 = lo > 0 ? lo - 1 : 0, hi2 = hi < L->nloc ? hi + 1 : L->nloc;
            const size_t n2 = (size_t)(hi2 - lo2) * plane;
            float *dd[SH_MAX_NDL];
            int rc = 0;
            if (hi <= lo)
                continue;
            for (s = 0; s < S->ngl; s++)
                g[s] = S->G[o][s].t + (size_t)lo2 * plane;
            for (s = 0; s < S->ndl; s++)
                dd[s] = S->D[o][s] + (size_t)lo2 * plane;
            if (!S->failed)
                rc = sift3d_hip_dog_stack(g, dd, S->ngl, n2, S->d_scalars + 8 + S->ndl * o, S->stream);
            if (rc == 1) {
                for (s = 0; s < S->ndl; s++)
                    SH_DO(sift3d_hip_subtract_absmax(g[s], g[s + 1], dd[s], n2,
                                                     S->d_scalars + 8 + S->ndl * o + s, S->stream));
            } else if (rc != SIFT3D_SUCCESS) {
                SH_DO(1);
            }
        }
    }
    if (S->world > 1)
        SH_COMM(S->T.allreduce_max(S->T.ctx, S->d_scalars + 8, S->ndl * S->num_octaves, S->stream));

    /* The windows of the orientation and descriptor kernels reach win_reach[s] planes into the
     * neighbours' slabs: the largest exchange of a step.  It travels on the communication stream
     * WHILE the extrema are found, and is awaited before the orientation kernel.  (No collective is
     * issued on the compute stream in between: one communicator, one order.) */
    if (S->world > 1 && S->o_shard > 0) {
        SH_COMM(sift3d_hip_event_record(S->ev_x, S->stream) ||
                sift3d_hip_stream_wait_event(S->comm_stream, S->ev_x));
        for (o = 0; o < S->o_shard; o++)
            for (s = 0; s < S->ngl; s++)
                SH_COMM(sh_halo_range(S, S->G[o][s].t, &S->G[o][s], sh_plane(S, o), 1, S->win_reach[s],
                                      S->comm_stream));
        SH_COMM(sift3d_hip_event_record(S->ev_halo, S->comm_stream));
    }

    /* detect_extrema (sift.c:735-871) on the owned planes, assign_orientations (sift.c:1109-1167)
     * for the local candidates */
    SH_DO(sh_ensure_cand(S, S->cand_cap ? S->cand_cap : (1u << 18)));
    for (attempt = 0; attempt < 2 && !S->failed; attempt++) {
        /* with every octave on the DoG-free sweep: the sweeps of octaves >= 1 (short launches) beside
         * octave 0's on a second stream, then scan + emission in octave order
         * (sift3d_hip_extrema_gauss6_phase); otherwise octave by octave */
        const int side = all_free && S->num_octaves > 1 && S->d_work2 != NULL;
        int phase;
        SH_DO(sift3d_hip_memset(S->d_scalars + 1, 0, sizeof(uint32_t), S->stream));
        if (side)
            SH_COMM(sift3d_hip_event_record(S->ev_fork, S->stream) ||
                    sift3d_hip_stream_wait_event(S->oct_stream, S->ev_fork));
        for (phase = side ? 1 : 0; phase <= (side ? 2 : 0); phase++) {
            for (o = 0; o < S->num_octaves; o++) {
                const sh_level *L = &S->G[o][0];
                const int nzo = S->dims[o][2];
                const int zl = (L->z0 > 1 ? L->z0 : 1) - L->off;
                int zh = (L->z1 < nzo - 1 ? L->z1 : nzo - 1) - L->off;
                const float *g[SH_MAX_NGL];
                void *wk = side && o > 0 ? (void *)((char *)S->d_work2 + S->work2_off[o]) : S->d_work;
                const size_t wb = side && o > 0 ? S->work2_sz[o] : S->work_bytes;
                if (zh < zl)
                    zh = zl;
                for (s = 0; s < S->ngl; s++)
                    g[s] = S->G[o][s].t;
                if (L->nloc < 3 || zh <= zl)
                    continue;                            /* no interior plane on this rank */
                if (S->dog_free[o]) {
                    SH_DO(sift3d_hip_extrema_gauss6_phase(g, S->d_scalars + 8 + S->ndl * o, S->dims[o][0],
                                                          S->dims[o][1], L->nloc, zl, zh, o * S->ngl + 1,
                                                          S->peak_thresh, S->d_cand, S->cand_cap,
                                                          (uint32_t *)(S->d_scalars + 1), wk, wb,
                                                          phase == 1 && o > 0 ? S->oct_stream : S->stream,
                                                          phase) != SIFT3D_SUCCESS);
                } else {
                    sift3d_hip_extrema_level lv[SH_MAX_K];
                    for (s = 0; s < S->K; s++) {
                        lv[s].prev = S->D[o][s];
                        lv[s].cur = S->D[o][s + 1];
                        lv[s].next = S->D[o][s + 2];
                        lv[s].d_absmax = S->d_scalars + 8 + S->ndl * o + s + 1;
                        lv[s].z_lo = zl;
                        lv[s].z_hi = zh;
                        lv[s].tag = o * S->ngl + s + 1;          /* Gaussian level (o, s) of the table */
                    }
                    SH_DO(sift3d_hip_extrema_mode(lv, S->K, S->dims[o][0], S->dims[o][1], L->nloc, S->peak_thresh,
                                                  S->cuboid, S->d_cand, S->cand_cap,
                                                  (uint32_t *)(S->d_scalars + 1), wk, wb, S->stream));
                }
            }
            if (phase == 1)
                SH_COMM(sift3d_hip_event_record(S->ev_join, S->oct_stream) ||
                        sift3d_hip_stream_wait_event(S->stream, S->ev_join));
        }
        SH_DO(sift3d_hip_memcpy_d2h(&count, S->d_scalars + 1, sizeof(count), S->stream));
        SH_COMM(sift3d_hip_stream_sync(S->stream));
        if (S->failed || count <= S->cand_cap)
            break;
        SH_DO(sh_ensure_cand(S, count + count / 4 + 1024));
    }
    if (S->inject == 2)
        S->failed = -2;
    if (S->failed)
        count = 0;
    sift3d_hip_event_record(S->ev2, S->stream);
    if (S->world > 1 && S->o_shard > 0)
        SH_COMM(sift3d_hip_stream_wait_event(S->stream, S->ev_halo));    /* the window halos have arrived */
    if (count) {
        const size_t need = sift3d_hip_orient_tab_bytes(S->num_octaves * S->ngl, S->cand_cap);
        if (need > S->otab_bytes) {
            sift3d_hip_free(S->d_otab);
            S->otab_bytes = 0;
            S->d_otab = sift3d_hip_malloc(need);
            /* zeroed once: the tables carry a validity mark (they are kept between calls) */
            if (!S->d_otab || sift3d_hip_memset(S->d_otab, 0, need, S->stream))
                SH_DO(1);
            else
                S->otab_bytes = need;
        }
        SH_DO(sift3d_hip_orient_tab(S->d_levels, S->num_octaves * S->ngl, S->d_cand, count, S->corner_thresh,
                                    S->d_R, S->d_keep, S->d_otab, S->cand_cap, S->stream));
        if (S->failed)
            count = 0;
    }
    sift3d_hip_event_record(S->ev3, S->stream);
    S->t_gather0 = now_s();

    /* The exchange: per-(octave, level) counts, the candidates' |DoG| values and the ORIENTED
     * keypoints -- device to device: the counts (with this rank's status word behind them) in a first
     * all-gather, whose result the host needs to size the second; values + records in ONE block per rank in
     * the second; the global list is then built by a kernel on every rank (sift3d_hip_slab_build: (o, s) major,
     * ranks in slab order inside a level -- their z ranges are disjoint and ascending --, a rank's own
     * (z, y, x) order inside its segment) and read back once. */
    {
        const size_t cnt_bytes = ((sizeof(int32_t) * 2 * (size_t)nkey + 15) & ~(size_t)15) + 16;
        const int W = S->world, nseg = nkey * W;
        int32_t *h_cnt, *h_stat;
        uint32_t *h_tab, *segk, *segc, *ck, *cc;
        size_t maxc = 0, maxk = 0, tot_c = 0, tot_k = 0, roff, toff, blk, tab_words;
        int any_failed = 0;
        /* [W blocks gathered][mine] */
        if (sh_ensure_dev(&S->d_cnt, &S->cnt_cap, cnt_bytes * (size_t)(W + 1)) ||
            sh_ensure_pinned(&S->h_cnt, &S->hcnt_cap, cnt_bytes * (size_t)W + 64))
            return SIFT3D_FAILURE;               /* (the exchange buffers themselves: nothing to send in) */
        h_cnt = (int32_t *)S->h_cnt;
        h_stat = (int32_t *)((char *)S->h_cnt + cnt_bytes * (size_t)W);      /* two words sent from here */
        {
            char *mine = (char *)S->d_cnt + cnt_bytes * (size_t)W;
            SH_DO(sift3d_hip_slab_count(S->d_cand, S->d_keep, count, S->ngl, S->K, nkey, (int32_t *)mine,
                                        S->stream));
            h_stat[0] = S->failed;
            SH_COMM(sift3d_hip_memcpy_h2d(mine + cnt_bytes - 16, h_stat, sizeof(int32_t), S->stream) ||
                    sh_allgather_dev(S, mine, S->d_cnt, cnt_bytes) ||
                    sift3d_hip_memcpy_d2h(h_cnt, S->d_cnt, cnt_bytes * (size_t)W, S->stream) ||
                    sift3d_hip_stream_sync(S->stream));
        }
        for (r = 0; r < W; r++)
            any_failed |= *(const int32_t *)((const char *)h_cnt + cnt_bytes * (size_t)(r + 1) - 16) != 0;
        if (any_failed) {
            if (S->failed)
                ERR("sift3d_amd_sharded_detect: rank %d failed locally (mark %d) \n", S->rank, S->failed);
            return SIFT3D_FAILURE;               /* on every rank, here */
        }
        /* prefix sums over keys per rank (ck kept, cc candidates), segment starts of the global orders */
        tab_words = 2 * ((size_t)nseg + 1) + 2 * (size_t)W * (nkey + 1);
        if (sh_ensure_pinned(&S->h_tab, &S->htab_cap, sizeof(uint32_t) * tab_words) ||
            sh_ensure_dev(&S->d_tab, &S->tab_cap, sizeof(uint32_t) * tab_words))
            return SIFT3D_FAILURE;
        h_tab = (uint32_t *)S->h_tab;
        segk = h_tab; segc = segk + nseg + 1; ck = segc + nseg + 1; cc = ck + (size_t)W * (nkey + 1);
        for (r = 0; r < W; r++) {
            const int32_t *a = (const int32_t *)((const char *)h_cnt + cnt_bytes * (size_t)r);
            ck[(size_t)r * (nkey + 1)] = cc[(size_t)r * (nkey + 1)] = 0;
            for (k = 0; k < nkey; k++) {
                cc[(size_t)r * (nkey + 1) + k + 1] = cc[(size_t)r * (nkey + 1) + k] + (uint32_t)a[2 * k];
                ck[(size_t)r * (nkey + 1) + k + 1] = ck[(size_t)r * (nkey + 1) + k] + (uint32_t)a[2 * k + 1];
            }
            if (cc[(size_t)r * (nkey + 1) + nkey] > maxc) maxc = cc[(size_t)r * (nkey + 1) + nkey];
            if (ck[(size_t)r * (nkey + 1) + nkey] > maxk) maxk = ck[(size_t)r * (nkey + 1) + nkey];
        }
        for (k = 0; k < nkey; k++)
            for (r = 0; r < W; r++) {
                segc[(size_t)k * W + r] = (uint32_t)tot_c;
                segk[(size_t)k * W + r] = (uint32_t)tot_k;
                tot_c += cc[(size_t)r * (nkey + 1) + k + 1] - cc[(size_t)r * (nkey + 1) + k];
                tot_k += ck[(size_t)r * (nkey + 1) + k + 1] - ck[(size_t)r * (nkey + 1) + k];
            }
        segc[nseg] = (uint32_t)tot_c;
        segk[nseg] = (uint32_t)tot_k;
        S->ncand = (int)tot_c;
        if (S->inject == 3)
            S->failed = -3;
        /* this rank's block: [values of all its candidates][its kept records][status word] */
        roff = (sizeof(float) * (maxc ? maxc : 1) + 15) & ~(size_t)15;
        toff = roff + ((56 * (maxk ? maxk : 1) + 15) & ~(size_t)15);
        blk = toff + 16;
        if (sh_ensure_dev(&S->d_xchg, &S->xchg_bytes, blk * (size_t)(W + 1)) ||
            sh_ensure_dev(&S->d_out, &S->out_cap, 64 + 64 * (tot_k ? tot_k : 1)) ||
            sh_ensure_pinned(&S->h_xchg, &S->hxchg_cap, 64 + 64 * (tot_k ? tot_k : 1)) ||
            sh_ensure_dev(&S->d_pack, &S->pack_cap, sift3d_hip_slab_pack_scratch_bytes(S->cand_cap)))
            return SIFT3D_FAILURE;
        /* keypoint store: dimensions of the first octave (sift.c:756-759) */
        kp->nx = S->nx; kp->ny = S->ny; kp->nz = S->nz;
        SH_DO(kp_store_resize(kp, tot_k));
        {
            char *mine = (char *)S->d_xchg + blk * (size_t)W;
            SH_DO(sift3d_hip_memcpy_h2d(S->d_tab, h_tab, sizeof(uint32_t) * tab_words, S->stream));
            SH_DO(sift3d_hip_slab_pack(S->d_levels, S->d_cand, S->d_keep, S->d_R, count, S->ngl, (float *)mine,
                                       mine + roff, S->d_pack, S->stream));
            h_stat[1] = S->failed;
            SH_COMM(sift3d_hip_memcpy_h2d(mine + toff, h_stat + 1, sizeof(int32_t), S->stream) ||
                    sh_allgather_dev(S, mine, S->d_xchg, blk));
            /* (copy_Keypoint omits `strength`, sift.c:372-384: slot j keeps GLOBAL candidate j's value) */
            SH_DO(sift3d_hip_slab_build(S->d_xchg, blk, roff, toff, W, nkey, (const uint32_t *)S->d_tab,
                                        (uint32_t)tot_k, (uint32_t)tot_c, S->d_out, S->stream));
            SH_COMM(sift3d_hip_memcpy_d2h(S->h_xchg, S->d_out, 64 + 64 * tot_k, S->stream) ||
                    sift3d_hip_stream_sync(S->stream));
        }
        if (S->failed || *(const int32_t *)S->h_xchg != 0) {
            if (S->failed)
                ERR("sift3d_amd_sharded_detect: rank %d failed locally (mark %d) \n", S->rank, S->failed);
            return SIFT3D_FAILURE;               /* on every rank: the status words are in everybody's list */
        }
        {
            /* the records into the store: a few threads, each on a contiguous part; a segment holds one
             * (octave, level), i.e. one scale (imutil.c:1578-1579) */
            const sh_out *rec = (const sh_out *)((const char *)S->h_xchg + 64);
            const int nt = host_threads(tot_k);
            double sdk[256];
            for (k = 0; k < nkey; k++)
                sdk[k] = sh_scale(S, k / S->K, k % S->K);
#pragma omp parallel num_threads(nt)
            {
                const int nth = omp_get_num_threads(), t = omp_get_thread_num();   /* (the team's real size) */
                const size_t lo = tot_k * (size_t)t / nth, hi = tot_k * (size_t)(t + 1) / nth;
                size_t g;
                for (g = lo; g < hi; g++) {
                    keypoint_t *kk = kp->buf + g;
                    kk->o = rec[g].o; kk->s = rec[g].s;
                    kk->xd = rec[g].x; kk->yd = rec[g].y; kk->zd = rec[g].z;
                    kk->sd = sdk[rec[g].o * S->K + rec[g].s];
                    memcpy(kk->R, rec[g].R, sizeof(kk->R));
                    kk->strength = rec[g].strength;
                }
            }
        }
    }
    S->t[0] = 1e-3 * sift3d_hip_event_elapsed_ms(S->ev0, S->ev1);
    S->t[3] = 1e-3 * sift3d_hip_event_elapsed_ms(S->ev1, S->ev2);
    S->t[4] = 1e-3 * sift3d_hip_event_elapsed_ms(S->ev2, S->ev3);
    S->t[5] = now_s() - S->t_gather0;
    S->t[6] = 1e-3 * sift3d_hip_event_elapsed_ms(S->ev4, S->ev0);
    S->t[1] = now_s() - t_start;
    return SIFT3D_SUCCESS;
}

/* ---- describe ------------------------------------------------------------------------------ */
/* Descriptors of the keypoints this rank owns (by z).  own_idx (capacity: the number of
 * keypoints) receives their positions in `kp`; desc their descriptors in that order.  The union
 * over ranks covers every keypoint exactly once. */
int sift3d_amd_sharded_describe(sift3d_amd_sharded *S, const sift3d_keypoint_store *kp,
                                sift3d_descriptor_store *desc, int *own_idx, int *n_own)
{
    const double t_start = now_s();
    int n = 0, num;
    size_t n_exact = 0;
    if (!S || !kp || !desc || !own_idx || !n_own)
        return SIFT3D_FAILURE;
    num = (int)kp->num;
    {
        /* the keypoints this rank owns, in list order: counted and placed by a few threads, each on a
         * contiguous part of the list */
        const int nt = host_threads((size_t)num);
        size_t pre[HOST_THREADS_MAX + 1], total = 0;
        int bad = 0;
        pre[0] = 0;
#pragma omp parallel num_threads(nt) reduction(| : bad)
        {
            const int nth = omp_get_num_threads(), t = omp_get_thread_num();  /* (the team's real size) */
            const size_t lo = (size_t)num * t / nth, hi = (size_t)num * (t + 1) / nth;
            size_t q, jj = 0;
#define SH_OWN(k_) (S->world == 1 || ((k_)->zd >= (double)S->bounds[(k_)->o][S->rank] &&          \
                                      (k_)->zd < (double)S->bounds[(k_)->o][S->rank + 1]))
            for (q = lo; q < hi; q++) {
                const keypoint_t *k = kp->buf + q;
                if (k->o < 0 || k->o >= S->num_octaves || k->s < 0 || k->s >= S->K) {
                    bad = 1;
                    continue;
                }
                jj += SH_OWN(k) ? 1 : 0;
            }
            pre[t + 1] = jj;
#pragma omp barrier
#pragma omp single
            {
                int u;
                for (u = 0; u < nth; u++)
                    pre[u + 1] += pre[u];
                total = pre[nth];
            }
            /* (implicit barrier) */
            jj = pre[t];
            for (q = lo; q < hi; q++) {
                const keypoint_t *k = kp->buf + q;
                if (k->o < 0 || k->o >= S->num_octaves || k->s < 0 || k->s >= S->K)
                    continue;
                if (SH_OWN(k))
                    own_idx[jj++] = (int)q;
            }
#undef SH_OWN
        }
        if (bad)
            return SIFT3D_FAILURE;
        n = (int)total;
    }
    *n_own = n;
    desc->nx = S->nx; desc->ny = S->ny; desc->nz = S->nz;
    if (!desc->pinned || (size_t)n > desc->cap) {
        const size_t cap = (size_t)n + (size_t)n / 8 + 64;
        desc_store_release(desc);
        desc->hist = (float *)sift3d_hip_host_alloc(sizeof(float) * DESC_NUMEL * cap);
        desc->xyzsd = (double *)malloc(sizeof(double) * 4 * cap);
        if (!desc->hist || !desc->xyzsd) {
            desc->pinned = desc->hist != NULL;
            desc_store_release(desc);
            return SIFT3D_FAILURE;
        }
        desc->pinned = 1;
        desc->cap = cap;
    }
    desc->num = (size_t)n;
    desc->d_num = 0;                 /* (rows are rewritten: a device copy kept by an earlier call is stale) */
    if (!n)
        return SIFT3D_SUCCESS;
    if ((uint32_t)n > S->kp_cap) {
        const uint32_t cap = (uint32_t)n + (uint32_t)n / 4 + 256;
        sift3d_hip_host_free(S->h_kp);
        S->kp_cap = 0;
        S->h_kp = (sift3d_hip_kp *)sift3d_hip_host_alloc(sizeof(sift3d_hip_kp) * (size_t)cap);
        if (!S->h_kp)
            return SIFT3D_FAILURE;
        S->kp_cap = cap;
    }
    /* launch order: widest windows first, each histogram to its own row (a stable counting sort by level,
     * as in sift3d_extract_descriptors) */
    {
        enum { LVM = 16 };
        const int nt = host_threads((size_t)n);
        /* (Gaussian index of the first level that takes the reference-order kernel; keypoint level s has
         * Gaussian index s + 1) */
        const int s_exact = exact_desc_first_level(S->exact_desc, S->ngl, S->K, S->sigma0, S->units) - 1;
        size_t cnt[HOST_THREADS_MAX][LVM], start[HOST_THREADS_MAX][LVM];
        if (S->K > LVM)
            return SIFT3D_FAILURE;
        memset(cnt, 0, sizeof(cnt));
#pragma omp parallel num_threads(nt)
        {
            const int nth = omp_get_num_threads(), t = omp_get_thread_num();
            const size_t lo = (size_t)n * t / nth, hi = (size_t)n * (t + 1) / nth;
            size_t q;
            for (q = lo; q < hi; q++)
                cnt[t][kp->buf[own_idx[q]].s]++;
#pragma omp barrier
#pragma omp single
            {
                size_t p = 0;
                int lv, u;
                for (lv = S->K - 1; lv >= 0; lv--) {
                    if (lv == s_exact - 1)
                        n_exact = p;           /* (widest windows first: the exact ones lead the list) */
                    for (u = 0; u < nth; u++) {
                        start[u][lv] = p;
                        p += cnt[u][lv];
                    }
                }
                if (s_exact <= 0)
                    n_exact = p;
            }
            /* (implicit barrier) */
            for (q = lo; q < hi; q++) {
                const keypoint_t *k = kp->buf + own_idx[q];
                const double f = ldexp(1.0, k->o);                 /* sift.c:1459, 1530-1533 */
                sift3d_hip_kp *r = S->h_kp + start[t][k->s]++;
                memcpy(r->R, k->R, sizeof(r->R));
                r->cx = (float)k->xd; r->cy = (float)k->yd; r->cz = (float)k->zd;   /* sift.c:1474-1476 */
                r->level = k->o * S->ngl + k->s + 1;
                r->row1 = (uint32_t)q + 1u;
                r->sd = k->sd;
                desc->xyzsd[4 * q] = k->xd * f;
                desc->xyzsd[4 * q + 1] = k->yd * f;
                desc->xyzsd[4 * q + 2] = k->zd * f;
                desc->xyzsd[4 * q + 3] = k->sd;
            }
        }
    }
    {
        /* (the kernel reads a keypoint's record once, as its wave starts: from the page-locked list in place) */
        float *dev_view = (float *)sift3d_hip_host_device_ptr(desc->hist);
        const sift3d_hip_kp *kp_view = (const sift3d_hip_kp *)sift3d_hip_host_device_ptr(S->h_kp);
        const size_t need = sift3d_hip_describe_part_bytes((uint32_t)n - (uint32_t)n_exact);
        if (need > S->dpart_bytes) {
            sift3d_hip_free(S->d_dpart);
            S->dpart_bytes = 0;
            S->d_dpart = sift3d_hip_malloc(need + need / 8);
            if (!S->d_dpart)
                return SIFT3D_FAILURE;
            S->dpart_bytes = need + need / 8;
        }
        if (!dev_view || !kp_view ||
            sift3d_hip_describe_parts(S->d_levels, S->num_octaves * S->ngl, kp_view, (uint32_t)n,
                                      (uint32_t)n_exact, dev_view, NULL, S->d_wlut, need ? S->d_dpart : NULL,
                                      S->stream) ||
            sift3d_hip_stream_sync(S->stream))
            return SIFT3D_FAILURE;
    }
    S->t[2] = now_s() - t_start;
    return SIFT3D_SUCCESS;
}

/* ---- descriptor gather ("computed by the owning rank and gathered (N x 771 f32)") ---- */
int sift3d_amd_sharded_gather_descriptors(sift3d_amd_sharded *S, const sift3d_keypoint_store *kp,
                                          const sift3d_descriptor_store *own, const int *own_idx, int n_own,
                                          sift3d_descriptor_store *all, int root)
{
    const int W = S ? S->world : 0;
    size_t num, maxn = 0, blk, q;
    size_t cnt[SH_MAX_WORLD];
    uint32_t *map = NULL;
    int32_t *h_stat;
    int r, want, failed = 0, any = 0;
    if (!S || !kp || !own || !all || (n_own > 0 && !own_idx) || root >= W)
        return SIFT3D_FAILURE;
    num = kp->num;
    want = root < 0 || root == S->rank;
    if (S->inject == 4)
        failed = -4;
    /* who owns which keypoint follows from the list and the slab bounds: every rank derives every rank's
     * rows (the rule of sift3d_amd_sharded_describe) */
    memset(cnt, 0, sizeof(cnt));
    map = (uint32_t *)malloc(sizeof(uint32_t) * (num ? num : 1));
    if (!map)
        failed = failed ? failed : __LINE__;
    /* (the counts size the exchange: every rank must arrive at the same ones, whatever failed locally) */
    for (q = 0; q < num; q++) {
        const keypoint_t *k = kp->buf + q;
        int rr = 0;
        if (k->o < 0 || k->o >= S->num_octaves) {
            failed = failed ? failed : __LINE__;       /* (the list is the same on every rank) */
            if (map)
                map[q] = 0;
            continue;
        }
        if (W > 1)
            for (rr = 0; rr < W - 1; rr++)
                if (k->zd < (double)S->bounds[k->o][rr + 1])
                    break;
        if (map)
            map[q] = ((uint32_t)rr << 24) | (uint32_t)(cnt[rr] & 0xffffffu);
        cnt[rr]++;
    }
    for (r = 0; r < W; r++)
        if (cnt[r] > maxn)
            maxn = cnt[r];
    if (maxn >= (1u << 24))
        return SIFT3D_FAILURE;                         /* (the same on every rank) */
    if (!failed && (size_t)(n_own < 0 ? 0 : n_own) != cnt[S->rank])
        failed = __LINE__;                             /* desc_own is not this rank's describe result */
    if (!failed && (own->num != cnt[S->rank] || (cnt[S->rank] && !own->hist)))
        failed = __LINE__;
    blk = DESC_NUMEL * sizeof(float) * (maxn ? maxn : 1) + 16;
    if (sh_ensure_dev(&S->d_gath, &S->gath_cap, blk * (size_t)(W + 1)) ||
        sh_ensure_pinned(&S->h_cnt, &S->hcnt_cap, 16 * (size_t)W + 64)) {
        free(map);
        return SIFT3D_FAILURE;                         /* (the exchange buffers themselves) */
    }
    h_stat = (int32_t *)((char *)S->h_cnt + 16 * (size_t)W);
    {
        char *mine = (char *)S->d_gath + blk * (size_t)W;
        if (!failed && cnt[S->rank] &&
            sift3d_hip_memcpy_h2d(mine, own->hist, DESC_NUMEL * sizeof(float) * cnt[S->rank], S->stream))
            failed = __LINE__;
        h_stat[0] = failed;
        if (sift3d_hip_memcpy_h2d(mine + blk - 16, h_stat, sizeof(int32_t), S->stream) ||
            sh_allgather_dev(S, mine, S->d_gath, blk) ||
            sift3d_hip_memcpy2d_d2h(S->h_cnt, 16, (char *)S->d_gath + blk - 16, blk, 16, (size_t)W, S->stream) ||
            sift3d_hip_stream_sync(S->stream)) {
            free(map);
            return SIFT3D_FAILURE;
        }
    }
    for (r = 0; r < W; r++)
        any |= *(const int32_t *)((const char *)S->h_cnt + 16 * (size_t)r) != 0;
    if (any) {
        if (failed)
            ERR("sift3d_amd_sharded_gather_descriptors: rank %d failed locally (mark %d) \n", S->rank, failed);
        free(map);
        return SIFT3D_FAILURE;                         /* on every rank, here */
    }
    if (want && !map)
        return SIFT3D_FAILURE;                         /* (unreachable: a missing map was reported above) */
    if (want) {
        /* rows into the global order on the device, one copy into the store's page-locked array */
        const size_t cap = num + num / 8 + 64;
        void *d_map = NULL, *d_rows = NULL;
        int rc = SIFT3D_SUCCESS;
        if (!all->pinned || num > all->cap) {
            desc_store_release(all);
            all->hist = (float *)sift3d_hip_host_alloc(sizeof(float) * DESC_NUMEL * cap);
            all->xyzsd = (double *)malloc(sizeof(double) * 4 * cap);
            if (!all->hist || !all->xyzsd) {
                all->pinned = all->hist != NULL;
                desc_store_release(all);
                free(map);
                return SIFT3D_FAILURE;
            }
            all->pinned = 1;
            all->cap = cap;
        }
        all->nx = S->nx; all->ny = S->ny; all->nz = S->nz;
        all->num = num;
        all->d_num = 0;
        if (num) {
            d_map = sift3d_hip_malloc(sizeof(uint32_t) * num);
            d_rows = sift3d_hip_malloc(sizeof(float) * DESC_NUMEL * num);
            if (!d_map || !d_rows ||
                sift3d_hip_memcpy_h2d(d_map, map, sizeof(uint32_t) * num, S->stream) ||
                sift3d_hip_rows_scatter((float *)d_rows, S->d_gath, blk, (const uint32_t *)d_map, (uint32_t)num,
                                        S->stream) ||
                sift3d_hip_memcpy_d2h(all->hist, d_rows, sizeof(float) * DESC_NUMEL * num, S->stream) ||
                sift3d_hip_stream_sync(S->stream))
                rc = SIFT3D_FAILURE;
            sift3d_hip_free(d_map);
            sift3d_hip_free(d_rows);
        }
        for (q = 0; q < num; q++) {
            const keypoint_t *k = kp->buf + q;
            const double f = ldexp(1.0, k->o);             /* sift.c:1459, 1530-1533 */
            all->xyzsd[4 * q] = k->xd * f;
            all->xyzsd[4 * q + 1] = k->yd * f;
            all->xyzsd[4 * q + 2] = k->zd * f;
            all->xyzsd[4 * q + 3] = k->sd;
        }
        free(map);
        return rc;
    }
    free(map);
    return SIFT3D_SUCCESS;
}

/* ---- RCCL transport (librccl is loaded at run time: single-GPU users do not need it) -------- */
typedef struct { char internal[128]; } sh_nccl_id;
typedef void *sh_nccl_comm;
typedef struct {
    void *lib;
    sh_nccl_comm comm;
    int rank, world;
    int (*GetUniqueId)(sh_nccl_id *);
    int (*CommInitRank)(sh_nccl_comm *, int, sh_nccl_id, int);
    int (*CommDestroy)(sh_nccl_comm);
    int (*Send)(const void *, size_t, int, int, sh_nccl_comm, void *);
    int (*Recv)(void *, size_t, int, int, sh_nccl_comm, void *);
    int (*AllReduce)(const void *, void *, size_t, int, int, sh_nccl_comm, void *);
    int (*AllGather)(const void *, void *, size_t, int, sh_nccl_comm, void *);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
} sh_rccl;

enum { SH_NCCL_INT8 = 0, SH_NCCL_FLOAT32 = 7, SH_NCCL_MAX = 2 };   /* rccl.h: ncclDataType_t, ncclRedOp_t */

static int sh_rccl_load(sh_rccl *R)
{
    static const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    size_t i;
    for (i = 0; i < sizeof(names) / sizeof(names[0]) && !R->lib; i++)
        R->lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!R->lib) {
        ERR("sift3d_amd: librccl could not be loaded: %s \n", dlerror());
        return SIFT3D_FAILURE;
    }
#define SH_SYM(field, name)                                             \
    do {                                                                \
        *(void **)(&R->field) = dlsym(R->lib, name);                    \
        if (!R->field) {                                                \
            ERR("sift3d_amd: librccl lacks %s \n", name);              \
            return SIFT3D_FAILURE;                                      \
        }                                                               \
    } while (0)
    SH_SYM(GetUniqueId, "ncclGetUniqueId");
    SH_SYM(CommInitRank, "ncclCommInitRank");
    SH_SYM(CommDestroy, "ncclCommDestroy");
    SH_SYM(Send, "ncclSend");
    SH_SYM(Recv, "ncclRecv");
    SH_SYM(AllReduce, "ncclAllReduce");
    SH_SYM(AllGather, "ncclAllGather");
    SH_SYM(GroupStart, "ncclGroupStart");
    SH_SYM(GroupEnd, "ncclGroupEnd");
#undef SH_SYM
    return SIFT3D_SUCCESS;
}

static int sh_rccl_halo(void *ctx, const void *send_lo, void *recv_lo, const void *send_hi, void *recv_hi,
                        size_t bytes, void *stream)
{
    sh_rccl *R = (sh_rccl *)ctx;
    int rc = 0;
    rc |= R->GroupStart();
    if (recv_lo) rc |= R->Recv(recv_lo, bytes, SH_NCCL_INT8, R->rank - 1, R->comm, stream);
    if (recv_hi) rc |= R->Recv(recv_hi, bytes, SH_NCCL_INT8, R->rank + 1, R->comm, stream);
    if (send_lo) rc |= R->Send(send_lo, bytes, SH_NCCL_INT8, R->rank - 1, R->comm, stream);
    if (send_hi) rc |= R->Send(send_hi, bytes, SH_NCCL_INT8, R->rank + 1, R->comm, stream);
    rc |= R->GroupEnd();
    return rc ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

static int sh_rccl_allreduce_max(void *ctx, float *d_buf, int n, void *stream)
{
    sh_rccl *R = (sh_rccl *)ctx;
    return R->AllReduce(d_buf, d_buf, (size_t)n, SH_NCCL_FLOAT32, SH_NCCL_MAX, R->comm, stream)
               ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

static int sh_rccl_allgather(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *stream)
{
    sh_rccl *R = (sh_rccl *)ctx;
    return R->AllGather(d_send, d_recv, bytes, SH_NCCL_INT8, R->comm, stream) ? SIFT3D_FAILURE
                                                                              : SIFT3D_SUCCESS;
}

int sift3d_amd_rccl_unique_id(void *id128)
{
    sh_rccl R;
    memset(&R, 0, sizeof(R));
    if (!id128 || sh_rccl_load(&R))
        return SIFT3D_FAILURE;
    return R.GetUniqueId((sh_nccl_id *)id128) ? SIFT3D_FAILURE : SIFT3D_SUCCESS;
}

int sift3d_amd_rccl_transport(sift3d_amd_transport *out, int world, int rank, const void *id128)
{
    sh_rccl *R;
    sh_nccl_id id;
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world)
        return SIFT3D_FAILURE;
    R = (sh_rccl *)calloc(1, sizeof(*R));
    if (!R || sh_rccl_load(R)) {
        free(R);
        return SIFT3D_FAILURE;
    }
    memcpy(&id, id128, sizeof(id));
    R->rank = rank; R->world = world;
    if (R->CommInitRank(&R->comm, world, id, rank)) {
        ERR("sift3d_amd: ncclCommInitRank failed (rank %d of %d) \n", rank, world);
        free(R);
        return SIFT3D_FAILURE;
    }
    memset(out, 0, sizeof(*out));
    out->rank = rank; out->world = world; out->ctx = R;
    out->halo = sh_rccl_halo;
    out->allreduce_max = sh_rccl_allreduce_max;
    out->allgather = sh_rccl_allgather;
    return SIFT3D_SUCCESS;
}

void sift3d_amd_rccl_transport_free(sift3d_amd_transport *t)
{
    sh_rccl *R;
    if (!t || !t->ctx)
        return;
    R = (sh_rccl *)t->ctx;
    if (R->comm)
        R->CommDestroy(R->comm);
    free(R);
    t->ctx = NULL;
}

/* ranks as threads of one process: the stream-ordered rehearsal transport */
#include "sift3d_thread_transport.c"

/* sift3d_thread_transport.c -- a STREAM-ORDERED transport for N slab drivers that run as N threads of
 * one process (included by sift3d_sharded.c).
 *
 * Purpose: rehearsal and test.  A one-GPU box cannot run ncclSend/ncclRecv between ranks, and transports
 * that drain the stream around every exchange (host-staged gloo, device copies between host barriers)
 * hide exactly the errors the RCCL transport would expose: a missing event edge between the stream an
 * exchange is enqueued on and the streams that produce or consume its buffers.  This transport has the
 * completion semantics of RCCL's calls and NO host-side stream synchronisation:
 *
 *   halo        the sender records `ready` on the stream it was handed and passes (pointer, ready) to the
 *               neighbour through a host mailbox; the receiver makes ITS stream wait for `ready`, enqueues
 *               the device-to-device copy there, records `consumed` and hands that back; the sender's
 *               stream waits for `consumed`.  So: the copy runs after everything the sender had enqueued
 *               before the call, the receiver's later work runs after the copy, and the sender's later
 *               work (which may overwrite the planes) runs after they have been read -- what
 *               ncclSend / ncclRecv in one group guarantee on their streams, and nothing more.
 *   all-gather  every rank publishes (pointer, ready), copies every block on its own stream behind the
 *               owners' `ready`, records `done`; every stream then waits for all `done` (a send buffer is
 *               not reused before everybody has read it).
 *   all-reduce  the same gather into a scratch row per rank, then one max kernel over the rows.
 *
 * The host threads only rendezvous to pass event handles (mutex + condition variable, with a timeout so
 * that a rank that failed does not leave the others hanging); they never wait for the device.  The
 * reference has no counterpart (no collectives; sift.c:1117 is its only parallelism). */

#include <pthread.h>
#include <time.h>

#define SH_T_TIMEOUT_S 300

typedef struct {
    const void *ptr;
    void *ready, *consumed;
    int full, acked;
} sh_tbox;

struct sift3d_amd_thread_group {
    int world;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int aborted;
    sh_tbox box[SH_MAX_WORLD][2];                 /* [sender][0: to rank - 1, 1: to rank + 1] */
    void *ev_ready[SH_MAX_WORLD][2];              /* recorded by the sender */
    void *ev_consumed[SH_MAX_WORLD][2];           /* recorded by the receiver of box[sender][dir] */
    /* collectives */
    int arrived;
    unsigned generation;
    const void *cptr[SH_MAX_WORLD];
    void *cready[SH_MAX_WORLD], *cdone[SH_MAX_WORLD];
    float *scratch[SH_MAX_WORLD];
    size_t scratch_floats[SH_MAX_WORLD];
};

typedef struct {
    sift3d_amd_thread_group *g;
    int rank;
} sh_tctx;

/* wait on the group's condition variable until *flag, the group is aborted, or the timeout: 0 when *flag */
static int sh_t_wait(sift3d_amd_thread_group *g, const volatile int *flag)
{
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += SH_T_TIMEOUT_S;
    while (!*flag && !g->aborted)
        if (pthread_cond_timedwait(&g->cv, &g->mu, &ts)) {
            g->aborted = 1;
            pthread_cond_broadcast(&g->cv);
            break;
        }
    return *flag ? 0 : -1;
}

static int sh_t_barrier(sift3d_amd_thread_group *g)
{
    int rc = 0;
    pthread_mutex_lock(&g->mu);
    if (g->aborted) {
        rc = -1;
    } else if (++g->arrived == g->world) {
        g->arrived = 0;
        g->generation++;
        pthread_cond_broadcast(&g->cv);
    } else {
        const unsigned gen = g->generation;
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_sec += SH_T_TIMEOUT_S;
        while (gen == g->generation && !g->aborted)
            if (pthread_cond_timedwait(&g->cv, &g->mu, &ts)) {
                g->aborted = 1;
                pthread_cond_broadcast(&g->cv);
                break;
            }
        rc = gen == g->generation ? -1 : 0;
    }
    pthread_mutex_unlock(&g->mu);
    return rc;
}

static int sh_t_halo(void *ctx, const void *send_lo, void *recv_lo, const void *send_hi, void *recv_hi,
                     size_t bytes, void *stream)
{
    sh_tctx *c = (sh_tctx *)ctx;
    sift3d_amd_thread_group *g = c->g;
    const int r = c->rank;
    const void *send[2];
    void *recv[2];
    int dir, rc = 0;
    send[0] = send_lo; send[1] = send_hi;
    recv[0] = recv_lo; recv[1] = recv_hi;
    if ((send_lo && r == 0) || (recv_lo && r == 0) || (send_hi && r == g->world - 1) ||
        (recv_hi && r == g->world - 1))
        return SIFT3D_FAILURE;
    /* 1. publish my planes: final once my stream has reached this point */
    for (dir = 0; dir < 2; dir++) {
        sh_tbox *b = &g->box[r][dir];
        if (!send[dir])
            continue;
        if (sift3d_hip_event_record(g->ev_ready[r][dir], stream))
            return SIFT3D_FAILURE;
        pthread_mutex_lock(&g->mu);
        b->ptr = send[dir];
        b->ready = g->ev_ready[r][dir];
        b->acked = 0;
        b->full = 1;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
    }
    /* 2. take the neighbours' planes on MY stream, behind their `ready` */
    for (dir = 0; dir < 2; dir++) {
        /* recv_lo comes from rank - 1, which sent it upwards (its dir 1); recv_hi from rank + 1 (its dir 0) */
        const int p = dir == 0 ? r - 1 : r + 1, pd = dir == 0 ? 1 : 0;
        sh_tbox *b;
        const void *src;
        void *ready;
        if (!recv[dir])
            continue;
        b = &g->box[p][pd];
        pthread_mutex_lock(&g->mu);
        rc = sh_t_wait(g, &b->full);
        src = b->ptr;
        ready = b->ready;
        pthread_mutex_unlock(&g->mu);
        if (rc)
            return SIFT3D_FAILURE;
        if (sift3d_hip_stream_wait_event(stream, ready) ||
            sift3d_hip_memcpy_d2d(recv[dir], src, bytes, stream) ||
            sift3d_hip_event_record(g->ev_consumed[p][pd], stream))
            return SIFT3D_FAILURE;
        pthread_mutex_lock(&g->mu);
        b->consumed = g->ev_consumed[p][pd];
        b->full = 0;
        b->acked = 1;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
    }
    /* 3. my stream may touch the planes I sent only after they have been read */
    for (dir = 0; dir < 2; dir++) {
        sh_tbox *b = &g->box[r][dir];
        void *consumed;
        if (!send[dir])
            continue;
        pthread_mutex_lock(&g->mu);
        rc = sh_t_wait(g, &b->acked);
        consumed = b->consumed;
        b->acked = 0;
        pthread_mutex_unlock(&g->mu);
        if (rc || sift3d_hip_stream_wait_event(stream, consumed))
            return SIFT3D_FAILURE;
    }
    return SIFT3D_SUCCESS;
}

/* every rank's block copied to dst + p * stride on MY stream, behind the owners' `ready`; then every
 * stream waits until all ranks have read (the send buffers may be reused) */
static int sh_t_gather_blocks(sh_tctx *c, const void *d_send, char *dst, size_t bytes, void *stream)
{
    sift3d_amd_thread_group *g = c->g;
    const int r = c->rank;
    int p;
    if (sift3d_hip_event_record(g->cready[r], stream))
        return SIFT3D_FAILURE;
    g->cptr[r] = d_send;
    if (sh_t_barrier(g))
        return SIFT3D_FAILURE;
    for (p = 0; p < g->world; p++)
        if ((p != r && sift3d_hip_stream_wait_event(stream, g->cready[p])) ||
            sift3d_hip_memcpy_d2d(dst + (size_t)p * bytes, g->cptr[p], bytes, stream))
            return SIFT3D_FAILURE;
    if (sift3d_hip_event_record(g->cdone[r], stream) || sh_t_barrier(g))
        return SIFT3D_FAILURE;
    for (p = 0; p < g->world; p++)
        if (p != r && sift3d_hip_stream_wait_event(stream, g->cdone[p]))
            return SIFT3D_FAILURE;
    return sh_t_barrier(g) ? SIFT3D_FAILURE : SIFT3D_SUCCESS;      /* (the handles may be reused) */
}

static int sh_t_allgather(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *stream)
{
    return sh_t_gather_blocks((sh_tctx *)ctx, d_send, (char *)d_recv, bytes, stream);
}

static int sh_t_allreduce_max(void *ctx, float *d_buf, int n, void *stream)
{
    sh_tctx *c = (sh_tctx *)ctx;
    sift3d_amd_thread_group *g = c->g;
    const int r = c->rank;
    const size_t need = (size_t)n * (size_t)g->world;
    if (n < 1)
        return SIFT3D_SUCCESS;
    if (need > g->scratch_floats[r]) {
        /* (an allocation is not stream work: the first call of a size pays it, like RCCL's own buffers) */
        sift3d_hip_free(g->scratch[r]);
        g->scratch_floats[r] = 0;
        if (!(g->scratch[r] = (float *)sift3d_hip_malloc(need * sizeof(float))))
            return SIFT3D_FAILURE;
        g->scratch_floats[r] = need;
    }
    /* the rows are complete, and every peer has read my values, before d_buf is overwritten */
    if (sh_t_gather_blocks(c, d_buf, (char *)g->scratch[r], (size_t)n * sizeof(float), stream))
        return SIFT3D_FAILURE;
    return sift3d_hip_max_rows(d_buf, g->scratch[r], g->world, n, stream);
}

sift3d_amd_thread_group *sift3d_amd_thread_group_create(int world)
{
    sift3d_amd_thread_group *g;
    int r, d;
    if (world < 1 || world > SH_MAX_WORLD || !sift3d_amd_device_available())
        return NULL;
    g = (sift3d_amd_thread_group *)calloc(1, sizeof(*g));
    if (!g)
        return NULL;
    g->world = world;
    pthread_mutex_init(&g->mu, NULL);
    pthread_cond_init(&g->cv, NULL);
    for (r = 0; r < world; r++) {
        for (d = 0; d < 2; d++)
            if (!(g->ev_ready[r][d] = sift3d_hip_event_create()) ||
                !(g->ev_consumed[r][d] = sift3d_hip_event_create()))
                goto fail;
        if (!(g->cready[r] = sift3d_hip_event_create()) || !(g->cdone[r] = sift3d_hip_event_create()))
            goto fail;
    }
    return g;
fail:
    sift3d_amd_thread_group_free(g);
    return NULL;
}

void sift3d_amd_thread_group_free(sift3d_amd_thread_group *g)
{
    int r, d;
    if (!g)
        return;
    for (r = 0; r < g->world; r++) {
        for (d = 0; d < 2; d++) {
            sift3d_hip_event_destroy(g->ev_ready[r][d]);
            sift3d_hip_event_destroy(g->ev_consumed[r][d]);
        }
        sift3d_hip_event_destroy(g->cready[r]);
        sift3d_hip_event_destroy(g->cdone[r]);
        sift3d_hip_free(g->scratch[r]);
    }
    pthread_cond_destroy(&g->cv);
    pthread_mutex_destroy(&g->mu);
    free(g);
}

/* wake every rank that waits in an exchange: their calls return SIFT3D_FAILURE (a rank has given up) */
void sift3d_amd_thread_group_abort(sift3d_amd_thread_group *g)
{
    if (!g)
        return;
    pthread_mutex_lock(&g->mu);
    g->aborted = 1;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
}

int sift3d_amd_thread_transport(sift3d_amd_transport *out, sift3d_amd_thread_group *g, int rank)
{
    sh_tctx *c;
    if (!out || !g || rank < 0 || rank >= g->world)
        return SIFT3D_FAILURE;
    c = (sh_tctx *)calloc(1, sizeof(*c));
    if (!c)
        return SIFT3D_FAILURE;
    c->g = g;
    c->rank = rank;
    memset(out, 0, sizeof(*out));
    out->rank = rank;
    out->world = g->world;
    out->ctx = c;
    out->halo = sh_t_halo;
    out->allreduce_max = sh_t_allreduce_max;
    out->allgather = sh_t_allgather;
    return SIFT3D_SUCCESS;
}

void sift3d_amd_thread_transport_free(sift3d_amd_transport *t)
{
    if (!t || !t->ctx)
        return;
    free(t->ctx);
    t->ctx = NULL;
}

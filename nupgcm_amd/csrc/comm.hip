// Multi-GPU plumbing: one process per GPU, RCCL over xGMI.  New work - the reference is single-device.
//   * npg_comm_*  : communicator bootstrap (the launcher broadcasts the 128-byte unique id, e.g. over torch.distributed)
//   * npg_halo_*  : interface exchange for a row-block distributed CSR.  A rank's vector is [owned | ghosts]; before a
//                   SpMV the owned entries its neighbours need are packed by one gather kernel and shipped with one
//                   grouped ncclSend/ncclRecv per neighbour straight into the ghost segment (xGMI is point-to-point, and
//                   RCM-ordered row blocks talk to <= 2 neighbours).
//
// Transport.  RCCL is the product transport.  RCCL refuses two ranks on one device, so to rehearse the N-rank code path
// on a single MI355X (development boxes have one GPU) NPG_COMM_TRANSPORT=shm selects a loop-back transport that moves
// the same messages through a POSIX shared-memory segment (device -> host -> peer -> device, host barriers).  Every
// kernel, the partition, the halo plan and the collective call sequence are identical; only the wire differs.  It is a
// rehearsal tool: slow by construction and never selected implicitly.
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>

#include "common.h"

namespace npg {

#define NPG_NCCL(call)                                                                            \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            npg::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return NPG_ECOMM;                                                                     \
        }                                                                                         \
    } while (0)

// inside ncclGroupStart() ... ncclGroupEnd(): close the group before reporting, or the communicator stays in group mode
#define NPG_NCCL_IN_GROUP(call)                                                                   \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            npg::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
            ncclGroupEnd();                                                                       \
            return NPG_ECOMM;                                                                     \
        }                                                                                         \
    } while (0)

// ---- shared-memory loop-back transport ---------------------------------------------------------------------------------
struct ShmHeader {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};

struct ShmComm {
    ShmHeader *hdr = nullptr;
    char *base = nullptr;       // mapping
    size_t bytes = 0, slot = 0; // total size, bytes per rank slot
    int rank = 0, nranks = 1;
    char name[64];
    char *slot_of(int r) const { return base + 4096 + (size_t)r * slot; }
};

// NPG_COMM_SELFTEST=1: a ONE-rank RCCL communicator goes through every RCCL call site (all-reduce, broadcast, grouped
// send/recv to itself) instead of the single-rank shortcuts, so that the product transport can be exercised on a
// one-GPU box (tests/test_gpu_rccl_selftest.py).  Never set in production.
static bool selftest() {
    static const bool on = [] {
        const char *t = getenv("NPG_COMM_SELFTEST");
        return t && atoi(t) != 0;
    }();
    return on;
}
static bool single_rank_shortcut(const npg_ctx *ctx) {
    if (!ctx->comm && !ctx->shm) return true;
    return ctx->nranks == 1 && !(selftest() && ctx->comm);
}

static bool use_shm() {
    const char *t = getenv("NPG_COMM_TRANSPORT");
    return t && strcmp(t, "shm") == 0;
}

static int shm_barrier(ShmComm *c) {
    const int gen = c->hdr->generation.load(std::memory_order_acquire);
    if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        c->hdr->arrived.store(0, std::memory_order_relaxed);
        c->hdr->generation.store(gen + 1, std::memory_order_release);
        return NPG_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
            set_error("shm transport: barrier timed out (a rank left the collective sequence)");
            return NPG_ECOMM;
        }
    }
    return NPG_OK;
}

#define NPG_SHM_BARRIER(c)              \
    do {                                \
        int rc_ = shm_barrier(c);       \
        if (rc_ != NPG_OK) return rc_;  \
    } while (0)

// in-place sum over ranks of n doubles at device pointer buf; every rank adds the slots in rank order (bitwise identical
// results on all ranks)
static int shm_allreduce(npg_ctx *ctx, double *buf, int n) {
    ShmComm *c = (ShmComm *)ctx->shm;
    NPG_REQUIRE((size_t)n * sizeof(double) <= c->slot, "shm transport: all-reduce of %d doubles exceeds the slot", n);
    NPG_HIP(hipMemcpyAsync(c->slot_of(c->rank), buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_SHM_BARRIER(c);
    std::vector<double> acc(n, 0.0);
    for (int r = 0; r < c->nranks; ++r) {
        const double *src = (const double *)c->slot_of(r);
        for (int i = 0; i < n; ++i) acc[i] += src[i];
    }
    NPG_SHM_BARRIER(c);
    NPG_HIP(hipMemcpy(buf, acc.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    return NPG_OK;
}

__global__ void k_pack(const double *x, const int32_t *idx, int64_t n, double *buf) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        buf[i] = x[idx[i]];
}

}  // namespace npg

using namespace npg;

static_assert(sizeof(ncclUniqueId) <= NPG_UNIQUE_ID_BYTES, "unique id does not fit the ABI buffer");

NPG_API int npg_comm_unique_id(void *id128) {
    NPG_REQUIRE(id128, "npg_comm_unique_id: NULL buffer");
    if (use_shm()) {
        memset(id128, 0, NPG_UNIQUE_ID_BYTES);
        const unsigned long long tag[2] = {(unsigned long long)getpid(),
                                           (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count()};
        memcpy(id128, tag, sizeof tag);
        return NPG_OK;
    }
    ncclUniqueId id;
    NPG_NCCL(ncclGetUniqueId(&id));
    memset(id128, 0, NPG_UNIQUE_ID_BYTES);
    memcpy(id128, &id, sizeof id);
    return NPG_OK;
}

NPG_API int npg_comm_init(npg_ctx *ctx, const void *id128, int rank, int nranks) {
    NPG_REQUIRE(ctx && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "npg_comm_init: bad argument");
    NPG_REQUIRE(ctx->comm == nullptr, "npg_comm_init: communicator already initialised");
    NPG_HIP(hipSetDevice(ctx->device));
    if (use_shm()) {
        NPG_REQUIRE(ctx->shm == nullptr, "npg_comm_init: communicator already initialised");
        ShmComm *c = new ShmComm();
        unsigned long long tag[2];
        memcpy(tag, id128, sizeof tag);
        snprintf(c->name, sizeof c->name, "/npg_%llx_%llx", tag[0], tag[1]);
        const char *mb = getenv("NPG_SHM_SLOT_MB");
        c->slot = (size_t)(mb ? atoi(mb) : 64) << 20;
        c->bytes = 4096 + (size_t)nranks * c->slot;
        c->rank = rank;
        c->nranks = nranks;
        const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);      // zero-filled on creation: header starts at 0
        NPG_REQUIRE(fd >= 0, "shm transport: shm_open(%s) failed", c->name);
        if (ftruncate(fd, (off_t)c->bytes) != 0) {
            close(fd);
            set_error("shm transport: ftruncate failed");
            return NPG_ECOMM;
        }
        void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        NPG_REQUIRE(m != MAP_FAILED, "shm transport: mmap failed");
        c->base = (char *)m;
        c->hdr = (ShmHeader *)m;
        ctx->shm = c;
        ctx->rank = rank;
        ctx->nranks = nranks;
        // once every rank has attached the name can go: the segment lives until the last mapping is dropped
        c->hdr->attached.fetch_add(1);
        int rc = shm_barrier(c);
        if (rc != NPG_OK) return rc;
        if (rank == 0) shm_unlink(c->name);
        return NPG_OK;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm;
    NPG_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    ctx->comm = (void *)comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return NPG_OK;
}

NPG_API int npg_comm_allreduce_sum(npg_ctx *ctx, double *host_inout, int n) {
    NPG_REQUIRE(ctx && host_inout && n > 0 && (size_t)n <= ctx->scratch_doubles, "npg_comm_allreduce_sum: bad argument");
    if (single_rank_shortcut(ctx)) return NPG_OK;
    NPG_HIP(hipMemcpyAsync(ctx->d_scratch, host_inout, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->shm) {
        int rc = shm_allreduce(ctx, ctx->d_scratch, n);
        if (rc != NPG_OK) return rc;
        NPG_HIP(hipMemcpy(host_inout, ctx->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost));
        return NPG_OK;
    }
    NPG_NCCL(ncclAllReduce(ctx->d_scratch, ctx->d_scratch, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    NPG_HIP(hipMemcpyAsync(host_inout, ctx->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    return NPG_OK;
}

NPG_API int npg_comm_allgather_segments(npg_ctx *ctx, const npg_vec *local, int nseg, const int32_t *seg_rank,
                                        const int64_t *seg_local_off, const int64_t *seg_global_off,
                                        const int64_t *seg_len, npg_vec *full) {
    NPG_REQUIRE(ctx && local && full && nseg >= 0 && (nseg == 0 || (seg_rank && seg_local_off && seg_global_off && seg_len)),
                "npg_comm_allgather_segments: bad argument");
    for (int s = 0; s < nseg; ++s) {
        NPG_REQUIRE(seg_rank[s] >= 0 && seg_rank[s] < ctx->nranks && seg_len[s] >= 0 && seg_global_off[s] >= 0 &&
                        seg_global_off[s] + seg_len[s] <= full->n,
                    "npg_comm_allgather_segments: segment %d out of range", s);
        if (seg_rank[s] == ctx->rank)
            NPG_REQUIRE(seg_local_off[s] >= 0 && seg_local_off[s] + seg_len[s] <= local->n,
                        "npg_comm_allgather_segments: local segment %d out of range", s);
    }
    if (single_rank_shortcut(ctx)) {
        for (int s = 0; s < nseg; ++s)
            NPG_HIP(hipMemcpyAsync(full->d + seg_global_off[s], local->d + seg_local_off[s],
                                   (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return NPG_OK;
    }
    if (ctx->shm) {
        // every rank lays its own segments end to end in its slot; readers recompute the same offsets
        ShmComm *c = (ShmComm *)ctx->shm;
        std::vector<size_t> fill(ctx->nranks, 0), off(nseg, 0);
        for (int s = 0; s < nseg; ++s) {
            off[s] = fill[seg_rank[s]];
            fill[seg_rank[s]] += (size_t)seg_len[s] * sizeof(double);
        }
        for (int r = 0; r < ctx->nranks; ++r)
            NPG_REQUIRE(fill[r] <= c->slot, "shm transport: all-gather payload exceeds the slot (NPG_SHM_SLOT_MB)");
        for (int s = 0; s < nseg; ++s)
            if (seg_rank[s] == ctx->rank && seg_len[s] > 0) {
                NPG_HIP(hipMemcpyAsync(c->slot_of(ctx->rank) + off[s], local->d + seg_local_off[s],
                                       (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                NPG_HIP(hipMemcpyAsync(full->d + seg_global_off[s], local->d + seg_local_off[s],
                                       (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            }
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        NPG_SHM_BARRIER(c);
        for (int s = 0; s < nseg; ++s)
            if (seg_rank[s] != ctx->rank && seg_len[s] > 0)
                NPG_HIP(hipMemcpy(full->d + seg_global_off[s], c->slot_of(seg_rank[s]) + off[s],
                                  (size_t)seg_len[s] * sizeof(double), hipMemcpyHostToDevice));
        NPG_SHM_BARRIER(c);
        return NPG_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    NPG_NCCL(ncclGroupStart());
    for (int s = 0; s < nseg; ++s) {
        if (seg_len[s] == 0) continue;
        const double *src = seg_rank[s] == ctx->rank ? local->d + seg_local_off[s] : full->d + seg_global_off[s];
        NPG_NCCL_IN_GROUP(ncclBroadcast(src, full->d + seg_global_off[s], (size_t)seg_len[s], ncclDouble, seg_rank[s], comm,
                                        ctx->stream));
    }
    NPG_NCCL(ncclGroupEnd());
    return NPG_OK;
}

NPG_API int npg_halo_create(npg_ctx *ctx, int64_t n_owned, int64_t n_ghost, int npeers, const int32_t *peer_rank,
                            const int64_t *send_ptr, const int32_t *send_idx, const int64_t *recv_ptr, npg_halo **out) {
    NPG_REQUIRE(ctx && out && n_owned >= 0 && n_ghost >= 0 && npeers >= 0, "npg_halo_create: bad argument");
    NPG_REQUIRE(npeers == 0 || (peer_rank && send_ptr && recv_ptr), "npg_halo_create: NULL plan arrays");
    npg_halo *h = new npg_halo();
    h->ctx = ctx;
    h->n_owned = n_owned;
    h->n_ghost = n_ghost;
    h->npeers = npeers;
    if (npeers > 0) {
        h->peer.assign(peer_rank, peer_rank + npeers);
        h->send_ptr.assign(send_ptr, send_ptr + npeers + 1);
        h->recv_ptr.assign(recv_ptr, recv_ptr + npeers + 1);
        NPG_REQUIRE(h->recv_ptr[npeers] == n_ghost, "npg_halo_create: recv_ptr must cover the ghost segment exactly");
        for (int p = 0; p < npeers; ++p)
            NPG_REQUIRE(peer_rank[p] >= 0 && peer_rank[p] < ctx->nranks && (peer_rank[p] != ctx->rank || selftest()),
                        "npg_halo_create: bad peer rank %d", peer_rank[p]);
        const int64_t ns = h->send_ptr[npeers];
        for (int64_t k = 0; k < ns; ++k)
            NPG_REQUIRE(send_idx[k] >= 0 && send_idx[k] < n_owned, "npg_halo_create: send index out of range");
        NPG_HIP(hipSetDevice(ctx->device));
        NPG_HIP(hipMalloc((void **)&h->send_idx, std::max<size_t>(1, (size_t)ns) * sizeof(int32_t)));
        NPG_HIP(hipMalloc((void **)&h->send_buf, std::max<size_t>(1, (size_t)ns) * sizeof(double)));
        if (ns) NPG_HIP(hipMemcpy(h->send_idx, send_idx, (size_t)ns * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    *out = h;
    return NPG_OK;
}

NPG_API int npg_halo_destroy(npg_halo *h) {
    if (!h) return NPG_OK;
    hipStreamSynchronize(h->ctx->stream);
    if (h->send_idx) hipFree(h->send_idx);
    if (h->send_buf) hipFree(h->send_buf);
    if (h->cstream) {
        hipStreamSynchronize(h->cstream);
        hipEventDestroy(h->ev_ready);
        hipEventDestroy(h->ev_done);
        hipStreamDestroy(h->cstream);
    }
    delete h;
    return NPG_OK;
}

static int halo_exchange_on(npg_halo *h, double *x, hipStream_t st);

int npg::halo_exchange_raw(npg_halo *h, double *x) { return halo_exchange_on(h, x, h->ctx->stream); }

int npg::halo_exchange_async(npg_halo *h, double *x) {
    npg_ctx *ctx = h->ctx;
    if (h->npeers == 0 && !ctx->shm) return NPG_OK;
    h->pending_x = x;
    if (ctx->shm) return NPG_OK;          // host-driven loop-back transport: the exchange happens in halo_exchange_wait()
    if (!h->cstream) {
        NPG_HIP(hipSetDevice(ctx->device));
        NPG_HIP(hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
        NPG_HIP(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
        NPG_HIP(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    }
    NPG_HIP(hipEventRecord(h->ev_ready, ctx->stream));
    NPG_HIP(hipStreamWaitEvent(h->cstream, h->ev_ready, 0));
    int rc = halo_exchange_on(h, x, h->cstream);
    if (rc != NPG_OK) return rc;
    NPG_HIP(hipEventRecord(h->ev_done, h->cstream));
    return NPG_OK;
}

int npg::halo_exchange_wait(npg_halo *h) {
    npg_ctx *ctx = h->ctx;
    if (h->npeers == 0 && !ctx->shm) return NPG_OK;
    if (ctx->shm) return halo_exchange_on(h, h->pending_x, ctx->stream);
    NPG_HIP(hipStreamWaitEvent(ctx->stream, h->ev_done, 0));
    return NPG_OK;
}

static int halo_exchange_on(npg_halo *h, double *x, hipStream_t st) {
    npg_ctx *ctx = h->ctx;
    if (h->npeers == 0 && !ctx->shm) return NPG_OK;
    NPG_REQUIRE(ctx->comm || ctx->shm, "halo exchange: communicator not initialised");
    const int64_t ns = h->npeers ? h->send_ptr[h->npeers] : 0;
    if (ns > 0) {
        const int grid = (int)std::min<int64_t>(1024, (ns + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_pack, dim3(grid), dim3(kBlock), 0, st, x, h->send_idx, ns, h->send_buf);
    }
    if (ctx->shm) {
        // slot = [npeers, {peer, offset, count} x npeers | packed send buffer]; a receiver looks its own entry up
        ShmComm *c = (ShmComm *)ctx->shm;
        const size_t dir = (size_t)(1 + 3 * h->npeers) * sizeof(int64_t);
        NPG_REQUIRE(dir + (size_t)ns * sizeof(double) <= c->slot, "shm transport: halo payload exceeds the slot");
        int64_t *d = (int64_t *)c->slot_of(c->rank);
        d[0] = h->npeers;
        for (int p = 0; p < h->npeers; ++p) {
            d[1 + 3 * p] = h->peer[p];
            d[2 + 3 * p] = h->send_ptr[p];
            d[3 + 3 * p] = h->send_ptr[p + 1] - h->send_ptr[p];
        }
        if (ns > 0)
            NPG_HIP(hipMemcpyAsync(c->slot_of(c->rank) + dir, h->send_buf, (size_t)ns * sizeof(double),
                                   hipMemcpyDeviceToHost, ctx->stream));
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        NPG_SHM_BARRIER(c);
        bool bad = false;                                   // a mismatch is reported AFTER the closing barrier: every rank
        for (int p = 0; p < h->npeers; ++p) {               // leaves the collective, none is left spinning
            const int64_t r0 = h->recv_ptr[p], r1 = h->recv_ptr[p + 1];
            if (r1 == r0) continue;
            const int64_t *sd = (const int64_t *)c->slot_of(h->peer[p]);
            const int64_t snp = sd[0];
            int64_t so = -1, sc = -1;
            for (int64_t q = 0; q < snp; ++q)
                if (sd[1 + 3 * q] == c->rank) {
                    so = sd[2 + 3 * q];
                    sc = sd[3 + 3 * q];
                }
            if (sc != r1 - r0) {
                set_error("shm transport: rank %d sends %lld values to rank %d, which expects %lld", h->peer[p], (long long)sc,
                          c->rank, (long long)(r1 - r0));
                bad = true;
                continue;
            }
            const char *payload = c->slot_of(h->peer[p]) + (size_t)(1 + 3 * snp) * sizeof(int64_t);
            NPG_HIP(hipMemcpy(x + h->n_owned + r0, payload + (size_t)so * sizeof(double), (size_t)sc * sizeof(double),
                              hipMemcpyHostToDevice));
        }
        NPG_SHM_BARRIER(c);
        return bad ? NPG_ECOMM : NPG_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    NPG_NCCL(ncclGroupStart());
    for (int p = 0; p < h->npeers; ++p) {
        const int64_t s0 = h->send_ptr[p], s1 = h->send_ptr[p + 1], r0 = h->recv_ptr[p], r1 = h->recv_ptr[p + 1];
        if (s1 > s0)
            NPG_NCCL_IN_GROUP(ncclSend(h->send_buf + s0, (size_t)(s1 - s0), ncclDouble, h->peer[p], comm, st));
        if (r1 > r0)
            NPG_NCCL_IN_GROUP(ncclRecv(x + h->n_owned + r0, (size_t)(r1 - r0), ncclDouble, h->peer[p], comm, st));
    }
    NPG_NCCL(ncclGroupEnd());
    return NPG_OK;
}

int npg::allreduce_sum_device(npg_ctx *ctx, double *buf, int n) {
    if (single_rank_shortcut(ctx)) return NPG_OK;
    if (ctx->shm) return shm_allreduce(ctx, buf, n);
    NPG_NCCL(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return NPG_OK;
}

NPG_API int npg_halo_exchange(npg_halo *h, npg_vec *x) {
    NPG_REQUIRE(h && x && x->n == h->n_owned + h->n_ghost, "npg_halo_exchange: vector must hold owned + ghost entries");
    return halo_exchange_raw(h, x->d);
}

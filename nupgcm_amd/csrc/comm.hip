// Multi-GPU plumbing: one process per GPU, RCCL over xGMI.  New work - the reference is single-device.
//   * npg_comm_*  : communicator bootstrap (the launcher broadcasts the 128-byte unique id, e.g. over torch.distributed)
//   * npg_halo_*  : interface exchange for a row-block distributed CSR.  A rank's vector is [owned | ghosts]; before a
//                   SpMV the owned entries its neighbours need are packed by one gather kernel and shipped with one
//                   grouped ncclSend/ncclRecv per neighbour straight into the ghost segment (xGMI is point-to-point, and
//                   RCM-ordered row blocks talk to <= 2 neighbours).
#include <rccl/rccl.h>

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

__global__ void k_pack(const double *x, const int32_t *idx, int64_t n, double *buf) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        buf[i] = x[idx[i]];
}

}  // namespace npg

using namespace npg;

static_assert(sizeof(ncclUniqueId) <= NPG_UNIQUE_ID_BYTES, "unique id does not fit the ABI buffer");

NPG_API int npg_comm_unique_id(void *id128) {
    NPG_REQUIRE(id128, "npg_comm_unique_id: NULL buffer");
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
    if (ctx->nranks == 1 || !ctx->comm) return NPG_OK;
    NPG_HIP(hipMemcpyAsync(ctx->d_scratch, host_inout, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
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
    if (ctx->nranks == 1 || !ctx->comm) {
        for (int s = 0; s < nseg; ++s)
            NPG_HIP(hipMemcpyAsync(full->d + seg_global_off[s], local->d + seg_local_off[s],
                                   (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return NPG_OK;
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    NPG_NCCL(ncclGroupStart());
    for (int s = 0; s < nseg; ++s) {
        if (seg_len[s] == 0) continue;
        const double *src = seg_rank[s] == ctx->rank ? local->d + seg_local_off[s] : full->d + seg_global_off[s];
        NPG_NCCL(ncclBroadcast(src, full->d + seg_global_off[s], (size_t)seg_len[s], ncclDouble, seg_rank[s], comm,
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
            NPG_REQUIRE(peer_rank[p] >= 0 && peer_rank[p] < ctx->nranks && peer_rank[p] != ctx->rank,
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
    delete h;
    return NPG_OK;
}

int npg::halo_exchange_raw(npg_halo *h, double *x) {
    if (h->npeers == 0) return NPG_OK;
    npg_ctx *ctx = h->ctx;
    NPG_REQUIRE(ctx->comm, "halo exchange: communicator not initialised");
    const int64_t ns = h->send_ptr[h->npeers];
    if (ns > 0) {
        const int grid = (int)std::min<int64_t>(1024, (ns + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_pack, dim3(grid), dim3(kBlock), 0, ctx->stream, x, h->send_idx, ns, h->send_buf);
    }
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    NPG_NCCL(ncclGroupStart());
    for (int p = 0; p < h->npeers; ++p) {
        const int64_t s0 = h->send_ptr[p], s1 = h->send_ptr[p + 1], r0 = h->recv_ptr[p], r1 = h->recv_ptr[p + 1];
        if (s1 > s0) NPG_NCCL(ncclSend(h->send_buf + s0, (size_t)(s1 - s0), ncclDouble, h->peer[p], comm, ctx->stream));
        if (r1 > r0) NPG_NCCL(ncclRecv(x + h->n_owned + r0, (size_t)(r1 - r0), ncclDouble, h->peer[p], comm, ctx->stream));
    }
    NPG_NCCL(ncclGroupEnd());
    return NPG_OK;
}

int npg::allreduce_sum_device(npg_ctx *ctx, double *buf, int n) {
    if (ctx->nranks == 1 || !ctx->comm) return NPG_OK;
    NPG_NCCL(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return NPG_OK;
}

NPG_API int npg_halo_exchange(npg_halo *h, npg_vec *x) {
    NPG_REQUIRE(h && x && x->n == h->n_owned + h->n_ghost, "npg_halo_exchange: vector must hold owned + ghost entries");
    return halo_exchange_raw(h, x->d);
}

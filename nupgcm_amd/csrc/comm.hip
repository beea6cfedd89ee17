// Multi-GPU plumbing: one process per GPU, RCCL over xGMI.  New work - the reference is single-device.
//   * npg_comm_*  : communicator bootstrap (the launcher broadcasts the 128-byte unique id, e.g. over torch.distributed)
//   * npg_halo_*  : interface exchange for a row-block distributed CSR.  A rank's vector is [owned | ghosts]; before a
//                   SpMV the owned entries its neighbours need are packed by one gather kernel and shipped with one
//                   grouped ncclSend/ncclRecv per neighbour straight into the ghost segment (xGMI is point-to-point, and
//                   RCM-ordered row blocks talk to <= 2 neighbours).
//
// Transports (NPG_COMM_TRANSPORT = auto | rccl | peer | shm; default auto).
//   rccl  RCCL collectives and grouped send/recv (library kernels on the stream).
//   peer  PEER-MAPPED WINDOWS for everything inside a Krylov cycle: every rank exports small uncached device buffers
//         through hipIpc, maps its peers' and then
//           * all-reduce (<= 32 doubles): ONE single-workgroup kernel folds the rank's partial rows, stores its row as 8-byte
//             {epoch, half-double} granules straight into every peer's window (xGMI stores; the data IS the flag) and polls
//             its own window until every peer's granules carry the epoch - bit-identical sums on all ranks (rank order),
//           * halo: a gather kernel stores the owned entries a neighbour needs straight into that neighbour's receive
//             window (system-scope write-through stores, one epoch flag per workgroup behind a vmcnt(0) drain); a small
//             wait kernel polls the flags, copies the window into the ghost segment and acknowledges.
//         Only kernels on HIP streams: no host call, no library kernel, so a whole distributed restart cycle replays from
//         ONE hipGraph like the single-GPU cycle.  Epochs live in device memory and are advanced by the kernels
//         themselves (a launch argument would be frozen under replay); every spin is bounded (NPG_PEER_TIMEOUT_S, default
//         120 s) and reports through a pinned status word.  The windows of N processes that share ONE device map the same
//         way, so the N-rank rehearsal on a one-GPU box runs the production kernels and protocol (RCCL refuses two ranks
//         on one device).  Bootstrap (handle exchange, host barriers) goes through a small POSIX shared-memory segment:
//         all ranks live on one node by construction.
//   auto  RCCL communicator (bulk collectives, rank evidence) + the peer windows for the in-cycle traffic when every
//         rank could map every peer and the start-up self-check passed on all ranks; RCCL alone otherwise.
//   shm   loop-back rehearsal transport: the same messages through a POSIX shared-memory segment (device -> host -> peer
//         -> device, host barriers).  Slow by construction and never selected implicitly.
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>

#include "common.h"
#include "peer_device.h"

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
    if (!ctx->comm && !ctx->shm && !ctx->peer) return true;
    return ctx->nranks == 1 && !(selftest() && (ctx->comm || ctx->peer));
}

static bool transport_is(const char *name) {
    const char *t = getenv("NPG_COMM_TRANSPORT");
    return t && strcmp(t, name) == 0;
}
static bool use_shm() { return transport_is("shm"); }
static bool use_peer_only() { return transport_is("peer"); }
static bool use_rccl_only() { return transport_is("rccl"); }

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

// ---- peer-mapped windows ---------------------------------------------------------------------------------------------------
constexpr int kMaxRanks = 16;      // ranks of one node
constexpr int kArSlots = 4;        // ring of all-reduce slots (2 would do: a rank cannot get two collectives ahead of a peer)
constexpr int kArGran = 2 * kPartStride;   // 8-byte granules per all-reduced row: {epoch, low half}, {epoch, high half}
constexpr int kMaxPeers = 15;      // neighbours of one halo plan

struct PeerArDev {
    uint64_t *const *win;      // device array [nranks]: every rank's all-reduce window (own included)
    uint32_t *epoch;           // device word: all-reduces completed so far on this communicator
    int *status;               // pinned host word
    unsigned long long ticks;
    int rank, nranks;
};

// One workgroup: fold `nrows` partial rows of kPartStride doubles (fixed order), push the row to every peer, collect the
// peers' rows, sum in rank order (identical bits on all ranks), out[0 .. kPartStride).  out may be part (nrows == 1).
__global__ void __launch_bounds__(1024) k_peer_fold_allreduce(const double *part, int nrows, double *out, PeerArDev P) {
    __shared__ double tmp[32 * kPartStride];
    __shared__ double all[kMaxRanks][kPartStride];
    const int k = threadIdx.x & (kPartStride - 1), slice = threadIdx.x >> 5;
    const uint32_t e = *P.epoch + 1;
    double s = 0.0;
    for (int b0 = slice; b0 < nrows; b0 += 32 * 8) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = b0 + 32 * i;
            v[i] = b < nrows ? part[(size_t)b * kPartStride + k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    tmp[slice * kPartStride + k] = s;
    __syncthreads();
    if (threadIdx.x < kPartStride) {
        double t = 0.0;
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) t += tmp[sl * kPartStride + threadIdx.x];
        all[P.rank][threadIdx.x] = t;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = (int)(e % kArSlots);
    // push: wave w serves peer w, w + 16, ...; lane l carries half (l & 1) of entry l / 2
    {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(all[P.rank][lane >> 1]);
        const uint32_t half = (lane & 1) ? (uint32_t)(bits >> 32) : (uint32_t)bits;
        const uint64_t g = ((uint64_t)e << 32) | half;
        for (int p = wave; p < P.nranks; p += 16)
            if (p != P.rank) st_sys(P.win[p] + ((size_t)slot * P.nranks + P.rank) * kArGran + lane, g);
    }
    // collect: the granules peer p stored in MY window
    for (int p = wave; p < P.nranks; p += 16) {
        if (p == P.rank) continue;
        const uint64_t *src = P.win[P.rank] + ((size_t)slot * P.nranks + p) * kArGran + lane;
        SpinGuard guard(P.ticks, P.status, 1);
        uint64_t g;
        for (;;) {
            g = ld_sys(src);
            if (__all((uint32_t)(g >> 32) == e)) break;
            if (__any(guard.expired())) break;
        }
        const uint32_t mine = (uint32_t)g, other = (uint32_t)__shfl_xor((int)mine, 1, 64);
        if (!(lane & 1)) all[p][lane >> 1] = __longlong_as_double((long long)(((unsigned long long)other << 32) | mine));
    }
    __syncthreads();
    if (threadIdx.x < kPartStride) {
        double t = 0.0;
        for (int p = 0; p < P.nranks; ++p) t += all[p][threadIdx.x];
        out[threadIdx.x] = t;
    }
    if (threadIdx.x == 0) *P.epoch = e;
}

// ONE kernel per exchange (phase 0).  Workgroup b < npush: gather x[send_idx[s]] for its chunk straight into the neighbour's
// window (write-through stores), drain, raise its flag.  Then workgroup b < nwait: wait for every neighbour's flags of this
// epoch, copy its share of the window behind the owned entries; the last one acknowledges and completes the epoch.  Every
// workgroup pushes BEFORE it waits, and pushes depend only on acknowledgements of epoch e - 2: no cycle of waits between
// ranks.  (grid = max(npush, nwait) <= 15 * 64 workgroups of 256 threads: all resident.)
// A solver that has work which needs no ghost value launches the two halves separately around it (phase 1 = push, phase 2 =
// wait + unpack), on ONE stream: the neighbours' stores land in this rank's window while that work runs - the overlap needs
// neither a second stream nor events.
__global__ void __launch_bounds__(256) k_halo_exchange(double *__restrict__ x, const int32_t *__restrict__ send_idx, int64_t n_owned,
                                                       int npush, int nwait, int phase, HaloPeerDev H, float *__restrict__ g32,
                                                       const int32_t *__restrict__ gslot = nullptr, float *__restrict__ xgb = nullptr) {
    const uint64_t e = *H.epoch + 1;
    if (phase != 2 && (int)blockIdx.x < npush) {
        const int4 t = H.tab[blockIdx.x];
        if (threadIdx.x == 0 && e > 2) {
            // the slot was last used by exchange e - 2: the peer must have copied it out
            SpinGuard guard(H.ticks, H.status, 2);
            while (ld_sys(H.ack + t.z) + 2 < e)
                if (guard.expired()) break;
        }
        __syncthreads();
        double *dst = H.dst[t.z] + (e & 1) * H.dst_stride[t.z] - H.seg0[t.z];
        // four independent index -> value chains per lane and trip
        for (int64_t s0 = t.x + (int)threadIdx.x; s0 < t.y; s0 += 4 * 256) {
            int32_t idx[4];
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) idx[u] = s0 + u * 256 < t.y ? send_idx[s0 + u * 256] : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x[idx[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (s0 + u * 256 < t.y) st_sys_f64(dst + s0 + u * 256, v[u]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
        __syncthreads();                                       // ... before one lane raises the workgroup's flag
        if (threadIdx.x == 0) st_sys(H.flag_dst[t.z] + t.w, e);
    }
    if (phase == 1) return;
    if ((int)blockIdx.x >= nwait) {          // push-only workgroup of a one-kernel exchange (phase 2 launches consumers only)
        halo_launch_done(H, e, gridDim.x);
        return;
    }
    halo_window_ready(H, e);
    __syncthreads();
    double *xg = x + n_owned;
    const double *src = H.rwin + (e & 1) * H.n_ghost;
    // eight loads in flight per lane (the window is uncached memory: every load is a round trip to HBM)
    const int64_t stride = nwait * 256LL;
    for (int64_t i0 = blockIdx.x * 256LL + threadIdx.x; i0 < H.n_ghost; i0 += 8 * stride) {
        double v[8];
        int32_t sl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * stride < H.n_ghost ? __builtin_nontemporal_load(src + i0 + u * stride) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) sl[u] = (gslot && i0 + u * stride < H.n_ghost) ? gslot[i0 + u * stride] : -1;      // (in flight with the window loads)
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * stride < H.n_ghost) {
                xg[i0 + u * stride] = v[u];
                if (g32) g32[i0 + u * stride] = (float)v[u];
                if (sl[u] >= 0) xgb[sl[u]] = (float)v[u];
            }
    }
    halo_consumed(H, e, (unsigned)nwait, phase == 2);
    if (phase == 0) halo_launch_done(H, e, gridDim.x);
}

// Sender workgroups of one (sender, receiver) pair; both sides derive the same count from the segment's length.  Round 5: an exchange
// is a chain of dependent round trips (index -> value -> remote store -> drain -> flag; flag -> uncached window load -> store), so it
// pays to make every chain ONE trip long: 512 entries per sender workgroup (two per lane) up to 64 workgroups per pair, and the
// consumers likewise (below).  Rank 4 of 8 of bowl3D h = 0.02 (81 k ghost entries, two neighbours): 12.4 -> see profiles/r05_dist_cycle.txt
// (NPG_HALO_CHUNK / NPG_HALO_WG / NPG_HALO_WAIT_WG: tuning; every rank must run with the same values.)
static int halo_wg_count(int64_t cnt) {
    static const int chunk = getenv("NPG_HALO_CHUNK") ? std::max(64, atoi(getenv("NPG_HALO_CHUNK"))) : 512;
    static const int wg = getenv("NPG_HALO_WG") ? std::min(kHaloWG, std::max(1, atoi(getenv("NPG_HALO_WG")))) : kHaloWG;
    return (int)std::min<int64_t>(wg, (cnt + chunk - 1) / chunk);
}

struct PeerComm {
    ShmComm *boot = nullptr;
    int rank = 0, nranks = 1;
    char *base = nullptr;                 // local window allocation: [all-reduce slots | staging]
    size_t ar_bytes = 0, stage_bytes = 0;
    std::vector<char *> peer_base;        // every rank's allocation as mapped here (own: base)
    uint64_t **d_win = nullptr;           // device array of all-reduce window pointers
    uint32_t *d_epoch = nullptr;
    int *h_status = nullptr;              // pinned, device-visible
    unsigned long long ticks = 0;
    bool ok = false;
    double **d_stage = nullptr;           // device array of the ranks' staging windows (allreduce_big_device)
    PeerArDev ar() const { return PeerArDev{d_win, d_epoch, h_status, ticks, rank, nranks}; }
    char *stage_of(int r) const { return peer_base[r] + ar_bytes; }
};

struct HaloPeer {
    char *win = nullptr;                  // local: [2 n_ghost doubles | npeers kHaloWG flags | npeers acks]
    std::vector<void *> opened;           // peers' windows mapped here
    void *d_tables = nullptr;             // one device allocation behind the tables below
    HaloPeerDev dev{};
    int nwg_push = 0, nwg_wait = 1;
};

static int boot_allgather(ShmComm *c, const void *mine, size_t bytes, void *all) {
    NPG_REQUIRE(bytes <= c->slot, "peer transport: bootstrap message of %zu bytes exceeds the slot", bytes);
    memcpy(c->slot_of(c->rank), mine, bytes);
    NPG_SHM_BARRIER(c);
    for (int r = 0; r < c->nranks; ++r) memcpy((char *)all + (size_t)r * bytes, c->slot_of(r), bytes);
    NPG_SHM_BARRIER(c);
    return NPG_OK;
}

static int alloc_window(void **p, size_t bytes) {
    // uncached device memory: remote stores and the owner's polls meet in memory, no L2 copy in between; fine-grained as
    // the fall-back spelling of the same intent
    hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("peer transport: hipExtMallocWithFlags(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return NPG_ENOMEM;
    }
    NPG_HIP(hipMemset(*p, 0, bytes));
    return NPG_OK;
}

static unsigned long long peer_ticks() {
    const char *t = getenv("NPG_PEER_TIMEOUT_S");
    const double sec = t ? atof(t) : 120.0;
    return (unsigned long long)(std::max(0.01, sec) * 1e8);       // s_memrealtime: 100 MHz
}

static int peer_status(const npg_ctx *ctx) {
    const PeerComm *pc = (const PeerComm *)ctx->peer;
    const int st = pc ? *(volatile int *)pc->h_status : 0;
    if (st == 0) return NPG_OK;
    set_error("peer transport: a device-side wait timed out (%s) - a rank left the collective sequence or its process died",
              st == 1 ? "all-reduce" : st == 2 ? "halo sender waiting for the receiver's acknowledgement" : "halo receiver");
    return NPG_ECOMM;
}

static ShmComm *boot_open(const void *id128, int rank, int nranks, size_t slot) {
    ShmComm *c = new ShmComm();
    unsigned long long tag[2] = {1469598103934665603ULL, 1099511628211ULL};       // FNV over the whole id
    for (int i = 0; i < NPG_UNIQUE_ID_BYTES; ++i) {
        tag[i & 1] ^= ((const unsigned char *)id128)[i];
        tag[i & 1] *= 1099511628211ULL;
    }
    snprintf(c->name, sizeof c->name, "/npg_%llx_%llx", tag[0], tag[1]);
    c->slot = slot;
    c->bytes = 4096 + (size_t)nranks * c->slot;
    c->rank = rank;
    c->nranks = nranks;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);      // zero-filled on creation: header starts at 0
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
        if (fd >= 0) close(fd);
        set_error("shm bootstrap: shm_open / ftruncate(%s) failed", c->name);
        delete c;
        return nullptr;
    }
    void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        set_error("shm bootstrap: mmap failed");
        delete c;
        return nullptr;
    }
    c->base = (char *)m;
    c->hdr = (ShmHeader *)m;
    // once every rank has attached the name can go: the segment lives until the last mapping is dropped
    c->hdr->attached.fetch_add(1);
    if (shm_barrier(c) != NPG_OK) return nullptr;
    if (rank == 0) shm_unlink(c->name);
    return c;
}

static void peer_free(PeerComm *pc) {
    if (!pc) return;
    for (int r = 0; r < (int)pc->peer_base.size(); ++r)
        if (r != pc->rank && pc->peer_base[r]) hipIpcCloseMemHandle(pc->peer_base[r]);
    if (pc->base) hipFree(pc->base);
    if (pc->d_win) hipFree(pc->d_win);
    if (pc->d_stage) hipFree(pc->d_stage);
    if (pc->d_epoch) hipFree(pc->d_epoch);
    if (pc->h_status) hipHostFree(pc->h_status);
    delete pc;
}

// Collective.  Leaves ctx->peer set only if EVERY rank mapped every peer and the self-check all-reduce gave the right sum
// everywhere; returns NPG_OK with ctx->peer == nullptr when the windows could not be set up (the caller decides whether
// that is an error), an error code only when the bootstrap itself broke.
static int peer_init(npg_ctx *ctx, ShmComm *boot, int rank, int nranks) {
    PeerComm *pc = new PeerComm();
    pc->boot = boot;
    pc->rank = rank;
    pc->nranks = nranks;
    pc->ticks = peer_ticks();
    pc->ar_bytes = (size_t)kArSlots * nranks * kArGran * sizeof(uint64_t);
    const char *mb = getenv("NPG_PEER_STAGE_MB");
    pc->stage_bytes = (size_t)(mb ? atoi(mb) : 32) << 20;
    struct Hello {
        hipIpcMemHandle_t handle;
        int ok, device;
        unsigned long long pid;
    } mine{}, all[kMaxRanks];
    bool good = nranks <= kMaxRanks;
    if (good && alloc_window((void **)&pc->base, pc->ar_bytes + pc->stage_bytes) != NPG_OK) good = false;
    if (good && hipIpcGetMemHandle(&mine.handle, pc->base) != hipSuccess) {
        (void)hipGetLastError();
        good = false;
    }
    mine.ok = good ? 1 : 0;
    mine.device = ctx->device;
    mine.pid = (unsigned long long)getpid();
    int rc = boot_allgather(boot, &mine, sizeof mine, all);
    if (rc) return rc;
    for (int r = 0; r < nranks; ++r) good = good && all[r].ok;
    pc->peer_base.assign(nranks, nullptr);
    if (good) {
        for (int r = 0; r < nranks && good; ++r) {
            if (r == rank) {
                pc->peer_base[r] = pc->base;
                continue;
            }
            void *q = nullptr;
            if (hipIpcOpenMemHandle(&q, all[r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                good = false;
            }
            pc->peer_base[r] = (char *)q;
        }
    }
    if (good) {
        good = hipMalloc((void **)&pc->d_win, nranks * sizeof(uint64_t *)) == hipSuccess &&
               hipMemcpy(pc->d_win, pc->peer_base.data(), nranks * sizeof(uint64_t *), hipMemcpyHostToDevice) == hipSuccess &&
               hipMalloc((void **)&pc->d_epoch, sizeof(uint32_t)) == hipSuccess &&
               hipMemset(pc->d_epoch, 0, sizeof(uint32_t)) == hipSuccess &&
               hipHostMalloc((void **)&pc->h_status, sizeof(int), hipHostMallocMapped) == hipSuccess;
        if (good) *pc->h_status = 0;
    }
    // self-check: sum over ranks of (rank + 1) k for k = 1 .. 32, twice (two epochs, two slots)
    int verdict = good ? 1 : 0, verdicts[kMaxRanks];
    if ((rc = boot_allgather(boot, &verdict, sizeof verdict, verdicts))) return rc;
    for (int r = 0; r < nranks; ++r) good = good && verdicts[r];
    if (good) {
        const unsigned long long keep = pc->ticks;
        pc->ticks = (unsigned long long)20e8;                       // 20 s for the self-check
        for (int rep = 0; rep < 2 && good; ++rep) {
            double h[kPartStride];
            for (int k = 0; k < kPartStride; ++k) h[k] = (rank + 1.0) * (k + 1.0) + rep;
            good = hipMemcpy(ctx->d_scratch, h, sizeof h, hipMemcpyHostToDevice) == hipSuccess;
            hipLaunchKernelGGL(k_peer_fold_allreduce, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_scratch, 1, ctx->d_scratch,
                               pc->ar());
            good = good && hipMemcpyAsync(h, ctx->d_scratch, sizeof h, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                   hipStreamSynchronize(ctx->stream) == hipSuccess && *pc->h_status == 0;
            for (int k = 0; k < kPartStride && good; ++k)
                good = h[k] == 0.5 * nranks * (nranks + 1.0) * (k + 1.0) + (double)rep * nranks;
        }
        pc->ticks = keep;
        verdict = good ? 1 : 0;
        if ((rc = boot_allgather(boot, &verdict, sizeof verdict, verdicts))) return rc;
        for (int r = 0; r < nranks; ++r) good = good && verdicts[r];
    }
    if (!good) {
        (void)hipGetLastError();
        peer_free(pc);
        return NPG_OK;
    }
    pc->ok = true;
    ctx->peer = pc;
    return NPG_OK;
}

}  // namespace npg

using namespace npg;

static_assert(sizeof(ncclUniqueId) <= NPG_UNIQUE_ID_BYTES, "unique id does not fit the ABI buffer");

NPG_API int npg_comm_unique_id(void *id128) {
    NPG_REQUIRE(id128, "npg_comm_unique_id: NULL buffer");
    if (use_shm() || use_peer_only()) {
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
    NPG_REQUIRE(ctx->comm == nullptr && ctx->shm == nullptr && ctx->peer == nullptr, "npg_comm_init: communicator already initialised");
    NPG_HIP(hipSetDevice(ctx->device));
    if (use_shm()) {
        const char *mb = getenv("NPG_SHM_SLOT_MB");
        ShmComm *c = boot_open(id128, rank, nranks, (size_t)(mb ? atoi(mb) : 64) << 20);
        if (!c) return NPG_ECOMM;
        ctx->shm = c;
        ctx->rank = rank;
        ctx->nranks = nranks;
        return NPG_OK;
    }
    if (use_peer_only()) {
        // peer windows without RCCL: the N-rank rehearsal on one device (RCCL refuses two ranks there), or a node where
        // RCCL is not wanted.  Failing to map the peers is an error here.
        ShmComm *boot = boot_open(id128, rank, nranks, 64 << 10);
        if (!boot) return NPG_ECOMM;
        ctx->rank = rank;
        ctx->nranks = nranks;
        int rc = peer_init(ctx, boot, rank, nranks);
        if (rc) return rc;
        NPG_REQUIRE(ctx->peer, "npg_comm_init: NPG_COMM_TRANSPORT=peer but the peer windows could not be set up on every rank "
                               "(hipIpc export / mapping or the self-check all-reduce failed)");
        return NPG_OK;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm;
    NPG_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    ctx->comm = (void *)comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    if (!use_rccl_only() && nranks > 1) {
        // auto: RCCL stays the communicator (bulk collectives, rank evidence); the traffic inside a Krylov cycle goes through
        // peer-mapped windows when they can be set up on EVERY rank, and through RCCL otherwise
        ShmComm *boot = boot_open(id128, rank, nranks, 64 << 10);
        if (boot) {
            int rc = peer_init(ctx, boot, rank, nranks);
            if (rc) return rc;
        }
        static bool said = false;
        if (rank == 0 && !said && (said = true))
            fprintf(stderr, "[npg comm] %d ranks: RCCL communicator + %s for the in-cycle halo / all-reduce\n", nranks,
                    ctx->peer ? "peer-mapped xGMI windows" : "RCCL (peer windows unavailable)");
    }
    return NPG_OK;
}

// Drop the peer windows and carry the in-cycle traffic through RCCL from here on (a caller whose end-to-end check of the
// peer transport failed on its hardware).  Needs the RCCL communicator of the auto transport, and no live halo plan or
// solver that was created while the windows existed.
NPG_API int npg_comm_disable_peer(npg_ctx *ctx) {
    NPG_REQUIRE(ctx, "npg_comm_disable_peer: NULL context");
    if (!ctx->peer) return NPG_OK;
    NPG_REQUIRE(ctx->comm, "npg_comm_disable_peer: no RCCL communicator to fall back on (NPG_COMM_TRANSPORT=peer)");
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    peer_free((PeerComm *)ctx->peer);
    ctx->peer = nullptr;
    return NPG_OK;
}

// one line of JSON describing the communicator (bench.py puts it into its record)
NPG_API int npg_comm_info(npg_ctx *ctx, char *buf, size_t cap) {
    NPG_REQUIRE(ctx && buf && cap > 0, "npg_comm_info: bad argument");
    int rccl_ranks = 0, rccl_rank = -1, rccl_dev = -1;
    if (ctx->comm) {
        NPG_NCCL(ncclCommCount((ncclComm_t)ctx->comm, &rccl_ranks));
        NPG_NCCL(ncclCommUserRank((ncclComm_t)ctx->comm, &rccl_rank));
        NPG_NCCL(ncclCommCuDevice((ncclComm_t)ctx->comm, &rccl_dev));
    }
    hipDeviceProp_t prop;
    NPG_HIP(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, cap,
             "{\"rank\": %d, \"nranks\": %d, \"rccl_ranks\": %d, \"rccl_rank\": %d, \"rccl_device\": %d, \"device\": %d, "
             "\"device_name\": \"%s\", \"pci_bus\": %d, \"in_cycle_transport\": \"%s\"}",
             ctx->rank, ctx->nranks, rccl_ranks, rccl_rank, rccl_dev, ctx->device, prop.name, prop.pciBusID,
             ctx->peer ? "peer" : ctx->shm ? "shm" : ctx->comm ? "rccl" : "none");
    return NPG_OK;
}

NPG_API int npg_comm_allreduce_sum(npg_ctx *ctx, double *host_inout, int n) {
    NPG_REQUIRE(ctx && host_inout && n > 0 && (size_t)n + kPartStride <= ctx->scratch_doubles, "npg_comm_allreduce_sum: bad argument");
    if (single_rank_shortcut(ctx)) return NPG_OK;
    NPG_HIP(hipMemcpyAsync(ctx->d_scratch, host_inout, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->shm) {
        int rc = shm_allreduce(ctx, ctx->d_scratch, n);
        if (rc != NPG_OK) return rc;
        NPG_HIP(hipMemcpy(host_inout, ctx->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost));
        return NPG_OK;
    }
    if (ctx->peer && !ctx->comm) {
        for (int k0 = 0; k0 < n; k0 += kPartStride) {          // rows of 32 doubles (the tail row sums scratch it does not return)
            int rc = allreduce_sum_device(ctx, ctx->d_scratch + k0, std::min(kPartStride, n - k0));
            if (rc) return rc;
        }
        NPG_HIP(hipMemcpyAsync(host_inout, ctx->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        return peer_status(ctx);
    }
    NPG_NCCL(ncclAllReduce(ctx->d_scratch, ctx->d_scratch, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    NPG_HIP(hipMemcpyAsync(host_inout, ctx->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    return NPG_OK;
}

// in-place sum over the ranks of a device vector of at most 32 doubles, enqueued on the context's stream (no host
// synchronisation): the collective the solvers use per iteration, exposed for host-side drivers and stress tests
NPG_API int npg_comm_allreduce_vec(npg_ctx *ctx, npg_vec *v) {
    NPG_REQUIRE(ctx && v && v->n >= 1 && v->n <= kPartStride, "npg_comm_allreduce_vec: 1 to %d entries", kPartStride);
    if (ctx->peer && !single_rank_shortcut(ctx)) {
        // the peer kernel always moves a whole row of 32 doubles: go through the context's scratch row
        NPG_HIP(hipMemsetAsync(ctx->d_scratch, 0, kPartStride * sizeof(double), ctx->stream));
        NPG_HIP(hipMemcpyAsync(ctx->d_scratch, v->d, (size_t)v->n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        int rc = allreduce_sum_device(ctx, ctx->d_scratch, kPartStride);
        if (rc) return rc;
        NPG_HIP(hipMemcpyAsync(v->d, ctx->d_scratch, (size_t)v->n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        return NPG_OK;
    }
    return allreduce_sum_device(ctx, v->d, (int)v->n);
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
    if (ctx->peer && !ctx->comm) {
        // every rank lays its own segments end to end in its staging window; readers copy them out device to device
        PeerComm *pc = (PeerComm *)ctx->peer;
        std::vector<size_t> fill(ctx->nranks, 0), off(nseg, 0);
        for (int s = 0; s < nseg; ++s) {
            off[s] = fill[seg_rank[s]];
            fill[seg_rank[s]] += (size_t)seg_len[s] * sizeof(double);
        }
        for (int r = 0; r < ctx->nranks; ++r)
            NPG_REQUIRE(fill[r] <= pc->stage_bytes, "peer transport: all-gather payload exceeds the staging window (NPG_PEER_STAGE_MB)");
        for (int s = 0; s < nseg; ++s)
            if (seg_rank[s] == ctx->rank && seg_len[s] > 0) {
                NPG_HIP(hipMemcpyAsync(pc->stage_of(ctx->rank) + off[s], local->d + seg_local_off[s],
                                       (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                NPG_HIP(hipMemcpyAsync(full->d + seg_global_off[s], local->d + seg_local_off[s],
                                       (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            }
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        NPG_SHM_BARRIER(pc->boot);
        for (int s = 0; s < nseg; ++s)
            if (seg_rank[s] != ctx->rank && seg_len[s] > 0)
                NPG_HIP(hipMemcpyAsync(full->d + seg_global_off[s], pc->stage_of(seg_rank[s]) + off[s],
                                       (size_t)seg_len[s] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        NPG_SHM_BARRIER(pc->boot);
        return peer_status(ctx);
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


// Peer-transport part of a halo plan.  Collective over the communicator: every rank publishes its window handle and
// what it expects from whom; a sender derives from the receiver's record where its segment lies in the receiver's window
// and checks that both sides agree on its length.
static int halo_peer_setup(npg_halo *h) {
    npg_ctx *ctx = h->ctx;
    PeerComm *pc = (PeerComm *)ctx->peer;
    NPG_REQUIRE(h->npeers <= kMaxPeers, "peer transport: %d neighbours in one halo plan (limit %d)", h->npeers, kMaxPeers);
    struct Hello {
        hipIpcMemHandle_t handle;
        long long n_ghost;
        int npeers, ok;
        int peer[kMaxPeers];
        long long recv_off[kMaxPeers], recv_cnt[kMaxPeers], send_cnt[kMaxPeers];
    };
    HaloPeer *w = new HaloPeer();
    h->pw = w;
    Hello mine{};
    std::vector<Hello> all(pc->nranks);
    const int np = h->npeers;
    const size_t rbytes = 2 * (size_t)std::max<int64_t>(h->n_ghost, 1) * sizeof(double);
    const size_t fbytes = (size_t)std::max(np, 1) * kHaloWG * sizeof(uint64_t), abytes = (size_t)std::max(np, 1) * sizeof(uint64_t);
    bool good = alloc_window((void **)&w->win, rbytes + fbytes + abytes) == NPG_OK;
    if (good && hipIpcGetMemHandle(&mine.handle, w->win) != hipSuccess) {
        (void)hipGetLastError();
        good = false;
    }
    mine.ok = good ? 1 : 0;
    mine.n_ghost = h->n_ghost;
    mine.npeers = np;
    for (int p = 0; p < np; ++p) {
        mine.peer[p] = h->peer[p];
        mine.recv_off[p] = h->recv_ptr[p];
        mine.recv_cnt[p] = h->recv_ptr[p + 1] - h->recv_ptr[p];
        mine.send_cnt[p] = h->send_ptr[p + 1] - h->send_ptr[p];
    }
    int rc = boot_allgather(pc->boot, &mine, sizeof mine, all.data());
    if (rc) return rc;
    for (int r = 0; r < pc->nranks; ++r)
        NPG_REQUIRE(all[r].ok, "peer transport: rank %d could not allocate / export its halo window", r);
    // tables
    std::vector<int4> tab;
    std::vector<int64_t> seg0(std::max(np, 1), 0), dst_stride(std::max(np, 1), 0);
    std::vector<double *> dst(std::max(np, 1), nullptr);
    std::vector<uint64_t *> flag_dst(std::max(np, 1), nullptr), ack_dst(std::max(np, 1), nullptr);
    std::vector<int> nflag(std::max(np, 1), 0);
    bool bad = false;
    for (int p = 0; p < np; ++p) {
        const int q = h->peer[p];
        const Hello &hq = all[q];
        int j = -1;
        for (int t = 0; t < hq.npeers; ++t)
            if (hq.peer[t] == ctx->rank) j = t;
        const int64_t scnt = mine.send_cnt[p], rcnt = mine.recv_cnt[p];
        if (j < 0 || hq.recv_cnt[j] != scnt || hq.send_cnt[j] != rcnt) {
            set_error("peer transport: halo plans of ranks %d and %d disagree (I send %lld / expect %lld; it expects %lld / sends %lld)",
                      ctx->rank, q, (long long)scnt, (long long)rcnt, j < 0 ? -1LL : hq.recv_cnt[j], j < 0 ? -1LL : hq.send_cnt[j]);
            bad = true;
            continue;
        }
        void *base = nullptr;
        if (q == ctx->rank) {
            base = w->win;                  // self-test: the rank is its own neighbour (a handle cannot be opened by its owner)
        } else if (hipIpcOpenMemHandle(&base, hq.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            set_error("peer transport: cannot map the halo window of rank %d", q);
            bad = true;
            continue;
        } else {
            w->opened.push_back(base);
        }
        const size_t q_r = 2 * (size_t)std::max<long long>(hq.n_ghost, 1) * sizeof(double);
        const size_t q_f = (size_t)std::max(hq.npeers, 1) * kHaloWG * sizeof(uint64_t);
        dst[p] = (double *)base + hq.recv_off[j];
        dst_stride[p] = hq.n_ghost;
        flag_dst[p] = (uint64_t *)((char *)base + q_r) + (size_t)j * kHaloWG;
        ack_dst[p] = (uint64_t *)((char *)base + q_r + q_f) + j;
        seg0[p] = h->send_ptr[p];
        nflag[p] = halo_wg_count(rcnt);
        // sender workgroups for this peer: equal chunks, the receiver computes the same count from the same length
        const int g = halo_wg_count(scnt);
        for (int k = 0; k < g; ++k) {
            const int64_t a = h->send_ptr[p] + scnt * k / g, b = h->send_ptr[p] + scnt * (k + 1) / g;
            tab.push_back(make_int4((int)a, (int)b, p, k));
        }
    }
    int verdict = bad ? 0 : 1;
    std::vector<int> verdicts(pc->nranks);
    if ((rc = boot_allgather(pc->boot, &verdict, sizeof verdict, verdicts.data()))) return rc;
    for (int r = 0; r < pc->nranks; ++r)
        if (!verdicts[r]) {
            if (!bad) set_error("peer transport: rank %d could not set up its part of a halo plan", r);
            return NPG_ECOMM;
        }
    NPG_REQUIRE(h->send_ptr.empty() || h->send_ptr[np] < INT32_MAX, "peer transport: send list too long");
    // one device allocation: [tab | seg0 | dst_stride | dst | flag_dst | ack_dst | nflag | arrive | epoch]
    const size_t nt = std::max<size_t>(tab.size(), 1), npp = (size_t)std::max(np, 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 15) & ~(size_t)15;
        return o;
    };
    const size_t o_tab = carve(nt * sizeof(int4)), o_seg = carve(npp * 8), o_str = carve(npp * 8), o_dst = carve(npp * 8),
                 o_fl = carve(npp * 8), o_ack = carve(npp * 8), o_nf = carve(npp * 4), o_arr = carve(16), o_ep = carve(8);
    std::vector<char> host(off, 0);
    if (!tab.empty()) memcpy(host.data() + o_tab, tab.data(), tab.size() * sizeof(int4));
    memcpy(host.data() + o_seg, seg0.data(), npp * 8);
    memcpy(host.data() + o_str, dst_stride.data(), npp * 8);
    memcpy(host.data() + o_dst, dst.data(), npp * 8);
    memcpy(host.data() + o_fl, flag_dst.data(), npp * 8);
    memcpy(host.data() + o_ack, ack_dst.data(), npp * 8);
    memcpy(host.data() + o_nf, nflag.data(), npp * 4);
    NPG_HIP(hipMalloc(&w->d_tables, off));
    NPG_HIP(hipMemcpy(w->d_tables, host.data(), off, hipMemcpyHostToDevice));
    char *dt = (char *)w->d_tables;
    HaloPeerDev &D = w->dev;
    D.tab = (const int4 *)(dt + o_tab);
    D.seg0 = (const int64_t *)(dt + o_seg);
    D.dst = (double *const *)(dt + o_dst);
    D.dst_stride = (const int64_t *)(dt + o_str);
    D.flag_dst = (uint64_t *const *)(dt + o_fl);
    D.ack = (const uint64_t *)(w->win + rbytes + fbytes);
    D.nflag = (const int *)(dt + o_nf);
    D.flags = (const uint64_t *)(w->win + rbytes);
    D.rwin = (const double *)w->win;
    D.ack_dst = (uint64_t *const *)(dt + o_ack);
    D.arrive = (unsigned long long *)(dt + o_arr);
    D.epoch = (uint64_t *)(dt + o_ep);
    D.status = pc->h_status;
    D.ticks = pc->ticks;
    D.n_ghost = h->n_ghost;
    D.npeers = np;
    w->nwg_push = (int)tab.size();
    {
        // consumers: 2 048 window entries per workgroup (eight uncached loads in flight per lane, one trip), at most 64 workgroups
        static const int wcap = getenv("NPG_HALO_WAIT_WG") ? std::max(1, atoi(getenv("NPG_HALO_WAIT_WG"))) : 64;
        w->nwg_wait = (int)std::max<int64_t>(1, std::min<int64_t>(wcap, (h->n_ghost + 2047) / 2048));
    }
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
    if (ctx->peer) {
        int rc = halo_peer_setup(h);        // collective: every rank creates its plans in the same order
        if (rc) {
            npg_halo_destroy(h);
            return rc;
        }
    }
    *out = h;
    return NPG_OK;
}

NPG_API int npg_halo_destroy(npg_halo *h) {
    if (!h) return NPG_OK;
    hipStreamSynchronize(h->ctx->stream);
    if (h->send_idx) hipFree(h->send_idx);
    if (h->send_buf) hipFree(h->send_buf);
    if (h->pw) {
        HaloPeer *w = (HaloPeer *)h->pw;
        if (h->cstream) hipStreamSynchronize(h->cstream);
        // No barrier here (destruction order across ranks is up to the host language's finalisers): this rank's own
        // exchanges have completed, and a peer's late acknowledgement store lands in memory that stays alive as long as
        // that peer keeps its mapping of the window open.
        for (void *q : w->opened) hipIpcCloseMemHandle(q);
        if (w->win) hipFree(w->win);
        if (w->d_tables) hipFree(w->d_tables);
        delete w;
        h->pw = nullptr;
    }
    if (h->cstream) {
        hipStreamSynchronize(h->cstream);
        hipEventDestroy(h->ev_ready);
        hipEventDestroy(h->ev_done);
        hipStreamDestroy(h->cstream);
    }
    delete h;
    return NPG_OK;
}

static int halo_exchange_on(npg_halo *h, double *x, hipStream_t st, float *g32, const int32_t *gslot, float *xgb);

__global__ void k_ghosts_to_f32(const double *__restrict__ src, float *__restrict__ dst, int64_t n, const int32_t *__restrict__ gslot,
                                float *__restrict__ xgb) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        dst[i] = (float)src[i];
        if (gslot && gslot[i] >= 0) xgb[gslot[i]] = (float)src[i];
    }
}

int npg::halo_exchange_raw(npg_halo *h, double *x, float *g32, const int32_t *gslot, float *xgb) {
    return halo_exchange_on(h, x, h->ctx->stream, g32, gslot, xgb);
}

int npg::halo_exchange_async(npg_halo *h, double *x, float *g32, const int32_t *gslot, float *xgb) {
    npg_ctx *ctx = h->ctx;
    if (h->npeers == 0 && !ctx->shm) return NPG_OK;
    h->pending_x = x;
    h->pending_g32 = g32;
    h->pending_gslot = gslot;
    h->pending_xgb = xgb;
    if (ctx->shm) return NPG_OK;          // host-driven loop-back transport: the exchange happens in halo_exchange_wait()
    // NPG_HALO_TWO_STREAM=1 (diagnostic; profiles/r04_overlap_rerun.txt): the peer exchange as ONE kernel on the plan's own stream,
    // ordered against the context's stream by two events - round 2's arrangement, which is what RCCL's overlap still uses
    static const int two_stream = getenv("NPG_HALO_TWO_STREAM") ? atoi(getenv("NPG_HALO_TWO_STREAM")) : 0;
    if (h->pw && !two_stream) {
        // peer windows: push now, on the context's own stream; whatever the caller enqueues next runs while the neighbours'
        // stores arrive in this rank's window; halo_exchange_wait() enqueues the wait + unpack half behind it
        HaloPeer *w = (HaloPeer *)h->pw;
        int rc = peer_status(ctx);
        if (rc) return rc;
        if (w->nwg_push > 0)
            hipLaunchKernelGGL(k_halo_exchange, dim3(w->nwg_push), dim3(256), 0, ctx->stream, x, (const int32_t *)h->send_idx,
                               h->n_owned, w->nwg_push, w->nwg_wait, 1, w->dev, (float *)nullptr, (const int32_t *)nullptr, (float *)nullptr);
        NPG_HIP(hipGetLastError());
        return NPG_OK;
    }
    if (!h->cstream) {
        NPG_HIP(hipSetDevice(ctx->device));
        NPG_HIP(hipStreamCreateWithFlags(&h->cstream, hipStreamNonBlocking));
        NPG_HIP(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
        NPG_HIP(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    }
    NPG_HIP(hipEventRecord(h->ev_ready, ctx->stream));
    NPG_HIP(hipStreamWaitEvent(h->cstream, h->ev_ready, 0));
    int rc = halo_exchange_on(h, x, h->cstream, g32, gslot, xgb);
    if (rc != NPG_OK) return rc;
    NPG_HIP(hipEventRecord(h->ev_done, h->cstream));
    return NPG_OK;
}

int npg::halo_exchange_wait(npg_halo *h) {
    npg_ctx *ctx = h->ctx;
    if (h->npeers == 0 && !ctx->shm) return NPG_OK;
    if (ctx->shm) return halo_exchange_on(h, h->pending_x, ctx->stream, h->pending_g32, h->pending_gslot, h->pending_xgb);
    static const int two_stream = getenv("NPG_HALO_TWO_STREAM") ? atoi(getenv("NPG_HALO_TWO_STREAM")) : 0;
    if (h->pw && !two_stream) {
        HaloPeer *w = (HaloPeer *)h->pw;
        hipLaunchKernelGGL(k_halo_exchange, dim3(w->nwg_wait), dim3(256), 0, ctx->stream, h->pending_x, (const int32_t *)h->send_idx,
                           h->n_owned, w->nwg_push, w->nwg_wait, 2, w->dev, h->pending_g32, h->pending_gslot, h->pending_xgb);
        NPG_HIP(hipGetLastError());
        return NPG_OK;
    }
    NPG_HIP(hipStreamWaitEvent(ctx->stream, h->ev_done, 0));
    return NPG_OK;
}

static int halo_exchange_on(npg_halo *h, double *x, hipStream_t st, float *g32, const int32_t *gslot, float *xgb) {
    npg_ctx *ctx = h->ctx;
    if (h->npeers == 0 && !ctx->shm) return NPG_OK;
    if (h->pw) {
        // peer windows: gather straight into the neighbours' windows, wait for theirs and copy it behind the owned entries.
        // One kernel launch, no host involvement: capturable, replayable.
        HaloPeer *w = (HaloPeer *)h->pw;
        int rc = peer_status(ctx);
        if (rc) return rc;
        hipLaunchKernelGGL(k_halo_exchange, dim3(std::max(w->nwg_push, w->nwg_wait)), dim3(256), 0, st, x,
                           (const int32_t *)h->send_idx, h->n_owned, w->nwg_push, w->nwg_wait, 0, w->dev, g32, gslot, xgb);
        NPG_HIP(hipGetLastError());
        return NPG_OK;
    }
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
        if (g32 && h->n_ghost > 0 && !bad)
            hipLaunchKernelGGL(k_ghosts_to_f32, dim3((unsigned)std::min<int64_t>(256, (h->n_ghost + 255) / 256)), dim3(256), 0, st,
                               (const double *)(x + h->n_owned), g32, h->n_ghost, gslot, xgb);
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
    if (g32 && h->n_ghost > 0)
        hipLaunchKernelGGL(k_ghosts_to_f32, dim3((unsigned)std::min<int64_t>(256, (h->n_ghost + 255) / 256)), dim3(256), 0, st,
                           (const double *)(x + h->n_owned), g32, h->n_ghost, gslot, xgb);
    return NPG_OK;
}

int npg::allreduce_sum_device(npg_ctx *ctx, double *buf, int n) {
    if (single_rank_shortcut(ctx)) return NPG_OK;
    if (ctx->shm) return shm_allreduce(ctx, buf, n);
    if (ctx->peer) {
        NPG_REQUIRE(n <= kPartStride, "peer all-reduce: %d values (one row of %d at most)", n, kPartStride);
        int rc = peer_status(ctx);
        if (rc) return rc;
        // the kernel always moves one row of 32 doubles: buf must be readable that far (every caller passes a row buffer or
        // the context's scratch); entries past n are summed and ignored
        hipLaunchKernelGGL(k_peer_fold_allreduce, dim3(1), dim3(1024), 0, ctx->stream, (const double *)buf, 1, buf,
                           ((PeerComm *)ctx->peer)->ar());
        return NPG_OK;
    }
    NPG_NCCL(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return NPG_OK;
}

__global__ void __launch_bounds__(1024) k_fold_rows(const double *__restrict__ part, int nrows, double *out) {
    __shared__ double tmp[32 * kPartStride];
    const int k = threadIdx.x & (kPartStride - 1), slice = threadIdx.x >> 5;      // 32 slices of 32 lanes
    double s = 0.0;
    for (int b0 = slice; b0 < nrows; b0 += 32 * 8) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = b0 + 32 * i;
            v[i] = b < nrows ? part[(size_t)b * kPartStride + k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    tmp[slice * kPartStride + k] = s;
    __syncthreads();
    if (threadIdx.x < kPartStride) {
        double t = 0.0;
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) t += tmp[sl * kPartStride + threadIdx.x];
        out[threadIdx.x] = t;
    }
}

int npg::fold_allreduce_rows(npg_ctx *ctx, const double *part, int nrows, double *out, hipStream_t st) {
    if (ctx->peer && !single_rank_shortcut(ctx)) {
        int rc = peer_status(ctx);
        if (rc) return rc;
        hipLaunchKernelGGL(k_peer_fold_allreduce, dim3(1), dim3(1024), 0, st, part, nrows, out, ((PeerComm *)ctx->peer)->ar());
        return NPG_OK;
    }
    hipLaunchKernelGGL(k_fold_rows, dim3(1), dim3(1024), 0, st, part, nrows, out);
    return allreduce_sum_device(ctx, out, kPartStride);
}

// out[i] = sum over ranks r (in rank order) of stage_r[i]: the ranks' staging windows are peer-mapped
__global__ void __launch_bounds__(256) k_sum_stages(double *out, const double *const *stage, int nranks, int64_t n) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        double s = 0.0;
        for (int r = 0; r < nranks; ++r) s += __builtin_nontemporal_load(stage[r] + i);
        out[i] = s;
    }
}

// In-place sum over the ranks of a LONG device vector (the restricted residual of a multigrid cycle whose coarse levels are
// replicated: one n_coarse-vector per cycle).  RCCL where there is a communicator; on the peer-only transport (the one-device
// rehearsal) through the staging windows with two host barriers - correct, not fast; shm: through the host.
int npg::allreduce_big_device(npg_ctx *ctx, double *buf, int64_t n) {
    if (single_rank_shortcut(ctx) || n <= 0) return NPG_OK;
    if (ctx->shm) return shm_allreduce(ctx, buf, (int)n);
    if (ctx->comm) {
        NPG_NCCL(ncclAllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
        return NPG_OK;
    }
    PeerComm *pc = (PeerComm *)ctx->peer;
    NPG_REQUIRE(pc, "allreduce_big_device: no communicator");
    NPG_REQUIRE((size_t)n * sizeof(double) <= pc->stage_bytes, "peer transport: %lld doubles exceed the staging window (NPG_PEER_STAGE_MB)",
                (long long)n);
    if (!pc->d_stage) {
        std::vector<double *> h(pc->nranks);
        for (int r = 0; r < pc->nranks; ++r) h[r] = (double *)pc->stage_of(r);
        NPG_HIP(hipMalloc((void **)&pc->d_stage, pc->nranks * sizeof(double *)));
        NPG_HIP(hipMemcpy(pc->d_stage, h.data(), pc->nranks * sizeof(double *), hipMemcpyHostToDevice));
    }
    NPG_HIP(hipMemcpyAsync(pc->stage_of(pc->rank), buf, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_SHM_BARRIER(pc->boot);
    hipLaunchKernelGGL(k_sum_stages, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, ctx->stream, buf,
                       (const double *const *)pc->d_stage, pc->nranks, n);
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_SHM_BARRIER(pc->boot);
    return peer_status(ctx);
}

bool npg::comm_is_kernel_only(const npg_ctx *ctx) { return ctx->peer != nullptr; }

int npg::comm_check(const npg_ctx *ctx) { return ctx->peer ? peer_status(ctx) : NPG_OK; }

NPG_API int npg_halo_exchange(npg_halo *h, npg_vec *x) {
    NPG_REQUIRE(h && x && x->n == h->n_owned + h->n_ghost, "npg_halo_exchange: vector must hold owned + ghost entries");
    return halo_exchange_raw(h, x->d);
}

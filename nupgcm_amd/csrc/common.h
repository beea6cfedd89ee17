// Internal definitions shared by the translation units of libnupgcm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nupgcm_hip.h"

#define NPG_API extern "C" __attribute__((visibility("default")))
// C++ internals that the separate tuning harness (libnupgcm_tune.so, tools/ only) links against; not part of the C ABI
#define NPG_SHARED __attribute__((visibility("default")))

namespace npg {

NPG_SHARED void set_error(const char *fmt, ...);

#define NPG_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            npg::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? NPG_ENOMEM : NPG_EHIP;                             \
        }                                                                                         \
    } while (0)

#define NPG_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            npg::set_error(__VA_ARGS__);    \
            return NPG_EINVAL;              \
        }                                   \
    } while (0)

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves per workgroup everywhere
constexpr int kNumCU = 256;        // MI355X
constexpr int kPartStride = 32;    // doubles per block in a partial-sum row (256 B)
constexpr int kTileRows = 256;     // most rows in one SpMV tile
constexpr int kMaxMem = 30;        // largest GMRES memory supported (partial rows hold mem + 2 values)

// one SpMV tile: rows [r0, r0 + nrows), its n CSR entries from `base` and (node-block rows) npe records from `pbase`.
// 32 bytes = one scalar load; kernels fetch the next tile's descriptor while they work on the current one.
struct TileDesc {
    int64_t base, pbase;
    int32_t r0, nrows, n, npe;
};
// a tile of the WINDOWED set (spmv_window.h): the same, plus (nw > 0) the tile's distinct column nodes wlist[woff .. woff + nw) and
// distinct other columns vlist[voff .. voff + nv), gathered ONCE into LDS and addressed by the records through 16-bit window
// indices; nw = 0: an ordinary tile.  (A type of its own: the twelve more bytes, carried by every kernel's tile loop, cost the
// ordinary kernels spilled scalar registers - k_spmv 191 -> 256 us when TileDesc itself had grown.)
struct WTileDesc {
    int64_t base, pbase;
    int32_t r0, nrows, n, npe;
    int32_t woff, voff, nw, nv;
};

}  // namespace npg

struct npg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cu = npg::kNumCU;
    // scratch for reductions (device) and their results (pinned host)
    double *d_scratch = nullptr;
    double *h_scratch = nullptr;
    size_t scratch_doubles = 0;
    // pinned staging buffer for perm uploads/downloads
    double *h_stage = nullptr;
    size_t stage_doubles = 0;
    // communicator (RCCL), opaque here
    void *comm = nullptr;
    void *shm = nullptr;    // loop-back rehearsal transport (NPG_COMM_TRANSPORT=shm, comm.hip)
    void *peer = nullptr;   // peer-mapped windows (comm.hip, PeerComm): in-cycle halo / all-reduce by xGMI stores + flags
    int rank = 0, nranks = 1;
};

struct npg_vec {
    npg_ctx *ctx = nullptr;
    int64_t n = 0;
    double *d = nullptr;
    bool owns = true;
};

struct npg_csr {
    npg_ctx *ctx = nullptr;
    int64_t m = 0, n = 0, nnz = 0;
    int64_t *rowptr = nullptr;   // device, m+1
    int32_t *col = nullptr;      // device, nnz
    double *val = nullptr;       // device, nnz
    bool owns_pattern = true;
    // nnz-balanced row tiles for the tiled SpMV kernels (device + host copy)
    npg::TileDesc *tile_ptr = nullptr; // device, ntiles descriptors
    int32_t ntiles = 0;
    int32_t ntiles_interior = 0; // row block of a distributed matrix (n > m: columns [owned | ghosts]): the first
                                 // ntiles_interior descriptors are tiles without ghost columns (= ntiles otherwise)
    int32_t lanes = 16;          // lanes per row chosen from the mean row length (lanes_default) or set by npg_csr_set_lanes
    int32_t lanes_default = 16;
    bool lanes_set = false;
    std::vector<int64_t> h_rowptr;
    // node-block part (npg_csr_block_nodes, spmv_device.h): the rows of the first nfull (x, y, z) nodes and of the nsurf
    // (x, y) nodes after them keep only their non-block entries in rowptr/col/val; `nnz` stays the LOGICAL entry count of
    // the matrix, `rnnz` counts what is left in col/val
    int64_t rnnz = 0;
    int32_t nfull = 0, nsurf = 0;
    int64_t *prow = nullptr;     // device, nfull + nsurf + 1
    int32_t *pcol = nullptr;     // device, column node of a record
    double *pkc = nullptr;       // device, {K, C} per record
    std::vector<int64_t> h_prow;
    // coupling records of the rows behind the block rows (the pressure rows of A_inversion; spmv_device.h): what such a row
    // holds in the block COLUMNS is stored as one record {c, d_x, d_y, d_z} per column node; rowptr/col/val keep the rest
    int64_t *drow = nullptr;     // device, (m - block_rows) + 1 record offsets; null without coupling records
    int32_t *dcol = nullptr;     // device, column node of a record
    double *dval = nullptr;      // device, [2 ndrec] (d_x, d_y) pairs followed by [ndrec] d_z
    float *dval32 = nullptr;     // optional fp32 copy, same layout
    int64_t ndrec = 0;
    std::vector<int64_t> h_drow;
    // column records of the block rows (spmv_device.h): what node q's rows hold OUTSIDE the block columns, one record
    // {m, a_x, a_y, a_z} per distinct column; the block rows then have no entries left in rowptr/col/val
    int64_t *grow = nullptr;     // device, nfull + nsurf + 1 record offsets; null without column records
    int32_t *gcol = nullptr;     // device, column of a record
    double *gval = nullptr;      // device, [2 ngrec] (a_x, a_y) pairs followed by [ngrec] a_z
    float *gval32 = nullptr;     // optional fp32 copy, same layout
    int64_t ngrec = 0;
    std::vector<int64_t> h_grow;
    // FULL node records (npg_csr_pack_nodes: function-valued viscosity, all nine component pairs per node pair): a SEPARATE
    // matrix object in record form hangs off the plain one (`packed`), which stays the assembly target; the packed object
    // holds, beside the record arrays, for every stored value the position in the plain matrix's `val` it is refreshed from
    // (-1: structural zero) - csr_repack() after every change of the plain values.  SpMV-side code goes through spmv_form().
    npg_csr *packed = nullptr;
    double *pk9 = nullptr;       // device, [9][npk9]: a_rs of record e at (3 r + s) npk9 + e (prow / pcol index the records)
    float *pk9_32 = nullptr;
    int64_t npk9 = 0;
    int64_t *map9 = nullptr, *mapd = nullptr, *mapg = nullptr, *maprem = nullptr;   // device, same layouts as pk9 / dval / gval / val
    // windowed tile set of the block rows (spmv_window.h, build_window_tiles): a SECOND tiling of the same matrix for the
    // kernels that gather from the fp32 gather-layout copy of their input - every node's record list padded to an even
    // count (zero records), 16-bit window indices beside pcol / gcol, per-tile lists of distinct columns
    npg::WTileDesc *wtile_ptr = nullptr; // device, nwtiles descriptors: windowed block tiles, then the ordinary tiles of the other rows
    int32_t nwtiles = 0, nwtiles_interior = 0;
    int32_t wlanes = 8;
    uint16_t *widx = nullptr;    // device, window index of every node record (same indexing as pcol)
    uint16_t *gidx = nullptr;    // device, window index of every column record (same indexing as gcol)
    int32_t *wlist = nullptr;    // device, concatenated per-tile lists of distinct column nodes (ascending per tile)
    int32_t *vlist = nullptr;    // device, concatenated per-tile lists of distinct columns of the column records
    // GHOST NODES of a rank's row block (npg_csr_set_ghost_nodes, round 5): ghost columns [gn_col[g], gn_col[g] + gn_ncomp[g]) are the
    // (x, y[, z]) components of ONE velocity node owned by a neighbour.  The windowed tile set then stores a row node's coupling to
    // such a node as ONE {c, K, C} node record with c = nnode() + g (its window entry is the node's 4-float slot behind the owned
    // nodes' slots in the gather-layout copy) instead of three column records - which is what kept a rank's boundary tiles at a
    // fifth of their size.  The ordinary tiles and the fp64 kernels keep the column records.
    std::vector<int32_t> gn_col, gn_ncomp;
    int32_t *gslot = nullptr;    // device, per ghost column (n - m entries): float position of its slot in the gather-layout copy, -1: none
    double *wgval = nullptr;     // device, the WINDOWED set's own column-record values when it differs from gval: [2 nw_grec] (a_x, a_y), [nw_grec] a_z
    int64_t nw_grec = 0, nw_rec = 0;   // column records / padded node records of the windowed set (= ngrec / h_prow[nnode] without ghost nodes)
    double *wdz = nullptr;       // device, d_z of the windowed set's own coupling records (rows behind the block rows, ghost nodes included)
    int64_t nw_drec = 0;         // its padded coupling records
    int64_t ngn() const { return (int64_t)gn_col.size(); }
    double *pkc2 = nullptr;      // device, the {K, C} values split by position in the record pair: [npairs] firsts, [npairs] seconds
    int64_t npairs = 0;
    double *dxy2 = nullptr;      // device, the (d_x, d_y) of the coupling records split the same way: [ndpairs] firsts, [ndpairs] seconds
    int64_t ndpairs = 0;
    int32_t *wbk = nullptr;      // device, [nnode][2]: tile-local end offsets of node q's column records / record pairs
    uint16_t *dwidx = nullptr;   // device, window index of every coupling record (windowed tiles of the rows behind the block rows)
    int32_t *dbk = nullptr;      // device, per row behind the block rows: tile-local end offset of its coupling record PAIRS
    int32_t nwrow_tiles = 0;     // windowed tiles of the rows behind the block rows (0: those rows keep their ordinary tiles)
    // block-diagonal matrix with arbitrary index sets as blocks (npg_csr_line_block_inverse): the blocks also packed DENSE,
    // column-major, one after the other - what products with it stream instead of the CSR arrays (k_line_apply, csr.hip)
    int64_t lb_nblocks = 0;
    int64_t *lb_ptr = nullptr;   // device, [nblocks + 1] offsets into lb_dofs
    int64_t *lb_dofs = nullptr;  // device, the blocks' rows / columns, ascending per block
    int64_t *lb_off = nullptr;   // device, [nblocks + 1] offsets of the dense blocks (sum of n^2)
    double *lb_val = nullptr;    // device, the dense blocks
    float *lb_val32 = nullptr;   //         and rounded to fp32 (products that ask for fp32 operator values)
    _Float16 *lb_val16 = nullptr; //        and to fp16, every column of a block divided by lb_scale[its unknown] = its largest magnitude
    double *lb_scale = nullptr;  // device, [nu] (what fp32-asking products read with NPG_LINE_FP16=1)
    int64_t ndrec_real = 0;      // coupling records without the zero records that pad a row's list to an even count
    int64_t nwlist = 0, nvlist = 0;
    int64_t nrec_real = 0;       // node records without the zero records that pad a node's list to an even count
    std::vector<npg::TileDesc> h_tiles;   // host copy of tile_ptr (final order)
    // npg_csr_block_nodes_dofs: the matrix was handed over in the CALLER's DoF order and is stored in the library's node-block
    // order; uperm[i] = caller index of internal row / column i.  npg_spmv and npg_gmres_solve take and return vectors in the
    // caller's order (one gather / scatter pass each way through the three scratch vectors); every other entry point refuses it
    int32_t *uperm = nullptr;    // device, m entries; null: no internal renumbering
    double *uvec[3] = {nullptr, nullptr, nullptr};      // device scratch, m doubles each: right-hand side, iterate, diagonal
    bool uperm_active = false;   // set while npg_gmres_solve runs on the internally ordered vectors
    // optional fp32 copies of the values (csr_refresh_fp32): read instead of val / pkc by SpMVs that ask for them
    // (SpmvEpi::f32 - the multigrid preconditioner's; results are still accumulated and returned in fp64)
    float *val32 = nullptr, *pkc32 = nullptr;
    // bumped whenever a pointer or count that kernels (and hence captured hipGraphs) bake in changes: build_tiles,
    // npg_csr_block_nodes, the first csr_refresh_fp32.  Holders of captured graphs compare it (mg.hip).
    uint64_t gen = 0;
    int64_t nnode() const { return (int64_t)nfull + nsurf; }
    int64_t block_rows() const { return 3 * (int64_t)nfull + 2 * (int64_t)nsurf; }
};

// interface exchange plan of a row-block distributed vector [owned | ghosts] (comm.hip)
struct npg_halo {
    npg_ctx *ctx = nullptr;
    int64_t n_owned = 0, n_ghost = 0;
    int npeers = 0;
    std::vector<int> peer;
    std::vector<int64_t> send_ptr, recv_ptr;
    int32_t *send_idx = nullptr;   // device
    double *send_buf = nullptr;    // device
    // overlapped exchange (halo_exchange_async): its own stream and the two events that order it against the compute stream
    hipStream_t cstream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    double *pending_x = nullptr;   // vector of the exchange begun by halo_exchange_async()
    float *pending_g32 = nullptr;  // ... and where its ghosts are ALSO stored as floats (null: nowhere)
    const int32_t *pending_gslot = nullptr;   // ... and their node slots in the gather-layout copy (halo_exchange_raw)
    float *pending_xgb = nullptr;
    void *pw = nullptr;            // peer-transport part of the plan (comm.hip, HaloPeer); null for RCCL / shm
};

namespace npg {
// enqueue the exchange of x's ghost segment / an in-place sum over ranks of n doubles on the context's stream
// g32 (optional): the received ghost values are also stored, rounded to fp32, in g32[0 .. n_ghost) - the Krylov kernels'
// gather-layout copy of their SpMV input (gmres.hip)
// gslot / xgb (optional, with g32): ghost entry i is ALSO stored, rounded, at xgb[gslot[i]] where gslot[i] >= 0 - its node's 4-float
// slot in the gather-layout copy (ghost nodes as record columns of the windowed tiles, npg_csr_set_ghost_nodes)
int halo_exchange_raw(npg_halo *h, double *x, float *g32 = nullptr, const int32_t *gslot = nullptr, float *xgb = nullptr);
// The same exchange in two halves, for a caller with work that needs no ghost value: kernels enqueued between the two calls
// run while the exchange is in flight and must not touch x's ghost segment.  Peer transport: the push half and the wait +
// unpack half are kernels on the context's own stream (the neighbours' stores arrive meanwhile).  RCCL: the exchange runs on
// the plan's own stream, ordered against the context's stream by two events.
int halo_exchange_async(npg_halo *h, double *x, float *g32 = nullptr, const int32_t *gslot = nullptr, float *xgb = nullptr);
int halo_exchange_wait(npg_halo *h);
int allreduce_sum_device(npg_ctx *ctx, double *buf, int n);
// in-place sum over the ranks of a long device vector (not a per-iteration collective: comm.hip)
int allreduce_big_device(npg_ctx *ctx, double *buf, int64_t n);
// fold `nrows` partial rows of kPartStride doubles into one row and sum it over the ranks into out[0 .. kPartStride):
// one kernel on the peer transport (fold + push + poll), fold kernel + collective otherwise
int fold_allreduce_rows(npg_ctx *ctx, const double *part, int nrows, double *out, hipStream_t st);
// true when every in-cycle communication call only launches kernels on HIP streams (the peer transport): a solver may
// then capture its distributed cycle into a hipGraph without any library's captured nodes
bool comm_is_kernel_only(const npg_ctx *ctx);
// NPG_OK, or NPG_ECOMM when a device-side wait of the peer transport has timed out since the communicator was created
int comm_check(const npg_ctx *ctx);
int ensure_stage(npg_ctx *ctx, size_t doubles);
int build_tiles(npg_csr *A);
// tile boundaries (consecutive whole rows, at most tile_slots LDS product slots) for any tile size: tuning harness
// the form the SpMV kernels read: the record-form companion of a plain matrix if it has one (npg_csr_pack_nodes)
inline const npg_csr *spmv_form(const npg_csr *A) { return (A && A->packed) ? A->packed : A; }
// Host-side consistency of what a tile kernel is about to index (O(1), before every launch that takes the matrix): the arrays its
// record loops read must exist and agree in length.  A kernel instance WITHOUT full-node-record support reading a companion
// (whose {K, C} array is null) is what the two memory faults of round 3 were - profiles/r04_round3_faults.txt.
inline int check_record_view(const npg_csr *A, bool n9_capable, const char *who) {
    if (!A) return NPG_OK;
    if (A->nnode() > 0) {
        NPG_REQUIRE(A->prow && A->pcol && (A->pkc != nullptr) != (A->pk9 != nullptr),
                    "%s: inconsistent record form (row offsets %p, columns %p, {K,C} %p, full records %p)", who, (void *)A->prow,
                    (void *)A->pcol, (void *)A->pkc, (void *)A->pk9);
        NPG_REQUIRE(!A->pk9 || n9_capable, "%s: this kernel does not read full node records (npg_csr_pack_nodes)", who);
        NPG_REQUIRE(!A->pk9 || A->npk9 == A->h_prow[(size_t)A->nnode()], "%s: %lld full node records but offsets for %lld", who,
                    (long long)A->npk9, (long long)A->h_prow[(size_t)A->nnode()]);
    } else {
        NPG_REQUIRE(!A->pk9 && !A->pkc && !A->grow, "%s: record arrays without block nodes", who);
    }
    NPG_REQUIRE(!A->grow || (A->gcol && A->gval && A->ngrec == A->h_grow[(size_t)A->nnode()]), "%s: inconsistent column records", who);
    NPG_REQUIRE(!A->drow || (A->dcol && A->dval && A->ndrec == A->h_drow[(size_t)(A->m - A->block_rows())]), "%s: inconsistent coupling records", who);
    return NPG_OK;
}
// refresh the companion's values from the plain matrix (no-op without one); enqueued on the context's stream
NPG_SHARED int csr_repack(const npg_csr *A);
NPG_SHARED int tile_boundaries(const npg_csr *A, int tile_slots, std::vector<int32_t> &tp, int max_rows = kTileRows);
struct CsrDev;
NPG_SHARED CsrDev csr_view(const npg_csr *A);
struct WinDev;
WinDev win_view(const npg_csr *A);
// epilogue of the tiled SpMV kernel: y = alpha (A x) + beta c   [c may be y itself; not read when beta == 0]
//                           and, if z:  z = zc zin + w dg .* y  [zin may be null; dg null: ones]
struct SpmvEpi {
    double alpha = 1.0, beta = 0.0;
    const double *c = nullptr;
    double *y = nullptr;
    double w = 0.0, zc = 0.0;
    const double *dg = nullptr, *zin = nullptr;
    double *z = nullptr;
    int f32 = 0;                 // read the matrix's fp32 value copies when it has them
};
// node-block epilogue of the tiled SpMV kernels (the multigrid smoother's t = Dinv r_u riding in the kernel that forms r = b - A x):
//   t[row] = sum_k Dinv[row, k] y[k] over the row's node block ; if xu: xu[row] = (zero ? 0 : xu[row]) + w t[row]      for row < rows
// (xu must not be what the product gathers from: other workgroups may still be reading it)
// Dinv = the node-block diagonal inverse in CSR (npg_csr_node_block_inverse); A's tiles must not split a node (node-blocked A).
struct NbEpi {
    const int64_t *drp;
    const int32_t *dcol;
    const double *dval;
    int32_t rows;
    double *t, *xu;
    double w;
    int zero;
};
// Byte accounting of an operator application (round 5: the multigrid bench line's roofline).  While `byte_sink` points somewhere,
// every product launched through spmv_epi / spmv_epi_gather32 adds the bytes its layout says it streams: the matrix in the form the
// kernel reads it (records, fp32 / fp16 value copies where the product takes them, the windowed set for the gather-layout product)
// + 8 B per input, output and epilogue-operand entry.  Host-side, free when null; a captured cycle is counted once, at capture.
extern thread_local int64_t *byte_sink;
int64_t spmv_stream_bytes(const npg_csr *A, const SpmvEpi &e, bool windowed);
int spmv_epi(const npg_csr *A, const double *x, const SpmvEpi &e, const NbEpi *nb = nullptr);
int spmv_raw(const npg_csr *A, const double *x, double *y, double alpha, double beta, int f32 = 0);
// the same product of a node-blocked matrix with a windowed tile set (spmv_window.h), its input rounded to fp32 into the gather
// layout first: xg = scratch of gather32_floats(A) floats (0: A cannot take this path)
int spmv_epi_gather32(const npg_csr *A, const double *x, float *xg, const SpmvEpi &e, const NbEpi *nb = nullptr);
// may the node-block epilogue ride in products with A?  (A stored by node blocks with Dinv's node counts: no tile splits a block)
bool nb_epilogue_ok(const npg_csr *A, const npg_csr *Dinv, int64_t nu);
int64_t gather32_floats(const npg_csr *A);
// 4-float node slots at the head of that copy: owned block nodes + the ghost nodes the windowed set uses as record columns
int64_t gather32_nodes(const npg_csr *A);
}  // namespace npg
struct npg_ilu0;
namespace npg {
// CG with the ILU(0) factors as M on raw device pointers (ilu.hip)
int ilu_pcg_raw(npg_ilu0 *m, const npg_csr *A, const double *b, double *x, double atol, double rtol, int64_t itmax,
                npg_solve_stats *stats);
// vectors of a matrix with an internal renumbering (npg_csr::uperm): dst[i] = src[uperm[i]] / dst[uperm[i]] = src[i]
void perm_gather(const npg_csr *A, double *dst, const double *src);
void perm_scatter(const npg_csr *A, double *dst, const double *src);
// (re)build the fp32 copies of A's values from val / pkc (enqueued on the context's stream)
int csr_refresh_fp32(const npg_csr *A);
// reductions that return a scalar to the host (synchronous)
int reduce_dot(npg_ctx *ctx, const double *x, const double *y, int64_t n, double *out);
int reduce_maxabs(npg_ctx *ctx, const double *x, int64_t n, double *out, int *has_nan);
}  // namespace npg

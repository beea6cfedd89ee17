// Device-resident preconditioned CG: replaces Krylov.krylov_solve!(::CgWorkspace, A, y, x; M=P, ...) as configured at
// /root/reference/src/evolution.jl:118-126 (SPD A = M + theta (Kh + Kv), Jacobi P) and driven from
// /root/reference/src/iterative_solvers.jl:58.
//
// Three kernels per iteration, scalars never leave the device:
//   CS  Ap = A p (tiled CSR SpMV), partial p'Ap
//   CU  alpha = gamma / p'Ap ; x += alpha p ; r -= alpha Ap ; z = P r ; partial r'z
//   CP  gamma' = r'z ; stopping test sqrt(gamma') <= atol + rtol sqrt(gamma_0) ; beta = gamma'/gamma ; p = z + beta p
// State snapshots alternate between two slots so that a workgroup never reads what another workgroup of the same launch
// writes.  The host looks at the state every `chunk` iterations only.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "common.h"
#include "spmv_device.h"

namespace npg {

struct CSnap {
    double gamma, eps, rnorm0, rnorm;
    int iter, done, first, pad;
};

struct CParams {
    double atol, rtol;
    long long itmax;
};

struct CDev {
    CsrDev A;
    const TileDesc *tile_ptr;
    int ntiles, n;
    int pkind;
    double pscalar;
    const double *pdiag;
    const double *b;
    double *x, *r, *z, *p, *Ap;
    double *Pg, *Pp;
    int G1, G2;
    const double *Qp;     // what k_cg_update reduces: Pp on one GPU, the all-reduced row when distributed
    int nQp;
    CSnap *S;   // two slots
    double *hist;
    int hist_cap;
    const CParams *prm;
};

constexpr int kKB = 1024;                 // threads per Krylov workgroup (see gmres.hip)
constexpr int kKW = kKB / 64;
constexpr int kNS = kKB / kPartStride;
constexpr int kMaxG = 256;
constexpr int kMaxI = kMaxG / kNS;

struct CShared {
    double tmp[kNS * kPartStride];
    double red[kPartStride];
    double wsum[kKW * kPartStride];
    CSnap S;
};

__device__ __forceinline__ double cg_precond(const CDev &d, int64_t row) {
    return d.pkind == NPG_PRECOND_SCALAR ? d.pscalar : (d.pkind == NPG_PRECOND_DIAG ? d.pdiag[row] : 1.0);
}

// r = b - A x ; z = P r ; partial r'z
template <int L>
__global__ void __launch_bounds__(kKB) k_cg_init(CDev d) {
    __shared__ double sh[kKW * kPartStride];
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    double acc[1] = {0.0};
    TileDesc nd = d.tile_ptr[blockIdx.x < (unsigned)d.ntiles ? blockIdx.x : 0];
    for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
        const TileDesc td = nd;
        if (t + (int)gridDim.x < d.ntiles) nd = d.tile_ptr[t + gridDim.x];      // in flight during this tile
        const int r0 = td.r0, r1 = td.r0 + td.nrows;
        spmv_tile<kKB, L>(d.A, PlainX{d.x}, td, tl, sw);
        if ((int)threadIdx.x < r1 - r0) {
            const int row = r0 + threadIdx.x;
            const double r = d.b[row] - sw[threadIdx.x];
            const double z = cg_precond(d, row) * r;
            d.r[row] = r;
            d.z[row] = z;
            acc[0] += r * z;
        }
    }
    block_store_partials<1, kKW>(acc, 1, sh, d.Pg);
}

// CP: reads slot `src`, writes slot `dst`.  ng = number of partial rows in Pg.
__global__ void __launch_bounds__(kKB) k_cg_direction(CDev d, int src, int dst, const double *qg, int ng) {
    __shared__ CShared sh;
    const CSnap prev = d.S[src];
    reduce_partials<kNS, kMaxI>(qg, ng, 1, sh.tmp, sh.red);
    if (threadIdx.x == 0) {
        CSnap s = prev;
        if (s.done == 0) {
            const double g = sh.red[0];
            if (s.first) {
                s.rnorm0 = sqrt(g);
                s.rnorm = s.rnorm0;
                s.eps = d.prm->atol + d.prm->rtol * s.rnorm0;
                s.first = 0;
                s.iter = 0;
                s.done = (g == 0.0) ? 4 : (s.rnorm0 <= s.eps ? 1 : 0);
                s.pad = 1;   // beta = 0 marker
                if (blockIdx.x == 0) d.hist[0] = s.rnorm0;
            } else {
                s.rnorm = sqrt(g);
                s.iter += 1;
                if (blockIdx.x == 0 && s.iter < d.hist_cap) d.hist[s.iter] = s.rnorm;
                const bool solved = (s.rnorm <= s.eps) || (s.rnorm + 1.0 <= 1.0);
                s.done = solved ? 1 : ((long long)s.iter >= d.prm->itmax ? 2 : (g != g ? 3 : 0));
                s.pad = 0;
            }
            sh.red[1] = s.pad ? 0.0 : g / prev.gamma;
            s.gamma = g;
        }
        sh.S = s;
        if (blockIdx.x == 0) d.S[dst] = s;
    }
    __syncthreads();
    if (sh.S.done != 0) return;
    const double beta = sh.red[1];
    for (int64_t row = blockIdx.x * (int64_t)kKB + threadIdx.x; row < d.n; row += (int64_t)gridDim.x * kKB)
        d.p[row] = d.z[row] + beta * d.p[row];
}

// CS
template <int L>
__global__ void __launch_bounds__(kKB) k_cg_spmv(CDev d, int slot) {
    __shared__ double sh[kKW * kPartStride];
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    if (d.S[slot].done != 0) return;
    double acc[1] = {0.0};
    TileDesc nd = d.tile_ptr[blockIdx.x < (unsigned)d.ntiles ? blockIdx.x : 0];
    for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
        const TileDesc td = nd;
        if (t + (int)gridDim.x < d.ntiles) nd = d.tile_ptr[t + gridDim.x];      // in flight during this tile
        const int r0 = td.r0, r1 = td.r0 + td.nrows;
        spmv_tile<kKB, L>(d.A, PlainX{d.p}, td, tl, sw);
        if ((int)threadIdx.x < r1 - r0) {
            const int row = r0 + threadIdx.x;
            const double ap = sw[threadIdx.x];
            d.Ap[row] = ap;
            acc[0] += d.p[row] * ap;
        }
    }
    block_store_partials<1, kKW>(acc, 1, sh, d.Pp);
}

// CU
__global__ void __launch_bounds__(kKB) k_cg_update(CDev d, int slot) {
    __shared__ CShared sh;
    const CSnap s = d.S[slot];
    reduce_partials<kNS, kMaxI>(d.Qp, d.nQp, 1, sh.tmp, sh.red);
    if (s.done != 0) return;
    const double pAp = sh.red[0];
    const double alpha = (pAp > 0.0) ? s.gamma / pAp : 0.0;
    double acc[1] = {0.0};
    for (int64_t row = blockIdx.x * (int64_t)kKB + threadIdx.x; row < d.n; row += (int64_t)gridDim.x * kKB) {
        const double pv = d.p[row];
        d.x[row] += alpha * pv;
        const double r = d.r[row] - alpha * d.Ap[row];
        const double z = cg_precond(d, row) * r;
        d.r[row] = r;
        d.z[row] = z;
        acc[0] += r * z;
    }
    block_store_partials<1, kKW>(acc, 1, sh.wsum, d.Pg);
}

__global__ void __launch_bounds__(kKB) k_cg_reduce_rows(const double *part, int nrows, double *out) {
    __shared__ double tmp[kNS * kPartStride];
    __shared__ double red[kPartStride];
    reduce_partials<kNS, kMaxI>(part, nrows, 1, tmp, red);
    if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace npg

using namespace npg;

struct npg_cg {
    npg_ctx *ctx = nullptr;
    int64_t n = 0;
    double *r = nullptr, *z = nullptr, *p = nullptr, *Ap = nullptr, *Pg = nullptr, *Pp = nullptr;
    CSnap *S = nullptr;
    CParams *prm = nullptr;
    double *hist = nullptr;
    int hist_cap = 0;
    CSnap *h_S = nullptr;
    CParams *h_prm = nullptr;
    int64_t hist_len = 0;
    npg_halo *halo = nullptr;
    double *Rg = nullptr;          // 2 rows of kPartStride doubles: all-reduced p'Ap and r'z in their first entries (distributed mode)
    int64_t n_ghost = 0;
    static constexpr int kMaxG = npg::kMaxG;
};

NPG_API int npg_cg_create(npg_ctx *ctx, int64_t n, npg_cg **out) {
    NPG_REQUIRE(ctx && out && n > 0 && n < INT32_MAX, "npg_cg_create: bad argument");
    npg_cg *ws = new npg_cg();
    ws->ctx = ctx;
    ws->n = n;
    NPG_HIP(hipSetDevice(ctx->device));
    const size_t vb = (size_t)n * sizeof(double);
    NPG_HIP(hipMalloc((void **)&ws->r, vb));
    NPG_HIP(hipMalloc((void **)&ws->z, vb));
    NPG_HIP(hipMalloc((void **)&ws->p, vb));
    NPG_HIP(hipMalloc((void **)&ws->Ap, vb));
    const size_t pb = (size_t)npg_cg::kMaxG * kPartStride * sizeof(double);
    NPG_HIP(hipMalloc((void **)&ws->Pg, pb));
    NPG_HIP(hipMalloc((void **)&ws->Pp, pb));
    NPG_HIP(hipMalloc((void **)&ws->S, 2 * sizeof(CSnap)));
    NPG_HIP(hipMalloc((void **)&ws->prm, sizeof(CParams)));
    ws->hist_cap = (int)std::min<int64_t>(2 * n + 2, 1 << 22);
    NPG_HIP(hipMalloc((void **)&ws->hist, sizeof(double) * ws->hist_cap));
    NPG_HIP(hipHostMalloc((void **)&ws->h_S, 2 * sizeof(CSnap), hipHostMallocDefault));
    NPG_HIP(hipHostMalloc((void **)&ws->h_prm, sizeof(CParams), hipHostMallocDefault));
    NPG_HIP(hipMemsetAsync(ws->p, 0, vb, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    *out = ws;
    return NPG_OK;
}

NPG_API int npg_cg_destroy(npg_cg *ws) {
    if (!ws) return NPG_OK;
    hipStreamSynchronize(ws->ctx->stream);
    void *ptrs[] = {ws->r, ws->z, ws->p, ws->Ap, ws->Pg, ws->Pp, ws->S, ws->prm, ws->hist, ws->Rg};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (ws->h_S) hipHostFree(ws->h_S);
    if (ws->h_prm) hipHostFree(ws->h_prm);
    delete ws;
    return NPG_OK;
}

NPG_API int npg_cg_set_halo(npg_cg *ws, npg_halo *h) {
    NPG_REQUIRE(ws, "npg_cg_set_halo: NULL workspace");
    NPG_REQUIRE(!h || h->n_owned == ws->n, "npg_cg_set_halo: the plan owns %lld rows, the workspace %lld",
                h ? (long long)h->n_owned : 0LL, (long long)ws->n);
    NPG_HIP(hipStreamSynchronize(ws->ctx->stream));
    ws->halo = h;
    ws->n_ghost = h ? h->n_ghost : 0;
    NPG_HIP(hipFree(ws->p));          // the SpMV input p needs room for the ghost entries
    const size_t nb = (size_t)(ws->n + ws->n_ghost) * sizeof(double);
    NPG_HIP(hipMalloc((void **)&ws->p, nb));
    NPG_HIP(hipMemset(ws->p, 0, nb));
    if (h && !ws->Rg) {
        NPG_HIP(hipMalloc((void **)&ws->Rg, 2 * kPartStride * sizeof(double)));
        NPG_HIP(hipMemset(ws->Rg, 0, 2 * kPartStride * sizeof(double)));
    }
    return NPG_OK;
}

// fold the partial rows and sum over the ranks: one kernel on the peer transport (comm.hip), fold + collective otherwise
static int cg_dist_reduce(npg_cg *ws, const double *part, int nrows, int slot, hipStream_t st) {
    return fold_allreduce_rows(ws->ctx, part, nrows, ws->Rg + slot * kPartStride, st);
}

template <int L>
static int cg_run(npg_cg *ws, const CDev &d, int64_t itmax, CSnap *last) {
    hipStream_t st = ws->ctx->stream;
    npg_cg *dist = ws->halo ? ws : nullptr;
    int rc = NPG_OK;
    if (dist && (rc = halo_exchange_raw(ws->halo, d.x))) return rc;
    hipLaunchKernelGGL(k_cg_init<L>, dim3(d.G1), dim3(kKB), 0, st, d);
    if (dist && (rc = cg_dist_reduce(ws, d.Pg, d.G1, 1, st))) return rc;
    // slot 0 = initial state; r'z comes from the init kernel's G1 partial rows (or from the all-reduced scalar)
    hipLaunchKernelGGL(k_cg_direction, dim3(d.G2), dim3(kKB), 0, st, d, 0, 1, dist ? ws->Rg + kPartStride : d.Pg, dist ? 1 : d.G1);
    int cur = 1;
    const int chunk = 4;
    int64_t it = 0;
    while (true) {
        NPG_HIP(hipMemcpyAsync(ws->h_S, ws->S + cur, sizeof(CSnap), hipMemcpyDeviceToHost, st));
        NPG_HIP(hipStreamSynchronize(st));
        *last = ws->h_S[0];
        if (last->done != 0 || it >= itmax) break;
        for (int k = 0; k < chunk; ++k, ++it) {
            if (dist && (rc = halo_exchange_raw(ws->halo, d.p))) return rc;
            hipLaunchKernelGGL(k_cg_spmv<L>, dim3(d.G1), dim3(kKB), 0, st, d, cur);
            if (dist && (rc = cg_dist_reduce(ws, d.Pp, d.G1, 0, st))) return rc;
            hipLaunchKernelGGL(k_cg_update, dim3(d.G2), dim3(kKB), 0, st, d, cur);
            if (dist && (rc = cg_dist_reduce(ws, d.Pg, d.G2, 1, st))) return rc;
            hipLaunchKernelGGL(k_cg_direction, dim3(d.G2), dim3(kKB), 0, st, d, cur, cur ^ 1, dist ? ws->Rg + kPartStride : d.Pg,
                               dist ? 1 : d.G2);
            cur ^= 1;
        }
        NPG_HIP(hipGetLastError());
    }
    return NPG_OK;
}

NPG_API int npg_cg_solve(npg_cg *ws, const npg_csr *A_in, int precond_kind, double precond_scalar,
                         const npg_vec *precond_diag, const npg_vec *y, npg_vec *x, double atol, double rtol,
                         int64_t itmax, npg_solve_stats *stats) {
    NPG_REQUIRE(ws && A_in && y && x, "npg_cg_solve: NULL argument");
    NPG_REQUIRE(!A_in->packed && !A_in->pk9, "npg_cg_solve: matrices with full node records are not served by the CG kernels");
    NPG_REQUIRE(!A_in->uperm, "npg_cg_solve: the matrix carries an internal renumbering (npg_csr_block_nodes_dofs): npg_spmv and npg_gmres_solve only");
    const npg_csr *A = A_in;
    if (int rc = check_record_view(A, false, "npg_cg_solve")) return rc;
    const int64_t nloc = ws->n + ws->n_ghost;     // distributed: vectors the SpMV reads hold [owned | ghosts]
    NPG_REQUIRE(A->m == ws->n && A->n == nloc && y->n == ws->n && x->n == nloc,
                "npg_cg_solve: workspace is for n=%lld (+%lld ghosts) but A is %lldx%lld, y has %lld, x has %lld",
                (long long)ws->n, (long long)ws->n_ghost, (long long)A->m, (long long)A->n, (long long)y->n,
                (long long)x->n);
    NPG_REQUIRE(precond_kind == NPG_PRECOND_NONE || precond_kind == NPG_PRECOND_SCALAR ||
                    (precond_kind == NPG_PRECOND_DIAG && precond_diag && precond_diag->n == ws->n),
                "npg_cg_solve: bad preconditioner");
    const auto t0 = std::chrono::steady_clock::now();
    npg_ctx *ctx = ws->ctx;
    CDev d;
    memset(&d, 0, sizeof d);
    d.A = csr_view(A);
    d.tile_ptr = A->tile_ptr;
    d.ntiles = A->ntiles;
    d.n = (int)ws->n;
    d.pkind = precond_kind;
    d.pscalar = precond_scalar;
    d.pdiag = precond_kind == NPG_PRECOND_DIAG ? precond_diag->d : nullptr;
    d.b = y->d;
    d.x = x->d;
    d.r = ws->r;
    d.z = ws->z;
    d.p = ws->p;
    d.Ap = ws->Ap;
    d.Pg = ws->Pg;
    d.Pp = ws->Pp;
    d.G1 = std::max(1, std::min<int>(A->ntiles, std::min(npg_cg::kMaxG, ctx->num_cu)));
    d.G2 = (int)std::max<int64_t>(1, std::min<int64_t>((ws->n + kKB - 1) / kKB, d.G1));
    d.Qp = ws->halo ? ws->Rg : d.Pp;
    d.nQp = ws->halo ? 1 : d.G1;
    d.S = ws->S;
    d.hist = ws->hist;
    d.hist_cap = ws->hist_cap;
    d.prm = ws->prm;
    if (itmax <= 0) itmax = 2 * ws->n;
    ws->h_prm->atol = atol;
    ws->h_prm->rtol = rtol;
    ws->h_prm->itmax = itmax;
    NPG_HIP(hipMemcpyAsync(ws->prm, ws->h_prm, sizeof(CParams), hipMemcpyHostToDevice, ctx->stream));
    CSnap s0{};
    s0.first = 1;
    ws->h_S[1] = s0;
    NPG_HIP(hipMemcpyAsync(ws->S, ws->h_S + 1, sizeof(CSnap), hipMemcpyHostToDevice, ctx->stream));
    CSnap last{};
    int rc;
    switch (A->lanes) {
        case 4: rc = cg_run<4>(ws, d, itmax, &last); break;
        case 8: rc = cg_run<8>(ws, d, itmax, &last); break;
        case 16: rc = cg_run<16>(ws, d, itmax, &last); break;
        default: rc = cg_run<32>(ws, d, itmax, &last); break;
    }
    if (rc) return rc;
    if (ws->halo && (rc = comm_check(ctx))) return rc;
    ws->hist_len = std::min<int64_t>((int64_t)last.iter + 1, ws->hist_cap);
    if (stats) {
        stats->solved = (last.done == 1 || last.done == 4) ? 1 : 0;
        stats->niter = last.iter;
        stats->npass = 1;
        stats->status = last.done;
        stats->nreorth = 0;
        stats->nflagged = 0;
        stats->rnorm0 = last.rnorm0;
        stats->rnorm = last.rnorm;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return NPG_OK;
}

NPG_API int64_t npg_cg_history(npg_cg *ws, double *buf, int64_t cap) {
    if (!ws || !buf || cap <= 0) return 0;
    const int64_t k = std::min<int64_t>(cap, ws->hist_len);
    if (hipMemcpy(buf, ws->hist, (size_t)k * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return k;
}

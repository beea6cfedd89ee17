// Tuning harness (not part of the public ABI): times stand-alone CSR-stream SpMV kernels with different workgroup sizes,
// tile sizes and load widths on an existing device matrix, so that the configuration used by the product kernels is
// chosen from measurements on the target matrix (see profiles/ and DESIGN.md).
#include <algorithm>
#include <map>

#include "../../nupgcm_amd/csrc/common.h"
#include "../../nupgcm_amd/csrc/spmv_device.h"
#include "spmv_pack.h"

namespace npg {

template <int NT, int L, int TNNZ, int U>
__global__ void __launch_bounds__(NT) k_spmv_var(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const double *__restrict__ val, const int32_t *__restrict__ tile_ptr,
                                                 int ntiles, int64_t nnz, const double *__restrict__ x,
                                                 double *__restrict__ y) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    const CsrDev A{rowptr, col, val, nnz, nullptr, nullptr, nullptr, 0, 0};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        spmv_tile<NT, L, PlainX, TNNZ, U>(A, PlainX{x}, r0, r1, tl, sw);
        for (int r = threadIdx.x; r < r1 - r0; r += NT) y[r0 + r] = sw[r];
    }
}

// wide loads: every lane reads two adjacent entries with one 16-byte val load and one 8-byte col load
template <int NT, int L, int TNNZ, int U2, bool NTL = false>
__global__ void __launch_bounds__(NT) k_spmv_wide(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                  const double *__restrict__ val, const int32_t *__restrict__ tile_ptr,
                                                  int ntiles, int64_t nnz, const double *__restrict__ x,
                                                  double *__restrict__ y) {
    __shared__ TileLdsT<TNNZ + 2> tl;
    __shared__ double sw[kTileRows];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        const int64_t base = rowptr[r0];
        const int n = (int)(rowptr[r1] - base);
        const int nrows = r1 - r0;
        const int64_t abase = base & ~1LL;
        const int off = (int)(base - abase);
        const int total = n + off;
        for (int r = threadIdx.x; r <= nrows; r += NT) tl.rp[r] = (int32_t)(rowptr[r0 + r] - base) + off;
        for (int k0 = 2 * threadIdx.x; k0 < total; k0 += 2 * NT * U2) {
            int2 c[U2];
            double2 v[U2];
            double xa[U2], xb[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                const bool ok = k < total && abase + k + 1 < nnz + (nnz & 1);
                if (ok && abase + k + 1 < nnz) {
                    if (NTL) {
                        const long long cc = __builtin_nontemporal_load(reinterpret_cast<const long long *>(col + abase + k));
                        c[u] = make_int2((int)(cc & 0xffffffffLL), (int)(cc >> 32));
                        v[u].x = __builtin_nontemporal_load(val + abase + k);
                        v[u].y = __builtin_nontemporal_load(val + abase + k + 1);
                    } else {
                        c[u] = *reinterpret_cast<const int2 *>(col + abase + k);
                        v[u] = *reinterpret_cast<const double2 *>(val + abase + k);
                    }
                } else if (k < total && abase + k < nnz) {
                    c[u] = make_int2(col[abase + k], 0);
                    v[u] = make_double2(val[abase + k], 0.0);
                } else {
                    c[u] = make_int2(0, 0);
                    v[u] = make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                xa[u] = x[c[u].x];
                xb[u] = x[c[u].y];
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                if (k < total) tl.prod[k] = (k >= off) ? v[u].x * xa[u] : 0.0;
                if (k + 1 < total) tl.prod[k + 1] = v[u].y * xb[u];
            }
        }
        __syncthreads();
        const int g = threadIdx.x / L, l = threadIdx.x % L;
        for (int r = g; r < nrows; r += NT / L) {
            double s = 0.0;
            const int e = tl.rp[r + 1];
            for (int k = tl.rp[r] + l; k < e; k += L) s += tl.prod[k];
            s = group_sum_dpp<L>(s);
            if (l == 0) sw[r] = s;
        }
        __syncthreads();
        for (int r = threadIdx.x; r < nrows; r += NT) y[r0 + r] = sw[r];
    }
}

struct VarTiles {
    int32_t *d = nullptr;
    int n = 0;
};

static std::map<std::pair<const void *, int>, VarTiles> g_tiles;

static int tiles_for(const npg_csr *A, int tnnz, VarTiles *out) {
    auto key = std::make_pair((const void *)A, tnnz);
    auto it = g_tiles.find(key);
    if (it != g_tiles.end()) {
        *out = it->second;
        return NPG_OK;
    }
    const int64_t *rp = A->h_rowptr.data();
    std::vector<int32_t> tp{0};
    int64_t r = 0;
    while (r < A->m) {
        int64_t r1 = r + 1;
        while (r1 < A->m && r1 - r < kTileRows && rp[r1 + 1] - rp[r] <= tnnz) ++r1;
        tp.push_back((int32_t)r1);
        r = r1;
    }
    VarTiles v;
    v.n = (int)tp.size() - 1;
    NPG_HIP(hipMalloc((void **)&v.d, tp.size() * sizeof(int32_t)));
    NPG_HIP(hipMemcpy(v.d, tp.data(), tp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    g_tiles[key] = v;
    *out = v;
    return NPG_OK;
}

template <int NT, int TNNZ, int U>
static int run_var(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    VarTiles t;
    int rc = tiles_for(A, TNNZ, &t);
    if (rc) return rc;
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min(t.n, bpc * ctx->num_cu));
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((k_spmv_var<NT, 16, TNNZ, U>), dim3(grid), dim3(NT), 0, ctx->stream, A->rowptr, A->col, A->val,
                           t.d, t.n, A->nnz, x, y);
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_spmv_var<NT, 16, TNNZ, U>), dim3(grid), dim3(NT), 0, ctx->stream, A->rowptr, A->col, A->val,
                           t.d, t.n, A->nnz, x, y);
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

template <int NT, int TNNZ, int U2, bool NTL = false>
static int run_wide(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    VarTiles t;
    int rc = tiles_for(A, TNNZ, &t);
    if (rc) return rc;
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min(t.n, bpc * ctx->num_cu));
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((k_spmv_wide<NT, 16, TNNZ, U2, NTL>), dim3(grid), dim3(NT), 0, ctx->stream, A->rowptr, A->col,
                           A->val, t.d, t.n, A->nnz, x, y);
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_spmv_wide<NT, 16, TNNZ, U2, NTL>), dim3(grid), dim3(NT), 0, ctx->stream, A->rowptr, A->col,
                           A->val, t.d, t.n, A->nnz, x, y);
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- variants on the product tile functions (work on xy-paired matrices too)
// diagnostic input: no memory access for x (prices the gathers)
struct FakeX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return 1e-9 * (double)c; }
    __device__ __forceinline__ double2 two(int c) const { return make_double2(1e-9 * (double)c, 2e-9 * (double)c); }
    __device__ __forceinline__ double third(int c) const { return 3e-9 * (double)c; }
};

// diagnostic input: everything gathered except the z component of the record columns (prices that one load)
struct NoZX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return x[c]; }
    __device__ __forceinline__ double2 two(int i) const {
        double2 r;
        __builtin_memcpy(&r, x + i, sizeof r);
        return r;
    }
    __device__ __forceinline__ double third(int c) const { return 3e-9 * (double)c; }
};


template <int NT, int L, int TNNZ, int U2, int WPE, class XF = FakeX>
__global__ void __launch_bounds__(NT, WPE) k_spmv_nogather(CsrDev A, const int32_t *__restrict__ tile_ptr, int ntiles,
                                                           const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        spmv_tile<NT, L, XF, TNNZ, U2>(A, XF{x}, r0, r1, tl, sw);
        for (int r = threadIdx.x; r < r1 - r0; r += NT) y[r0 + r] = sw[r];
    }
}

// diagnostic: what a tile-local gather list would cost - x values served from an LDS stage filled by ~1000 coalesced
// gathers per tile (wrong results on purpose: the stage holds x[r0 ...], the indices are folded into it)
struct StageX {
    const double *xs;
    __device__ __forceinline__ double operator()(int c) const { return xs[c & 1023]; }
    __device__ __forceinline__ double2 two(int c) const { return make_double2(xs[c & 1023], xs[(c + 1) & 1023]); }
    __device__ __forceinline__ double third(int c) const { return xs[c & 1023]; }
};

template <int NT, int L, int TNNZ, int U2, int WPE>
__global__ void __launch_bounds__(NT, WPE) k_spmv_stage(CsrDev A, const int32_t *__restrict__ tile_ptr, int ntiles,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        const int32_t *__restrict__ glist) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    __shared__ double xs[1024];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        for (int i = threadIdx.x; i < 1024; i += NT) xs[i] = x[glist[(size_t)t * 1024 + i]];
        __syncthreads();
        spmv_tile<NT, L, StageX, TNNZ, U2>(A, StageX{xs}, r0, r1, tl, sw);
        for (int r = threadIdx.x; r < r1 - r0; r += NT) y[r0 + r] = sw[r];
    }
}

// ---- "wide" variant: two workgroups per CU, 128 registers ------------------------------------------------------------------
// Same tile, same products, same summation order; but ALL loads of a tile (UP records and UC entry pairs per lane) are issued
// before the first gather and ALL gathers before the first product: two dependent memory round trips per tile instead of
// four.  Needs ~115 VGPRs, i.e. two 512-thread workgroups per CU, which in turn have room for tiles twice as large.
template <int NT, int L, class XF, int TNNZ, int UP, int UC, class PROF = NoProf>
__device__ __forceinline__ void spmv_tile_wide(const CsrDev &A, const XF x, const TileDesc &td, TileLdsT<TNNZ> &t,
                                               double *__restrict__ out, PROF prof = PROF()) {
    const int r0 = td.r0, nrows = td.nrows, r1 = r0 + nrows;
    const int64_t base = td.base;
    const int n = td.n;
    const bool blk = r0 < block_rows(A);
    const bool full = r0 < 3 * A.nfull;
    const int ncomp = full ? 3 : 2;
    const int npe = td.npe;
    const int64_t pbase = td.pbase;
    int nnode = 0, q0 = 0;
    if (blk) {
        q0 = node_of_row(A, r0);
        nnode = node_of_row(A, r1) - q0;
    }
    const int64_t abase = base & ~1LL;
    const int off = (int)(base - abase);
    const int total = n + off;
    const int slot0 = blk ? ncomp * npe : 0;
    if (slot0 + total > TNNZ + 2 || npe > UP * NT || total > 2 * NT * UC || abase + total > A.nnz) {
        spmv_tile<NT, L, XF, TNNZ, 4, PROF>(A, x, td, t, out, prof);       // generic path (long rows, odd shapes)
        return;
    }
    const int tid = threadIdx.x;
    int32_t rc[UP];
    double2 rkc[UP];
    int2 cc[UC];
    double2 cv[UC];
    // ---- every load of the tile
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int e = tid + u * NT;
        if (e < npe) {
            rc[u] = __builtin_nontemporal_load(A.pcol + pbase + e);
            const double *p = reinterpret_cast<const double *>(A.pkc + pbase + e);
            rkc[u].x = __builtin_nontemporal_load(p);
            rkc[u].y = __builtin_nontemporal_load(p + 1);
        } else {
            rc[u] = 0;
            rkc[u] = make_double2(0.0, 0.0);
        }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
        const int k = 2 * tid + u * 2 * NT;
        if (k + 1 < total) {
            const long long c2 = __builtin_nontemporal_load(reinterpret_cast<const long long *>(A.col + abase + k));
            cc[u] = make_int2((int)(c2 & 0xffffffffLL), (int)(c2 >> 32));
            cv[u].x = __builtin_nontemporal_load(A.val + abase + k);
            cv[u].y = __builtin_nontemporal_load(A.val + abase + k + 1);
        } else if (k < total) {
            cc[u] = make_int2(A.col[abase + k], 0);
            cv[u] = make_double2(A.val[abase + k], 0.0);
        } else {
            cc[u] = make_int2(0, 0);
            cv[u] = make_double2(0.0, 0.0);
        }
    }
    for (int r = tid; r <= nrows; r += NT) t.rp[r] = (int32_t)(A.rowptr[r0 + r] - base) + off + slot0;
    if (blk)
        for (int q = tid; q <= nnode; q += NT) t.prp[q] = (int32_t)(A.prow[q0 + q] - pbase);
    // ---- every gather of the tile
    double2 xx[UP];
    double zz[UP], xa[UC], xb[UC];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int cf = rc[u] < A.nfull ? rc[u] : A.nfull;
        const int xo = 2 * rc[u] + cf;
        xx[u] = x.two(xo);
        zz[u] = (full && rc[u] < A.nfull) ? x.third(xo + 2) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
        xa[u] = x(cc[u].x);
        xb[u] = x(cc[u].y);
    }
    // ---- products
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int e = tid + u * NT;
        if (e < npe) {
            t.prod[e] = rkc[u].x * xx[u].x + rkc[u].y * xx[u].y;
            t.prod[npe + e] = rkc[u].x * xx[u].y - rkc[u].y * xx[u].x;
            if (full) t.prod[2 * npe + e] = rkc[u].x * zz[u];
        }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
        const int k = 2 * tid + u * 2 * NT;
        if (k < total) t.prod[slot0 + k] = (k >= off) ? cv[u].x * xa[u] : 0.0;
        if (k + 1 < total) t.prod[slot0 + k + 1] = cv[u].y * xb[u];
    }
    prof.stamp(0);
    __syncthreads();
    prof.stamp(1);
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    for (int r = g; r < nrows; r += NT / L) {
        double s = 0.0;
        const int e = t.rp[r + 1];
        for (int k = t.rp[r] + 2 * l; k < e; k += 2 * L) {
            const double a = t.prod[k], b = t.prod[k + 1];
            s += a + (k + 1 < e ? b : 0.0);
        }
        if (blk) {
            const int q = full ? (r * 21846) >> 16 : r >> 1;
            const int pb = (r - q * ncomp) * npe, pe = pb + t.prp[q + 1];
            for (int k = pb + t.prp[q] + 2 * l; k < pe; k += 2 * L) {
                const double a = t.prod[k], b = t.prod[k + 1];
                s += a + (k + 1 < pe ? b : 0.0);
            }
        }
        s = group_sum_dpp<L>(s);
        if (l == 0) out[r] = s;
    }
    prof.stamp(2);
    __syncthreads();
}

template <int NT, int L, class XF, int TNNZ, int UP, int UC>
__device__ __forceinline__ void spmv_tile_wide(const CsrDev &A, const XF x, int r0, int r1, TileLdsT<TNNZ> &t,
                                               double *__restrict__ out) {
    TileDesc td;
    td.r0 = r0;
    td.nrows = r1 - r0;
    td.base = A.rowptr[r0];
    td.n = (int)(A.rowptr[r1] - td.base);
    td.pbase = 0;
    td.npe = 0;
    if (r0 < block_rows(A)) {
        td.pbase = A.prow[node_of_row(A, r0)];
        td.npe = (int)(A.prow[node_of_row(A, r1)] - td.pbase);
    }
    spmv_tile_wide<NT, L, XF, TNNZ, UP, UC>(A, x, td, t, out);
}

// the "wide" tile function: 2 workgroups per CU, everything in flight before the first gather
template <int NT, int L, int TNNZ, int UP, int UC, int WPE>
__global__ void __launch_bounds__(NT, WPE) k_spmv_wide2(CsrDev A, const int32_t *__restrict__ tile_ptr, int ntiles,
                                                        const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        spmv_tile_wide<NT, L, PlainX, TNNZ, UP, UC>(A, PlainX{x}, r0, r1, tl, sw);
        for (int r = threadIdx.x; r < r1 - r0; r += NT) y[r0 + r] = sw[r];
    }
}

template <int NT, int TNNZ, int UP, int UC, int WPE>
static int run_wide2(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms);

template <int NT, int L, int TNNZ, int U2, int WPE, bool MERGED>
__global__ void __launch_bounds__(NT, WPE) k_spmv_prod(CsrDev A, const int32_t *__restrict__ tile_ptr, int ntiles,
                                                       const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        spmv_tile<NT, L, PlainX, TNNZ, U2>(A, PlainX{x}, r0, r1, tl, sw);
        for (int r = threadIdx.x; r < r1 - r0; r += NT) y[r0 + r] = sw[r];
    }
}

static std::map<std::pair<const void *, int>, VarTiles> g_ptiles;

template <int NT, int TNNZ, int U2, int WPE, bool MERGED, int NOGATHER = 0>
static int run_prod(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    auto key = std::make_pair((const void *)A, TNNZ);
    if (!g_ptiles.count(key)) {
        std::vector<int32_t> tp;
        int rc = tile_boundaries(A, TNNZ, tp);
        if (rc) return rc;
        VarTiles v;
        v.n = (int)tp.size() - 1;
        NPG_HIP(hipMalloc((void **)&v.d, tp.size() * sizeof(int32_t)));
        NPG_HIP(hipMemcpy(v.d, tp.data(), tp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        g_ptiles[key] = v;
    }
    const VarTiles t = g_ptiles[key];
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min(t.n, bpc * ctx->num_cu));
    const CsrDev Av = csr_view(A);
    static std::map<std::pair<const void *, int>, int32_t *> g_lists;
    int32_t *glist = nullptr;
    if (NOGATHER == 3) {
        // a plausible gather list per tile: the tile's own rows first, then columns spread over a window around them
        if (!g_lists.count(key)) {
            std::vector<int32_t> tp((size_t)t.n + 1);
            NPG_HIP(hipMemcpy(tp.data(), t.d, tp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            std::vector<int32_t> gl((size_t)t.n * 1024);
            for (int q = 0; q < t.n; ++q)
                for (int i = 0; i < 1024; ++i) {
                    const int64_t c = (int64_t)tp[q] + (i < 256 ? i : (i - 640) * 37);
                    gl[(size_t)q * 1024 + i] = (int32_t)std::min<int64_t>(std::max<int64_t>(c, 0), A->n - 1);
                }
            for (int q = 0; q < t.n; ++q) std::sort(gl.begin() + (size_t)q * 1024, gl.begin() + (size_t)(q + 1) * 1024);
            int32_t *dg;
            NPG_HIP(hipMalloc((void **)&dg, gl.size() * sizeof(int32_t)));
            NPG_HIP(hipMemcpy(dg, gl.data(), gl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            g_lists[key] = dg;
        }
        glist = g_lists[key];
    }
    auto go = [&]() {
        if (NOGATHER == 3)
            hipLaunchKernelGGL((k_spmv_stage<NT, 16, TNNZ, U2, WPE>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.d, t.n, x, y,
                               glist);
        else if (NOGATHER == 1)
            hipLaunchKernelGGL((k_spmv_nogather<NT, 16, TNNZ, U2, WPE>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.d, t.n,
                               x, y);
        else if (NOGATHER == 2)
            hipLaunchKernelGGL((k_spmv_nogather<NT, 16, TNNZ, U2, WPE, NoZX>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.d,
                               t.n, x, y);
        else
            hipLaunchKernelGGL((k_spmv_prod<NT, 16, TNNZ, U2, WPE, MERGED>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.d,
                               t.n, x, y);
    };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- phase timing (diagnostic): the PRODUCT tile function with cycle stamps taken by every wave's lane 0 of workgroup
// thread 0.  acc[0..4] += cycles in: descriptor wait | stream loads + gathers + products | barrier 1 | segmented sums |
// barrier 2 + output ; acc[6] = tiles
struct CycleProf {
    unsigned long long *c;      // c[0..2] stamps
    __device__ __forceinline__ void stamp(int i) const {
        __builtin_amdgcn_s_waitcnt(0);
        c[i] = clock64();
    }
};

template <int NT, int L, int TNNZ, int U2, bool GATHER>
__global__ void __launch_bounds__(NT, 6) k_spmv_timed(CsrDev A, const TileDesc *__restrict__ tile_ptr, int ntiles,
                                                      const double *__restrict__ x, double *__restrict__ y,
                                                      unsigned long long *acc) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    unsigned long long a[5] = {0, 0, 0, 0, 0}, nt = 0, st[3];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    TileDesc td = tile_ptr[t];
    while (true) {
        const unsigned long long c0 = clock64();
        const int tn = t + gridDim.x;
        TileDesc nd = td;
        if (tn < ntiles) nd = tile_ptr[tn];
        const unsigned long long c1 = clock64();
        if (GATHER)
            spmv_tile<NT, L, PlainX, TNNZ, U2, CycleProf>(A, PlainX{x}, td, tl, sw, CycleProf{st});
        else
            spmv_tile<NT, L, FakeX, TNNZ, U2, CycleProf>(A, FakeX{x}, td, tl, sw, CycleProf{st});
        for (int r = threadIdx.x; r < td.nrows; r += NT) y[td.r0 + r] = sw[r];
        const unsigned long long c5 = clock64();
        a[0] += c1 - c0;
        a[1] += st[0] - c1;
        a[2] += st[1] - st[0];
        a[3] += st[2] - st[1];
        a[4] += c5 - st[2];
        ++nt;
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(acc + i, a[i]);
        atomicAdd(acc + 6, nt);
    }
}

template <int NT, int TNNZ, int UP, int UC, int WPE>
static int run_wide2(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    auto key = std::make_pair((const void *)A, TNNZ);
    if (!g_ptiles.count(key)) {
        std::vector<int32_t> tp;
        int rc = tile_boundaries(A, TNNZ, tp);
        if (rc) return rc;
        VarTiles v;
        v.n = (int)tp.size() - 1;
        NPG_HIP(hipMalloc((void **)&v.d, tp.size() * sizeof(int32_t)));
        NPG_HIP(hipMemcpy(v.d, tp.data(), tp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        g_ptiles[key] = v;
    }
    const VarTiles t = g_ptiles[key];
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min(t.n, bpc * ctx->num_cu));
    const CsrDev Av = csr_view(A);
    auto go = [&]() {
        hipLaunchKernelGGL((k_spmv_wide2<NT, 16, TNNZ, UP, UC, WPE>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.d, t.n, x,
                           y);
    };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- tile-packed blobs + LDS-DMA pipelined tiles (spmv_pack.h) ---------------------------------------------------------
// a cycle stamp the compiler keeps in place relative to memory operations
__device__ __forceinline__ unsigned long long stamp_here() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}

template <int NT, int L, int TNNZ, int TROWS, int WPE, bool NTL, bool PERM>
__global__ void __launch_bounds__(NT, WPE) k_spmv_pack(CsrDev A, const char *__restrict__ blob,
                                                       const PackDesc *__restrict__ tiles, int ntiles,
                                                       const double *__restrict__ x, double *__restrict__ y,
                                                       unsigned long long *acc) {
    constexpr int BUFB = pack_buf_bytes(TNNZ, TROWS, PERM);
    static_assert(2 * BUFB + 8 * TROWS <= 65536, "LDS-DMA addresses its destination through 16 bits of M0");
    __shared__ __attribute__((aligned(16))) char buf[2][BUFB];
    __shared__ double sw[TROWS];
    unsigned long long a[5] = {0, 0, 0, 0, 0}, c0, c1, c2, c3, c4, c5;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    int tn = t + gridDim.x;
    PackDesc td = tiles[t];
    PackDesc nd = td;
    if (tn < ntiles) nd = tiles[tn];
    pack_issue<NT, NTL, PERM>(blob, td, buf[0]);
    int b = 0;
    while (true) {
        const int tnn = tn + gridDim.x;
        PackDesc nnd = nd;
        if (tnn < ntiles) nnd = tiles[tnn];                       // scalar load, in flight during this tile
        if (acc) c0 = stamp_here();
        wait_vm0_barrier();                                       // A
        if (acc) c1 = stamp_here();
        if (td.flags & 4) {
            pack_long_row<NT>(A, PlainX{x}, td, buf[b], sw);
            if (tn < ntiles) pack_issue<NT, NTL, PERM>(blob, nd, buf[b ^ 1]);
        } else {
            pack_products<NT, PlainX, TNNZ, PERM>(A, PlainX{x}, td, buf[b]);       // B, C
            if (acc) c2 = stamp_here();
            if (tn < ntiles) pack_issue<NT, NTL, PERM>(blob, nd, buf[b ^ 1]);      // D
            if (acc) c3 = stamp_here();
            pack_sums<NT, L, PERM>(td, buf[b], sw);                     // E
        }
        if (acc) c4 = stamp_here();
        wait_lds_barrier();                                       // F
        for (int r = threadIdx.x; r < td.nrows; r += NT) y[td.r0 + r] = sw[r];
        if (acc) {
            c5 = stamp_here();
            a[0] += c1 - c0;
            a[1] += c2 - c1;
            a[2] += c3 - c2;
            a[3] += c4 - c3;
            a[4] += c5 - c4;
        }
        if (tn >= ntiles) break;
        t = tn;
        tn = tnn;
        td = nd;
        nd = nnd;
        b ^= 1;
    }
    if (acc && threadIdx.x == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(acc + i, a[i]);
}


// three LDS buffers: tile t is multiplied and summed while tile t+1's x values are in flight to registers and tile t+2's
// blob is in flight to LDS
template <int NT, int L, int TNNZ, int TROWS, int WPE, bool NTL>
__global__ void __launch_bounds__(NT, WPE) k_spmv_pack3(CsrDev A, const char *__restrict__ blob,
                                                        const PackDesc *__restrict__ tiles, int ntiles,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        unsigned long long *acc) {
    constexpr int BUFB = pack_buf_bytes(TNNZ, TROWS);
    static_assert(3 * BUFB + 8 * TROWS <= 65536, "LDS-DMA addresses its destination through 16 bits of M0");
    __shared__ __attribute__((aligned(16))) char buf[3][BUFB];
    __shared__ double sw[TROWS];
    unsigned long long a[5] = {0, 0, 0, 0, 0}, c0, c1, c2, c3, c4, c5;
    int t = blockIdx.x;
    if (t >= ntiles) return;
    const int stride = gridDim.x;
    PackDesc td = tiles[t], nd = td, nnd = td;
    if (t + stride < ntiles) nd = tiles[t + stride];
    if (t + 2 * stride < ntiles) nnd = tiles[t + 2 * stride];
    pack_issue<NT, NTL>(blob, td, buf[0]);
    if (t + stride < ntiles) pack_issue<NT, NTL>(blob, nd, buf[1]);
    wait_vm0_barrier();
    PackX<NT, TNNZ> G;
    pack_gather<NT, PlainX, TNNZ>(A, PlainX{x}, td, buf[0], G);
    int b = 0;
    while (true) {
        const int b1 = b == 2 ? 0 : b + 1, b2 = b1 == 2 ? 0 : b1 + 1;
        PackDesc n3 = nnd;
        if (t + 3 * stride < ntiles) n3 = tiles[t + 3 * stride];      // scalar load, in flight during this tile
        if (acc) c0 = stamp_here();
        wait_vm0_barrier();                                       // A: x values of tile t here, blob of tile t+1 landed
        if (acc) c1 = stamp_here();
        pack_multiply<NT, TNNZ>(td, buf[b], G);
        if (t + stride < ntiles) pack_gather<NT, PlainX, TNNZ>(A, PlainX{x}, nd, buf[b1], G);
        wait_lds_barrier();                                       // C
        if (acc) c2 = stamp_here();
        if (t + 2 * stride < ntiles) pack_issue<NT, NTL>(blob, nnd, buf[b2]);      // D
        if (acc) c3 = stamp_here();
        pack_sums<NT, L>(td, buf[b], sw);                         // E
        if (acc) c4 = stamp_here();
        wait_lds_barrier();                                       // F
        for (int r = threadIdx.x; r < td.nrows; r += NT) y[td.r0 + r] = sw[r];
        if (acc) {
            c5 = stamp_here();
            a[0] += c1 - c0;
            a[1] += c2 - c1;
            a[2] += c3 - c2;
            a[3] += c4 - c3;
            a[4] += c5 - c4;
        }
        t += stride;
        if (t >= ntiles) break;
        td = nd;
        nd = nnd;
        nnd = n3;
        b = b1;
    }
    if (acc && threadIdx.x == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(acc + i, a[i]);
}

// one workgroup per tile: the tile's pieces of the CSR / record arrays -> its blob
template <bool PERM>
__global__ void __launch_bounds__(256) k_pack_tiles(CsrDev A, const PackDesc *__restrict__ tiles, int ntiles,
                                                    char *__restrict__ blob, const uint16_t *__restrict__ pdst,
                                                    const uint16_t *__restrict__ cdst) {
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const PackDesc pd = tiles[t];
        if (pd.flags & 4) continue;
        const bool blk = pd.flags & 1;
        const PackGeo g = pack_geo<PERM>(pd.nrows, pd.nnode, pd.npe, pd.n, blk);
        int32_t *I = reinterpret_cast<int32_t *>(blob + pd.off16 * 16);
        uint16_t *PD = reinterpret_cast<uint16_t *>(I) + 2 * g.i_pdst, *CD = reinterpret_cast<uint16_t *>(I) + 2 * g.i_cdst;
        double *D = reinterpret_cast<double *>(blob + pd.off16 * 16 + 4 * g.nints);
        const int64_t base = A.rowptr[pd.r0];
        const int q0 = blk ? node_of_row(A, pd.r0) : 0;
        const int64_t pbase = blk ? A.prow[q0] : 0;
        for (int r = threadIdx.x; r <= pd.nrows; r += blockDim.x) I[r] = 2 * g.npeA + (int32_t)(A.rowptr[pd.r0 + r] - base);
        if (blk)
            for (int q = threadIdx.x; q <= pd.nnode; q += blockDim.x) I[g.i_prp + q] = (int32_t)(A.prow[q0 + q] - pbase);
        for (int e = threadIdx.x; e < pd.npe; e += blockDim.x) {
            I[g.i_pcol + e] = A.pcol[pbase + e];
            const double2 kc = A.pkc[pbase + e];
            D[e] = kc.x;
            D[g.npeA + e] = kc.y;
            if (PERM) PD[e] = pdst[pbase + e];
        }
        for (int k = threadIdx.x; k < pd.n; k += blockDim.x) {
            I[g.i_col + k] = A.col[base + k];
            D[2 * g.npeA + k] = A.val[base + k];
            if (PERM) CD[k] = cdst[base + k];
        }
    }
}

struct PackTiles {
    PackDesc *d = nullptr;
    char *blob = nullptr;
    int n = 0;
    size_t bytes = 0;
};
static std::map<std::pair<const void *, std::pair<int, int>>, PackTiles> g_packs;

static int sorted_streams(const npg_csr *A, const std::vector<std::pair<int64_t, int>> &rec, const std::vector<std::pair<int64_t, int>> &ent,
                          CsrDev *Av, uint16_t **pdst_out, uint16_t **cdst_out);

template <int NT, int L, int TNNZ, int TROWS, int WPE, bool NTL, bool PERM = false, int DEPTH = 2>
static int run_pack(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    npg_ctx *ctx = A->ctx;
    auto key = std::make_pair((const void *)A, std::make_pair(TNNZ + (PERM ? 1 : 0), TROWS));
    CsrDev Av = csr_view(A);
    if (!g_packs.count(key)) {
        std::vector<int32_t> tp;
        int rc = tile_boundaries(A, TNNZ, tp, TROWS);
        if (rc) return rc;
        const int64_t *rp = A->h_rowptr.data();
        const int64_t nf3 = 3 * (int64_t)A->nfull, nbr = A->block_rows();
        auto node = [&](int64_t r) { return r < nf3 ? r / 3 : A->nfull + (r - nf3) / 2; };
        std::vector<PackDesc> td(tp.size() - 1);
        int64_t off16 = 0;
        int nlong = 0;
        for (size_t t = 0; t + 1 < tp.size(); ++t) {
            const int64_t r0 = tp[t], r1 = tp[t + 1];
            PackDesc &q = td[t];
            q.off16 = off16;
            q.r0 = (int32_t)r0;
            q.nrows = (int32_t)(r1 - r0);
            q.n = (int32_t)(rp[r1] - rp[r0]);
            q.npe = 0;
            q.nnode = 0;
            q.flags = 0;
            if (r0 < nbr) {
                q.flags = 1 | (r0 < nf3 ? 2 : 0);
                q.nnode = (int32_t)(node(r1) - node(r0));
                q.npe = (int32_t)(A->h_prow[node(r1)] - A->h_prow[node(r0)]);
            }
            const PackGeo g = pack_geo<PERM>(q.nrows, q.nnode, q.npe, q.n, q.flags & 1);
            if (4 * g.nints + 8 * (g.ndbl + ((q.flags & 2) ? g.npeA : 0)) > pack_buf_bytes(TNNZ, TROWS, PERM)) {
                NPG_REQUIRE(q.nrows == 1 && !(q.flags & 1), "run_pack: a tile does not fit its LDS buffer");
                q.flags = 4;
                ++nlong;
                continue;
            }
            off16 += (4 * (int64_t)g.nints + 8 * (int64_t)g.ndbl) / 16;
        }
        PackTiles v;
        v.n = (int)td.size();
        v.bytes = (size_t)off16 * 16;
        NPG_HIP(hipMalloc((void **)&v.d, td.size() * sizeof(PackDesc)));
        NPG_HIP(hipMemcpy(v.d, td.data(), td.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
        NPG_HIP(hipMalloc((void **)&v.blob, v.bytes + 64));
        NPG_HIP(hipMemsetAsync(v.blob, 0, v.bytes + 64, ctx->stream));
        uint16_t *pdst = nullptr, *cdst = nullptr;
        CsrDev Ap = Av;
        if (PERM) {
            std::vector<std::pair<int64_t, int>> rec, ent;      // (start, count) of every tile's record / CSR stream
            for (const PackDesc &q : td) {
                const int64_t q0 = (q.flags & 1) ? node(q.r0) : 0;
                rec.emplace_back((q.flags & 1) ? A->h_prow[q0] : 0, (q.flags & 4) ? 0 : q.npe);
                ent.emplace_back(rp[q.r0], (q.flags & 4) ? 0 : q.n);
            }
            rc = sorted_streams(A, rec, ent, &Ap, &pdst, &cdst);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_pack_tiles<PERM>, dim3(std::min(v.n, 4096)), dim3(256), 0, ctx->stream, Ap, v.d, v.n, v.blob, pdst,
                           cdst);
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "  packed %d tiles (<= %d slots, <= %d rows; %d long rows left in CSR): %.1f MB\n", v.n, TNNZ, TROWS, nlong,
                v.bytes / 1e6);
        g_packs[key] = v;
    }
    const PackTiles t = g_packs[key];
    const int grid = std::max(1, std::min(t.n, bpc * ctx->num_cu));
    auto go = [&](unsigned long long *acc) {
        if constexpr (DEPTH == 3)
            hipLaunchKernelGGL((k_spmv_pack3<NT, L, TNNZ, TROWS, WPE, NTL>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.blob,
                               t.d, t.n, x, y, acc);
        else
            hipLaunchKernelGGL((k_spmv_pack<NT, L, TNNZ, TROWS, WPE, NTL, PERM>), dim3(grid), dim3(NT), 0, ctx->stream, Av, t.blob,
                               t.d, t.n, x, y, acc);
    };
    if (getenv("NPG_DMA_PHASES")) {
        unsigned long long *acc, h[5];
        NPG_HIP(hipMalloc((void **)&acc, sizeof h));
        NPG_HIP(hipMemset(acc, 0, sizeof h));
        go(acc);
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        NPG_HIP(hipMemcpy(h, acc, sizeof h, hipMemcpyDeviceToHost));
        NPG_HIP(hipFree(acc));
        fprintf(stderr, "  cycles per tile (%d tiles): A wait %.0f | B+C gather %.0f | D issue %.0f | E sums %.0f | F out %.0f\n", t.n,
                (double)h[0] / t.n, (double)h[1] / t.n, (double)h[2] / t.n, (double)h[3] / t.n, (double)h[4] / t.n);
    }
    for (int i = 0; i < 2; ++i) go(nullptr);
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go(nullptr);
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- column-sorted tiles (experiment): inside every tile the record stream and the CSR stream are sorted by column, and
// every entry carries the product slot it belongs to (uint16), so that adjacent lanes gather adjacent (or the same)
// components of x: the vector L1 (94 % busy in k_spmv, 83 % of its tag lookups are gathers: profiles/r02_spmv_pmc.txt)
// sees a quarter of the cache lines per gather instruction.  Products land in the same LDS slots as in spmv_tile, so
// the segmented sums - and the result, bit for bit - are unchanged.
struct PermDev {
    const uint16_t *pdst, *cdst;      // product slot (tile-local record index / CSR entry index) of a stream position
};

template <int NT, int L, class XF, int TNNZ = kTileNnz, int U2 = 4>
__device__ __forceinline__ void spmv_tile_perm(const CsrDev &A, const PermDev &P, const XF x, const TileDesc &td,
                                               TileLdsT<TNNZ> &t, double *__restrict__ out) {
    const int r0 = td.r0, nrows = td.nrows, r1 = r0 + nrows;
    const int64_t base = td.base;
    const int n = td.n;
    const bool blk = r0 < block_rows(A);
    const bool full = r0 < 3 * A.nfull;
    const int ncomp = full ? 3 : 2;
    const int npe = td.npe;
    const int64_t pbase = td.pbase;
    int nnode = 0, q0 = 0;
    if (blk) {
        q0 = node_of_row(A, r0);
        nnode = node_of_row(A, r1) - q0;
    }
    const int64_t abase = base & ~1LL;
    const int off = (int)(base - abase);
    const int total = n + off;
    const int slot0 = blk ? ncomp * npe : 0;
    for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(A.rowptr[r0 + r] - base) + off + slot0;
    if (blk)
        for (int q = threadIdx.x; q <= nnode; q += NT) t.prp[q] = (int32_t)(A.prow[q0 + q] - pbase);
    constexpr int UP = U2 > 3 ? 3 : U2;
    for (int e0 = threadIdx.x; e0 < npe; e0 += UP * NT) {
        int32_t c[UP];
        int d[UP];
        double2 kc[UP], xx[UP];
        double zz[UP];
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int e = e0 + u * NT;
            if (e < npe) {
                c[u] = __builtin_nontemporal_load(A.pcol + pbase + e);
                d[u] = __builtin_nontemporal_load(P.pdst + pbase + e);
                const double *p = reinterpret_cast<const double *>(A.pkc + pbase + e);
                kc[u].x = __builtin_nontemporal_load(p);
                kc[u].y = __builtin_nontemporal_load(p + 1);
            } else {
                c[u] = 0;
                d[u] = 0;
                kc[u] = make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int cf = c[u] < A.nfull ? c[u] : A.nfull;
            const int xo = 2 * c[u] + cf;
            xx[u] = x.two(xo);
            zz[u] = (full && c[u] < A.nfull) ? x.third(xo + 2) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int e = e0 + u * NT;
            if (e < npe) {
                t.prod[d[u]] = kc[u].x * xx[u].x + kc[u].y * xx[u].y;
                t.prod[npe + d[u]] = kc[u].x * xx[u].y - kc[u].y * xx[u].x;
                if (full) t.prod[2 * npe + d[u]] = kc[u].x * zz[u];
            }
        }
    }
    for (int k0 = 2 * threadIdx.x; k0 < total; k0 += 2 * NT * U2) {
        int2 c[U2];
        int d0[U2], d1[U2];
        double2 v[U2];
        double xa[U2], xb[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int k = k0 + u * 2 * NT;
            if (k < total && abase + k + 1 < A.nnz) {
                const long long cc = __builtin_nontemporal_load(reinterpret_cast<const long long *>(A.col + abase + k));
                c[u] = make_int2((int)(cc & 0xffffffffLL), (int)(cc >> 32));
                const unsigned dd = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(P.cdst + abase + k));
                d0[u] = (int)(dd & 0xffffu);
                d1[u] = (int)(dd >> 16);
                v[u].x = __builtin_nontemporal_load(A.val + abase + k);
                v[u].y = __builtin_nontemporal_load(A.val + abase + k + 1);
            } else if (k < total && abase + k < A.nnz) {
                c[u] = make_int2(A.col[abase + k], 0);
                d0[u] = P.cdst[abase + k];
                d1[u] = 0;
                v[u] = make_double2(A.val[abase + k], 0.0);
            } else {
                c[u] = make_int2(0, 0);
                d0[u] = d1[u] = 0;
                v[u] = make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            xa[u] = x(c[u].x);
            xb[u] = x(c[u].y);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int k = k0 + u * 2 * NT;
            // (an entry in front of the tile's first one belongs to the previous tile: its slot is a pad)
            if (k < total) t.prod[slot0 + (k >= off ? off + d0[u] : 0)] = (k >= off) ? v[u].x * xa[u] : 0.0;
            if (k + 1 < total) t.prod[slot0 + off + d1[u]] = v[u].y * xb[u];
        }
    }
    __syncthreads();
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    for (int r = g; r < nrows; r += NT / L) {
        double s = 0.0;
        const int e = t.rp[r + 1];
        for (int k = t.rp[r] + 2 * l; k < e; k += 2 * L) {
            const double a = t.prod[k], b = t.prod[k + 1];
            s += a + (k + 1 < e ? b : 0.0);
        }
        if (blk) {
            const int q = (r * (full ? 21846 : 32768)) >> 16;
            const int pb = (r - q * ncomp) * npe, pe = pb + t.prp[q + 1];
            for (int k = pb + t.prp[q] + 2 * l; k < pe; k += 2 * L) {
                const double a = t.prod[k], b = t.prod[k + 1];
                s += a + (k + 1 < pe ? b : 0.0);
            }
        }
        s = group_sum_dpp<L>(s);
        if (l == 0) out[r] = s;
    }
    __syncthreads();
}

template <int L>
__global__ void __launch_bounds__(512, 6) k_spmv_perm(CsrDev A, PermDev P, const TileDesc *__restrict__ tile_ptr, int ntiles,
                                                      const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    TileDesc td = tile_ptr[t];
    while (true) {
        const int tn = t + gridDim.x;
        TileDesc nd = td;
        if (tn < ntiles) nd = tile_ptr[tn];
        spmv_tile_perm<512, L>(A, P, PlainX{x}, td, tl, sw);
        for (int r = threadIdx.x; r < td.nrows; r += 512) y[td.r0 + r] = sw[r];
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
}

struct PermMat {
    int32_t *pcol = nullptr, *col = nullptr;
    double *pkc = nullptr, *val = nullptr;
    uint16_t *pdst = nullptr, *cdst = nullptr;
};
static std::map<const void *, PermMat> g_perm;

static int run_perm(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms, bool sorted) {
    npg_ctx *ctx = A->ctx;
    auto key = (const void *)((const char *)A + (sorted ? 1 : 0));
    if (!g_perm.count(key)) {
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        const int64_t nrec = A->nnode() ? A->h_prow[A->nnode()] : 0, nz = A->rnnz;
        std::vector<TileDesc> td((size_t)A->ntiles);
        NPG_HIP(hipMemcpy(td.data(), A->tile_ptr, td.size() * sizeof(TileDesc), hipMemcpyDeviceToHost));
        std::vector<int32_t> pcol((size_t)nrec), col((size_t)nz);
        std::vector<double> pkc((size_t)2 * nrec), val((size_t)nz);
        if (nrec) {
            NPG_HIP(hipMemcpy(pcol.data(), A->pcol, pcol.size() * 4, hipMemcpyDeviceToHost));
            NPG_HIP(hipMemcpy(pkc.data(), A->pkc, pkc.size() * 8, hipMemcpyDeviceToHost));
        }
        NPG_HIP(hipMemcpy(col.data(), A->col, col.size() * 4, hipMemcpyDeviceToHost));
        NPG_HIP(hipMemcpy(val.data(), A->val, val.size() * 8, hipMemcpyDeviceToHost));
        std::vector<int32_t> pcol2(pcol.size()), col2(col.size());
        std::vector<double> pkc2(pkc.size()), val2(val.size());
        std::vector<uint16_t> pdst(pcol.size() + 8), cdst(col.size() + 8);
        std::vector<int> idx;
        for (const TileDesc &q : td) {
            idx.resize((size_t)q.npe);
            for (int e = 0; e < q.npe; ++e) idx[e] = e;
            if (sorted) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return pcol[q.pbase + a] < pcol[q.pbase + b]; });
            for (int e = 0; e < q.npe; ++e) {
                pcol2[q.pbase + e] = pcol[q.pbase + idx[e]];
                pkc2[2 * (q.pbase + e)] = pkc[2 * (q.pbase + idx[e])];
                pkc2[2 * (q.pbase + e) + 1] = pkc[2 * (q.pbase + idx[e]) + 1];
                pdst[q.pbase + e] = (uint16_t)idx[e];
            }
            idx.resize((size_t)q.n);
            for (int k = 0; k < q.n; ++k) idx[k] = k;
            if (sorted) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return col[q.base + a] < col[q.base + b]; });
            for (int k = 0; k < q.n; ++k) {
                col2[q.base + k] = col[q.base + idx[k]];
                val2[q.base + k] = val[q.base + idx[k]];
                cdst[q.base + k] = (uint16_t)idx[k];
            }
        }
        PermMat m;
        auto up = [&](auto **d, const auto &h) {
            NPG_HIP(hipMalloc((void **)d, std::max<size_t>(16, h.size() * sizeof(h[0]) + 16)));
            if (!h.empty()) NPG_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(h[0]), hipMemcpyHostToDevice));
            return 0;
        };
        if (up(&m.pcol, pcol2) || up(&m.col, col2) || up(&m.pkc, pkc2) || up(&m.val, val2) || up(&m.pdst, pdst) ||
            up(&m.cdst, cdst))
            return NPG_EHIP;
        g_perm[key] = m;
    }
    const PermMat m = g_perm[key];
    CsrDev Av = csr_view(A);
    Av.pcol = m.pcol;
    Av.col = m.col;
    Av.pkc = reinterpret_cast<const double2 *>(m.pkc);
    Av.val = m.val;
    const PermDev P{m.pdst, m.cdst};
    const int grid = std::max(1, std::min<int>(A->ntiles, bpc * ctx->num_cu));
    auto go = [&]() {
        hipLaunchKernelGGL(k_spmv_perm<8>, dim3(grid), dim3(512), 0, ctx->stream, Av, P, A->tile_ptr, A->ntiles, x, y);
    };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// every tile's record stream and CSR stream sorted by column; *Av gets the permuted arrays, pdst / cdst the tile-local
// index every stream position came from (= its product slot)
static int sorted_streams(const npg_csr *A, const std::vector<std::pair<int64_t, int>> &rec, const std::vector<std::pair<int64_t, int>> &ent,
                          CsrDev *Av, uint16_t **pdst_out, uint16_t **cdst_out) {
    npg_ctx *ctx = A->ctx;
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t nrec = A->nnode() ? A->h_prow[A->nnode()] : 0, nz = A->rnnz;
    std::vector<int32_t> pcol((size_t)nrec), col((size_t)nz);
    std::vector<double> pkc((size_t)2 * nrec), val((size_t)nz);
    if (nrec) {
        NPG_HIP(hipMemcpy(pcol.data(), A->pcol, pcol.size() * 4, hipMemcpyDeviceToHost));
        NPG_HIP(hipMemcpy(pkc.data(), A->pkc, pkc.size() * 8, hipMemcpyDeviceToHost));
    }
    NPG_HIP(hipMemcpy(col.data(), A->col, col.size() * 4, hipMemcpyDeviceToHost));
    NPG_HIP(hipMemcpy(val.data(), A->val, val.size() * 8, hipMemcpyDeviceToHost));
    std::vector<int32_t> pcol2(pcol), col2(col);
    std::vector<double> pkc2(pkc), val2(val);
    std::vector<uint16_t> pdst(pcol.size() + 8), cdst(col.size() + 8);
    std::vector<int> idx;
    for (size_t t = 0; t < rec.size(); ++t) {
        const int64_t pb = rec[t].first, b = ent[t].first;
        const int npe = rec[t].second, n = ent[t].second;
        idx.resize((size_t)npe);
        for (int e = 0; e < npe; ++e) idx[e] = e;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return pcol[pb + a] < pcol[pb + c]; });
        for (int e = 0; e < npe; ++e) {
            pcol2[pb + e] = pcol[pb + idx[e]];
            pkc2[2 * (pb + e)] = pkc[2 * (pb + idx[e])];
            pkc2[2 * (pb + e) + 1] = pkc[2 * (pb + idx[e]) + 1];
            pdst[pb + e] = (uint16_t)idx[e];
        }
        idx.resize((size_t)n);
        for (int k = 0; k < n; ++k) idx[k] = k;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return col[b + a] < col[b + c]; });
        for (int k = 0; k < n; ++k) {
            col2[b + k] = col[b + idx[k]];
            val2[b + k] = val[b + idx[k]];
            cdst[b + k] = (uint16_t)idx[k];
        }
    }
    int32_t *dpcol, *dcol;
    double *dpkc, *dval;
    NPG_HIP(hipMalloc((void **)&dpcol, pcol2.size() * 4 + 16));
    NPG_HIP(hipMalloc((void **)&dcol, col2.size() * 4 + 16));
    NPG_HIP(hipMalloc((void **)&dpkc, pkc2.size() * 8 + 16));
    NPG_HIP(hipMalloc((void **)&dval, val2.size() * 8 + 16));
    NPG_HIP(hipMalloc((void **)pdst_out, pdst.size() * 2 + 16));
    NPG_HIP(hipMalloc((void **)cdst_out, cdst.size() * 2 + 16));
    if (nrec) {
        NPG_HIP(hipMemcpy(dpcol, pcol2.data(), pcol2.size() * 4, hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(dpkc, pkc2.data(), pkc2.size() * 8, hipMemcpyHostToDevice));
    }
    NPG_HIP(hipMemcpy(dcol, col2.data(), col2.size() * 4, hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(dval, val2.data(), val2.size() * 8, hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(*pdst_out, pdst.data(), pdst.size() * 2, hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(*cdst_out, cdst.data(), cdst.size() * 2, hipMemcpyHostToDevice));
    Av->pcol = dpcol;
    Av->col = dcol;
    Av->pkc = reinterpret_cast<const double2 *>(dpkc);
    Av->val = dval;
    return NPG_OK;
}

// ---- CU-local tile queues (experiment): the workgroups resident on ONE CU take ADJACENT tiles from that CU's own
// contiguous chunk of the tile sequence, so that they share their x window in the CU's 32 KiB L1 and a workgroup's next
// tile reuses most of the lines of its last one.  (Round-robin dealing gives a CU three unrelated windows of ~19 KiB.)
// A workgroup finds its CU from HW_ID / XCC_ID, the first one to arrive on a CU claims the next chunk; chunks are
// balanced by stored bytes; a workgroup whose chunk is empty steals from the following chunks (every wave reaches the
// exit: the queues only ever advance).
struct CuQueues {
    int *tab;          // [2048] physical CU key -> chunk + 1 (0: not claimed yet)
    int *nclaimed;     // chunks claimed so far
    int *next;         // [nchunk] next tile of the chunk
    const int *end;    // [nchunk] end of the chunk
    int nchunk;
};

__device__ __forceinline__ int my_cu_key() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID
    return (int)(((xcc & 7u) << 8) | ((hw >> 8) & 0xffu));                          // XCC, SE, SH, CU
}

template <int L>
__global__ void __launch_bounds__(512, 6) k_spmv_cuq(CsrDev A, const TileDesc *__restrict__ tile_ptr, int ntiles, CuQueues Q,
                                                     const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    __shared__ int s_chunk, s_tile;
    if (threadIdx.x == 0) {
        const int key = my_cu_key();
        int c = atomicAdd(&Q.tab[key], 0);
        if (c == 0) {
            if (atomicCAS(&Q.tab[key], 0, -1) == 0) {
                c = atomicAdd(Q.nclaimed, 1) % Q.nchunk + 1;
                atomicExch(&Q.tab[key], c);
            } else {
                while ((c = atomicAdd(&Q.tab[key], 0)) <= 0) __builtin_amdgcn_s_sleep(2);
            }
        } else if (c < 0) {
            while ((c = atomicAdd(&Q.tab[key], 0)) <= 0) __builtin_amdgcn_s_sleep(2);
        }
        s_chunk = c - 1;
    }
    __syncthreads();
    const int home = s_chunk;
    auto claim = [&]() {          // thread 0: next tile for this workgroup, -1 when every queue is empty
        for (int k = 0; k < Q.nchunk; ++k) {
            const int c = home + k < Q.nchunk ? home + k : home + k - Q.nchunk;
            if (atomicAdd(&Q.next[c], 0) >= Q.end[c]) continue;
            const int t = atomicAdd(&Q.next[c], 1);
            if (t < Q.end[c]) return t;
        }
        return -1;
    };
    __shared__ TileDesc s_td;
    if (threadIdx.x == 0) {
        s_tile = claim();
        if (s_tile >= 0) s_td = tile_ptr[s_tile];
    }
    __syncthreads();
    int t = s_tile;
    while (t >= 0) {
        const TileDesc td = s_td;
        __syncthreads();                       // everyone has read s_tile / s_td
        if (threadIdx.x == 0) {                // the next tile and its descriptor, fetched while this one is worked on
            s_tile = claim();
            if (s_tile >= 0) s_td = tile_ptr[s_tile];
        }
        spmv_tile<512, L>(A, PlainX{x}, td, tl, sw);
        for (int r = threadIdx.x; r < td.nrows; r += 512) y[td.r0 + r] = sw[r];
        __syncthreads();
        t = s_tile;
    }
}

__global__ void k_cuq_reset(CuQueues Q, const int *__restrict__ begin) {
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) Q.tab[i] = 0;
    for (int i = threadIdx.x; i < Q.nchunk; i += blockDim.x) Q.next[i] = begin[i];
    if (threadIdx.x == 0) *Q.nclaimed = 0;
}

static int run_cuq(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    npg_ctx *ctx = A->ctx;
    static std::map<const void *, std::pair<CuQueues, int *>> cache;
    if (!cache.count(A)) {
        std::vector<TileDesc> td((size_t)A->ntiles);
        NPG_HIP(hipMemcpy(td.data(), A->tile_ptr, td.size() * sizeof(TileDesc), hipMemcpyDeviceToHost));
        const int nchunk = ctx->num_cu;
        std::vector<double> cum(td.size() + 1, 0.0);
        for (size_t t = 0; t < td.size(); ++t) cum[t + 1] = cum[t] + 20.0 * td[t].npe + 12.0 * td[t].n + 12.0 * td[t].nrows;
        std::vector<int> begin(nchunk), end(nchunk);
        int t0 = 0;
        for (int c = 0; c < nchunk; ++c) {
            const double target = cum.back() * (c + 1) / nchunk;
            int t1 = t0;
            while (t1 < (int)td.size() && cum[t1 + 1] <= target + 1e-9) ++t1;
            if (c == nchunk - 1) t1 = (int)td.size();
            begin[c] = t0;
            end[c] = t1;
            t0 = t1;
        }
        CuQueues Q;
        int *dbegin, *dend;
        NPG_HIP(hipMalloc((void **)&Q.tab, 2048 * sizeof(int)));
        NPG_HIP(hipMalloc((void **)&Q.nclaimed, sizeof(int)));
        NPG_HIP(hipMalloc((void **)&Q.next, nchunk * sizeof(int)));
        NPG_HIP(hipMalloc((void **)&dend, nchunk * sizeof(int)));
        NPG_HIP(hipMalloc((void **)&dbegin, nchunk * sizeof(int)));
        NPG_HIP(hipMemcpy(dend, end.data(), nchunk * sizeof(int), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(dbegin, begin.data(), nchunk * sizeof(int), hipMemcpyHostToDevice));
        Q.end = dend;
        Q.nchunk = nchunk;
        cache[A] = std::make_pair(Q, dbegin);
    }
    const CuQueues Q = cache[A].first;
    const int *dbegin = cache[A].second;
    const int grid = std::max(1, std::min<int>(A->ntiles, bpc * ctx->num_cu));
    const CsrDev Av = csr_view(A);
    auto go = [&]() {
        hipLaunchKernelGGL(k_cuq_reset, dim3(1), dim3(256), 0, ctx->stream, Q, dbegin);
        hipLaunchKernelGGL(k_spmv_cuq<8>, dim3(grid), dim3(512), 0, ctx->stream, Av, A->tile_ptr, A->ntiles, Q, x, y);
    };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// the same idea with the global order kept: at step k the whole chip works on tiles [768 k, 768 (k + 1)) as it does
// with round-robin dealing (one moving window over the matrix), but the three workgroups of a CU take three ADJACENT
// tiles of it.  Needs exactly 3 workgroups on each of the CUs (grid = 3 x CUs, 51 KB of LDS each).
template <int L>
__global__ void __launch_bounds__(512, 6) k_spmv_cuadj(CsrDev A, const TileDesc *__restrict__ tile_ptr, int ntiles, CuQueues Q,
                                                       const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    __shared__ int s_lane;
    if (threadIdx.x == 0) {
        const int key = my_cu_key();
        int c = atomicAdd(&Q.tab[key], 0);
        if (c == 0 && atomicCAS(&Q.tab[key], 0, -1) == 0) {
            c = atomicAdd(Q.nclaimed, 1) + 1;
            atomicExch(&Q.tab[key], c);
        } else {
            while ((c = atomicAdd(&Q.tab[key], 0)) <= 0) __builtin_amdgcn_s_sleep(2);
        }
        const int slot = atomicAdd(&Q.next[c - 1], 1);          // arrival order on this CU
        s_lane = (slot < 3 && c - 1 < Q.nchunk) ? 3 * (c - 1) + slot : -1;
    }
    __syncthreads();
    const int lane = s_lane, stride = 3 * Q.nchunk;
    if (lane < 0) return;
    int t = lane;
    if (t >= ntiles) return;
    TileDesc td = tile_ptr[t];
    while (true) {
        const int tn = t + stride;
        TileDesc nd = td;
        if (tn < ntiles) nd = tile_ptr[tn];
        spmv_tile<512, L>(A, PlainX{x}, td, tl, sw);
        for (int r = threadIdx.x; r < td.nrows; r += 512) y[td.r0 + r] = sw[r];
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
}

__global__ void k_cuadj_reset(CuQueues Q) {
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) Q.tab[i] = 0;
    for (int i = threadIdx.x; i < Q.nchunk; i += blockDim.x) Q.next[i] = 0;
    if (threadIdx.x == 0) *Q.nclaimed = 0;
}

static int run_cuadj(const npg_csr *A, const double *x, double *y, int reps, double *ms) {
    npg_ctx *ctx = A->ctx;
    static CuQueues Q{};
    if (!Q.tab) {
        NPG_HIP(hipMalloc((void **)&Q.tab, 2048 * sizeof(int)));
        NPG_HIP(hipMalloc((void **)&Q.nclaimed, sizeof(int)));
        NPG_HIP(hipMalloc((void **)&Q.next, 2048 * sizeof(int)));
        Q.end = nullptr;
        Q.nchunk = ctx->num_cu;
    }
    const CsrDev Av = csr_view(A);
    auto go = [&]() {
        hipLaunchKernelGGL(k_cuadj_reset, dim3(1), dim3(256), 0, ctx->stream, Q);
        hipLaunchKernelGGL(k_spmv_cuadj<8>, dim3(3 * ctx->num_cu), dim3(512), 0, ctx->stream, Av, A->tile_ptr, A->ntiles, Q, x, y);
    };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- diagnostic: the same gather instructions on a SMALL window of x (column index wrapped): prices the gathers' cache
// misses apart from their instructions (results are meaningless)
template <int MASK>
struct WrapX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return x[c & MASK]; }
    __device__ __forceinline__ double2 two(int i) const {
        double2 r;
        __builtin_memcpy(&r, x + (i & MASK & ~1), sizeof r);
        return r;
    }
    __device__ __forceinline__ double third(int i) const { return x[i & MASK]; }
};

template <int L, class XF>
__global__ void __launch_bounds__(512, 6) k_spmv_xf(CsrDev A, const TileDesc *__restrict__ tile_ptr, int ntiles,
                                                    const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    TileDesc td = tile_ptr[t];
    while (true) {
        const int tn = t + gridDim.x;
        TileDesc nd = td;
        if (tn < ntiles) nd = tile_ptr[tn];
        spmv_tile<512, L>(A, XF{x}, td, tl, sw);
        for (int r = threadIdx.x; r < td.nrows; r += 512) y[td.r0 + r] = sw[r];
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
}

template <class XF>
static int run_xf(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min<int>(A->ntiles, bpc * ctx->num_cu));
    const CsrDev Av = csr_view(A);
    auto go = [&]() { hipLaunchKernelGGL((k_spmv_xf<8, XF>), dim3(grid), dim3(512), 0, ctx->stream, Av, A->tile_ptr, A->ntiles, x, y); };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- diagnostic: the gathers served by LDS reads from a (fake) x window instead of the texture path: what an x window
// staged in LDS could buy at most (results are meaningless; the window is never filled)
struct LdsX {
    const double *w;          // LDS
    __device__ __forceinline__ double operator()(int c) const { return w[c & 511]; }
    __device__ __forceinline__ double2 two(int i) const { return make_double2(w[i & 511], w[(i + 1) & 511]); }
    __device__ __forceinline__ double third(int i) const { return w[i & 511]; }
};

template <int L, int TNNZ, int MODE>
__global__ void __launch_bounds__(512, 6) k_spmv_ldsx(CsrDev A, const int32_t *__restrict__ tile_ptr, int ntiles,
                                                      const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLdsT<TNNZ> tl;
    __shared__ double sw[kTileRows];
    __shared__ double xw[512];
    xw[threadIdx.x] = x[threadIdx.x];
    __syncthreads();
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int r0 = tile_ptr[t], r1 = tile_ptr[t + 1];
        if (MODE == 0)
            spmv_tile<512, L, PlainX, TNNZ, 4>(A, PlainX{x}, r0, r1, tl, sw);
        else if (MODE == 1)
            spmv_tile<512, L, LdsX, TNNZ, 4>(A, LdsX{xw}, r0, r1, tl, sw);
        else
            spmv_tile<512, L, FakeX, TNNZ, 4>(A, FakeX{x}, r0, r1, tl, sw);
        for (int r = threadIdx.x; r < r1 - r0; r += 512) y[r0 + r] = sw[r];
    }
}

template <int MODE>
static int run_ldsx(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms) {
    constexpr int TNNZ = 5120;
    auto key = std::make_pair((const void *)A, TNNZ);
    if (!g_ptiles.count(key)) {
        std::vector<int32_t> tp;
        int rc = tile_boundaries(A, TNNZ, tp);
        if (rc) return rc;
        VarTiles v;
        v.n = (int)tp.size() - 1;
        NPG_HIP(hipMalloc((void **)&v.d, tp.size() * sizeof(int32_t)));
        NPG_HIP(hipMemcpy(v.d, tp.data(), tp.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        g_ptiles[key] = v;
    }
    const VarTiles t = g_ptiles[key];
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min(t.n, bpc * ctx->num_cu));
    const CsrDev Av = csr_view(A);
    auto go = [&]() { hipLaunchKernelGGL((k_spmv_ldsx<8, TNNZ, MODE>), dim3(grid), dim3(512), 0, ctx->stream, Av, t.d, t.n, x, y); };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

// ---- x windows in LDS (experiment): diagnostic 87 says gathers served by LDS reads cost a sixth of gathers on the
// texture path.  Per tile up to 16 contiguous index ranges of x (<= WIN doubles in total; 88 % of the gather accesses of
// A_inversion on bowl3D h = 0.02 at 3072-slot tiles, counted on the host) are copied into LDS by coalesced loads, and the
// tile's column indices are rewritten at build time: bit 31 set = offset into the window (bit 30: the node has a z
// component), else the global index as before.
constexpr int kWin = 2560, kWinRanges = 16, kWinTile = 3072;
struct WinRange {
    int32_t lo;            // first index of x
    uint16_t len, off;     // length, offset into the window
};

template <int NT, int L, int TNNZ, int U2 = 4>
__device__ __forceinline__ void spmv_tile_win(const CsrDev &A, const double *__restrict__ x, const WinRange *__restrict__ wr,
                                              const TileDesc &td, TileLdsT<TNNZ> &t, double *__restrict__ xw,
                                              double *__restrict__ out) {
    const int r0 = td.r0, nrows = td.nrows, r1 = r0 + nrows;
    const int64_t base = td.base;
    const int n = td.n;
    const bool blk = r0 < block_rows(A);
    const bool full = r0 < 3 * A.nfull;
    const int ncomp = full ? 3 : 2;
    const int npe = td.npe;
    const int64_t pbase = td.pbase;
    int nnode = 0, q0 = 0;
    if (blk) {
        q0 = node_of_row(A, r0);
        nnode = node_of_row(A, r1) - q0;
    }
    const int64_t abase = base & ~1LL;
    const int off = (int)(base - abase);
    const int total = n + off;
    const int slot0 = blk ? ncomp * npe : 0;
    for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(A.rowptr[r0 + r] - base) + off + slot0;
    if (blk)
        for (int q = threadIdx.x; q <= nnode; q += NT) t.prp[q] = (int32_t)(A.prow[q0 + q] - pbase);
    // ---- the record stream's loads first (they do not need the window), then the window
    constexpr int UP = (TNNZ / 2 + NT - 1) / NT;      // (x, y)-node tiles: two slots per record
    int32_t c[UP];
    double2 kc[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int e = threadIdx.x + u * NT;
        if (e < npe) {
            c[u] = __builtin_nontemporal_load(A.pcol + pbase + e);
            const double *p = reinterpret_cast<const double *>(A.pkc + pbase + e);
            kc[u].x = __builtin_nontemporal_load(p);
            kc[u].y = __builtin_nontemporal_load(p + 1);
        } else {
            c[u] = 0;
            kc[u] = make_double2(0.0, 0.0);
        }
    }
    {
        const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
        if (g < kWinRanges) {
            const WinRange w = wr[g];
            for (int i = l; i < w.len; i += 32) xw[w.off + i] = x[w.lo + i];
        }
    }
    __syncthreads();
    {
        double2 xx[UP];
        double zz[UP];
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            if (c[u] < 0) {
                const int o = c[u] & 0xffff;
                xx[u] = make_double2(xw[o], xw[o + 1]);
                zz[u] = (c[u] & 0x40000000) ? xw[o + 2] : 0.0;
            } else {
                const int cf = c[u] < A.nfull ? c[u] : A.nfull;
                const int xo = 2 * c[u] + cf;
                double2 r;
                __builtin_memcpy(&r, x + xo, sizeof r);
                xx[u] = r;
                zz[u] = (full && c[u] < A.nfull) ? x[xo + 2] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int e = threadIdx.x + u * NT;
            if (e < npe) {
                t.prod[e] = kc[u].x * xx[u].x + kc[u].y * xx[u].y;
                t.prod[npe + e] = kc[u].x * xx[u].y - kc[u].y * xx[u].x;
                if (full) t.prod[2 * npe + e] = kc[u].x * zz[u];
            }
        }
    }
    for (int k0 = 2 * threadIdx.x; k0 < total; k0 += 2 * NT * U2) {
        int2 cc[U2];
        double2 v[U2];
        double xa[U2], xb[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int k = k0 + u * 2 * NT;
            if (k < total && abase + k + 1 < A.nnz) {
                const long long q = __builtin_nontemporal_load(reinterpret_cast<const long long *>(A.col + abase + k));
                cc[u] = make_int2((int)(q & 0xffffffffLL), (int)(q >> 32));
                v[u].x = __builtin_nontemporal_load(A.val + abase + k);
                v[u].y = __builtin_nontemporal_load(A.val + abase + k + 1);
            } else if (k < total && abase + k < A.nnz) {
                cc[u] = make_int2(A.col[abase + k], 0);
                v[u] = make_double2(A.val[abase + k], 0.0);
            } else {
                cc[u] = make_int2(0, 0);
                v[u] = make_double2(0.0, 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int k = k0 + u * 2 * NT;
            // (the entry in front of the tile's first one belongs to the previous tile: its index is that tile's)
            xa[u] = (k >= off) ? (cc[u].x < 0 ? xw[cc[u].x & 0xffff] : x[cc[u].x]) : 0.0;
            xb[u] = cc[u].y < 0 ? xw[cc[u].y & 0xffff] : x[cc[u].y];
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int k = k0 + u * 2 * NT;
            if (k < total) t.prod[slot0 + k] = (k >= off) ? v[u].x * xa[u] : 0.0;
            if (k + 1 < total) t.prod[slot0 + k + 1] = v[u].y * xb[u];
        }
    }
    __syncthreads();
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    for (int r = g; r < nrows; r += NT / L) {
        double s = 0.0;
        const int e = t.rp[r + 1];
        for (int k = t.rp[r] + 2 * l; k < e; k += 2 * L) {
            const double a = t.prod[k], b = t.prod[k + 1];
            s += a + (k + 1 < e ? b : 0.0);
        }
        if (blk) {
            const int q = (r * (full ? 21846 : 32768)) >> 16;
            const int pb = (r - q * ncomp) * npe, pe = pb + t.prp[q + 1];
            for (int k = pb + t.prp[q] + 2 * l; k < pe; k += 2 * L) {
                const double a = t.prod[k], b = t.prod[k + 1];
                s += a + (k + 1 < pe ? b : 0.0);
            }
        }
        s = group_sum_dpp<L>(s);
        if (l == 0) out[r] = s;
    }
    __syncthreads();
}

template <int L>
__global__ void __launch_bounds__(512, 6) k_spmv_win(CsrDev A, const WinRange *__restrict__ wr, const TileDesc *__restrict__ tile_ptr,
                                                     int ntiles, const double *__restrict__ x, double *__restrict__ y) {
    __shared__ TileLdsT<kWinTile> tl;
    __shared__ double sw[kTileRows];
    __shared__ double xw[kWin + 2];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    TileDesc td = tile_ptr[t];
    while (true) {
        const int tn = t + gridDim.x;
        TileDesc nd = td;
        if (tn < ntiles) nd = tile_ptr[tn];
        spmv_tile_win<512, L, kWinTile>(A, x, wr + (size_t)t * kWinRanges, td, tl, xw, sw);
        for (int r = threadIdx.x; r < td.nrows; r += 512) y[td.r0 + r] = sw[r];
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
}

struct WinMat {
    int32_t *pcol = nullptr, *col = nullptr;
    WinRange *wr = nullptr;
    TileDesc *td = nullptr;
    int ntiles = 0;
};

static int run_win(const npg_csr *A, const double *x, double *y, int bpc, int reps, double *ms, bool windows) {
    npg_ctx *ctx = A->ctx;
    static std::map<std::pair<const void *, bool>, WinMat> cache;
    auto key = std::make_pair((const void *)A, windows);
    if (!cache.count(key)) {
        NPG_HIP(hipStreamSynchronize(ctx->stream));
        std::vector<int32_t> tp;
        int rc = tile_boundaries(A, kWinTile, tp);
        if (rc) return rc;
        const int64_t *rp = A->h_rowptr.data();
        const int64_t nfull = A->nfull, nf3 = 3 * nfull, nbr = A->block_rows();
        auto node = [&](int64_t r) { return r < nf3 ? r / 3 : nfull + (r - nf3) / 2; };
        auto first = [&](int64_t c) { return c < nfull ? 3 * c : nf3 + 2 * (c - nfull); };
        const int nt = (int)tp.size() - 1;
        std::vector<TileDesc> td((size_t)nt);
        const int64_t nrec = A->nnode() ? A->h_prow[A->nnode()] : 0, nz = A->rnnz;
        std::vector<int32_t> pcol((size_t)nrec), col((size_t)nz);
        if (nrec) NPG_HIP(hipMemcpy(pcol.data(), A->pcol, pcol.size() * 4, hipMemcpyDeviceToHost));
        NPG_HIP(hipMemcpy(col.data(), A->col, col.size() * 4, hipMemcpyDeviceToHost));
        std::vector<WinRange> wr((size_t)nt * kWinRanges, WinRange{0, 0, 0});
        double lanes_all = 0, lanes_win = 0, staged = 0;
        std::vector<std::pair<int64_t, int>> need;      // (index of x, lanes)
        for (int t = 0; t < nt; ++t) {
            const int64_t r0 = tp[t], r1 = tp[t + 1];
            TileDesc &q = td[t];
            q.r0 = (int32_t)r0;
            q.nrows = (int32_t)(r1 - r0);
            q.base = rp[r0];
            q.n = (int32_t)(rp[r1] - rp[r0]);
            q.pbase = 0;
            q.npe = 0;
            if (r0 < nbr) {
                q.pbase = A->h_prow[node(r0)];
                q.npe = (int32_t)(A->h_prow[node(r1)] - q.pbase);
            }
            if (!windows) continue;
            need.clear();
            for (int e = 0; e < q.npe; ++e) need.emplace_back(first(pcol[q.pbase + e]), 2);
            for (int k = 0; k < q.n; ++k) need.emplace_back(col[q.base + k], 1);
            std::sort(need.begin(), need.end());
            // clusters of needed indices (gap <= 64), whole nodes
            struct Cl { int64_t lo, hi; double w; };
            std::vector<Cl> cl;
            for (auto &pr : need) {
                const int64_t i0 = pr.first, i1 = i0 + (pr.second == 2 ? (i0 < nf3 ? 3 : 2) : 1);
                if (!cl.empty() && i0 <= cl.back().hi + 64) {
                    cl.back().hi = std::max(cl.back().hi, i1);
                    cl.back().w += pr.second;
                } else
                    cl.push_back(Cl{i0, i1, (double)pr.second});
                lanes_all += pr.second;
            }
            std::sort(cl.begin(), cl.end(), [](const Cl &a, const Cl &b) { return a.w > b.w; });
            int used = 0, nr = 0;
            std::vector<Cl> sel;
            for (auto &c : cl) {
                const int len = (int)(c.hi - c.lo) + ((c.hi - c.lo) & 1);
                if (nr >= kWinRanges || used + len > kWin) continue;
                wr[(size_t)t * kWinRanges + nr] = WinRange{(int32_t)c.lo, (uint16_t)(c.hi - c.lo), (uint16_t)used};
                sel.push_back(Cl{c.lo, c.hi, (double)used});
                used += len;
                ++nr;
                lanes_win += c.w;
            }
            staged += used;
            auto lookup = [&](int64_t i) -> int {
                for (auto &c : sel)
                    if (i >= c.lo && i < c.hi) return (int)c.w + (int)(i - c.lo);
                return -1;
            };
            for (int e = 0; e < q.npe; ++e) {
                const int64_t cn = pcol[q.pbase + e];
                const int o = lookup(first(cn));
                if (o >= 0) pcol[q.pbase + e] = (int32_t)(0x80000000u | (cn < nfull ? 0x40000000u : 0u) | (unsigned)o);
            }
            for (int k = 0; k < q.n; ++k) {
                const int o = lookup(col[q.base + k]);
                if (o >= 0) col[q.base + k] = (int32_t)(0x80000000u | (unsigned)o);
            }
        }
        if (windows)
            fprintf(stderr, "  x windows: %d tiles of <= %d slots, %.1f %% of the gather accesses inside, %.0f doubles staged per tile\n", nt,
                    kWinTile, 100.0 * lanes_win / lanes_all, staged / nt);
        WinMat m;
        m.ntiles = nt;
        NPG_HIP(hipMalloc((void **)&m.pcol, pcol.size() * 4 + 16));
        NPG_HIP(hipMalloc((void **)&m.col, col.size() * 4 + 16));
        NPG_HIP(hipMalloc((void **)&m.wr, wr.size() * sizeof(WinRange)));
        NPG_HIP(hipMalloc((void **)&m.td, td.size() * sizeof(TileDesc)));
        if (nrec) NPG_HIP(hipMemcpy(m.pcol, pcol.data(), pcol.size() * 4, hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(m.col, col.data(), col.size() * 4, hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(m.wr, wr.data(), wr.size() * sizeof(WinRange), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(m.td, td.data(), td.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        cache[key] = m;
    }
    const WinMat m = cache[key];
    CsrDev Av = csr_view(A);
    Av.pcol = m.pcol;
    Av.col = m.col;
    const int grid = std::max(1, std::min<int>(m.ntiles, bpc * ctx->num_cu));
    auto go = [&]() { hipLaunchKernelGGL(k_spmv_win<8>, dim3(grid), dim3(512), 0, ctx->stream, Av, m.wr, m.td, m.ntiles, x, y); };
    for (int i = 0; i < 2; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) go();
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f / reps;
    return NPG_OK;
}

}  // namespace npg

using namespace npg;

// phase timing of the product configuration: out7 = cycles per phase summed over workgroups (thread 0), [6] = tiles
NPG_API int npg_spmv_phase_cycles(const npg_csr *A, const npg_vec *x, npg_vec *y, int blocks_per_cu, int gather,
                                  unsigned long long *out7) {
    NPG_REQUIRE(A && x && y && out7 && x->n == A->n && y->n == A->m, "npg_spmv_phase_cycles: bad argument");
    unsigned long long *acc;
    NPG_HIP(hipMalloc((void **)&acc, 8 * sizeof(unsigned long long)));
    NPG_HIP(hipMemset(acc, 0, 8 * sizeof(unsigned long long)));
    npg_ctx *ctx = A->ctx;
    const int grid = std::max(1, std::min<int>(A->ntiles, blocks_per_cu * ctx->num_cu));
    const CsrDev Av = csr_view(A);
    if (gather)
        hipLaunchKernelGGL((k_spmv_timed<512, 16, kTileNnz, 4, true>), dim3(grid), dim3(512), 0, ctx->stream, Av,
                           A->tile_ptr, A->ntiles, x->d, y->d, acc);
    else
        hipLaunchKernelGGL((k_spmv_timed<512, 16, kTileNnz, 4, false>), dim3(grid), dim3(512), 0, ctx->stream, Av,
                           A->tile_ptr, A->ntiles, x->d, y->d, acc);
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_HIP(hipMemcpy(out7, acc, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    NPG_HIP(hipFree(acc));
    return NPG_OK;
}

NPG_API int npg_spmv_variant(const npg_csr *A, const npg_vec *x, npg_vec *y, int variant, int blocks_per_cu, int reps,
                             double *ms) {
    NPG_REQUIRE(A && x && y && ms && x->n == A->n && y->n == A->m && reps > 0, "npg_spmv_variant: bad argument");
    NPG_REQUIRE(A->nnode() == 0 || variant >= 30, "npg_spmv_variant: variants < 30 take plain CSR matrices only");
    switch (variant) {
        case 30: return run_prod<512, 4096, 4, 6, false>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 31: return run_prod<512, 2048, 2, 6, false>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 32: return run_prod<1024, 4096, 2, 8, false>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 40: return run_prod<512, 4096, 4, 6, false, 1>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 42: return run_prod<512, 4096, 4, 6, false, 2>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 50: return run_wide2<512, 5824, 4, 3, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 51: return run_wide2<512, 8192, 6, 4, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 52: return run_wide2<512, 9216, 6, 4, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 53: return run_wide2<512, 8192, 6, 3, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 54: return run_wide2<1024, 9216, 3, 2, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 43: return run_prod<512, 4800, 4, 6, false, 3>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 44: return run_prod<512, 4800, 4, 6, false, 0>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 45: return run_prod<512, 4800, 4, 6, false, 1>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 60: return run_pack<512, 8, 2048, 128, 6, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 61: return run_pack<512, 8, 2048, 128, 6, false>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 62: return run_pack<512, 8, 2560, 160, 4, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 66: return run_pack<64, 8, 256, 16, 6, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 67: return run_pack<64, 8, 384, 24, 4, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 68: return run_pack<128, 8, 512, 32, 6, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 69: return run_pack<64, 8, 512, 32, 3, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 63: return run_pack<256, 8, 1024, 64, 6, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 64: return run_pack<1024, 8, 2560, 160, 4, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 65: return run_pack<256, 8, 2048, 128, 3, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 72: return run_pack<512, 8, 1792, 128, 6, true, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 73: return run_pack<256, 8, 896, 64, 6, true, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 74: return run_pack<512, 8, 1792, 128, 6, true, false>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 75: return run_pack<512, 8, 1280, 96, 6, true, false, 3>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 76: return run_pack<256, 8, 640, 48, 6, true, false, 3>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 77: return run_pack<512, 8, 1664, 128, 4, true, false, 3>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 82: return run_xf<PlainX>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 83: return run_xf<WrapX<2047>>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 84: return run_xf<WrapX<32767>>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 85: return run_xf<WrapX<524287>>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 86: return run_ldsx<0>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 87: return run_ldsx<1>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 88: return run_ldsx<2>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 90: return run_win(A, x->d, y->d, blocks_per_cu, reps, ms, true);
        case 91: return run_win(A, x->d, y->d, blocks_per_cu, reps, ms, false);
        case 81: return run_cuadj(A, x->d, y->d, reps, ms);
        case 80: return run_cuq(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 70: return run_perm(A, x->d, y->d, blocks_per_cu, reps, ms, true);
        case 71: return run_perm(A, x->d, y->d, blocks_per_cu, reps, ms, false);
        case 0: return run_var<512, 4096, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 1: return run_var<512, 4096, 8>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 2: return run_var<1024, 8192, 8>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 3: return run_var<256, 2048, 8>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 4: return run_var<512, 8192, 8>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 5: return run_var<256, 4096, 8>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 6: return run_var<1024, 4096, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 7: return run_wide<512, 4096, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 8: return run_wide<256, 4096, 8>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 9: return run_wide<1024, 8192, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 10: return run_wide<256, 2048, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 11: return run_wide<1024, 4096, 2>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 12: return run_wide<1024, 8192, 4, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 13: return run_wide<512, 4096, 4, true>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 14: return run_wide<1024, 8192, 2>(A, x->d, y->d, blocks_per_cu, reps, ms);
        case 15: return run_wide<512, 8192, 4>(A, x->d, y->d, blocks_per_cu, reps, ms);
        default: NPG_REQUIRE(false, "npg_spmv_variant: unknown variant %d", variant);
    }
}

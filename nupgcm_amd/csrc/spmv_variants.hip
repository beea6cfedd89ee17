// Tuning harness (not part of the public ABI): times stand-alone CSR-stream SpMV kernels with different workgroup sizes,
// tile sizes and load widths on an existing device matrix, so that the configuration used by the product kernels is
// chosen from measurements on the target matrix (see profiles/ and DESIGN.md).
#include <algorithm>
#include <map>

#include "common.h"
#include "spmv_device.h"

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

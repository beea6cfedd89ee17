// CSR SpMV building blocks shared by the stand-alone SpMV and by the fused Krylov kernels.
//
// Layout in HBM: rowptr int64[m+1], col int32[nnz], val fp64[nnz] - 12 bytes streamed per stored entry - plus, for the
// inversion matrix with constant viscosity, an "xy-paired" part (below).
//
// "CSR-stream" tiles.  The host groups consecutive whole rows into tiles of at most TNNZ LDS product slots and kTileRows
// rows.  A workgroup handles a tile in two phases:
//   1. every thread streams PAIRS of adjacent entries of the tile's contiguous [rowptr[r0], rowptr[r1]) range: one
//      16-byte `val` load and one 8-byte `col` load per lane (the tile start is rounded down to an even entry so the
//      16-byte loads are aligned), U2 independent pairs in flight per lane, issued non-temporally - the matrix is read
//      exactly once per SpMV and must not evict the x vector from L2; then gathers x[col] (served by L2 / Infinity
//      Cache: the RCM ordering keeps a tile's columns close) and puts the products into LDS;
//   2. a sub-group of L lanes per row sums that row's segment of the LDS products (in-register DPP reduction) -
//      the wavefront-segmented sum.
// Streaming is therefore independent of the row-length distribution: short velocity rows and long pressure rows cost the
// same per stored entry.  A row longer than TNNZ forms a tile of its own and is handled by the whole workgroup.
//
// xy-paired part.  With constant viscosity the velocity block of A_inversion is [K -C; C K] on the (x, y) components
// (/root/reference/src/inversion.jl:183-192: same-component friction + f z-cross-u): rows 2q and 2q+1 (the DoF ordering
// interleaves the two components of a node, fe.py) hold the same K_qc and +-C_qc at columns 2c, 2c+1.  Those four CSR
// entries (48 bytes, four 8-byte gathers) are stored ONCE as {col c, K, C} (20 bytes) and cost one 16-byte gather of
// (x[2c], x[2c+1]).  Everything else (z rows, pressure rows, the u-p couplings of the paired rows) stays plain CSR.
//
// Configuration (512 threads, 4096 product slots, 4 pairs per lane, 3 workgroups per CU: fp64 + 64-bit addressing needs
// ~80 VGPRs, which rules out two 1024-thread workgroups per CU) chosen from the sweep in profiles/r01_spmv_variants.txt
// (tools/spmv_tune.py).
#pragma once
#include "device_utils.h"

namespace npg {

constexpr int kTileNnz = 4096;   // LDS product slots per tile: 32 KiB of fp64

// device view of a matrix: the (remainder) CSR arrays + the optional xy-paired part
struct CsrDev {
    const int64_t *rowptr;
    const int32_t *col;
    const double *val;
    int64_t nnz;              // entries in col/val
    const int64_t *prow;      // [npairs + 1] offsets of pair-row q (= rows 2q, 2q+1) into pcol / pkc; null if npairs == 0
    const int32_t *pcol;      // pair index c of the column node (columns 2c, 2c+1)
    const double2 *pkc;       // {K, C}: A[2q,2c] = A[2q+1,2c+1] = K ; A[2q,2c+1] = C ; A[2q+1,2c] = -C
    int npairs;
};

// SpMV input accessor: a plain contiguous vector (the Krylov kernels also plug in an on-the-fly corrected input)
struct PlainX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return x[c]; }
    __device__ __forceinline__ double2 pair(int c) const { return *reinterpret_cast<const double2 *>(x + 2 * (size_t)c); }
};

template <int TNNZ>
struct TileLdsT {
    double prod[TNNZ + 2];           // +2: the CSR part of a tile may start on an odd entry
    int32_t rp[kTileRows + 1];       // CSR row offsets into prod
    int32_t prp[kTileRows / 2 + 1];  // pair-row offsets into the paired products
};
using TileLds = TileLdsT<kTileNnz>;

// Phase 1 + 2 for one tile of rows [r0, r1).  On return (after the trailing barrier) out[r - r0] holds (A x)[r].
// NT = threads in the workgroup, L = lanes per row, U2 = independent entry pairs per lane and trip.
// Tiles never straddle the end of the paired region and start on even rows inside it.
template <int NT, int L, class XF, int TNNZ = kTileNnz, int U2 = 4>
__device__ __forceinline__ void spmv_tile(const CsrDev &A, const XF x, int r0, int r1, TileLdsT<TNNZ> &t,
                                          double *__restrict__ out) {
    const int64_t base = A.rowptr[r0];
    const int n = (int)(A.rowptr[r1] - base);
    const int nrows = r1 - r0;
    const bool paired = r0 < 2 * A.npairs;
    int npe = 0;                      // paired entries of this tile
    int64_t pbase = 0;
    if (paired) {
        pbase = A.prow[r0 >> 1];
        npe = (int)(A.prow[r1 >> 1] - pbase);
    }
    const int64_t abase = base & ~1LL;
    const int off = (int)(base - abase);
    const int total = n + off;
    const int slot0 = 2 * npe;        // CSR products live behind the two paired product arrays
    for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(A.rowptr[r0 + r] - base) + off + slot0;
    if (paired)
        for (int q = threadIdx.x; q <= (nrows >> 1); q += NT) t.prp[q] = (int32_t)(A.prow[(r0 >> 1) + q] - pbase);
    if (slot0 + total <= TNNZ + 2) {
        // ---- paired stream: {c, K, C} -> products for row 2q (first npe slots) and row 2q+1 (next npe slots)
        for (int e0 = threadIdx.x; e0 < npe; e0 += U2 * NT) {
            int32_t c[U2];
            double2 kc[U2], xx[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int e = e0 + u * NT;
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
#pragma unroll
            for (int u = 0; u < U2; ++u) xx[u] = x.pair(c[u]);
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int e = e0 + u * NT;
                if (e < npe) {
                    t.prod[e] = kc[u].x * xx[u].x + kc[u].y * xx[u].y;
                    t.prod[npe + e] = kc[u].x * xx[u].y - kc[u].y * xx[u].x;
                }
            }
        }
        // ---- CSR stream
        for (int k0 = 2 * threadIdx.x; k0 < total; k0 += 2 * NT * U2) {
            int2 c[U2];
            double2 v[U2];
            double xa[U2], xb[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                if (k < total && abase + k + 1 < A.nnz) {
                    const long long cc = __builtin_nontemporal_load(reinterpret_cast<const long long *>(A.col + abase + k));
                    c[u] = make_int2((int)(cc & 0xffffffffLL), (int)(cc >> 32));
                    v[u].x = __builtin_nontemporal_load(A.val + abase + k);
                    v[u].y = __builtin_nontemporal_load(A.val + abase + k + 1);
                } else if (k < total && abase + k < A.nnz) {
                    c[u] = make_int2(A.col[abase + k], 0);
                    v[u] = make_double2(A.val[abase + k], 0.0);
                } else {
                    c[u] = make_int2(0, 0);
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
                if (k < total) t.prod[slot0 + k] = (k >= off) ? v[u].x * xa[u] : 0.0;
                if (k + 1 < total) t.prod[slot0 + k + 1] = v[u].y * xb[u];
            }
        }
        __syncthreads();
        const int g = threadIdx.x / L, l = threadIdx.x % L;
        for (int r = g; r < nrows; r += NT / L) {
            double s = 0.0;
            const int e = t.rp[r + 1];
            for (int k = t.rp[r] + l; k < e; k += L) s += t.prod[k];
            if (paired) {
                const int q = r >> 1, pb = (r & 1) ? npe : 0, pe = t.prp[q + 1];
                for (int k = t.prp[q] + l; k < pe; k += L) s += t.prod[pb + k];
            }
            s = group_sum_dpp<L>(s);
            if (l == 0) out[r] = s;
        }
    } else {
        // one very long (unpaired) row: the whole workgroup strides over it, tree-reduce through LDS
        double s = 0.0;
        for (int k = threadIdx.x; k < n; k += NT) s += A.val[base + k] * x(A.col[base + k]);
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) t.prod[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < NT / 64; ++w) tot += t.prod[w];
            out[0] = tot;
        }
    }
    __syncthreads();
}

// Variant of spmv_tile that issues the loads of BOTH streams of a tile (paired records and CSR entries) before the first
// gather, so that a paired tile has all of its bytes in flight at once instead of in two dependent phases.
template <int NT, int L, class XF, int TNNZ = kTileNnz, int U2 = 4>
__device__ __forceinline__ void spmv_tile2(const CsrDev &A, const XF x, int r0, int r1, TileLdsT<TNNZ> &t,
                                           double *__restrict__ out) {
    const int64_t base = A.rowptr[r0];
    const int n = (int)(A.rowptr[r1] - base);
    const int nrows = r1 - r0;
    const bool paired = r0 < 2 * A.npairs;
    int npe = 0;
    int64_t pbase = 0;
    if (paired) {
        pbase = A.prow[r0 >> 1];
        npe = (int)(A.prow[r1 >> 1] - pbase);
    }
    const int64_t abase = base & ~1LL;
    const int off = (int)(base - abase);
    const int total = n + off;
    const int slot0 = 2 * npe;
    if (slot0 + total > TNNZ + 2 || npe > NT * U2 || total > 2 * NT * U2) {
        spmv_tile<NT, L, XF, TNNZ, U2>(A, x, r0, r1, t, out);      // generic path (long rows)
        return;
    }
    // ---- issue: paired records, CSR entry pairs, row offsets
    int32_t pc[U2];
    double2 kc[U2];
    int2 c[U2];
    double2 v[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
        const int e = threadIdx.x + u * NT;
        if (e < npe) {
            pc[u] = __builtin_nontemporal_load(A.pcol + pbase + e);
            const double *p = reinterpret_cast<const double *>(A.pkc + pbase + e);
            kc[u].x = __builtin_nontemporal_load(p);
            kc[u].y = __builtin_nontemporal_load(p + 1);
        } else {
            pc[u] = 0;
            kc[u] = make_double2(0.0, 0.0);
        }
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) {
        const int k = 2 * threadIdx.x + u * 2 * NT;
        if (k < total && abase + k + 1 < A.nnz) {
            const long long cc = __builtin_nontemporal_load(reinterpret_cast<const long long *>(A.col + abase + k));
            c[u] = make_int2((int)(cc & 0xffffffffLL), (int)(cc >> 32));
            v[u].x = __builtin_nontemporal_load(A.val + abase + k);
            v[u].y = __builtin_nontemporal_load(A.val + abase + k + 1);
        } else if (k < total && abase + k < A.nnz) {
            c[u] = make_int2(A.col[abase + k], 0);
            v[u] = make_double2(A.val[abase + k], 0.0);
        } else {
            c[u] = make_int2(0, 0);
            v[u] = make_double2(0.0, 0.0);
        }
    }
    for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(A.rowptr[r0 + r] - base) + off + slot0;
    if (paired)
        for (int q = threadIdx.x; q <= (nrows >> 1); q += NT) t.prp[q] = (int32_t)(A.prow[(r0 >> 1) + q] - pbase);
    // ---- gathers + products
    {
        double2 xx[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) xx[u] = x.pair(pc[u]);
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int e = threadIdx.x + u * NT;
            if (e < npe) {
                t.prod[e] = kc[u].x * xx[u].x + kc[u].y * xx[u].y;
                t.prod[npe + e] = kc[u].x * xx[u].y - kc[u].y * xx[u].x;
            }
        }
    }
    {
        double xa[U2], xb[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            xa[u] = x(c[u].x);
            xb[u] = x(c[u].y);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int k = 2 * threadIdx.x + u * 2 * NT;
            if (k < total) t.prod[slot0 + k] = (k >= off) ? v[u].x * xa[u] : 0.0;
            if (k + 1 < total) t.prod[slot0 + k + 1] = v[u].y * xb[u];
        }
    }
    __syncthreads();
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    for (int r = g; r < nrows; r += NT / L) {
        double s = 0.0;
        const int e = t.rp[r + 1];
        for (int k = t.rp[r] + l; k < e; k += L) s += t.prod[k];
        if (paired) {
            const int q = r >> 1, pb = (r & 1) ? npe : 0, pe = t.prp[q + 1];
            for (int k = t.prp[q] + l; k < pe; k += L) s += t.prod[pb + k];
        }
        s = group_sum_dpp<L>(s);
        if (l == 0) out[r] = s;
    }
    __syncthreads();
}

}  // namespace npg

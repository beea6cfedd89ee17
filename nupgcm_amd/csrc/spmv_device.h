// CSR SpMV building blocks shared by the stand-alone SpMV and by the fused Krylov kernels.
//
// Layout in HBM: rowptr int64[m+1], col int32[nnz], val fp64[nnz] - 12 bytes streamed per stored entry.
//
// "CSR-stream" tiles.  The host groups consecutive whole rows into tiles of at most TNNZ stored entries and kTileRows
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
// Configuration (512 threads, 4096-entry tiles, 4 pairs per lane, 3 workgroups per CU: fp64 + 64-bit addressing needs
// ~80 VGPRs, which rules out two 1024-thread workgroups per CU) chosen from the sweep in profiles/r01_spmv_variants.txt
// (tools/spmv_tune.py): 4.1-4.3 TB/s on the 1.56 GB bowl3D h=0.02 matrix.
#pragma once
#include "device_utils.h"

namespace npg {

constexpr int kTileNnz = 4096;   // stored entries per tile: 32 KiB of fp64 products in LDS

// SpMV input accessor: a plain contiguous vector (the Krylov kernels also plug in an on-the-fly corrected input)
struct PlainX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return x[c]; }
};

template <int TNNZ>
struct TileLdsT {
    double prod[TNNZ + 2];       // +2: the tile may start on an odd entry
    int32_t rp[kTileRows + 1];   // row offsets of the tile relative to its (even-aligned) first loaded entry
};
using TileLds = TileLdsT<kTileNnz>;

// Phase 1 + 2 for one tile.  On return (after the trailing barrier) out[r - r0] holds sum_k val[k] x[col[k]] for every
// row r of the tile.  NT = threads in the workgroup, L = lanes per row, U2 = independent entry pairs per lane and trip,
// nnz = total stored entries of the matrix (bounds the last pair).
template <int NT, int L, class XF, int TNNZ = kTileNnz, int U2 = 4>
__device__ __forceinline__ void spmv_tile(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                          const double *__restrict__ val, int64_t nnz, const XF x, int r0, int r1,
                                          TileLdsT<TNNZ> &t, double *__restrict__ out) {
    const int64_t base = rowptr[r0];
    const int n = (int)(rowptr[r1] - base);
    const int nrows = r1 - r0;
    const int64_t abase = base & ~1LL;
    const int off = (int)(base - abase);
    const int total = n + off;
    for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(rowptr[r0 + r] - base) + off;
    if (n <= TNNZ) {
        for (int k0 = 2 * threadIdx.x; k0 < total; k0 += 2 * NT * U2) {
            int2 c[U2];
            double2 v[U2];
            double xa[U2], xb[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                if (k < total && abase + k + 1 < nnz) {
                    const long long cc = __builtin_nontemporal_load(reinterpret_cast<const long long *>(col + abase + k));
                    c[u] = make_int2((int)(cc & 0xffffffffLL), (int)(cc >> 32));
                    v[u].x = __builtin_nontemporal_load(val + abase + k);
                    v[u].y = __builtin_nontemporal_load(val + abase + k + 1);
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
                xa[u] = x(c[u].x);
                xb[u] = x(c[u].y);
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                if (k < total) t.prod[k] = (k >= off) ? v[u].x * xa[u] : 0.0;
                if (k + 1 < total) t.prod[k + 1] = v[u].y * xb[u];
            }
        }
        __syncthreads();
        const int g = threadIdx.x / L, l = threadIdx.x % L;
        for (int r = g; r < nrows; r += NT / L) {
            double s = 0.0;
            const int e = t.rp[r + 1];
            for (int k = t.rp[r] + l; k < e; k += L) s += t.prod[k];
            s = group_sum_dpp<L>(s);
            if (l == 0) out[r] = s;
        }
    } else {
        // one very long row: the whole workgroup strides over it, tree-reduce through LDS
        double s = 0.0;
        for (int k = threadIdx.x; k < n; k += NT) s += val[base + k] * x(col[base + k]);
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

}  // namespace npg

// CSR SpMV building blocks shared by the stand-alone SpMV and by the fused Krylov kernels.
//
// Layout in HBM: rowptr int64[m+1], col int32[nnz], val fp64[nnz] - 12 bytes streamed per stored entry.
//
// "CSR-stream" tiles.  The host groups consecutive whole rows into tiles of at most kTileNnz stored entries and
// kTileRows rows.  A workgroup handles a tile in two phases:
//   1. every thread streams entries k = tid, tid + NT, ... of the tile's contiguous [rowptr[r0], rowptr[r1]) range:
//      consecutive lanes read consecutive `col`/`val` entries (perfectly coalesced 4 B / 8 B per lane, several independent
//      loads in flight per lane), gather x[col] (served by L2 / Infinity Cache: the RCM ordering keeps a tile's columns
//      close), and put the products into LDS;
//   2. a sub-group of L lanes per row sums that row's segment of the LDS products (in-register DPP reduction) -
//      the wavefront-segmented sum.
// Streaming is therefore independent of the row-length distribution: short velocity rows and long pressure rows cost the
// same per stored entry.  A row longer than kTileNnz forms a tile of its own and is handled by the whole workgroup.
#pragma once
#include "device_utils.h"

namespace npg {

constexpr int kTileNnz = 4096;   // stored entries per tile: 32 KiB of fp64 products in LDS (two workgroups per CU)

// SpMV input accessor: a plain contiguous vector (the Krylov kernels also plug in an on-the-fly corrected input)
struct PlainX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return x[c]; }
};

struct TileLds {
    double prod[kTileNnz];
    int32_t rp[kTileRows + 1];   // row offsets of the tile relative to its first entry
};

// Phase 1 + 2 for one ordinary tile.  On return (after the trailing barrier) out[r - r0] holds sum_k val[k] x[col[k]]
// for every row r of the tile; out may alias nothing in `t`.  NT = threads in the workgroup, L = lanes per row.
template <int NT, int L, class XF>
__device__ __forceinline__ void spmv_tile(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                          const double *__restrict__ val, const XF x, int r0, int r1, TileLds &t,
                                          double *__restrict__ out) {
    const int64_t base = rowptr[r0];
    const int n = (int)(rowptr[r1] - base);
    const int nrows = r1 - r0;
    for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(rowptr[r0 + r] - base);
    if (n <= kTileNnz) {
        constexpr int U = 4;
        for (int k0 = threadIdx.x; k0 < n; k0 += U * NT) {
            int32_t c[U];
            double v[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * NT;
                c[u] = (k < n) ? col[base + k] : 0;
                v[u] = (k < n) ? val[base + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = x(c[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * NT;
                if (k < n) t.prod[k] = v[u] * xv[u];
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

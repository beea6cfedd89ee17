// Device-side helpers: 64-lane wave reductions, block reductions, fixed-order partial-sum reduction across blocks.
#pragma once
#include "common.h"

namespace npg {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// sum over a sub-group of L consecutive lanes (L a power of two <= 64); every lane of the group gets the result
template <int L>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// fp64 lane exchange through the DPP crossbar of the VALU (no LDS traffic, unlike ds_bpermute-based __shfl):
// CTRL 0xB1 = quad_perm[1,0,3,2] (xor 1), 0x4E = quad_perm[2,3,0,1] (xor 2), 0x141 = row_half_mirror, 0x140 = row_mirror.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// Sum over aligned sub-groups of L consecutive lanes (L = 4, 8, 16, 32 or 64); every lane gets its group's total.
// The first four doublings stay inside a 16-lane DPP row; only 32- and 64-lane groups need a cross-row permute.
template <int L>
__device__ __forceinline__ double group_sum_dpp(double v) {
    if (L >= 2) v += dpp_f64<0xB1>(v);
    if (L >= 4) v += dpp_f64<0x4E>(v);
    if (L >= 8) v += dpp_f64<0x141>(v);
    if (L >= 16) v += dpp_f64<0x140>(v);
    if (L >= 32) v += __shfl_xor(v, 16, 64);
    if (L >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// Block-wide sum of one value for a 256-thread block; result valid in every thread.  `sh` holds >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Reduce `nvals` (<= kPartStride) per-thread accumulators acc[0..nvals) over a block of NW waves and store them to
// part[blockIdx.x * kPartStride + k].  sh: NW * kPartStride doubles.  Fixed summation order => deterministic.
template <int NV, int NW>
__device__ __forceinline__ void block_store_partials(const double (&acc)[NV], int nvals, double *sh, double *part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k < nvals) {
            double s = wave_sum(acc[k]);
            if (lane == 0) sh[wave * kPartStride + k] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < nvals) {
        const int k = threadIdx.x;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sh[w * kPartStride + k];
        part[(size_t)blockIdx.x * kPartStride + k] = s;
    }
}

// Every block sums the rows part[b][0..nvals) over b < nblocks (<= NS * MAXI) in one fixed order and leaves the totals
// in sh_out[k] (visible to all threads after the trailing barrier).  The block has NS * kPartStride threads: thread
// (slice, k) loads rows slice, slice + NS, ... - all MAXI loads are issued back to back before any is consumed, so the
// prologue costs one memory round trip instead of nblocks/NS dependent ones.  sh_tmp: NS * kPartStride doubles.
template <int NS, int MAXI>
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, int nblocks, int nvals, double *sh_tmp,
                                                double *sh_out) {
    const int k = threadIdx.x & (kPartStride - 1);
    const int slice = threadIdx.x >> 5;
    constexpr int CH = MAXI < 8 ? MAXI : 8;      // loads in flight per thread
    static_assert(MAXI % CH == 0, "MAXI must be a multiple of the chunk");
    double s = 0.0;
    for (int i0 = 0; i0 < MAXI; i0 += CH) {
        if (slice + NS * i0 >= nblocks) break;   // uniform per 32-lane slice
        double v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int b = slice + NS * (i0 + i);
            v[i] = (k < nvals && b < nblocks) ? part[(size_t)b * kPartStride + k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) s += v[i];
    }
    sh_tmp[slice * kPartStride + k] = s;
    __syncthreads();
    if (threadIdx.x < kPartStride) {
        double t = 0.0;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) t += sh_tmp[sl * kPartStride + threadIdx.x];
        sh_out[threadIdx.x] = t;
    }
    __syncthreads();
}

}  // namespace npg

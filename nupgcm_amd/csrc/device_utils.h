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

// Block-wide sum of one value for a 256-thread block; result valid in every thread.  `sh` holds >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// Reduce `nvals` (<= kPartStride) per-thread accumulators acc[0..nvals) over the block and store them to
// part[blockIdx.x * kPartStride + k].  sh: 4 * kPartStride doubles.  Fixed summation order => deterministic.
template <int NV>
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
        part[(size_t)blockIdx.x * kPartStride + k] =
            (sh[k] + sh[kPartStride + k]) + (sh[2 * kPartStride + k] + sh[3 * kPartStride + k]);
    }
}

// Every block sums the rows part[b][0..nvals) over b < nblocks in one fixed order and leaves the totals in sh_out[k]
// (visible to all threads after the trailing barrier).  sh_tmp: 8 * kPartStride doubles.
__device__ __forceinline__ void reduce_partials(const double *part, int nblocks, int nvals, double *sh_tmp,
                                                double *sh_out) {
    const int k = threadIdx.x & (kPartStride - 1);
    const int slice = threadIdx.x >> 5;   // 8 slices of 32 threads
    double s = 0.0;
    if (k < nvals)
        for (int b = slice; b < nblocks; b += 8) s += part[(size_t)b * kPartStride + k];
    sh_tmp[slice * kPartStride + k] = s;
    __syncthreads();
    if (threadIdx.x < kPartStride) {
        double t = 0.0;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) t += sh_tmp[sl * kPartStride + threadIdx.x];
        sh_out[threadIdx.x] = t;
    }
    __syncthreads();
}

}  // namespace npg

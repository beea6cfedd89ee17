// CSR row kernels shared by the stand-alone SpMV and by the fused Krylov kernels.
//
// Layout: rowptr int64[m+1], col int32[nnz], val fp64[nnz] - 12 bytes streamed per stored entry.  A row is handled by a
// sub-group of L consecutive lanes of a 64-lane wave (L = 4..64 picked from the mean row length), so one wave reads
// 64/L neighbouring rows at once: consecutive lanes read consecutive entries of `val`/`col` (coalesced 8 B / 4 B per
// lane), and the L partial products are summed with in-register xor-shuffles.  Rows are grouped on the host into
// nnz-balanced *tiles* of <= 256 consecutive rows, so every workgroup streams the same number of bytes whatever mix of
// short velocity rows and long pressure rows it gets.
#pragma once
#include "device_utils.h"

namespace npg {

template <int L>
__device__ __forceinline__ double csr_row_dot(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                              const double *__restrict__ val, const double *__restrict__ x, int row,
                                              int lane_in_group) {
    const int64_t start = rowptr[row], end = rowptr[row + 1];
    double s0 = 0.0, s1 = 0.0;
    int64_t k = start + lane_in_group;
    for (; k + L < end; k += 2 * L) {
        const int32_t c0 = col[k], c1 = col[k + L];
        const double v0 = val[k], v1 = val[k + L];
        s0 += v0 * x[c0];
        s1 += v1 * x[c1];
    }
    if (k < end) s0 += val[k] * x[col[k]];
    return group_sum<L>(s0 + s1);
}

}  // namespace npg

// Device-resident restarted GMRES(m): replaces Krylov.krylov_solve!(::GmresWorkspace, A, y, x; M=P, ...) as configured at
// /root/reference/src/inversion.jl:74-94 and driven from /root/reference/src/iterative_solvers.jl:58.
//
// MI355X design.  On the reference's GPU path one inner iteration is ~23 library calls and ~11 host-blocking scalar
// reductions (modified Gram-Schmidt: j dependent dot->axpy pairs).  Here a restart cycle is ONE hipGraph of 3 kernels per
// inner iteration, no scalar ever visits the host inside a cycle, and the kernel boundary (the cheapest grid-wide
// synchronisation on this chip, ~1.5 us) is the only global barrier:
//
//   K1(j)  finalise column j-1 (Givens, residual estimate, stopping test - redundantly in every workgroup, from
//          fixed-order partial sums), v_j = wt/beta, w = P A v_j (tiled CSR SpMV, rows staged in LDS),
//          partial h1 = V_{0..j}' w and ||w||^2                                    -> P1
//   K2(j)  h1 = sum P1 ; wt = w - V h1 ; partial h2 = V' wt and ||wt||^2            -> P2
//   K3(j)  h2 = sum P2 ; if ||wt|| < eta ||w|| (DGKS test): wt -= V h2, ||wt||^2    -> P3   (else: returns at once)
//   XU     finalise the last column, back-substitute R y = z, x += V y
//   R1     wt = P (b - A x), ||wt||^2                                              -> PR   (true residual for next cycle)
//
// i.e. classical Gram-Schmidt with a selective second pass: one reduction per pass instead of MGS's j sequential ones;
// the Hessenberg column is h1 (+ h2 when the second pass ran).  Results agree with MGS to rounding, not bitwise.
// All partial sums are reduced in a fixed order, so a solve is bit-reproducible run to run.
//
// State hand-off between kernels goes through write-once snapshot slots T[j] ("state after j finalised columns"): a slot
// is written by workgroup 0 of one kernel and only read by LATER kernels, so no workgroup ever reads a location another
// workgroup of the same launch writes.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "common.h"
#include "spmv_device.h"

namespace npg {

struct Snap {
    double eps, rnorm0, rnorm, beta, zeta;
    int iter, inner, done, npass, first, nreorth, pad0, pad1;
};

struct ColInfo {
    double wnorm2, n2p;
    int reorth, pad;
};

struct GParams {
    double atol, rtol, eta2, btol;
    long long itmax;
};

struct GDev {
    const int64_t *rowptr;
    const int32_t *col;
    const double *val;
    const int32_t *tile_ptr;
    int ntiles, n, ld, mem;
    int pkind;
    double pscalar;
    const double *pdiag;
    const double *b;
    double *x, *V, *w, *wt;
    double *P1, *P2, *P3, *PR;
    int G1, G2;
    Snap *C, *T;
    double *c, *s, *z, *R, *hcol1, *hcol2;
    ColInfo *ci;
    double *hist;
    int hist_cap;
    const GParams *prm;
};

__device__ __forceinline__ void sym_givens(double a, double b, double &c, double &s, double &rho) {
    if (b == 0.0) {
        c = (a == 0.0) ? 1.0 : copysign(1.0, a);
        s = 0.0;
        rho = fabs(a);
    } else if (a == 0.0) {
        c = 0.0;
        s = copysign(1.0, b);
        rho = fabs(b);
    } else if (fabs(b) > fabs(a)) {
        const double t = a / b;
        s = copysign(1.0, b) / sqrt(1.0 + t * t);
        c = s * t;
        rho = b / s;
    } else {
        const double t = b / a;
        c = copysign(1.0, a) / sqrt(1.0 + t * t);
        s = c * t;
        rho = a / c;
    }
}

// Shared scratch of the Krylov kernels
struct KShared {
    double tmp[8 * kPartStride];
    double red[kPartStride];
    double wsum[4 * kPartStride];
    double h[kPartStride], cc[kPartStride], ss[kPartStride];
    Snap T;
    double y[kPartStride];
};

// Finalise Hessenberg column `colj` (all threads call; thread 0 does the serial part).  Leaves the new snapshot in sh.T.
__device__ void finalize_column(const GDev &d, int colj, KShared &sh) {
    const Snap prev = d.T[colj];
    const ColInfo ci = d.ci[colj];
    if (prev.done == 0 && ci.reorth) reduce_partials(d.P3, d.G2, 1, sh.tmp, sh.red);
    if (threadIdx.x <= colj && threadIdx.x < kPartStride) {
        sh.h[threadIdx.x] = d.hcol1[colj * kPartStride + threadIdx.x] + d.hcol2[colj * kPartStride + threadIdx.x];
        sh.cc[threadIdx.x] = d.c[threadIdx.x];
        sh.ss[threadIdx.x] = d.s[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        Snap t = prev;
        if (t.done == 0) {
            const double n2 = ci.reorth ? sh.red[0] : ci.n2p;
            const double hbis = sqrt(n2);
            for (int i = 0; i < colj; ++i) {
                const double tmp = sh.cc[i] * sh.h[i] + sh.ss[i] * sh.h[i + 1];
                sh.h[i + 1] = sh.ss[i] * sh.h[i] - sh.cc[i] * sh.h[i + 1];
                sh.h[i] = tmp;
            }
            double cj, sj, rho;
            sym_givens(sh.h[colj], hbis, cj, sj, rho);
            const double zeta_next = sj * t.zeta;
            const double zcol = cj * t.zeta;
            t.rnorm = fabs(zeta_next);
            t.iter += 1;
            t.inner += 1;
            t.nreorth += ci.reorth;
            const bool solved = (t.rnorm <= t.eps) || (t.rnorm + 1.0 <= 1.0);
            t.done = solved ? 1 : ((long long)t.iter >= d.prm->itmax ? 2 : (hbis <= d.prm->btol ? 3 : 0));
            t.beta = hbis;
            t.zeta = zeta_next;
            if (blockIdx.x == 0) {
                d.c[colj] = cj;
                d.s[colj] = sj;
                d.z[colj] = zcol;
                const int base = colj * (colj + 1) / 2;
                for (int i = 0; i < colj; ++i) d.R[base + i] = sh.h[i];
                d.R[base + colj] = rho;
                if (t.iter < d.hist_cap) d.hist[t.iter] = t.rnorm;
            }
        }
        sh.T = t;
        if (blockIdx.x == 0) d.T[colj + 1] = t;
    }
    __syncthreads();
}

__device__ __forceinline__ double precond_row(const GDev &d, int row) {
    return d.pkind == NPG_PRECOND_SCALAR ? d.pscalar : (d.pkind == NPG_PRECOND_DIAG ? d.pdiag[row] : 1.0);
}

// ---- R1: wt = P (b - A x), partial ||wt||^2 ---------------------------------------------------------------------------
template <int L>
__global__ void __launch_bounds__(kBlock) k_gmres_residual(GDev d) {
    __shared__ double sh[4 * kPartStride];
    const Snap c = *d.C;
    double acc[1] = {0.0};
    if (c.done == 0) {
        const int g = threadIdx.x / L, l = threadIdx.x % L;
        for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
            const int r0 = d.tile_ptr[t], r1 = d.tile_ptr[t + 1];
            for (int row = r0 + g; row < r1; row += kBlock / L) {
                const double ax = csr_row_dot<L>(d.rowptr, d.col, d.val, d.x, row, l);
                if (l == 0) {
                    const double r = precond_row(d, row) * (d.b[row] - ax);
                    d.wt[row] = r;
                    acc[0] += r * r;
                }
            }
        }
    }
    block_store_partials<1>(acc, 1, sh, d.PR);
}

// ---- K1 ---------------------------------------------------------------------------------------------------------------
template <int L, int JB>
__global__ void __launch_bounds__(kBlock) k_gmres_arnoldi(GDev d, int j) {
    __shared__ KShared sh;
    __shared__ double sw[kBlock];
    if (j == 0) {
        const Snap c = *d.C;
        if (c.done == 0) reduce_partials(d.PR, d.G1, 1, sh.tmp, sh.red);
        if (threadIdx.x == 0) {
            Snap t = c;
            if (t.done == 0) {
                const double beta = sqrt(sh.red[0]);
                if (t.first) {
                    t.rnorm0 = beta;
                    t.rnorm = beta;
                    t.eps = d.prm->atol + d.prm->rtol * beta;
                    t.first = 0;
                    if (blockIdx.x == 0) d.hist[0] = beta;
                    if (beta == 0.0) t.done = 4;
                }
                t.beta = beta;
                t.zeta = beta;
                t.inner = 0;
                t.npass += 1;
            } else {
                t.inner = 0;
            }
            sh.T = t;
            if (blockIdx.x == 0) d.T[0] = t;
        }
        __syncthreads();
    } else {
        finalize_column(d, j - 1, sh);
    }
    const Snap T = sh.T;
    if (T.done != 0) return;

    const double inv_beta = 1.0 / T.beta;
    double acc[JB + 1];
#pragma unroll
    for (int k = 0; k <= JB; ++k) acc[k] = 0.0;
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    double *Vj = d.V + (size_t)j * d.ld;
    for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
        const int r0 = d.tile_ptr[t], r1 = d.tile_ptr[t + 1];
        __syncthreads();
        // phase A: SpMV rows of the tile -> LDS
        for (int row = r0 + g; row < r1; row += kBlock / L) {
            const double s = csr_row_dot<L>(d.rowptr, d.col, d.val, d.wt, row, l);
            if (l == 0) sw[row - r0] = s * inv_beta * precond_row(d, row);
        }
        __syncthreads();
        // phase B: one thread per row: normalised basis vector, w, partial dot products
        const int row = r0 + threadIdx.x;
        if (row < r1) {
            const double wv = sw[threadIdx.x];
            const double vj = d.wt[row] * inv_beta;
            Vj[row] = vj;
            d.w[row] = wv;
#pragma unroll
            for (int k = 0; k < JB; ++k)
                if (k < j) acc[k] += d.V[(size_t)k * d.ld + row] * wv;
            acc[JB] += wv * wv;
            // the k == j term uses the value just computed
            double vjw = vj * wv;
#pragma unroll
            for (int k = 0; k < JB; ++k)
                if (k == j) acc[k] += vjw;
        }
    }
    // partial row layout: [0..j] = h1, [j+1] = ||w||^2
    double out[JB + 1];
#pragma unroll
    for (int k = 0; k < JB; ++k) out[k] = acc[k];
    out[JB] = 0.0;
#pragma unroll
    for (int k = 0; k <= JB; ++k)
        if (k == j + 1) out[k] = acc[JB];
    block_store_partials<JB + 1>(out, j + 2, sh.wsum, d.P1);
}

// ---- K2 ---------------------------------------------------------------------------------------------------------------
template <int JB>
__global__ void __launch_bounds__(kBlock) k_gmres_orth1(GDev d, int j) {
    __shared__ KShared sh;
    const Snap T = d.T[j];
    if (T.done != 0) return;
    reduce_partials(d.P1, d.G1, j + 2, sh.tmp, sh.red);
    if (blockIdx.x == 0 && threadIdx.x < kPartStride) {
        d.hcol1[j * kPartStride + threadIdx.x] = (threadIdx.x <= j) ? sh.red[threadIdx.x] : 0.0;
        if (threadIdx.x == 0) d.ci[j].wnorm2 = sh.red[j + 1];
    }
    double h[JB], acc[JB + 1];
#pragma unroll
    for (int k = 0; k < JB; ++k) {
        h[k] = (k <= j) ? sh.red[k] : 0.0;
        acc[k] = 0.0;
    }
    acc[JB] = 0.0;
    for (int64_t row = blockIdx.x * (int64_t)kBlock + threadIdx.x; row < d.n; row += (int64_t)gridDim.x * kBlock) {
        double v[JB];
#pragma unroll
        for (int k = 0; k < JB; ++k) v[k] = (k <= j) ? d.V[(size_t)k * d.ld + row] : 0.0;
        double wv = d.w[row];
#pragma unroll
        for (int k = 0; k < JB; ++k) wv -= h[k] * v[k];
        d.wt[row] = wv;
#pragma unroll
        for (int k = 0; k < JB; ++k) acc[k] += v[k] * wv;
        acc[JB] += wv * wv;
    }
    double out[JB + 1];
#pragma unroll
    for (int k = 0; k < JB; ++k) out[k] = acc[k];
    out[JB] = 0.0;
#pragma unroll
    for (int k = 0; k <= JB; ++k)
        if (k == j + 1) out[k] = acc[JB];
    block_store_partials<JB + 1>(out, j + 2, sh.wsum, d.P2);
}

// ---- K3 ---------------------------------------------------------------------------------------------------------------
template <int JB>
__global__ void __launch_bounds__(kBlock) k_gmres_orth2(GDev d, int j) {
    __shared__ KShared sh;
    const Snap T = d.T[j];
    if (T.done != 0) return;
    const double wnorm2 = d.ci[j].wnorm2;
    reduce_partials(d.P2, d.G2, j + 2, sh.tmp, sh.red);
    const double n2p = sh.red[j + 1];
    const bool reorth = n2p < d.prm->eta2 * wnorm2;
    if (blockIdx.x == 0 && threadIdx.x < kPartStride) {
        d.hcol2[j * kPartStride + threadIdx.x] = (reorth && threadIdx.x <= j) ? sh.red[threadIdx.x] : 0.0;
        if (threadIdx.x == 0) {
            d.ci[j].n2p = n2p;
            d.ci[j].reorth = reorth ? 1 : 0;
        }
    }
    if (!reorth) return;
    double h[JB];
#pragma unroll
    for (int k = 0; k < JB; ++k) h[k] = (k <= j) ? sh.red[k] : 0.0;
    double acc[1] = {0.0};
    for (int64_t row = blockIdx.x * (int64_t)kBlock + threadIdx.x; row < d.n; row += (int64_t)gridDim.x * kBlock) {
        double wv = d.wt[row];
#pragma unroll
        for (int k = 0; k < JB; ++k)
            if (k <= j) wv -= h[k] * d.V[(size_t)k * d.ld + row];
        d.wt[row] = wv;
        acc[0] += wv * wv;
    }
    block_store_partials<1>(acc, 1, sh.wsum, d.P3);
}

// ---- XU ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_gmres_update(GDev d) {
    __shared__ KShared sh;
    finalize_column(d, d.mem - 1, sh);     // no-op copy if the pass already ended
    const Snap T = sh.T;
    const int k = T.inner;
    if (threadIdx.x < kPartStride) sh.y[threadIdx.x] = (threadIdx.x < k) ? d.z[threadIdx.x] : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        // back substitution on the packed upper-triangular R (column-major packed, column c starts at c(c+1)/2)
        for (int i = k - 1; i >= 0; --i) {
            double yi = sh.y[i];
            for (int c = k - 1; c > i; --c) yi -= d.R[c * (c + 1) / 2 + i] * sh.y[c];
            const double rii = d.R[i * (i + 1) / 2 + i];
            sh.y[i] = (fabs(rii) <= d.prm->btol) ? 0.0 : yi / rii;
        }
        if (blockIdx.x == 0) {
            Snap c = T;
            c.inner = 0;
            *d.C = c;
        }
    }
    __syncthreads();
    if (k == 0) return;
    for (int64_t row = blockIdx.x * (int64_t)kBlock + threadIdx.x; row < d.n; row += (int64_t)gridDim.x * kBlock) {
        double xv = d.x[row];
        for (int i = 0; i < k; ++i) xv += sh.y[i] * d.V[(size_t)i * d.ld + row];
        d.x[row] = xv;
    }
}

}  // namespace npg

using namespace npg;

struct npg_gmres {
    npg_ctx *ctx = nullptr;
    int64_t n = 0;
    int mem = 20, ld = 0;
    double *V = nullptr, *w = nullptr, *wt = nullptr;
    double *P1 = nullptr, *P2 = nullptr, *P3 = nullptr, *PR = nullptr;
    Snap *C = nullptr, *T = nullptr;
    double *c = nullptr, *s = nullptr, *z = nullptr, *R = nullptr, *hcol1 = nullptr, *hcol2 = nullptr;
    ColInfo *ci = nullptr;
    double *hist = nullptr;
    int hist_cap = 0;
    GParams *prm = nullptr;
    Snap *h_C = nullptr;          // pinned, two slots (one per graph of the ping-pong pair)
    GParams *h_prm = nullptr;     // pinned
    int64_t hist_len = 0;
    // graph cache: two instances of the cycle graph that differ only in where the carried state is copied for the host,
    // so that cycle c+1 can be enqueued before the host has looked at the outcome of cycle c
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    GDev key;
    bool have_graph = false;
    // profile mode: eager launches with HIP events around every Arnoldi (SpMV) kernel
    bool profile = false;
    std::vector<hipEvent_t> pev;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
    npg_halo *halo = nullptr;
    static constexpr int kMaxG = 512;
};

template <int L, int JB>
static void launch_arnoldi(const GDev &d, int j, hipStream_t st) {
    hipLaunchKernelGGL((k_gmres_arnoldi<L, JB>), dim3(d.G1), dim3(kBlock), 0, st, d, j);
}

template <int L>
static void launch_cycle_L(const GDev &d, hipStream_t st, hipEvent_t *pev) {
    for (int j = 0; j < d.mem; ++j) {
        const int nb = j + 1;
        if (pev) hipEventRecord(pev[2 * j], st);
        if (nb <= 4) {
            launch_arnoldi<L, 4>(d, j, st);
            if (pev) hipEventRecord(pev[2 * j + 1], st);
            hipLaunchKernelGGL(k_gmres_orth1<4>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
            hipLaunchKernelGGL(k_gmres_orth2<4>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
        } else if (nb <= 8) {
            launch_arnoldi<L, 8>(d, j, st);
            if (pev) hipEventRecord(pev[2 * j + 1], st);
            hipLaunchKernelGGL(k_gmres_orth1<8>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
            hipLaunchKernelGGL(k_gmres_orth2<8>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
        } else if (nb <= 12) {
            launch_arnoldi<L, 12>(d, j, st);
            if (pev) hipEventRecord(pev[2 * j + 1], st);
            hipLaunchKernelGGL(k_gmres_orth1<12>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
            hipLaunchKernelGGL(k_gmres_orth2<12>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
        } else if (nb <= 16) {
            launch_arnoldi<L, 16>(d, j, st);
            if (pev) hipEventRecord(pev[2 * j + 1], st);
            hipLaunchKernelGGL(k_gmres_orth1<16>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
            hipLaunchKernelGGL(k_gmres_orth2<16>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
        } else if (nb <= 20) {
            launch_arnoldi<L, 20>(d, j, st);
            if (pev) hipEventRecord(pev[2 * j + 1], st);
            hipLaunchKernelGGL(k_gmres_orth1<20>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
            hipLaunchKernelGGL(k_gmres_orth2<20>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
        } else {
            launch_arnoldi<L, 30>(d, j, st);
            if (pev) hipEventRecord(pev[2 * j + 1], st);
            hipLaunchKernelGGL(k_gmres_orth1<30>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
            hipLaunchKernelGGL(k_gmres_orth2<30>, dim3(d.G2), dim3(kBlock), 0, st, d, j);
        }
    }
    hipLaunchKernelGGL(k_gmres_update, dim3(d.G2), dim3(kBlock), 0, st, d);
    hipLaunchKernelGGL(k_gmres_residual<L>, dim3(d.G1), dim3(kBlock), 0, st, d);
}

static void launch_residual(const GDev &d, int lanes, hipStream_t st) {
    switch (lanes) {
        case 4: hipLaunchKernelGGL(k_gmres_residual<4>, dim3(d.G1), dim3(kBlock), 0, st, d); break;
        case 8: hipLaunchKernelGGL(k_gmres_residual<8>, dim3(d.G1), dim3(kBlock), 0, st, d); break;
        case 16: hipLaunchKernelGGL(k_gmres_residual<16>, dim3(d.G1), dim3(kBlock), 0, st, d); break;
        default: hipLaunchKernelGGL(k_gmres_residual<32>, dim3(d.G1), dim3(kBlock), 0, st, d); break;
    }
}

static void launch_cycle(const GDev &d, int lanes, hipStream_t st, hipEvent_t *pev) {
    switch (lanes) {
        case 4: launch_cycle_L<4>(d, st, pev); break;
        case 8: launch_cycle_L<8>(d, st, pev); break;
        case 16: launch_cycle_L<16>(d, st, pev); break;
        default: launch_cycle_L<32>(d, st, pev); break;
    }
}

NPG_API int npg_gmres_create(npg_ctx *ctx, int64_t n, int memory, npg_gmres **out) {
    NPG_REQUIRE(ctx && out && n > 0, "npg_gmres_create: bad argument");
    NPG_REQUIRE(memory >= 1 && memory <= kMaxMem, "npg_gmres_create: memory must be in [1,%d]", kMaxMem);
    NPG_REQUIRE(n < INT32_MAX, "npg_gmres_create: n exceeds int32 row indices");
    npg_gmres *ws = new npg_gmres();
    ws->ctx = ctx;
    ws->n = n;
    ws->mem = memory;
    ws->ld = (int)((n + 31) / 32 * 32);
    NPG_HIP(hipSetDevice(ctx->device));
    const size_t vb = (size_t)ws->ld * sizeof(double);
    NPG_HIP(hipMalloc((void **)&ws->V, vb * memory));
    NPG_HIP(hipMalloc((void **)&ws->w, vb));
    NPG_HIP(hipMalloc((void **)&ws->wt, vb));
    const size_t pb = (size_t)npg_gmres::kMaxG * kPartStride * sizeof(double);
    NPG_HIP(hipMalloc((void **)&ws->P1, pb));
    NPG_HIP(hipMalloc((void **)&ws->P2, pb));
    NPG_HIP(hipMalloc((void **)&ws->P3, pb));
    NPG_HIP(hipMalloc((void **)&ws->PR, pb));
    NPG_HIP(hipMalloc((void **)&ws->C, sizeof(Snap)));
    NPG_HIP(hipMalloc((void **)&ws->T, sizeof(Snap) * (memory + 1)));
    NPG_HIP(hipMalloc((void **)&ws->c, sizeof(double) * kPartStride));
    NPG_HIP(hipMalloc((void **)&ws->s, sizeof(double) * kPartStride));
    NPG_HIP(hipMalloc((void **)&ws->z, sizeof(double) * kPartStride));
    NPG_HIP(hipMalloc((void **)&ws->R, sizeof(double) * kPartStride * (kPartStride + 1) / 2));
    NPG_HIP(hipMalloc((void **)&ws->hcol1, sizeof(double) * kPartStride * kPartStride));
    NPG_HIP(hipMalloc((void **)&ws->hcol2, sizeof(double) * kPartStride * kPartStride));
    NPG_HIP(hipMalloc((void **)&ws->ci, sizeof(ColInfo) * kPartStride));
    ws->hist_cap = (int)std::min<int64_t>(2 * n + 2, 1 << 22);
    NPG_HIP(hipMalloc((void **)&ws->hist, sizeof(double) * ws->hist_cap));
    NPG_HIP(hipMalloc((void **)&ws->prm, sizeof(GParams)));
    NPG_HIP(hipHostMalloc((void **)&ws->h_C, 2 * sizeof(Snap), hipHostMallocDefault));
    NPG_HIP(hipEventCreateWithFlags(&ws->ev[0], hipEventDisableTiming));
    NPG_HIP(hipEventCreateWithFlags(&ws->ev[1], hipEventDisableTiming));
    NPG_HIP(hipHostMalloc((void **)&ws->h_prm, sizeof(GParams), hipHostMallocDefault));
    NPG_HIP(hipMemsetAsync(ws->V, 0, vb * memory, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->w, 0, vb, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->wt, 0, vb, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->c, 0, sizeof(double) * kPartStride, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->s, 0, sizeof(double) * kPartStride, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->z, 0, sizeof(double) * kPartStride, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->hcol1, 0, sizeof(double) * kPartStride * kPartStride, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->hcol2, 0, sizeof(double) * kPartStride * kPartStride, ctx->stream));
    NPG_HIP(hipMemsetAsync(ws->ci, 0, sizeof(ColInfo) * kPartStride, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    *out = ws;
    return NPG_OK;
}

NPG_API int npg_gmres_destroy(npg_gmres *ws) {
    if (!ws) return NPG_OK;
    hipStreamSynchronize(ws->ctx->stream);
    for (int k = 0; k < 2; ++k) {
        if (ws->exec[k]) hipGraphExecDestroy(ws->exec[k]);
        if (ws->graph[k]) hipGraphDestroy(ws->graph[k]);
        if (ws->ev[k]) hipEventDestroy(ws->ev[k]);
    }
    for (hipEvent_t e : ws->pev) hipEventDestroy(e);
    void *ptrs[] = {ws->V, ws->w, ws->wt, ws->P1, ws->P2, ws->P3, ws->PR, ws->C, ws->T, ws->c, ws->s,
                    ws->z, ws->R, ws->hcol1, ws->hcol2, ws->ci, ws->hist, ws->prm};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (ws->h_C) hipHostFree(ws->h_C);
    if (ws->h_prm) hipHostFree(ws->h_prm);
    delete ws;
    return NPG_OK;
}

NPG_API int npg_gmres_set_halo(npg_gmres *ws, npg_halo *h) {
    NPG_REQUIRE(ws, "npg_gmres_set_halo: NULL workspace");
    ws->halo = h;
    ws->have_graph = false;
    return NPG_OK;
}

NPG_API int npg_gmres_solve(npg_gmres *ws, const npg_csr *A, int precond_kind, double precond_scalar,
                            const npg_vec *precond_diag, const npg_vec *y, npg_vec *x, double atol, double rtol,
                            int64_t itmax, double reorth_eta, npg_solve_stats *stats) {
    NPG_REQUIRE(ws && A && y && x, "npg_gmres_solve: NULL argument");
    NPG_REQUIRE(A->m == ws->n && A->n == ws->n && y->n == ws->n && x->n == ws->n,
                "npg_gmres_solve: workspace is for n=%lld but A is %lldx%lld, y has %lld, x has %lld", (long long)ws->n,
                (long long)A->m, (long long)A->n, (long long)y->n, (long long)x->n);
    NPG_REQUIRE(precond_kind == NPG_PRECOND_NONE || precond_kind == NPG_PRECOND_SCALAR ||
                    (precond_kind == NPG_PRECOND_DIAG && precond_diag && precond_diag->n == ws->n),
                "npg_gmres_solve: bad preconditioner");
    NPG_REQUIRE(ws->halo == nullptr, "npg_gmres_solve: distributed solves go through npg_dist_* (halo set)");
    const auto t0 = std::chrono::steady_clock::now();
    npg_ctx *ctx = ws->ctx;
    hipStream_t st = ctx->stream;

    GDev d;
    memset(&d, 0, sizeof d);
    d.rowptr = A->rowptr;
    d.col = A->col;
    d.val = A->val;
    d.tile_ptr = A->tile_ptr;
    d.ntiles = A->ntiles;
    d.n = (int)ws->n;
    d.ld = ws->ld;
    d.mem = ws->mem;
    d.pkind = precond_kind;
    d.pscalar = precond_scalar;
    d.pdiag = precond_kind == NPG_PRECOND_DIAG ? precond_diag->d : nullptr;
    d.b = y->d;
    d.x = x->d;
    d.V = ws->V;
    d.w = ws->w;
    d.wt = ws->wt;
    d.P1 = ws->P1;
    d.P2 = ws->P2;
    d.P3 = ws->P3;
    d.PR = ws->PR;
    d.G1 = std::max(1, std::min<int>(A->ntiles, std::min(npg_gmres::kMaxG, 2 * ctx->num_cu)));
    d.G2 = (int)std::max<int64_t>(1, std::min<int64_t>((ws->n + kBlock - 1) / kBlock, d.G1));
    d.C = ws->C;
    d.T = ws->T;
    d.c = ws->c;
    d.s = ws->s;
    d.z = ws->z;
    d.R = ws->R;
    d.hcol1 = ws->hcol1;
    d.hcol2 = ws->hcol2;
    d.ci = ws->ci;
    d.hist = ws->hist;
    d.hist_cap = ws->hist_cap;
    d.prm = ws->prm;

    if (itmax <= 0) itmax = 2 * ws->n;
    ws->h_prm->atol = atol;
    ws->h_prm->rtol = rtol;
    ws->h_prm->eta2 = reorth_eta <= 0.0 ? -1.0 : reorth_eta * reorth_eta;
    ws->h_prm->btol = std::pow(2.220446049250313e-16, 0.75);
    ws->h_prm->itmax = itmax;
    NPG_HIP(hipMemcpyAsync(ws->prm, ws->h_prm, sizeof(GParams), hipMemcpyHostToDevice, st));
    Snap c0{};
    c0.first = 1;
    *ws->h_C = c0;
    NPG_HIP(hipMemcpyAsync(ws->C, ws->h_C, sizeof(Snap), hipMemcpyHostToDevice, st));

    // (re)capture the per-cycle graph when any baked-in argument changed
    if (!ws->have_graph || memcmp(&ws->key, &d, sizeof(GDev)) != 0) {
        NPG_HIP(hipStreamSynchronize(st));
        for (int k = 0; k < 2; ++k) {
            if (ws->exec[k]) hipGraphExecDestroy(ws->exec[k]);
            if (ws->graph[k]) hipGraphDestroy(ws->graph[k]);
            ws->exec[k] = nullptr;
            ws->graph[k] = nullptr;
            NPG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            launch_cycle(d, A->lanes, st, nullptr);
            NPG_HIP(hipMemcpyAsync(ws->h_C + k, ws->C, sizeof(Snap), hipMemcpyDeviceToHost, st));
            NPG_HIP(hipStreamEndCapture(st, &ws->graph[k]));
            NPG_HIP(hipGraphInstantiate(&ws->exec[k], ws->graph[k], nullptr, nullptr, 0));
        }
        memcpy(&ws->key, &d, sizeof d);
        ws->have_graph = true;
    }

    // first true residual, then cycles until the carried state says done
    launch_residual(d, A->lanes, st);
    NPG_HIP(hipGetLastError());
    const int64_t max_cycles = (itmax + ws->mem - 1) / ws->mem + 1;
    Snap last{};
    if (!ws->profile) {
        // cycle c+1 is enqueued before the host reads the outcome of cycle c: the device never idles waiting for the
        // host, and a cycle launched after convergence costs only its early-exit kernels
        NPG_HIP(hipGraphLaunch(ws->exec[0], st));
        NPG_HIP(hipEventRecord(ws->ev[0], st));
        for (int64_t cyc = 0;; ++cyc) {
            const int cur = (int)(cyc & 1), nxt = cur ^ 1;
            const bool more = cyc + 1 < max_cycles;
            if (more) {
                NPG_HIP(hipGraphLaunch(ws->exec[nxt], st));
                NPG_HIP(hipEventRecord(ws->ev[nxt], st));
            }
            NPG_HIP(hipEventSynchronize(ws->ev[cur]));
            last = ws->h_C[cur];
            if (last.done != 0 || !more) break;
        }
        NPG_HIP(hipStreamSynchronize(st));
    } else {
        if (ws->pev.empty()) {
            ws->pev.resize(2 * ws->mem);
            for (auto &e : ws->pev) NPG_HIP(hipEventCreate(&e));
        }
        for (int64_t cyc = 0; cyc < max_cycles; ++cyc) {
            launch_cycle(d, A->lanes, st, ws->pev.data());
            NPG_HIP(hipMemcpyAsync(ws->h_C, ws->C, sizeof(Snap), hipMemcpyDeviceToHost, st));
            NPG_HIP(hipStreamSynchronize(st));
            last = ws->h_C[0];
            if (last.done != 0) break;      // the last (partial) cycle is not counted: some of its kernels exit early
            for (int j = 0; j < ws->mem; ++j) {
                float ms = 0.f;
                NPG_HIP(hipEventElapsedTime(&ms, ws->pev[2 * j], ws->pev[2 * j + 1]));
                ws->prof_ms += ms;
                ws->prof_launches += 1;
            }
        }
    }
    ws->hist_len = std::min<int64_t>((int64_t)last.iter + 1, ws->hist_cap);
    if (stats) {
        stats->solved = (last.done == 1 || last.done == 4) ? 1 : 0;
        stats->niter = last.iter;
        stats->npass = last.npass;
        stats->status = last.done;
        stats->nreorth = last.nreorth;
        stats->reserved = 0;
        stats->rnorm0 = last.rnorm0;
        stats->rnorm = last.rnorm;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return NPG_OK;
}

NPG_API int npg_gmres_set_profile(npg_gmres *ws, int on) {
    NPG_REQUIRE(ws, "npg_gmres_set_profile: NULL workspace");
    ws->profile = on != 0;
    ws->prof_ms = 0.0;
    ws->prof_launches = 0;
    return NPG_OK;
}

NPG_API int npg_gmres_get_profile(npg_gmres *ws, double *ms_total, int64_t *launches) {
    NPG_REQUIRE(ws && ms_total && launches, "npg_gmres_get_profile: NULL argument");
    *ms_total = ws->prof_ms;
    *launches = ws->prof_launches;
    return NPG_OK;
}

NPG_API int64_t npg_gmres_history(npg_gmres *ws, double *buf, int64_t cap) {
    if (!ws || !buf || cap <= 0) return 0;
    const int64_t k = std::min<int64_t>(cap, ws->hist_len);
    if (hipMemcpy(buf, ws->hist, (size_t)k * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return k;
}

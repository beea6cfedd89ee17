// ILU(0) on the device: the inner preconditioner of the reference's block-diagonal preconditioner on the GPU,
//     P_prec = KrylovPreconditioners.kp_ilu0(P);  CgPreconditioner(P, P_prec; ldiv=true, itmax=100)
// (/root/reference/src/preconditioners.jl:101-107; KrylovPreconditioners.jl 0.3.7 hands that to the vendor library's csrilu02 +
// two csrsv2 triangular solves).  Same algorithm here, written out:
//   * analysis (host, index work on the sparsity pattern): row i of the factorisation and of the lower solve waits for the rows
//     j < i in its pattern, row i of the upper solve for the rows j > i - rows are grouped into LEVELS of mutually independent
//     rows (level of i = 1 + the highest level among the rows it waits for);
//   * factorisation (IKJ, in place on a copy of the values, no fill outside A's pattern): one launch per level, one wavefront per
//     row - for every k < i of the row in ascending order: l_ik = a_ik / u_kk, then a_ij -= l_ik u_kj for the j > k that row i
//     holds; the row lives in LDS while it is worked on;
//   * z = U^-1 L^-1 r: one launch per level and triangle (L has a unit diagonal), sixteen lanes per row; the launch sequence of
//     an (r, z) pair is captured into a hipGraph and replayed;
//   * preconditioned CG around it (Krylov.jl's cg with M = the factors, ldiv = true: gamma = r'z, stop at sqrt(gamma) <= atol +
//     rtol sqrt(gamma_0)), scalars through the host - an iteration is thousands of level launches, two 8-byte copies do not show.
// A level-scheduled triangular solve is bound by the number of levels (one dependent launch each), not by bandwidth: on the
// RCM-ordered P2 friction block that is thousands of levels of a few hundred rows.  It is here because the reference has it; the
// multigrid preconditioner (mg.hip) is what this library recommends for the inversion.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "common.h"

namespace npg {

constexpr int kIluMaxRow = 1024;          // longest row the factorisation kernel stages in LDS
constexpr int kTriLanes = 16;             // lanes per row in the triangular solves

// rows[0, nrows) of one level: row i <- its ILU(0) row.  One wavefront (workgroup of 64) per row.
__global__ void __launch_bounds__(64) k_ilu0_level(const int32_t *__restrict__ rows, int nrows, const int64_t *__restrict__ rp,
                                                   const int32_t *__restrict__ col, const int64_t *__restrict__ diag,
                                                   double *__restrict__ lu, int *__restrict__ bad) {
    __shared__ double v[kIluMaxRow];
    __shared__ int32_t c[kIluMaxRow];
    __shared__ double lik_s;
    for (int q = blockIdx.x; q < nrows; q += gridDim.x) {
        const int i = rows[q];
        const int64_t a = rp[i];
        const int len = (int)(rp[i + 1] - a), nlow = (int)(diag[i] - a);
        __syncthreads();
        for (int e = threadIdx.x; e < len; e += 64) {
            v[e] = lu[a + e];
            c[e] = col[a + e];
        }
        __syncthreads();
        for (int kk = 0; kk < nlow; ++kk) {
            const int k = c[kk];
            const int64_t dk = diag[k];
            if (threadIdx.x == 0) {
                const double l = v[kk] / lu[dk];
                v[kk] = l;
                lik_s = l;
            }
            __syncthreads();
            const double lik = lik_s;
            const int64_t e1 = rp[k + 1];
            for (int64_t jj = dk + 1 + threadIdx.x; jj < e1; jj += 64) {
                const int j = col[jj];
                int lo = kk + 1, hi = len - 1;            // the columns of a row ascend: j > k sits behind position kk
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int cm = c[mid];
                    if (cm == j) {
                        v[mid] -= lik * lu[jj];           // (distinct j, distinct positions: no two lanes meet)
                        break;
                    }
                    if (cm < j) lo = mid + 1; else hi = mid - 1;
                }
            }
            __syncthreads();
        }
        for (int e = threadIdx.x; e < len; e += 64) lu[a + e] = v[e];
        // the row's own pivot: rows of later levels divide by it.  A zero or non-finite pivot is counted (cuSPARSE's csrilu02 reports
        // the position of a zero pivot; KrylovPreconditioners.kp_ilu0 of the reference goes through it) - the host fails the call
        if (threadIdx.x == 0) {
            const double ukk = v[nlow];
            if (!(fabs(ukk) > 0.0) || !(fabs(ukk) < 1e300)) atomicAdd(bad, 1);
        }
    }
}

// one level of a triangular solve, kTriLanes lanes per row.  lower: out_i = in_i - sum_{j<i} L_ij out_j (unit diagonal);
// upper: out_i = (in_i - sum_{j>i} U_ij out_j) / U_ii
template <bool LOWER>
__global__ void __launch_bounds__(256) k_tri_level(const int32_t *__restrict__ rows, int nrows, const int64_t *__restrict__ rp,
                                                   const int32_t *__restrict__ col, const int64_t *__restrict__ diag,
                                                   const double *__restrict__ lu, const double *__restrict__ in,
                                                   double *__restrict__ out) {
    const int lane = threadIdx.x & (kTriLanes - 1);
    const int q = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kTriLanes);
    const bool live = q < nrows;
    const int i = live ? rows[q] : 0;
    double s = 0.0;
    if (live) {
        const int64_t b = LOWER ? rp[i] : diag[i] + 1, e = LOWER ? diag[i] : rp[i + 1];
        for (int64_t k = b + lane; k < e; k += kTriLanes) s += lu[k] * out[col[k]];
    }
#pragma unroll
    for (int o = kTriLanes / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kTriLanes);
    if (live && lane == 0) out[i] = LOWER ? in[i] - s : (in[i] - s) / lu[diag[i]];
}

__global__ void __launch_bounds__(256) k_dot_part(const double *__restrict__ a, const double *__restrict__ b, int64_t n,
                                                  double *__restrict__ part) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += a[i] * b[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(256) k_dot_final(const double *__restrict__ part, int np, double *__restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}
// x += alpha p ; r -= alpha Ap
__global__ void k_pcg_update(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ p, const double *__restrict__ Ap,
                             double alpha, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * p[i];
        r[i] -= alpha * Ap[i];
    }
}
// p = z + beta p
__global__ void k_pcg_direction(double *__restrict__ p, const double *__restrict__ z, double beta, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = z[i] + beta * p[i];
}

}  // namespace npg

using namespace npg;

struct npg_ilu0 {
    npg_ctx *ctx = nullptr;
    int64_t n = 0, nnz = 0;
    int64_t *rowptr = nullptr, *diag = nullptr;          // device; diag[i] = position of (i, i)
    int32_t *col = nullptr;
    double *lu = nullptr;                                // L (strictly lower, unit diagonal implied) and U in A's pattern
    int32_t *rows_l = nullptr, *rows_u = nullptr;        // rows in level order (lower: ascending dependencies, upper: descending)
    std::vector<int32_t> lp, up;                         // level offsets into rows_l / rows_u (host)
    double *t = nullptr;                                 // L^-1 r
    // captured z = U^-1 L^-1 r of the last (r, z) pair
    hipGraphExec_t graph = nullptr;
    const double *g_r = nullptr;
    double *g_z = nullptr;
    bool use_graph = true;
    // CG workspace
    double *r = nullptr, *z = nullptr, *p = nullptr, *Ap = nullptr, *part = nullptr, *scal = nullptr;
    double *h_scal = nullptr;
};

static void ilu_drop_graph(npg_ilu0 *m) {
    if (m->graph) hipGraphExecDestroy(m->graph);
    m->graph = nullptr;
    m->g_r = nullptr;
    m->g_z = nullptr;
}

NPG_API int npg_ilu0_destroy(npg_ilu0 *m) {
    if (!m) return NPG_OK;
    hipStreamSynchronize(m->ctx->stream);
    ilu_drop_graph(m);
    for (void *p : {(void *)m->rowptr, (void *)m->diag, (void *)m->col, (void *)m->lu, (void *)m->rows_l, (void *)m->rows_u, (void *)m->t,
                    (void *)m->r, (void *)m->z, (void *)m->p, (void *)m->Ap, (void *)m->part, (void *)m->scal})
        if (p) hipFree(p);
    if (m->h_scal) hipHostFree(m->h_scal);
    delete m;
    return NPG_OK;
}

static int ilu_factor(npg_ilu0 *m, const npg_csr *A) {
    hipStream_t st = m->ctx->stream;
    NPG_HIP(hipMemcpyAsync(m->lu, A->val, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
    int *bad = nullptr, nbad = 0;
    NPG_HIP(hipMalloc((void **)&bad, sizeof(int)));
    NPG_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
    for (size_t l = 0; l + 1 < m->lp.size(); ++l) {
        const int nr = m->lp[l + 1] - m->lp[l];
        hipLaunchKernelGGL(k_ilu0_level, dim3(std::min(nr, 65535)), dim3(64), 0, st, (const int32_t *)m->rows_l + m->lp[l], nr,
                           (const int64_t *)m->rowptr, (const int32_t *)m->col, (const int64_t *)m->diag, m->lu, bad);
    }
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, st);
    const hipError_t e3 = hipStreamSynchronize(st);
    hipFree(bad);
    NPG_HIP(e1);
    NPG_HIP(e2);
    NPG_HIP(e3);
    NPG_REQUIRE(nbad == 0, "ILU(0): %d zero or non-finite pivot(s) - the matrix has no ILU(0) factorisation in its own pattern "
                           "(csrilu02 would report a zero pivot)", nbad);
    return NPG_OK;
}

NPG_API int npg_ilu0_create(npg_ctx *ctx, const npg_csr *A, npg_ilu0 **out) {
    NPG_REQUIRE(ctx && A && out, "npg_ilu0_create: NULL argument");
    NPG_REQUIRE(A->m == A->n && A->m > 0, "npg_ilu0_create: the matrix must be square");
    NPG_REQUIRE(A->nnode() == 0 && !A->packed && !A->uperm, "npg_ilu0_create: a plain-CSR matrix is required");
    NPG_REQUIRE(A->m < (int64_t)1 << 31, "npg_ilu0_create: too many rows");
    const int64_t n = A->m, nnz = A->nnz;
    NPG_HIP(hipSetDevice(ctx->device));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<int32_t> col((size_t)nnz);
    NPG_HIP(hipMemcpy(col.data(), A->col, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int64_t *rp = A->h_rowptr.data();
    std::vector<int64_t> diag((size_t)n);
    std::vector<int32_t> levl((size_t)n), levu((size_t)n);
    int32_t nl = 0, nu = 0;
    for (int64_t i = 0; i < n; ++i) {
        NPG_REQUIRE(rp[i + 1] - rp[i] <= kIluMaxRow, "npg_ilu0_create: row %lld has %lld entries (limit %d)", (long long)i,
                    (long long)(rp[i + 1] - rp[i]), kIluMaxRow);
        int64_t d = -1;
        int32_t lev = 0;
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
            NPG_REQUIRE(k == rp[i] || col[k] > col[k - 1], "npg_ilu0_create: the columns of row %lld do not ascend", (long long)i);
            if (col[k] < i) lev = std::max(lev, levl[col[k]] + 1);
            if (col[k] == i) d = k;
        }
        NPG_REQUIRE(d >= 0, "npg_ilu0_create: row %lld has no diagonal entry", (long long)i);
        diag[i] = d;
        levl[i] = lev;
        nl = std::max(nl, lev + 1);
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        int32_t lev = 0;
        for (int64_t k = diag[i] + 1; k < rp[i + 1]; ++k) lev = std::max(lev, levu[col[k]] + 1);
        levu[i] = lev;
        nu = std::max(nu, lev + 1);
    }
    npg_ilu0 *m = new npg_ilu0();
    m->ctx = ctx;
    m->n = n;
    m->nnz = nnz;
    // rows by level (counting sort; rows of a level keep their ascending order: neighbouring rows read neighbouring data)
    auto by_level = [&](const std::vector<int32_t> &lev, int32_t nlev, std::vector<int32_t> &ptr, std::vector<int32_t> &rows) {
        ptr.assign((size_t)nlev + 1, 0);
        for (int64_t i = 0; i < n; ++i) ptr[(size_t)lev[i] + 1]++;
        for (int32_t l = 0; l < nlev; ++l) ptr[(size_t)l + 1] += ptr[l];
        rows.resize((size_t)n);
        std::vector<int32_t> at(ptr.begin(), ptr.end() - 1);
        for (int64_t i = 0; i < n; ++i) rows[(size_t)at[lev[i]]++] = (int32_t)i;
    };
    std::vector<int32_t> rl, ru;
    by_level(levl, nl, m->lp, rl);
    by_level(levu, nu, m->up, ru);
    struct Guard {
        npg_ilu0 *m;
        bool keep = false;
        ~Guard() { if (!keep) npg_ilu0_destroy(m); }
    } guard{m};
    NPG_HIP(hipMalloc((void **)&m->rowptr, (size_t)(n + 1) * sizeof(int64_t)));
    NPG_HIP(hipMalloc((void **)&m->diag, (size_t)n * sizeof(int64_t)));
    NPG_HIP(hipMalloc((void **)&m->col, std::max<size_t>(1, (size_t)nnz) * sizeof(int32_t)));
    NPG_HIP(hipMalloc((void **)&m->lu, std::max<size_t>(1, (size_t)nnz) * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&m->rows_l, (size_t)n * sizeof(int32_t)));
    NPG_HIP(hipMalloc((void **)&m->rows_u, (size_t)n * sizeof(int32_t)));
    for (double **p : {&m->t, &m->r, &m->z, &m->p, &m->Ap}) NPG_HIP(hipMalloc((void **)p, (size_t)n * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&m->part, 256 * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&m->scal, 8 * sizeof(double)));
    NPG_HIP(hipHostMalloc((void **)&m->h_scal, 8 * sizeof(double)));
    NPG_HIP(hipMemcpy(m->rowptr, rp, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(m->diag, diag.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(m->col, col.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(m->rows_l, rl.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    NPG_HIP(hipMemcpy(m->rows_u, ru.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
    // (graph replay under rocprofv3's tracer: see mg.hip)
    const bool traced = getenv("ROCPROFILER_LIBRARY_CTOR") || getenv("ROCPROF_OUTPUT_PATH") || getenv("ROCP_TOOL_LIBRARIES");
    m->use_graph = getenv("NPG_ILU_EAGER") ? atoi(getenv("NPG_ILU_EAGER")) == 0 : !traced;
    int rc = ilu_factor(m, A);
    if (rc) return rc;
    guard.keep = true;
    *out = m;
    return NPG_OK;
}

// the values of A changed (same pattern): factorise again
NPG_API int npg_ilu0_refactor(npg_ilu0 *m, const npg_csr *A) {
    NPG_REQUIRE(m && A, "npg_ilu0_refactor: NULL argument");
    NPG_REQUIRE(A->m == m->n && A->n == m->n && A->nnz == m->nnz && A->nnode() == 0, "npg_ilu0_refactor: not the matrix's pattern");
    NPG_HIP(hipStreamSynchronize(m->ctx->stream));
    return ilu_factor(m, A);
}

NPG_API int npg_ilu0_info(const npg_ilu0 *m, int64_t *levels_lower, int64_t *levels_upper, int64_t *nnz) {
    NPG_REQUIRE(m, "npg_ilu0_info: NULL handle");
    if (levels_lower) *levels_lower = (int64_t)m->lp.size() - 1;
    if (levels_upper) *levels_upper = (int64_t)m->up.size() - 1;
    if (nnz) *nnz = m->nnz;
    return NPG_OK;
}

// the factors' values in A's pattern (strictly lower part: L without its unit diagonal; the rest: U)
NPG_API int npg_ilu0_factors(const npg_ilu0 *m, double *host_values) {
    NPG_REQUIRE(m && host_values, "npg_ilu0_factors: NULL argument");
    NPG_HIP(hipStreamSynchronize(m->ctx->stream));
    NPG_HIP(hipMemcpy(host_values, m->lu, (size_t)m->nnz * sizeof(double), hipMemcpyDeviceToHost));
    return NPG_OK;
}

static void ilu_enqueue(npg_ilu0 *m, const double *r, double *z) {
    hipStream_t st = m->ctx->stream;
    for (size_t l = 0; l + 1 < m->lp.size(); ++l) {
        const int nr = m->lp[l + 1] - m->lp[l];
        hipLaunchKernelGGL(k_tri_level<true>, dim3((nr * kTriLanes + 255) / 256), dim3(256), 0, st, (const int32_t *)m->rows_l + m->lp[l], nr,
                           (const int64_t *)m->rowptr, (const int32_t *)m->col, (const int64_t *)m->diag, (const double *)m->lu, r, m->t);
    }
    for (size_t l = 0; l + 1 < m->up.size(); ++l) {
        const int nr = m->up[l + 1] - m->up[l];
        hipLaunchKernelGGL(k_tri_level<false>, dim3((nr * kTriLanes + 255) / 256), dim3(256), 0, st, (const int32_t *)m->rows_u + m->up[l], nr,
                           (const int64_t *)m->rowptr, (const int32_t *)m->col, (const int64_t *)m->diag, (const double *)m->lu,
                           (const double *)m->t, z);
    }
}

// z = U^-1 L^-1 r on raw device pointers (r and z must not alias)
static int ilu_apply_raw(npg_ilu0 *m, const double *r, double *z) {
    hipStream_t st = m->ctx->stream;
    if (!m->use_graph) {
        ilu_enqueue(m, r, z);
        NPG_HIP(hipGetLastError());
        return NPG_OK;
    }
    if (!m->graph || m->g_r != r || m->g_z != z) {
        ilu_drop_graph(m);
        hipGraph_t g = nullptr;
        NPG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        ilu_enqueue(m, r, z);
        const hipError_t ec = hipStreamEndCapture(st, &g);
        if (ec != hipSuccess && g) hipGraphDestroy(g);
        NPG_HIP(ec);
        const hipError_t ei = hipGraphInstantiate(&m->graph, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        NPG_HIP(ei);
        m->g_r = r;
        m->g_z = z;
    }
    NPG_HIP(hipGraphLaunch(m->graph, st));
    return NPG_OK;
}

NPG_API int npg_ilu0_apply(npg_ilu0 *m, const npg_vec *r, npg_vec *z) {
    NPG_REQUIRE(m && r && z, "npg_ilu0_apply: NULL argument");
    NPG_REQUIRE(r->n == m->n && z->n == m->n && r->d != z->d, "npg_ilu0_apply: vectors must have %lld entries and not alias", (long long)m->n);
    return ilu_apply_raw(m, r->d, z->d);
}

static int ilu_dot(npg_ilu0 *m, const double *a, const double *b, double *out) {
    hipStream_t st = m->ctx->stream;
    const int g = (int)std::min<int64_t>(256, (m->n + 255) / 256);
    hipLaunchKernelGGL(k_dot_part, dim3(g), dim3(256), 0, st, a, b, m->n, m->part);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(256), 0, st, (const double *)m->part, g, m->scal);
    NPG_HIP(hipMemcpyAsync(m->h_scal, m->scal, sizeof(double), hipMemcpyDeviceToHost, st));
    NPG_HIP(hipStreamSynchronize(st));
    *out = m->h_scal[0];
    return NPG_OK;
}

namespace npg {
// CG on A x = b with M = the ILU(0) factors (ldiv), warm start x - Krylov.jl's cg as the reference's CgPreconditioner calls it
// (src/preconditioners.jl:24-37): gamma = r'z, stop when sqrt(gamma) <= atol + rtol sqrt(gamma_0).  Raw device pointers.
int ilu_pcg_raw(npg_ilu0 *m, const npg_csr *A, const double *b, double *x, double atol, double rtol, int64_t itmax,
                npg_solve_stats *stats) {
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t st = m->ctx->stream;
    const int64_t n = m->n;
    const int grid = (int)std::min<int64_t>(2048, (n + 255) / 256);
    if (itmax <= 0) itmax = 2 * n;
    int rc;
    SpmvEpi e{};                                   // r = b - A x
    e.alpha = -1.0; e.beta = 1.0; e.c = b; e.y = m->r;
    if ((rc = spmv_epi(A, x, e))) return rc;
    if ((rc = ilu_apply_raw(m, m->r, m->z))) return rc;
    NPG_HIP(hipMemcpyAsync(m->p, m->z, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    double gamma = 0.0;
    if ((rc = ilu_dot(m, m->r, m->z, &gamma))) return rc;
    const double rnorm0 = std::sqrt(std::max(gamma, 0.0)), eps = atol + rtol * rnorm0;
    double rnorm = rnorm0;
    int64_t it = 0;
    int status = gamma == 0.0 ? 4 : (rnorm0 <= eps ? 1 : 0);
    while (status == 0) {
        if ((rc = spmv_raw(A, m->p, m->Ap, 1.0, 0.0))) return rc;
        double pAp = 0.0;
        if ((rc = ilu_dot(m, m->p, m->Ap, &pAp))) return rc;
        if (!(pAp > 0.0)) { status = 3; break; }                    // not positive definite along p / breakdown
        const double alpha = gamma / pAp;
        hipLaunchKernelGGL(k_pcg_update, dim3(grid), dim3(256), 0, st, x, m->r, (const double *)m->p, (const double *)m->Ap, alpha, n);
        if ((rc = ilu_apply_raw(m, m->r, m->z))) return rc;
        double g2 = 0.0;
        if ((rc = ilu_dot(m, m->r, m->z, &g2))) return rc;
        ++it;
        rnorm = std::sqrt(std::max(g2, 0.0));
        if (g2 != g2) { status = 3; break; }
        if (rnorm <= eps) { status = 1; break; }
        if (it >= itmax) { status = 2; break; }
        hipLaunchKernelGGL(k_pcg_direction, dim3(grid), dim3(256), 0, st, m->p, (const double *)m->z, g2 / gamma, n);
        gamma = g2;
    }
    NPG_HIP(hipGetLastError());
    if (stats) {
        *stats = npg_solve_stats{};
        stats->solved = (status == 1 || status == 4) ? 1 : 0;
        stats->niter = (int32_t)it;
        stats->status = status;
        stats->rnorm0 = rnorm0;
        stats->rnorm = rnorm;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return NPG_OK;
}
}  // namespace npg

NPG_API int npg_cg_ilu0_solve(npg_ilu0 *m, const npg_csr *A, const npg_vec *b, npg_vec *x, double atol, double rtol, int64_t itmax,
                              npg_solve_stats *stats) {
    NPG_REQUIRE(m && A && b && x, "npg_cg_ilu0_solve: NULL argument");
    NPG_REQUIRE(A->m == m->n && A->n == m->n && b->n == m->n && x->n == m->n && b->d != x->d,
                "npg_cg_ilu0_solve: A must be %lld x %lld, b and x of that length and not aliased", (long long)m->n, (long long)m->n);
    NPG_REQUIRE(atol >= 0.0 && rtol >= 0.0, "npg_cg_ilu0_solve: negative tolerance");
    return ilu_pcg_raw(m, A, b->d, x->d, atol, rtol, itmax, stats);
}

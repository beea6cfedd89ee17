// libnupgcm_host.so - the C ABI of include/nupgcm_hip.h on the HOST (plain C++17 + OpenMP, no HIP), for the reference's CPU()
// architecture (BASELINE.json configs[0]: "bowl3D h = 0.1 mesh, CPU() architecture, 5 timesteps of bowl_mixing - plumbing, runs
// without a GPU"; /root/reference/src/architectures.jl:4-20, src/iterative_solvers.jl:42-58).
//
// What it is: the subset of the entry points that nupgcm_amd's Model(CPU(), ...) drives - context, vectors, plain-CSR matrices,
// SpMV, the element kernels of the path (src/inversion.jl:133-249, src/evolution.jl:209-296, src/model.jl:269-300,
// src/inputs.jl:87-137, src/timesteppers.jl:108-119) and Krylov.jl's restarted GMRES / CG for the sizes and closures where the
// reference's CPU() path iterates instead of factorising (src/iterative_solvers.jl:58).  Same names, argument meaning and
// error behaviour as the HIP library; handles are host memory.  The sparse LU the reference's CPU() path uses wherever it can
// (lu(A) + ldiv!, src/inversion.jl:55-58, src/evolution.jl:150-153) is a LIBRARY there too (UMFPACK): the Python host calls
// SuperLU through scipy for it (nupgcm_amd/iterative_solvers.py).
//
// What it is not: the oracle (oracle/ is numpy test infrastructure; nothing here derives from it) and not a fallback for GPU() -
// a process runs on one architecture, chosen explicitly (nupgcm_amd/_lib.py: select("host") through npg.CPU()).
//
// The element integrals restate csrc/fe.hip statement by statement in scalar loops: same tables, same order of the sums over
// quadrature points and over the cells of a row (row-owner assembly through the inverted index), so the host matrices agree with
// the device's to rounding and repeat bit for bit from run to run whatever OMP_NUM_THREADS is.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <omp.h>
#include <unistd.h>

#include "../../include/nupgcm_hip.h"

#define NPG_API extern "C" __attribute__((visibility("default")))

namespace {
thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
#define REQUIRE(cond, ...)          \
    do {                            \
        if (!(cond)) {              \
            set_error(__VA_ARGS__); \
            return NPG_EINVAL;      \
        }                           \
    } while (0)
}  // namespace

struct npg_ctx {
    std::chrono::steady_clock::time_point t0;
};
struct npg_vec {
    npg_ctx *ctx = nullptr;
    int64_t n = 0;
    double *d = nullptr;
    bool owns = true;
};
struct npg_csr {
    npg_ctx *ctx = nullptr;
    int64_t m = 0, n = 0, nnz = 0;
    std::vector<int64_t> rowptr;
    std::vector<int32_t> col;
    std::vector<double> val;
};

// ---- context -----------------------------------------------------------------------------------------------------------------
NPG_API const char *npg_last_error(void) { return g_err.c_str(); }
NPG_API int npg_ctx_create(int device, npg_ctx **out) {
    REQUIRE(out, "npg_ctx_create: NULL argument");
    (void)device;                       // the host has one "device"
    *out = new npg_ctx();
    return NPG_OK;
}
NPG_API int npg_ctx_destroy(npg_ctx *ctx) {
    delete ctx;
    return NPG_OK;
}
NPG_API int npg_ctx_sync(npg_ctx *) { return NPG_OK; }
NPG_API void *npg_ctx_stream(npg_ctx *) { return nullptr; }
NPG_API int npg_mem_status(npg_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    REQUIRE(ctx && free_bytes && total_bytes, "npg_mem_status: NULL argument");
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    *total_bytes = page * (size_t)sysconf(_SC_PHYS_PAGES);
    *free_bytes = page * (size_t)sysconf(_SC_AVPHYS_PAGES);
    return NPG_OK;
}
NPG_API int npg_device_name(npg_ctx *ctx, char *buf, size_t cap) {
    REQUIRE(ctx && buf && cap > 0, "npg_device_name: bad argument");
    snprintf(buf, cap, "host CPU (libnupgcm_host, %d OpenMP threads)", omp_get_max_threads());
    return NPG_OK;
}
NPG_API int npg_timer_start(npg_ctx *ctx) {
    REQUIRE(ctx, "npg_timer_start: NULL context");
    ctx->t0 = std::chrono::steady_clock::now();
    return NPG_OK;
}
NPG_API int npg_timer_stop(npg_ctx *ctx, double *ms) {
    REQUIRE(ctx && ms, "npg_timer_stop: NULL argument");
    *ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - ctx->t0).count();
    return NPG_OK;
}

// ---- vectors -----------------------------------------------------------------------------------------------------------------
NPG_API int npg_vec_create(npg_ctx *ctx, int64_t n, npg_vec **out) {
    REQUIRE(ctx && out && n >= 0, "npg_vec_create: bad argument");
    npg_vec *v = new npg_vec();
    v->ctx = ctx;
    v->n = n;
    v->d = (double *)calloc((size_t)std::max<int64_t>(n, 1), sizeof(double));
    if (!v->d) {
        delete v;
        set_error("npg_vec_create: out of memory (%lld doubles)", (long long)n);
        return NPG_ENOMEM;
    }
    *out = v;
    return NPG_OK;
}
NPG_API int npg_vec_destroy(npg_vec *v) {
    if (!v) return NPG_OK;
    if (v->owns) free(v->d);
    delete v;
    return NPG_OK;
}
NPG_API int npg_vec_view(npg_vec *v, int64_t offset, int64_t n, npg_vec **out) {
    REQUIRE(v && out && offset >= 0 && n >= 0 && offset + n <= v->n, "npg_vec_view: window out of range");
    npg_vec *w = new npg_vec();
    w->ctx = v->ctx;
    w->n = n;
    w->d = v->d + offset;
    w->owns = false;
    *out = w;
    return NPG_OK;
}
NPG_API int64_t npg_vec_len(const npg_vec *v) { return v ? v->n : -1; }
NPG_API int npg_vec_upload(npg_vec *v, const double *host) {
    REQUIRE(v && host, "npg_vec_upload: NULL argument");
    memcpy(v->d, host, (size_t)v->n * sizeof(double));
    return NPG_OK;
}
NPG_API int npg_vec_download(const npg_vec *v, double *host) {
    REQUIRE(v && host, "npg_vec_download: NULL argument");
    memcpy(host, v->d, (size_t)v->n * sizeof(double));
    return NPG_OK;
}
// v[i] = host[perm[i]]   /   host[i] = v[perm[i]]        (as the HIP library: the gathers of src/model.jl:274-275,282,312)
NPG_API int npg_vec_upload_perm(npg_vec *v, const double *host, const int64_t *perm) {
    REQUIRE(v && host && perm, "npg_vec_upload_perm: NULL argument");
    for (int64_t i = 0; i < v->n; ++i) v->d[i] = host[perm[i]];
    return NPG_OK;
}
NPG_API int npg_vec_download_perm(const npg_vec *v, double *host, const int64_t *perm) {
    REQUIRE(v && host && perm, "npg_vec_download_perm: NULL argument");
    for (int64_t i = 0; i < v->n; ++i) {
        REQUIRE(perm[i] >= 0 && perm[i] < v->n, "npg_vec_download_perm: index out of range");
        host[i] = v->d[perm[i]];
    }
    return NPG_OK;
}
NPG_API int npg_vec_fill(npg_vec *v, double a) {
    REQUIRE(v, "npg_vec_fill: NULL vector");
    std::fill(v->d, v->d + v->n, a);
    return NPG_OK;
}
NPG_API int npg_vec_copy(npg_vec *dst, const npg_vec *src) {
    REQUIRE(dst && src && dst->n == src->n, "npg_vec_copy: length mismatch");
    if (dst->d != src->d) memmove(dst->d, src->d, (size_t)dst->n * sizeof(double));
    return NPG_OK;
}
NPG_API int npg_vec_axpby(npg_vec *y, double a, const npg_vec *x, double b) {
    REQUIRE(y && x && y->n == x->n, "npg_vec_axpby: length mismatch");
    const int64_t n = y->n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y->d[i] = a * x->d[i] + (b == 0.0 ? 0.0 : b * y->d[i]);
    return NPG_OK;
}
static double dot_fixed(const double *x, const double *y, int64_t n) {
    // fixed blocking (independent of the thread count): bit-reproducible whatever OMP_NUM_THREADS is
    constexpr int64_t B = 4096;
    const int64_t nb = (n + B - 1) / B;
    std::vector<double> part((size_t)nb);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; ++b) {
        double s = 0.0;
        for (int64_t i = b * B; i < std::min(n, (b + 1) * B); ++i) s += x[i] * y[i];
        part[(size_t)b] = s;
    }
    double s = 0.0;
    for (double p : part) s += p;
    return s;
}
NPG_API int npg_vec_dot(const npg_vec *x, const npg_vec *y, double *out) {
    REQUIRE(x && y && out && x->n == y->n, "npg_vec_dot: length mismatch");
    *out = dot_fixed(x->d, y->d, x->n);
    return NPG_OK;
}
NPG_API int npg_vec_nrm2(const npg_vec *x, double *out) {
    REQUIRE(x && out, "npg_vec_nrm2: NULL argument");
    *out = std::sqrt(dot_fixed(x->d, x->d, x->n));
    return NPG_OK;
}
NPG_API int npg_vec_maxabs(const npg_vec *x, double *out, int *has_nan) {
    REQUIRE(x && out, "npg_vec_maxabs: NULL argument");
    double m = 0.0;
    int nan = 0;
    for (int64_t i = 0; i < x->n; ++i) {
        const double a = std::fabs(x->d[i]);
        if (a != a) nan = 1;
        else m = std::max(m, a);
    }
    *out = m;
    if (has_nan) *has_nan = nan;
    return NPG_OK;
}
NPG_API int npg_vec_is_constant(const npg_vec *x, double *value, int *is_constant) {
    REQUIRE(x && value && is_constant && x->n > 0, "npg_vec_is_constant: bad argument");
    double lo = x->d[0], hi = x->d[0];
    for (int64_t i = 1; i < x->n; ++i) {
        lo = std::min(lo, x->d[i]);
        hi = std::max(hi, x->d[i]);
    }
    *value = lo;
    *is_constant = lo == hi;
    return NPG_OK;
}
NPG_API int npg_vec_lincomb(npg_vec *y, int nterms, const double *coef, const npg_vec *const *xs) {
    REQUIRE(y && nterms >= 1 && nterms <= 8 && coef && xs, "npg_vec_lincomb: 1 to 8 terms");
    for (int k = 0; k < nterms; ++k) REQUIRE(xs[k] && xs[k]->n == y->n, "npg_vec_lincomb: length mismatch in term %d", k);
    const int64_t n = y->n;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < nterms; ++k) s += coef[k] * xs[k]->d[i];
        y->d[i] = s;
    }
    return NPG_OK;
}
NPG_API int npg_vec_mul(npg_vec *y, const npg_vec *d, const npg_vec *x) {
    REQUIRE(y && d && x && y->n == d->n && y->n == x->n, "npg_vec_mul: length mismatch");
    for (int64_t i = 0; i < y->n; ++i) y->d[i] = d->d[i] * x->d[i];
    return NPG_OK;
}

// ---- CSR matrices ------------------------------------------------------------------------------------------------------------------
static int csr_check(int64_t m, int64_t n, const std::vector<int64_t> &rp, const std::vector<int32_t> &col, const char *who) {
    REQUIRE(rp.size() == (size_t)m + 1 && rp[0] == 0, "%s: bad row offsets", who);
    for (int64_t r = 0; r < m; ++r) {
        REQUIRE(rp[r] <= rp[r + 1], "%s: row offsets must not decrease", who);
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            REQUIRE(col[k] >= 0 && col[k] < n, "%s: column index %d out of range in row %lld", who, col[k], (long long)r);
            REQUIRE(k == rp[r] || col[k - 1] < col[k], "%s: columns of row %lld are not strictly ascending", who, (long long)r);
        }
    }
    return NPG_OK;
}
NPG_API int npg_csr_create(npg_ctx *ctx, int64_t m, int64_t n, const int64_t *rowptr, const int32_t *colind, const double *val,
                           npg_csr **out) {
    REQUIRE(ctx && out && rowptr && m >= 0 && n >= 0 && n < INT32_MAX, "npg_csr_create: bad argument");
    npg_csr *A = new npg_csr();
    A->ctx = ctx;
    A->m = m;
    A->n = n;
    A->rowptr.assign(rowptr, rowptr + m + 1);
    A->nnz = rowptr[m];
    REQUIRE(A->nnz == 0 || colind, "npg_csr_create: NULL column array");
    A->col.assign(colind, colind + A->nnz);
    if (val) A->val.assign(val, val + A->nnz);
    else A->val.assign((size_t)A->nnz, 0.0);
    const int rc = csr_check(m, n, A->rowptr, A->col, "npg_csr_create");
    if (rc) {
        delete A;
        return rc;
    }
    *out = A;
    return NPG_OK;
}
NPG_API int npg_csr_create_from_csc(npg_ctx *ctx, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                                    const double *nzval, int drop_zeros, npg_csr **out) {
    REQUIRE(ctx && out && colptr && m >= 0 && n >= 0 && n < INT32_MAX, "npg_csr_create_from_csc: bad argument");
    const int64_t nz = colptr[n];
    REQUIRE(nz == 0 || (rowval && nzval), "npg_csr_create_from_csc: NULL arrays");
    npg_csr *A = new npg_csr();
    A->ctx = ctx;
    A->m = m;
    A->n = n;
    A->rowptr.assign((size_t)m + 1, 0);
    for (int64_t j = 0; j < n; ++j)
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            if (rowval[k] < 0 || rowval[k] >= m) {
                delete A;
                set_error("npg_csr_create_from_csc: row index out of range");
                return NPG_EINVAL;
            }
            if (!drop_zeros || nzval[k] != 0.0) ++A->rowptr[(size_t)rowval[k] + 1];
        }
    for (int64_t r = 0; r < m; ++r) A->rowptr[r + 1] += A->rowptr[r];
    A->nnz = A->rowptr[m];
    A->col.resize((size_t)A->nnz);
    A->val.resize((size_t)A->nnz);
    std::vector<int64_t> next(A->rowptr.begin(), A->rowptr.end() - 1);
    for (int64_t j = 0; j < n; ++j)            // columns ascending => every row's entries arrive in ascending column order
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k)
            if (!drop_zeros || nzval[k] != 0.0) {
                const int64_t s = next[(size_t)rowval[k]]++;
                A->col[(size_t)s] = (int32_t)j;
                A->val[(size_t)s] = nzval[k];
            }
    const int rc = csr_check(m, n, A->rowptr, A->col, "npg_csr_create_from_csc");
    if (rc) {
        delete A;
        return rc;
    }
    *out = A;
    return NPG_OK;
}
NPG_API int npg_csr_destroy(npg_csr *A) {
    delete A;
    return NPG_OK;
}
NPG_API int npg_csr_shape(const npg_csr *A, int64_t *m, int64_t *n, int64_t *nnz) {
    REQUIRE(A, "npg_csr_shape: NULL matrix");
    if (m) *m = A->m;
    if (n) *n = A->n;
    if (nnz) *nnz = A->nnz;
    return NPG_OK;
}
NPG_API int npg_csr_storage(const npg_csr *A, int64_t *nodes, int64_t *records, int64_t *csr_entries) {
    REQUIRE(A, "npg_csr_storage: NULL matrix");
    if (nodes) *nodes = 0;              // plain CSR only on the host
    if (records) *records = 0;
    if (csr_entries) *csr_entries = A->nnz;
    return NPG_OK;
}
NPG_API int npg_csr_spmv_bytes(const npg_csr *A, int64_t *matrix_bytes) {
    REQUIRE(A && matrix_bytes, "npg_csr_spmv_bytes: NULL argument");
    *matrix_bytes = 8 * (A->m + 1) + 12 * A->nnz;
    return NPG_OK;
}
NPG_API int npg_csr_block_nodes(npg_csr *A, int64_t, int64_t, double, int *blocked) {
    REQUIRE(A && blocked, "npg_csr_block_nodes: NULL argument");
    *blocked = 0;                       // the record layouts are HBM layouts: the host keeps plain CSR
    return NPG_OK;
}
NPG_API int npg_csr_download(const npg_csr *A, int64_t *rowptr, int32_t *colind, double *val) {
    REQUIRE(A, "npg_csr_download: NULL matrix");
    if (rowptr) memcpy(rowptr, A->rowptr.data(), A->rowptr.size() * sizeof(int64_t));
    if (colind && A->nnz) memcpy(colind, A->col.data(), (size_t)A->nnz * sizeof(int32_t));
    if (val && A->nnz) memcpy(val, A->val.data(), (size_t)A->nnz * sizeof(double));
    return NPG_OK;
}
NPG_API int npg_csr_to_csc(const npg_csr *A, int64_t *colptr, int64_t *rowval, double *nzval) {
    REQUIRE(A && colptr && (A->nnz == 0 || (rowval && nzval)), "npg_csr_to_csc: NULL argument");
    std::fill(colptr, colptr + A->n + 1, 0);
    for (int64_t k = 0; k < A->nnz; ++k) ++colptr[A->col[(size_t)k] + 1];
    for (int64_t j = 0; j < A->n; ++j) colptr[j + 1] += colptr[j];
    std::vector<int64_t> next(colptr, colptr + A->n);
    for (int64_t r = 0; r < A->m; ++r)
        for (int64_t k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) {
            const int64_t s = next[(size_t)A->col[(size_t)k]]++;
            rowval[s] = r;
            nzval[s] = A->val[(size_t)k];
        }
    return NPG_OK;
}
NPG_API int npg_csr_clone(const npg_csr *A, npg_csr **out) {
    REQUIRE(A && out, "npg_csr_clone: NULL argument");
    *out = new npg_csr(*A);
    return NPG_OK;
}
NPG_API int npg_csr_zero_values(npg_csr *A) {
    REQUIRE(A, "npg_csr_zero_values: NULL matrix");
    std::fill(A->val.begin(), A->val.end(), 0.0);
    return NPG_OK;
}
NPG_API int npg_csr_combine(npg_csr *out, double a, const npg_csr *X, double b, const npg_csr *Y, const npg_csr *Z) {
    REQUIRE(out && X && Y && Z, "npg_csr_combine: NULL argument");
    REQUIRE(out->nnz == X->nnz && X->nnz == Y->nnz && Y->nnz == Z->nnz && out->m == X->m && X->m == Y->m && Y->m == Z->m,
            "npg_csr_combine: operands must share one sparsity pattern");
    for (int64_t k = 0; k < out->nnz; ++k) out->val[(size_t)k] = a * X->val[(size_t)k] + b * (Y->val[(size_t)k] + Z->val[(size_t)k]);
    return NPG_OK;
}
NPG_API int npg_csr_inv_diag(const npg_csr *A, npg_vec *d) {
    REQUIRE(A && d && A->m <= A->n && d->n == A->m, "npg_csr_inv_diag: shape mismatch");
    for (int64_t r = 0; r < A->m; ++r) {
        double v = 0.0;
        for (int64_t k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k)
            if (A->col[(size_t)k] == r) v = A->val[(size_t)k];
        d->d[r] = 1.0 / v;
    }
    return NPG_OK;
}
static void spmv_raw(const npg_csr *A, const double *x, double *y, double alpha, double beta) {
    const int64_t m = A->m;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; ++r) {
        double s = 0.0;
        for (int64_t k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k) s += A->val[(size_t)k] * x[A->col[(size_t)k]];
        y[r] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[r]);
    }
}
NPG_API int npg_spmv(const npg_csr *A, const npg_vec *x, npg_vec *y, double alpha, double beta) {
    REQUIRE(A && x && y && x->n == A->n && y->n == A->m && x->d != y->d, "npg_spmv: shape mismatch (A %lldx%lld, x %lld, y %lld)",
            A ? (long long)A->m : 0LL, A ? (long long)A->n : 0LL, x ? (long long)x->n : 0LL, y ? (long long)y->n : 0LL);
    spmv_raw(A, x->d, y->d, alpha, beta);
    return NPG_OK;
}

// ---- Krylov.jl's gmres! / cg! as the reference configures them (src/inversion.jl:74-94, src/evolution.jl:118-126) ---------------------
struct npg_gmres {
    npg_ctx *ctx = nullptr;
    int64_t n = 0;
    int mem = 20;
    std::vector<double> V, w, q, dx, hist;
};
struct npg_cg {
    npg_ctx *ctx = nullptr;
    int64_t n = 0;
    std::vector<double> r, z, p, Ap, hist;
};
static inline double pre(int kind, double scalar, const npg_vec *diag, int64_t i) {
    return kind == NPG_PRECOND_SCALAR ? scalar : (kind == NPG_PRECOND_DIAG ? diag->d[i] : 1.0);
}
NPG_API int npg_gmres_create(npg_ctx *ctx, int64_t n, int memory, npg_gmres **out) {
    REQUIRE(ctx && out && n > 0 && memory >= 1 && memory <= 31, "npg_gmres_create: bad argument");
    npg_gmres *ws = new npg_gmres();
    ws->ctx = ctx;
    ws->n = n;
    ws->mem = memory;
    ws->V.assign((size_t)n * (size_t)memory, 0.0);
    ws->w.assign((size_t)n, 0.0);
    ws->q.assign((size_t)n, 0.0);
    ws->dx.assign((size_t)n, 0.0);
    *out = ws;
    return NPG_OK;
}
NPG_API int npg_gmres_destroy(npg_gmres *ws) {
    delete ws;
    return NPG_OK;
}
// knobs of the HBM layouts: accepted and ignored on the host
NPG_API int npg_gmres_set_split(npg_gmres *ws, int) { return ws ? NPG_OK : NPG_EINVAL; }
NPG_API int npg_gmres_set_basis(npg_gmres *ws, int) { return ws ? NPG_OK : NPG_EINVAL; }
NPG_API int npg_gmres_set_gather(npg_gmres *ws, int) { return ws ? NPG_OK : NPG_EINVAL; }
NPG_API int npg_gmres_set_profile(npg_gmres *ws, int) { return ws ? NPG_OK : NPG_EINVAL; }
NPG_API int npg_gmres_get_profile(npg_gmres *ws, double *ms_total, int64_t *launches) {
    REQUIRE(ws && ms_total && launches, "npg_gmres_get_profile: NULL argument");
    *ms_total = 0.0;
    *launches = 0;
    return NPG_OK;
}
NPG_API int64_t npg_gmres_history(npg_gmres *ws, double *buf, int64_t cap) {
    if (!ws || !buf || cap <= 0) return 0;
    const int64_t k = std::min<int64_t>(cap, (int64_t)ws->hist.size());
    memcpy(buf, ws->hist.data(), (size_t)k * sizeof(double));
    return k;
}
static void sym_givens(double a, double b, double &c, double &s, double &rho) {
    if (b == 0.0) {
        c = a == 0.0 ? 1.0 : std::copysign(1.0, a);
        s = 0.0;
        rho = std::fabs(a);
    } else if (a == 0.0) {
        c = 0.0;
        s = std::copysign(1.0, b);
        rho = std::fabs(b);
    } else if (std::fabs(b) > std::fabs(a)) {
        const double t = a / b;
        s = std::copysign(1.0, b) / std::sqrt(1.0 + t * t);
        c = s * t;
        rho = b / s;
    } else {
        const double t = b / a;
        c = std::copysign(1.0, a) / std::sqrt(1.0 + t * t);
        s = c * t;
        rho = a / c;
    }
}
// Left-preconditioned restarted GMRES(memory) with modified Gram-Schmidt: residual estimate |zeta|, stop at
// ||M r|| <= atol + rtol ||M r0||, warm start (x in/out), itmax == 0 means 2 n, breakdown tolerance eps^(3/4).
NPG_API int npg_gmres_solve(npg_gmres *ws, const npg_csr *A, int precond_kind, double precond_scalar, const npg_vec *precond_diag,
                            const npg_vec *y, npg_vec *x, double atol, double rtol, int64_t itmax, double, npg_solve_stats *stats) {
    REQUIRE(ws && A && y && x, "npg_gmres_solve: NULL argument");
    REQUIRE(A->m == ws->n && A->n == ws->n && y->n == ws->n && x->n == ws->n, "npg_gmres_solve: workspace is for n=%lld",
            (long long)ws->n);
    REQUIRE(precond_kind == NPG_PRECOND_NONE || precond_kind == NPG_PRECOND_SCALAR ||
                (precond_kind == NPG_PRECOND_DIAG && precond_diag && precond_diag->n == ws->n),
            "npg_gmres_solve: bad preconditioner");
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t n = ws->n;
    const int mem = ws->mem;
    if (itmax <= 0) itmax = 2 * n;
    const double btol = std::pow(2.220446049250313e-16, 0.75);
    double *w = ws->w.data(), *q = ws->q.data();
    std::vector<double> c((size_t)mem), s((size_t)mem), z((size_t)mem + 1), R((size_t)mem * (mem + 1) / 2), yk((size_t)mem);
    ws->hist.clear();
    int64_t iter = 0;
    int npass = 0, status = 0;
    double rnorm0 = 0.0, rnorm = 0.0, eps = 0.0;
    bool first = true;
    while (status == 0) {
        // true residual q = M (b - A x)
        spmv_raw(A, x->d, w, 1.0, 0.0);
        for (int64_t i = 0; i < n; ++i) q[i] = pre(precond_kind, precond_scalar, precond_diag, i) * (y->d[i] - w[i]);
        const double beta = std::sqrt(dot_fixed(q, q, n));
        if (first) {
            rnorm0 = rnorm = beta;
            eps = atol + rtol * beta;
            ws->hist.push_back(beta);
            first = false;
            if (beta == 0.0) {
                status = 4;
                break;
            }
            if (beta <= eps) {      // (atol alone can satisfy the rule at the start)
                status = 1;
                break;
            }
        }
        ++npass;
        double zeta = beta;
        z[0] = beta;
        double *V = ws->V.data();
        for (int64_t i = 0; i < n; ++i) V[i] = q[i] / beta;
        int k = 0;
        for (; k < mem && status == 0; ++k) {
            double *vk = V + (size_t)k * (size_t)n;
            spmv_raw(A, vk, w, 1.0, 0.0);
            for (int64_t i = 0; i < n; ++i) q[i] = pre(precond_kind, precond_scalar, precond_diag, i) * w[i];
            double *Rk = R.data() + (size_t)k * (k + 1) / 2;
            for (int i = 0; i <= k; ++i) {          // modified Gram-Schmidt
                const double *vi = V + (size_t)i * (size_t)n;
                const double h = dot_fixed(vi, q, n);
                Rk[i] = h;
                for (int64_t t = 0; t < n; ++t) q[t] -= h * vi[t];
            }
            const double hbis = std::sqrt(dot_fixed(q, q, n));
            for (int i = 0; i < k; ++i) {           // previous rotations
                const double hi = Rk[i], hn = Rk[i + 1];
                Rk[i] = c[(size_t)i] * hi + s[(size_t)i] * hn;
                Rk[i + 1] = s[(size_t)i] * hi - c[(size_t)i] * hn;
            }
            double ck, sk, rho;
            sym_givens(Rk[k], hbis, ck, sk, rho);
            c[(size_t)k] = ck;
            s[(size_t)k] = sk;
            Rk[k] = rho;
            const double znext = sk * zeta;
            z[(size_t)k] = ck * zeta;
            zeta = znext;
            rnorm = std::fabs(zeta);
            ++iter;
            ws->hist.push_back(rnorm);
            const bool solved = rnorm <= eps || rnorm + 1.0 <= 1.0;
            if (solved) status = 1;
            else if (iter >= itmax) status = 2;
            else if (hbis <= btol) status = 3;
            if (status == 0 && k + 1 < mem) {
                double *vn = V + (size_t)(k + 1) * (size_t)n;
                for (int64_t t = 0; t < n; ++t) vn[t] = q[t] / hbis;
            }
        }
        const int kk = std::min(k, mem);
        // back substitution R y = z, x += V y
        for (int i = kk - 1; i >= 0; --i) {
            double v = z[(size_t)i];
            for (int j = i + 1; j < kk; ++j) v -= R[(size_t)j * (j + 1) / 2 + i] * yk[(size_t)j];
            const double rii = R[(size_t)i * (i + 1) / 2 + i];
            yk[(size_t)i] = std::fabs(rii) <= btol ? 0.0 : v / rii;
        }
        for (int i = 0; i < kk; ++i) {
            const double *vi = V + (size_t)i * (size_t)n;
            const double a = yk[(size_t)i];
            for (int64_t t = 0; t < n; ++t) x->d[t] += a * vi[t];
        }
    }
    if (stats) {
        stats->solved = (status == 1 || status == 4) ? 1 : 0;
        stats->niter = (int32_t)iter;
        stats->npass = npass;
        stats->status = status;
        stats->nreorth = 0;
        stats->nflagged = 0;
        stats->rnorm0 = rnorm0;
        stats->rnorm = rnorm;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return NPG_OK;
}
NPG_API int npg_cg_create(npg_ctx *ctx, int64_t n, npg_cg **out) {
    REQUIRE(ctx && out && n > 0, "npg_cg_create: bad argument");
    npg_cg *ws = new npg_cg();
    ws->ctx = ctx;
    ws->n = n;
    for (auto *v : {&ws->r, &ws->z, &ws->p, &ws->Ap}) v->assign((size_t)n, 0.0);
    *out = ws;
    return NPG_OK;
}
NPG_API int npg_cg_destroy(npg_cg *ws) {
    delete ws;
    return NPG_OK;
}
NPG_API int64_t npg_cg_history(npg_cg *ws, double *buf, int64_t cap) {
    if (!ws || !buf || cap <= 0) return 0;
    const int64_t k = std::min<int64_t>(cap, (int64_t)ws->hist.size());
    memcpy(buf, ws->hist.data(), (size_t)k * sizeof(double));
    return k;
}
// Krylov.jl cg!: gamma = r'z, stop at sqrt(gamma) <= atol + rtol sqrt(gamma_0); warm start; itmax == 0 means 2 n
NPG_API int npg_cg_solve(npg_cg *ws, const npg_csr *A, int precond_kind, double precond_scalar, const npg_vec *precond_diag,
                         const npg_vec *y, npg_vec *x, double atol, double rtol, int64_t itmax, npg_solve_stats *stats) {
    REQUIRE(ws && A && y && x, "npg_cg_solve: NULL argument");
    REQUIRE(A->m == ws->n && A->n == ws->n && y->n == ws->n && x->n == ws->n, "npg_cg_solve: workspace is for n=%lld", (long long)ws->n);
    REQUIRE(precond_kind == NPG_PRECOND_NONE || precond_kind == NPG_PRECOND_SCALAR ||
                (precond_kind == NPG_PRECOND_DIAG && precond_diag && precond_diag->n == ws->n),
            "npg_cg_solve: bad preconditioner");
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t n = ws->n;
    if (itmax <= 0) itmax = 2 * n;
    double *r = ws->r.data(), *z = ws->z.data(), *p = ws->p.data(), *Ap = ws->Ap.data();
    spmv_raw(A, x->d, Ap, 1.0, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        r[i] = y->d[i] - Ap[i];
        z[i] = pre(precond_kind, precond_scalar, precond_diag, i) * r[i];
        p[i] = z[i];
    }
    double gamma = dot_fixed(r, z, n);
    double rnorm = std::sqrt(std::max(gamma, 0.0));
    const double rnorm0 = rnorm, eps = atol + rtol * rnorm;
    ws->hist.assign(1, rnorm);
    int64_t iter = 0;
    int status = rnorm == 0.0 ? 4 : (rnorm <= eps ? 1 : 0);
    while (status == 0) {
        spmv_raw(A, p, Ap, 1.0, 0.0);
        const double pAp = dot_fixed(p, Ap, n);
        if (!(pAp > 0.0)) {
            status = 3;             // not positive definite / breakdown
            break;
        }
        const double alpha = gamma / pAp;
        for (int64_t i = 0; i < n; ++i) {
            x->d[i] += alpha * p[i];
            r[i] -= alpha * Ap[i];
            z[i] = pre(precond_kind, precond_scalar, precond_diag, i) * r[i];
        }
        const double gnext = dot_fixed(r, z, n);
        rnorm = std::sqrt(std::max(gnext, 0.0));
        ++iter;
        ws->hist.push_back(rnorm);
        if (rnorm <= eps) status = 1;
        else if (iter >= itmax) status = 2;
        const double beta = gnext / gamma;
        gamma = gnext;
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    if (stats) {
        stats->solved = (status == 1 || status == 4) ? 1 : 0;
        stats->niter = (int32_t)iter;
        stats->npass = 0;
        stats->status = status;
        stats->nreorth = stats->nflagged = 0;
        stats->rnorm0 = rnorm0;
        stats->rnorm = rnorm;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return NPG_OK;
}

// ---- element kernels ---------------------------------------------------------------------------------------------------------------------
struct npg_fe {
    npg_ctx *ctx = nullptr;
    int64_t ncell = 0, n_inv = 0, n_b = 0;
    int nq = 0, nb = 0;
    std::vector<double> G, wdet, qw, N2, dN2, Nb, dNb, N1, u_diri, b_diri;       // G: [ncell][12] (4 gradients x 3 components)
    std::vector<int32_t> cu, cp, cb;                                              // [ncell][30], [ncell][4], [ncell][nb]
    std::vector<int64_t> gptr, iptr;                                              // inverted indices: row -> (cell, local DoF)
    std::vector<int64_t> gidx, iidx;                                              // cell * nb + i   /   cell * 34 + l
    std::vector<double> coef[4], kv0, hcell, loc;                                 // nu, kappa_h, kappa_v, f: [ncell][nq]
};
static inline double fval(const double *x, const std::vector<double> &diri, int32_t idx) { return idx >= 0 ? x[idx] : diri[(size_t)(-1 - idx)]; }

NPG_API int npg_fe_create(npg_ctx *ctx, const npg_fe_desc *d, npg_fe **out) {
    REQUIRE(ctx && d && out, "npg_fe_create: NULL argument");
    REQUIRE(d->ncell > 0 && d->nq > 0 && d->nq <= 16, "npg_fe_create: need 1 <= nq <= 16");
    REQUIRE(d->nloc_b == 10 || d->nloc_b == 4, "npg_fe_create: nloc_b must be 10 (P2) or 4 (P1)");
    REQUIRE(d->grad_lambda && d->wdet && d->qw && d->N2 && d->dN2 && d->Nb && d->dNb && d->N1 && d->cell_u && d->cell_p && d->cell_b,
            "npg_fe_create: NULL table");
    const int64_t nc = d->ncell;
    const int nb = d->nloc_b, nq = d->nq;
    for (int64_t k = 0; k < nc * 30; ++k)
        REQUIRE(d->cell_u[k] < d->n_inv && (d->cell_u[k] >= 0 || -1 - (int64_t)d->cell_u[k] < d->n_u_diri),
                "npg_fe_create: cell_u[%lld] = %d out of range", (long long)k, d->cell_u[k]);
    for (int64_t k = 0; k < nc * 4; ++k) REQUIRE(d->cell_p[k] < d->n_inv, "npg_fe_create: cell_p[%lld] = %d out of range", (long long)k, d->cell_p[k]);
    for (int64_t k = 0; k < nc * nb; ++k)
        REQUIRE(d->cell_b[k] < d->n_b && (d->cell_b[k] >= 0 || -1 - (int64_t)d->cell_b[k] < d->n_b_diri),
                "npg_fe_create: cell_b[%lld] = %d out of range", (long long)k, d->cell_b[k]);
    npg_fe *fe = new npg_fe();
    fe->ctx = ctx;
    fe->ncell = nc;
    fe->nq = nq;
    fe->nb = nb;
    fe->n_inv = d->n_inv;
    fe->n_b = d->n_b;
    fe->G.assign(d->grad_lambda, d->grad_lambda + nc * 12);
    fe->wdet.assign(d->wdet, d->wdet + nc);
    fe->qw.assign(d->qw, d->qw + nq);
    fe->N2.assign(d->N2, d->N2 + nq * 10);
    fe->dN2.assign(d->dN2, d->dN2 + nq * 40);
    fe->Nb.assign(d->Nb, d->Nb + nq * nb);
    fe->dNb.assign(d->dNb, d->dNb + nq * nb * 4);
    fe->N1.assign(d->N1, d->N1 + nq * 4);
    fe->cu.assign(d->cell_u, d->cell_u + nc * 30);
    fe->cp.assign(d->cell_p, d->cell_p + nc * 4);
    fe->cb.assign(d->cell_b, d->cell_b + nc * nb);
    fe->u_diri.assign(d->u_diri, d->u_diri + std::max<int64_t>(0, d->n_u_diri));
    fe->b_diri.assign(d->b_diri, d->b_diri + std::max<int64_t>(0, d->n_b_diri));
    fe->u_diri.push_back(0.0);
    fe->b_diri.push_back(0.0);
    // inverted indices, cell-ascending: the order in which a row's cell contributions are added
    fe->gptr.assign((size_t)d->n_b + 1, 0);
    for (int64_t k = 0; k < nc * nb; ++k)
        if (fe->cb[(size_t)k] >= 0) ++fe->gptr[(size_t)fe->cb[(size_t)k] + 1];
    for (int64_t r = 0; r < d->n_b; ++r) fe->gptr[r + 1] += fe->gptr[r];
    fe->gidx.resize((size_t)fe->gptr[d->n_b]);
    {
        std::vector<int64_t> next(fe->gptr.begin(), fe->gptr.end() - 1);
        for (int64_t k = 0; k < nc * nb; ++k)
            if (fe->cb[(size_t)k] >= 0) fe->gidx[(size_t)next[(size_t)fe->cb[(size_t)k]]++] = k;
    }
    fe->iptr.assign((size_t)d->n_inv + 1, 0);
    for (int64_t k = 0; k < nc * 30; ++k)
        if (fe->cu[(size_t)k] >= 0) ++fe->iptr[(size_t)fe->cu[(size_t)k] + 1];
    for (int64_t k = 0; k < nc * 4; ++k)
        if (fe->cp[(size_t)k] >= 0) ++fe->iptr[(size_t)fe->cp[(size_t)k] + 1];
    for (int64_t r = 0; r < d->n_inv; ++r) fe->iptr[r + 1] += fe->iptr[r];
    fe->iidx.resize((size_t)fe->iptr[d->n_inv]);
    {
        std::vector<int64_t> next(fe->iptr.begin(), fe->iptr.end() - 1);
        for (int64_t c = 0; c < nc; ++c) {
            for (int l = 0; l < 30; ++l) {
                const int32_t r = fe->cu[(size_t)c * 30 + l];
                if (r >= 0) fe->iidx[(size_t)next[(size_t)r]++] = c * 34 + l;
            }
            for (int m = 0; m < 4; ++m) {
                const int32_t r = fe->cp[(size_t)c * 4 + m];
                if (r >= 0) fe->iidx[(size_t)next[(size_t)r]++] = c * 34 + 30 + m;
            }
        }
    }
    fe->loc.assign((size_t)nc * nb, 0.0);
    *out = fe;
    return NPG_OK;
}
NPG_API int npg_fe_destroy(npg_fe *fe) {
    delete fe;
    return NPG_OK;
}
NPG_API int npg_fe_set_precision(npg_fe *fe, int precision) {
    REQUIRE(fe, "npg_fe_set_precision: NULL handle");
    REQUIRE(precision == NPG_FE_FP64, "npg_fe_set_precision: the host element kernels are fp64 (Gridap's arithmetic); the fp32-local mode "
                                      "is a device option");
    return NPG_OK;
}
NPG_API int npg_fe_get_precision(const npg_fe *fe) { return fe ? NPG_FE_FP64 : NPG_EINVAL; }
static int coef_index(const char *name) {
    return !strcmp(name, "nu") ? 0 : !strcmp(name, "kappa_h") ? 1 : !strcmp(name, "kappa_v") ? 2 : !strcmp(name, "f") ? 3 : -1;
}
NPG_API int npg_fe_set_coeff(npg_fe *fe, const char *name, const double *values) {
    REQUIRE(fe && name && values, "npg_fe_set_coeff: NULL argument");
    const int k = coef_index(name);
    REQUIRE(k >= 0, "npg_fe_set_coeff: unknown coefficient '%s'", name);
    fe->coef[k].assign(values, values + fe->ncell * fe->nq);
    if (k == 2) fe->kv0 = fe->coef[2];
    return NPG_OK;
}
// physical gradient of local function i (table dN: [nq][nloc][4]) at quadrature point q of a cell with gradients G[4][3]
static inline void grad_of(const double *dN, int nloc, const double *G, int q, int i, double g[3]) {
    const double *dn = dN + ((size_t)q * nloc + i) * 4;
    for (int a = 0; a < 3; ++a) g[a] = dn[0] * G[a] + dn[1] * G[3 + a] + dn[2] * G[6 + a] + dn[3] * G[9 + a];
}
// pass 2 of the vector assembly: every destination row adds its cells' local entries in cell order
static void gather_rows(const npg_fe *fe, double theta, double dt, const npg_vec *rhs_diff, const npg_vec *rhs_flux, const npg_vec *rhs_M,
                        const npg_vec *rhs_h, const npg_vec *rhs_v, double *out) {
    const int64_t n = fe->n_b;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        double s = 0.0;
        for (int64_t k = fe->gptr[r]; k < fe->gptr[r + 1]; ++k) s += fe->loc[(size_t)fe->gidx[(size_t)k]];
        if (rhs_diff) s += theta * rhs_diff->d[r];
        if (rhs_flux) s += dt * rhs_flux->d[r];
        double lift = 0.0;
        if (rhs_h) lift += rhs_h->d[r];
        if (rhs_v) lift += rhs_v->d[r];
        lift *= theta;
        if (rhs_M) lift += rhs_M->d[r];
        out[r] = s - lift;
    }
}
// loc[cell][i] = int ( c1 b + c2 b_prev - cdt ( u~ . grad b~ + u~_z N2 ) ) phi_i         (src/model.jl:292-300)
static int advection_pass1(npg_fe *fe, int scheme, double dt, double N2, const npg_vec *b, const npg_vec *bp, const npg_vec *xi,
                           const npg_vec *xip) {
    REQUIRE(scheme == NPG_BDF1 || scheme == NPG_BDF2, "fe: scheme must be NPG_BDF1 or NPG_BDF2");
    REQUIRE(b && bp && xi && xip, "fe: NULL state vector");
    REQUIRE(b->n == fe->n_b && bp->n == fe->n_b, "fe: buoyancy vectors must have %lld entries", (long long)fe->n_b);
    REQUIRE(xi->n == fe->n_inv && xip->n == fe->n_inv, "fe: inversion vectors must have %lld entries", (long long)fe->n_inv);
    const bool bdf2 = scheme == NPG_BDF2;
    const double c1 = bdf2 ? 4.0 / 3.0 : 1.0, c2 = bdf2 ? -1.0 / 3.0 : 0.0, e1 = bdf2 ? 2.0 : 1.0, e2 = bdf2 ? -1.0 : 0.0;
    const double cdt = bdf2 ? 2.0 / 3.0 * dt : dt;
    const int nb = fe->nb, nq = fe->nq;
    const int64_t nc = fe->ncell;
#pragma omp parallel for schedule(static)
    for (int64_t cell = 0; cell < nc; ++cell) {
        const double *G = &fe->G[(size_t)cell * 12];
        double bm[10], bt[10], ut[30], acc[10];
        for (int i = 0; i < nb; ++i) {
            const int32_t idx = fe->cb[(size_t)cell * nb + i];
            const double v = fval(b->d, fe->b_diri, idx), vp = fval(bp->d, fe->b_diri, idx);
            bm[i] = c1 * v + c2 * vp;
            bt[i] = e1 * v + e2 * vp;
            acc[i] = 0.0;
        }
        for (int k = 0; k < 30; ++k) {
            const int32_t idx = fe->cu[(size_t)cell * 30 + k];
            ut[k] = e1 * fval(xi->d, fe->u_diri, idx) + e2 * fval(xip->d, fe->u_diri, idx);
        }
        for (int q = 0; q < nq; ++q) {
            double bq = 0.0, gl[4] = {0, 0, 0, 0};
            for (int i = 0; i < nb; ++i) {
                bq += fe->Nb[(size_t)q * nb + i] * bm[i];
                const double *dn = &fe->dNb[((size_t)q * nb + i) * 4];
                for (int k = 0; k < 4; ++k) gl[k] += dn[k] * bt[i];
            }
            double u[3] = {0, 0, 0};
            for (int i = 0; i < 10; ++i) {
                const double nn = fe->N2[(size_t)q * 10 + i];
                u[0] += nn * ut[3 * i];
                u[1] += nn * ut[3 * i + 1];
                u[2] += nn * ut[3 * i + 2];
            }
            double g[3];
            for (int a = 0; a < 3; ++a) g[a] = gl[0] * G[a] + gl[1] * G[3 + a] + gl[2] * G[6 + a] + gl[3] * G[9 + a];
            const double integrand = bq - cdt * (u[0] * g[0] + u[1] * g[1] + u[2] * g[2] + u[2] * N2);
            const double wq = fe->qw[(size_t)q] * fe->wdet[(size_t)cell] * integrand;
            for (int i = 0; i < nb; ++i) acc[i] += wq * fe->Nb[(size_t)q * nb + i];
        }
        for (int i = 0; i < nb; ++i) fe->loc[(size_t)cell * nb + i] = acc[i];
    }
    return NPG_OK;
}
NPG_API int npg_fe_advection_rhs(npg_fe *fe, int scheme, double dt, double N2, const npg_vec *b, const npg_vec *b_prev, const npg_vec *x_inv,
                                 const npg_vec *x_inv_prev, npg_vec *out) {
    REQUIRE(fe && out && out->n == fe->n_b, "npg_fe_advection_rhs: bad argument");
    const int rc = advection_pass1(fe, scheme, dt, N2, b, b_prev, x_inv, x_inv_prev);
    if (rc) return rc;
    gather_rows(fe, 0.0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, out->d);
    return NPG_OK;
}
NPG_API int npg_fe_evolution_rhs(npg_fe *fe, int scheme, double dt, double N2, double theta, const npg_vec *b, const npg_vec *b_prev,
                                 const npg_vec *x_inv, const npg_vec *x_inv_prev, const npg_vec *rhs_diff, const npg_vec *rhs_flux,
                                 const npg_vec *rhs_M, const npg_vec *rhs_h, const npg_vec *rhs_v, npg_vec *y) {
    REQUIRE(fe && y && y->n == fe->n_b, "npg_fe_evolution_rhs: bad argument");
    for (const npg_vec *v : {rhs_diff, rhs_flux, rhs_M, rhs_h, rhs_v})
        REQUIRE(!v || v->n == fe->n_b, "npg_fe_evolution_rhs: rhs_* vectors must have %lld entries", (long long)fe->n_b);
    const int rc = advection_pass1(fe, scheme, dt, N2, b, b_prev, x_inv, x_inv_prev);
    if (rc) return rc;
    gather_rows(fe, theta, dt, rhs_diff, rhs_flux, rhs_M, rhs_h, rhs_v, y->d);
    return NPG_OK;
}
// rhs_diff = -N2 int kappa_v d_z phi_i        (src/evolution.jl:269-278)
NPG_API int npg_fe_assemble_rhs_diff(npg_fe *fe, double N2, npg_vec *out) {
    REQUIRE(fe && out && out->n == fe->n_b, "npg_fe_assemble_rhs_diff: bad argument");
    REQUIRE(!fe->coef[2].empty(), "npg_fe_assemble_rhs_diff: coefficient kappa_v has not been set");
    const int nb = fe->nb, nq = fe->nq;
#pragma omp parallel for schedule(static)
    for (int64_t cell = 0; cell < fe->ncell; ++cell) {
        const double *G = &fe->G[(size_t)cell * 12];
        double acc[10] = {0};
        for (int q = 0; q < nq; ++q) {
            const double wq = -N2 * fe->qw[(size_t)q] * fe->wdet[(size_t)cell] * fe->coef[2][(size_t)cell * nq + q];
            for (int i = 0; i < nb; ++i) {
                const double *dn = &fe->dNb[((size_t)q * nb + i) * 4];
                acc[i] += wq * (dn[0] * G[2] + dn[1] * G[5] + dn[2] * G[8] + dn[3] * G[11]);
            }
        }
        for (int i = 0; i < nb; ++i) fe->loc[(size_t)cell * nb + i] = acc[i];
    }
    gather_rows(fe, 0.0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, out->d);
    return NPG_OK;
}
static inline int64_t slot_of(const npg_csr *A, int64_t row, int32_t c) {
    const auto b = A->col.begin() + A->rowptr[(size_t)row], e = A->col.begin() + A->rowptr[(size_t)row + 1];
    const auto it = std::lower_bound(b, e, c);
    return (it != e && *it == c) ? (int64_t)(it - A->col.begin()) : -1;
}
// Row-owner matrix assembly (as csrc/fe.hip: the owner of a CSR row walks the (cell, local DoF) pairs that carry the row's DoF,
// cell-ascending, and adds the local rows into its own row - no races, the same bits on every run).
NPG_API int npg_fe_assemble_matrix(npg_fe *fe, int which, double scale, int full_stress, npg_csr *A, npg_vec *lift) {
    REQUIRE(fe && A, "npg_fe_assemble_matrix: NULL argument");
    const int nb = fe->nb, nq = fe->nq;
    std::fill(A->val.begin(), A->val.end(), 0.0);
    if (lift) std::fill(lift->d, lift->d + lift->n, 0.0);
    int64_t missing = 0;
    if (which == NPG_MAT_M || which == NPG_MAT_KH || which == NPG_MAT_KV) {
        REQUIRE(A->m <= fe->n_b && A->n <= fe->n_b && A->m <= A->n, "npg_fe_assemble_matrix: matrix must be n_b x n_b");
        REQUIRE(!lift || (lift->n >= A->m && lift->n <= fe->n_b), "npg_fe_assemble_matrix: lift must have n_b entries");
        const std::vector<double> *kap = which == NPG_MAT_KH ? &fe->coef[1] : which == NPG_MAT_KV ? &fe->coef[2] : nullptr;
        REQUIRE(which == NPG_MAT_M || !kap->empty(), "npg_fe_assemble_matrix: diffusivity coefficient has not been set");
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : missing)
        for (int64_t r = 0; r < A->m; ++r) {
            double lf = 0.0;
            for (int64_t k = fe->gptr[r]; k < fe->gptr[r + 1]; ++k) {
                const int64_t cell = fe->gidx[(size_t)k] / nb;
                const int i = (int)(fe->gidx[(size_t)k] % nb);
                const double *G = &fe->G[(size_t)cell * 12];
                double acc[10] = {0};
                for (int q = 0; q < nq; ++q) {
                    double wq = fe->qw[(size_t)q] * fe->wdet[(size_t)cell];
                    if (which == NPG_MAT_M) {
                        wq *= fe->Nb[(size_t)q * nb + i];
                        for (int j = 0; j < nb; ++j) acc[j] += wq * fe->Nb[(size_t)q * nb + j];
                    } else {
                        double gi[3];
                        grad_of(fe->dNb.data(), nb, G, q, i, gi);
                        wq *= (*kap)[(size_t)cell * nq + q];
                        for (int j = 0; j < nb; ++j) {
                            double gj[3];
                            grad_of(fe->dNb.data(), nb, G, q, j, gj);
                            acc[j] += wq * (which == NPG_MAT_KH ? gi[0] * gj[0] + gi[1] * gj[1] : gi[2] * gj[2]);
                        }
                    }
                }
                for (int j = 0; j < nb; ++j) {
                    const int32_t c = fe->cb[(size_t)cell * nb + j];
                    if (c >= 0) {
                        const int64_t s = slot_of(A, r, c);
                        if (s >= 0) A->val[(size_t)s] += acc[j];
                        else if (acc[j] != 0.0) ++missing;
                    } else {
                        lf += acc[j] * fe->b_diri[(size_t)(-1 - c)];
                    }
                }
            }
            if (lift) lift->d[r] = lf;
        }
    } else if (which == NPG_MAT_B) {
        // rows (u node i, component z), columns buoyancy nodes: scale * int phi_i phib_j          (src/inversion.jl:208)
        REQUIRE(A->m <= fe->n_inv && A->n == fe->n_b, "npg_fe_assemble_matrix: B must be n_inv x n_b");
        REQUIRE(!lift || (lift->n >= A->m && lift->n <= fe->n_inv), "npg_fe_assemble_matrix: lift must have n_inv entries");
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : missing)
        for (int64_t r = 0; r < A->m; ++r) {
            double lf = 0.0;
            for (int64_t k = fe->iptr[r]; k < fe->iptr[r + 1]; ++k) {
                const int64_t cell = fe->iidx[(size_t)k] / 34;
                const int l = (int)(fe->iidx[(size_t)k] % 34);
                if (l >= 30 || l % 3 != 2) continue;
                const int i = l / 3;
                double acc[10] = {0};
                for (int q = 0; q < nq; ++q) {
                    const double wq = fe->qw[(size_t)q] * fe->wdet[(size_t)cell] * scale * fe->N2[(size_t)q * 10 + i];
                    for (int j = 0; j < nb; ++j) acc[j] += wq * fe->Nb[(size_t)q * nb + j];
                }
                for (int j = 0; j < nb; ++j) {
                    const int32_t c = fe->cb[(size_t)cell * nb + j];
                    if (c >= 0) {
                        const int64_t s = slot_of(A, r, c);
                        if (s >= 0) A->val[(size_t)s] += acc[j];
                        else if (acc[j] != 0.0) ++missing;
                    } else {
                        lf += acc[j] * fe->b_diri[(size_t)(-1 - c)];
                    }
                }
            }
            if (lift) lift->d[r] = lf;
        }
    } else if (which == NPG_MAT_A) {
        //   [(i,a),(j,c)] += a2e2 int nu ( d_ac grad phi_i . grad phi_j  [+ d_c phi_i d_a phi_j  if full_stress] )
        //   [(i,x),(j,y)] -= int f phi_i phi_j ; [(i,y),(j,x)] += int f phi_i phi_j
        //   [(i,a), p_m ] -= int d_a phi_i psi_m ; [p_m, (i,a)] += int psi_m d_a phi_i          (src/inversion.jl:172-192)
        REQUIRE(A->m <= fe->n_inv && A->n <= fe->n_inv && A->m <= A->n, "npg_fe_assemble_matrix: A must be n_inv x n_inv");
        REQUIRE(!fe->coef[0].empty() && !fe->coef[3].empty(), "npg_fe_assemble_matrix: coefficients nu and f must be set");
        auto add = [&](int64_t r, int32_t c, double v, int64_t &miss) {
            const int64_t s = slot_of(A, r, c);
            if (s >= 0) A->val[(size_t)s] += v;
            else if (v != 0.0) ++miss;
        };
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : missing)
        for (int64_t r = 0; r < A->m; ++r) {
            for (int64_t k = fe->iptr[r]; k < fe->iptr[r + 1]; ++k) {
                const int64_t cell = fe->iidx[(size_t)k] / 34;
                const int l = (int)(fe->iidx[(size_t)k] % 34);
                const double *G = &fe->G[(size_t)cell * 12];
                const double *nu = &fe->coef[0][(size_t)cell * nq], *ff = &fe->coef[3][(size_t)cell * nq];
                if (l < 30) {
                    const int i = l / 3, a = l % 3;
                    for (int j = 0; j < 10; ++j) {
                        double kk = 0.0, cc = 0.0, fs[3] = {0, 0, 0};
                        for (int q = 0; q < nq; ++q) {
                            const double wq = fe->qw[(size_t)q] * fe->wdet[(size_t)cell];
                            double gi[3], gj[3];
                            grad_of(fe->dN2.data(), 10, G, q, i, gi);
                            grad_of(fe->dN2.data(), 10, G, q, j, gj);
                            const double wn = wq * scale * nu[q];
                            kk += wn * (gi[0] * gj[0] + gi[1] * gj[1] + gi[2] * gj[2]);
                            cc += wq * ff[q] * fe->N2[(size_t)q * 10 + i] * fe->N2[(size_t)q * 10 + j];
                            if (full_stress)
                                for (int c = 0; c < 3; ++c) fs[c] += wn * gi[c] * gj[a];
                        }
                        for (int c = 0; c < 3; ++c) {
                            const int32_t cj = fe->cu[(size_t)cell * 30 + 3 * j + c];
                            if (cj < 0) continue;               // homogeneous velocity Dirichlet data: no lift
                            double v = full_stress ? fs[c] : 0.0;
                            if (a == c) v += kk;
                            if (a == 0 && c == 1) v -= cc;
                            if (a == 1 && c == 0) v += cc;
                            if (a == c || full_stress || (a < 2 && c < 2)) add(r, cj, v, missing);
                        }
                    }
                    for (int m = 0; m < 4; ++m) {
                        const int32_t pm = fe->cp[(size_t)cell * 4 + m];
                        if (pm < 0) continue;
                        double dd = 0.0;
                        for (int q = 0; q < nq; ++q) {
                            double gi[3];
                            grad_of(fe->dN2.data(), 10, G, q, i, gi);
                            dd += fe->qw[(size_t)q] * fe->wdet[(size_t)cell] * fe->N1[(size_t)q * 4 + m] * gi[a];
                        }
                        add(r, pm, -dd, missing);
                    }
                } else {
                    const int m = l - 30;
                    for (int i = 0; i < 10; ++i) {
                        double dsum[3] = {0, 0, 0};
                        for (int q = 0; q < nq; ++q) {
                            double gi[3];
                            grad_of(fe->dN2.data(), 10, G, q, i, gi);
                            const double wm = fe->qw[(size_t)q] * fe->wdet[(size_t)cell] * fe->N1[(size_t)q * 4 + m];
                            for (int a = 0; a < 3; ++a) dsum[a] += wm * gi[a];
                        }
                        for (int a = 0; a < 3; ++a) {
                            const int32_t ci = fe->cu[(size_t)cell * 30 + 3 * i + a];
                            if (ci >= 0) add(r, ci, dsum[a], missing);
                        }
                    }
                }
            }
        }
    } else {
        REQUIRE(false, "npg_fe_assemble_matrix: unknown matrix id %d", which);
    }
    REQUIRE(missing == 0, "npg_fe_assemble_matrix: %lld non-zero local entries fall outside the CSR pattern", (long long)missing);
    return NPG_OK;
}
// closures at the quadrature points from d_z b          (src/inputs.jl:87-91, 130-137)
static void coeff_from_bz(const npg_fe *fe, int mode, const npg_vec *b, const std::vector<double> &base, double p0, double p1, double alpha,
                          double N2, double p2, double p3, std::vector<double> &out) {
    const int nb = fe->nb, nq = fe->nq;
    out.resize((size_t)fe->ncell * nq);
#pragma omp parallel for schedule(static)
    for (int64_t cell = 0; cell < fe->ncell; ++cell) {
        const double *G = &fe->G[(size_t)cell * 12];
        double bn[10];
        for (int i = 0; i < nb; ++i) bn[i] = fval(b->d, fe->b_diri, fe->cb[(size_t)cell * nb + i]);
        for (int q = 0; q < nq; ++q) {
            double bz = 0.0;
            for (int i = 0; i < nb; ++i) {
                const double *dn = &fe->dNb[((size_t)q * nb + i) * 4];
                bz += bn[i] * (dn[0] * G[2] + dn[1] * G[5] + dn[2] * G[8] + dn[3] * G[11]);
            }
            const double abz = alpha * (N2 + bz);
            const size_t o = (size_t)cell * nq + q;
            if (mode == 0) {
                out[o] = base[o] + p0 * (1.0 + std::tanh(-abz / p1)) / 2.0;
            } else {
                const double f = fe->coef[3][o];
                const double nu = f * (f / std::sqrt(p1 * p1 + abz * abz));
                const double m = std::fmax(p2 * p3, p2 * nu);
                out[o] = (m + std::log(std::exp(p2 * p3 - m) + std::exp(p2 * nu - m))) / p2;
            }
        }
    }
}
NPG_API int npg_fe_update_kappa_convection(npg_fe *fe, const double *kappa_v0_host_or_null, double kappa_c, double N2min, double alpha,
                                           double N2, const npg_vec *b) {
    REQUIRE(fe && b && b->n == fe->n_b, "npg_fe_update_kappa_convection: bad argument");
    if (kappa_v0_host_or_null) {
        const int rc = npg_fe_set_coeff(fe, "kappa_v", kappa_v0_host_or_null);
        if (rc) return rc;
    }
    REQUIRE(!fe->kv0.empty(), "npg_fe_update_kappa_convection: background kappa_v has not been set");
    coeff_from_bz(fe, 0, b, fe->kv0, kappa_c, N2min, alpha, N2, 0.0, 0.0, fe->coef[2]);
    return NPG_OK;
}
NPG_API int npg_fe_update_nu_eddy(npg_fe *fe, double N2min, double alpha, double N2, double smoothing, double nu_min, const npg_vec *b) {
    REQUIRE(fe && b && b->n == fe->n_b, "npg_fe_update_nu_eddy: bad argument");
    REQUIRE(!fe->coef[3].empty(), "npg_fe_update_nu_eddy: coefficient f has not been set");
    coeff_from_bz(fe, 1, b, fe->coef[3], 0.0, N2min, alpha, N2, smoothing, nu_min, fe->coef[0]);
    return NPG_OK;
}
// min_K h_K / max(max_q |u|, u_min)          (update_dt!, src/timesteppers.jl:108-119)
NPG_API int npg_fe_cfl_ratio(npg_fe *fe, const double *h_cells_host, double u_min, const npg_vec *x_inv, double *out) {
    REQUIRE(fe && x_inv && out && x_inv->n == fe->n_inv, "npg_fe_cfl_ratio: bad argument");
    if (h_cells_host) fe->hcell.assign(h_cells_host, h_cells_host + fe->ncell);
    REQUIRE(!fe->hcell.empty(), "npg_fe_cfl_ratio: cell sizes have not been provided");
    double best = 1e300;
    for (int64_t cell = 0; cell < fe->ncell; ++cell) {
        double un[30];
        for (int k = 0; k < 30; ++k) un[k] = fval(x_inv->d, fe->u_diri, fe->cu[(size_t)cell * 30 + k]);
        double smax = 0.0;
        for (int q = 0; q < fe->nq; ++q) {
            double u[3] = {0, 0, 0};
            for (int i = 0; i < 10; ++i) {
                const double nn = fe->N2[(size_t)q * 10 + i];
                u[0] += nn * un[3 * i];
                u[1] += nn * un[3 * i + 1];
                u[2] += nn * un[3 * i + 2];
            }
            smax = std::fmax(smax, std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]));
        }
        best = std::fmin(best, fe->hcell[(size_t)cell] / std::fmax(smax, u_min));
    }
    *out = best;
    return NPG_OK;
}

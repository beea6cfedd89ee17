/* ORACLE (test infrastructure, not product code) - C / OpenMP restatement of the two Krylov.jl 0.10.6 solvers the reference
 * drives, for timing the reference's host Krylov branch on all cores of the GPU box (bench.py cpu_baseline) and for
 * cross-checking oracle/krylov_oracle.py.  Same algorithm as krylov_oracle.gmres / .cg (see that file's header for the
 * provenance: call site /root/reference/src/iterative_solvers.jl:58, workspaces src/inversion.jl:74-94 and
 * src/evolution.jl:118-126): left-preconditioned restarted GMRES(m) with MODIFIED Gram-Schmidt and Givens QR, stop when
 * ||M r|| <= atol + rtol ||M r0||, warm start; Jacobi-preconditioned CG, stop on sqrt(r'z).  M is a diagonal (vector).
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load the library built from this file. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_get_max_threads(void) { return omp_get_max_threads(); }

static void spmv(int64_t n, const int64_t *rp, const int32_t *ci, const double *v, const double *x, double *y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
        y[i] = s;
    }
}

static double dot(int64_t n, const double *a, const double *b) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static void sym_givens(double a, double b, double *c, double *s, double *rho) {
    if (b == 0.0) { *c = a == 0.0 ? 1.0 : (a > 0 ? 1.0 : -1.0); *s = 0.0; *rho = fabs(a); return; }
    if (a == 0.0) { *c = 0.0; *s = b > 0 ? 1.0 : -1.0; *rho = fabs(b); return; }
    if (fabs(b) > fabs(a)) {
        double t = a / b;
        *s = (b > 0 ? 1.0 : -1.0) / sqrt(1.0 + t * t);
        *c = *s * t;
        *rho = b / *s;
    } else {
        double t = b / a;
        *c = (a > 0 ? 1.0 : -1.0) / sqrt(1.0 + t * t);
        *s = *c * t;
        *rho = a / *c;
    }
}

/* returns the number of inner iterations; x in/out (warm start); hist (cap >= itmax + 1) receives the residual estimates;
 * *solved is set.  Md: diagonal of the (inverse-action) preconditioner. */
int64_t orc_gmres(int64_t n, const int64_t *rp, const int32_t *ci, const double *val, const double *b, double *x,
                  const double *Md, int mem, double atol, double rtol, int64_t itmax, double *hist, int *solved) {
    double *V = malloc((size_t)mem * n * sizeof(double)), *w = malloc(n * sizeof(double)), *q = malloc(n * sizeof(double)),
           *dx = malloc(n * sizeof(double));
    double *c = calloc(mem, sizeof(double)), *s = calloc(mem, sizeof(double)), *z = calloc(mem, sizeof(double)),
           *R = calloc((size_t)mem * (mem + 1) / 2, sizeof(double)), *y = calloc(mem, sizeof(double));
    const double btol = pow(2.220446049250313e-16, 0.75);
    if (itmax == 0) itmax = 2 * n;
    spmv(n, rp, ci, val, x, w);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) q[i] = Md[i] * (b[i] - w[i]);           /* r0 = M (b - A x0) */
    double beta = sqrt(dot(n, q, q)), rnorm = beta;
    const double eps = atol + rtol * rnorm;
    int64_t it = 0, nh = 0;
    hist[nh++] = beta;
    int ok = rnorm <= eps, breakdown = 0;
    int64_t inner_itmax = itmax;
    int npass = 0;
    while (!ok && it < itmax && !breakdown && beta != 0.0) {
        if (npass >= 1) {
            spmv(n, rp, ci, val, x, w);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) q[i] = Md[i] * (b[i] - w[i]);
            beta = sqrt(dot(n, q, q));
        }
        memset(z, 0, mem * sizeof(double));
        z[0] = beta;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) { V[i] = q[i] / beta; dx[i] = 0.0; }
        ++npass;
        int inner = 0, inner_tired = 0;
        int64_t nr = 0;
        while (!ok && !inner_tired && !breakdown) {
            ++inner;
            const double *vk = V + (size_t)(inner - 1) * n;
            spmv(n, rp, ci, val, vk, w);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) q[i] = Md[i] * w[i];
            for (int j = 0; j < inner; ++j) {                                /* modified Gram-Schmidt */
                const double *vj = V + (size_t)j * n;
                const double h = dot(n, vj, q);
                R[nr + j] = h;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; ++i) q[i] -= h * vj[i];
            }
            const double hbis = sqrt(dot(n, q, q));
            for (int j = 0; j < inner - 1; ++j) {
                const double t = c[j] * R[nr + j] + s[j] * R[nr + j + 1];
                R[nr + j + 1] = s[j] * R[nr + j] - c[j] * R[nr + j + 1];
                R[nr + j] = t;
            }
            sym_givens(R[nr + inner - 1], hbis, &c[inner - 1], &s[inner - 1], &R[nr + inner - 1]);
            const double zeta = s[inner - 1] * z[inner - 1];
            z[inner - 1] = c[inner - 1] * z[inner - 1];
            rnorm = fabs(zeta);
            hist[nh++] = rnorm;
            nr += inner;
            ok = (rnorm <= eps) || (rnorm + 1.0 <= 1.0);
            breakdown = hbis <= btol;
            inner_tired = inner >= (mem < inner_itmax ? mem : inner_itmax);
            if (!(ok || inner_tired || breakdown)) {
                double *vn = V + (size_t)inner * n;
#pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; ++i) vn[i] = q[i] / hbis;
                z[inner] = zeta;
            }
        }
        memcpy(y, z, mem * sizeof(double));
        for (int i = inner - 1; i >= 0; --i) {
            int64_t pos = nr + i - inner;
            for (int j = inner - 1; j > i; --j) { y[i] -= R[pos] * y[j]; pos -= j; }
            y[i] = fabs(R[pos]) <= btol ? 0.0 : y[i] / R[pos];
        }
        for (int j = 0; j < inner; ++j) {
            const double *vj = V + (size_t)j * n;
            const double a = y[j];
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) dx[i] += a * vj[i];
        }
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) x[i] += dx[i];
        inner_itmax -= inner;
        it += inner;
    }
    *solved = ok;
    free(V); free(w); free(q); free(dx); free(c); free(s); free(z); free(R); free(y);
    return it;
}

/* Jacobi-preconditioned CG (Krylov.jl cg!): stop when sqrt(r'z) <= atol + rtol sqrt(r0'z0); x in/out */
int64_t orc_cg(int64_t n, const int64_t *rp, const int32_t *ci, const double *val, const double *b, double *x,
               const double *Md, double atol, double rtol, int64_t itmax, int *solved) {
    double *r = malloc(n * sizeof(double)), *z = malloc(n * sizeof(double)), *p = malloc(n * sizeof(double)),
           *Ap = malloc(n * sizeof(double));
    if (itmax == 0) itmax = 2 * n;
    spmv(n, rp, ci, val, x, Ap);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) { r[i] = b[i] - Ap[i]; z[i] = Md[i] * r[i]; p[i] = z[i]; }
    double gamma = dot(n, r, z);
    double rnorm = sqrt(gamma);
    const double eps = atol + rtol * rnorm;
    int64_t it = 0;
    int ok = rnorm <= eps;
    while (!ok && it < itmax) {
        spmv(n, rp, ci, val, p, Ap);
        const double pAp = dot(n, p, Ap);
        const double a = gamma / pAp;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) { x[i] += a * p[i]; r[i] -= a * Ap[i]; z[i] = Md[i] * r[i]; }
        const double gnew = dot(n, r, z);
        const double bt = gnew / gamma;
        gamma = gnew;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) p[i] = z[i] + bt * p[i];
        rnorm = sqrt(gamma);
        ++it;
        ok = rnorm <= eps;
    }
    *solved = ok;
    free(r); free(z); free(p); free(Ap);
    return it;
}

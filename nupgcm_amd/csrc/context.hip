// Context, device vectors and the BLAS-1 / broadcast entry points of the C ABI.
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cmath>

#include "common.h"
#include "device_utils.h"

namespace npg {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int ensure_stage(npg_ctx *ctx, size_t doubles) {
    if (ctx->stage_doubles >= doubles) return NPG_OK;
    if (ctx->h_stage) NPG_HIP(hipHostFree(ctx->h_stage));
    ctx->h_stage = nullptr;
    ctx->stage_doubles = 0;
    NPG_HIP(hipHostMalloc((void **)&ctx->h_stage, doubles * sizeof(double), hipHostMallocDefault));
    ctx->stage_doubles = doubles;
    return NPG_OK;
}

static inline int grid_for(int64_t n, int cap = 2048) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- elementwise kernels ------------------------------------------------------------------------------------------
__global__ void k_fill(double *x, double a, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = a;
}

__global__ void k_axpby(double *y, double a, const double *x, double b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

__global__ void k_mul(double *y, const double *d, const double *x, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = d[i] * x[i];
}

struct LinComb {
    int nterms;
    double coef[8];
    const double *x[8];
};

__global__ void k_lincomb(double *y, LinComb lc, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < lc.nterms) s += lc.coef[k] * lc.x[k][i];
        y[i] = s;
    }
}

// ---- reductions to the host -----------------------------------------------------------------------------------------
__global__ void k_dot_partial(const double *x, const double *y, int64_t n, double *part) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += x[i] * y[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void k_maxabs_partial(const double *x, int64_t n, double *part) {
    __shared__ double sh[8];
    double m = 0.0, nan = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        if (v != v) nan = 1.0;
        m = fmax(m, fabs(v));
    }
    m = wave_max(m);
    nan = wave_max(nan);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[wave] = m;
        sh[4 + wave] = nan;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        part[2 * blockIdx.x + 1] = fmax(fmax(sh[4], sh[5]), fmax(sh[6], sh[7]));
    }
}

__global__ void k_minmax_partial(const double *x, int64_t n, double *part) {
    __shared__ double sh[8];
    double lo = 1e300, hi = -1e300;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[wave] = lo;
        sh[4 + wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
        part[2 * blockIdx.x + 1] = fmax(fmax(sh[4], sh[5]), fmax(sh[6], sh[7]));
    }
}

__global__ void k_final_sum(const double *part, int np, double *out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

int reduce_dot(npg_ctx *ctx, const double *x, const double *y, int64_t n, double *out) {
    const int g = grid_for(n, 1024);
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(kBlock), 0, ctx->stream, x, y, n, ctx->d_scratch + 8);
    hipLaunchKernelGGL(k_final_sum, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->d_scratch + 8, g, ctx->d_scratch);
    NPG_HIP(hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    *out = ctx->h_scratch[0];
    return NPG_OK;
}

int reduce_maxabs(npg_ctx *ctx, const double *x, int64_t n, double *out, int *has_nan) {
    const int g = grid_for(n, 1024);
    hipLaunchKernelGGL(k_maxabs_partial, dim3(g), dim3(kBlock), 0, ctx->stream, x, n, ctx->d_scratch + 8);
    NPG_HIP(hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch + 8, 2 * g * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    double m = 0.0, nan = 0.0;
    for (int i = 0; i < g; ++i) {
        m = std::fmax(m, ctx->h_scratch[2 * i]);
        nan = std::fmax(nan, ctx->h_scratch[2 * i + 1]);
    }
    *out = m;
    if (has_nan) *has_nan = nan > 0.0;
    return NPG_OK;
}

}  // namespace npg

using namespace npg;

// ---- context ----------------------------------------------------------------------------------------------------------
NPG_API const char *npg_last_error(void) { return g_err; }

// NPG_SEGV_BACKTRACE=1 (diagnostics; used to symbolise the rocprofv3 hipGraphLaunch fault): on SIGSEGV
// print the faulting address, the mapping that holds / ends at it and every frame with its module and offset, then exit.
static void maps_line(const void *addr) {
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return;
    char line[512];
    while (fgets(line, sizeof line, f)) {
        unsigned long lo, hi;
        if (sscanf(line, "%lx-%lx", &lo, &hi) == 2 && (unsigned long)addr >= lo && (unsigned long)addr < hi) {
            fprintf(stderr, "      maps: %s", line);
            break;
        }
    }
    fclose(f);
}
static void on_segv(int, siginfo_t *si, void *) {
    fprintf(stderr, "\n*** [npg] SIGSEGV at address %p\n    mappings below / at the faulting address:\n", si->si_addr);
    maps_line((const char *)si->si_addr - 1);
    maps_line(si->si_addr);
    void *fr[64];
    const int n = backtrace(fr, 64);
    for (int i = 0; i < n; ++i) {
        Dl_info di;
        if (dladdr(fr[i], &di) && di.dli_fname)
            fprintf(stderr, "  #%d %p  %s + 0x%lx  (%s)\n", i, fr[i], di.dli_fname,
                    (unsigned long)((char *)fr[i] - (char *)di.dli_fbase), di.dli_sname ? di.dli_sname : "?");
        else
            fprintf(stderr, "  #%d %p  ?\n", i, fr[i]);
    }
    _exit(139);
}
static void maybe_install_segv_handler() {
    static bool done = false;
    if (done) return;
    done = true;
    const char *t = getenv("NPG_SEGV_BACKTRACE");
    if (!t || atoi(t) == 0) return;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
}

NPG_API int npg_ctx_create(int device, npg_ctx **out) {
    NPG_REQUIRE(out != nullptr, "npg_ctx_create: out is NULL");
    maybe_install_segv_handler();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        set_error("npg_ctx_create: no HIP device visible - libnupgcm_hip has no CPU fallback");
        return NPG_ENODEV;
    }
    NPG_REQUIRE(device >= 0 && device < count, "npg_ctx_create: device %d out of range [0,%d)", device, count);
    NPG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    NPG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("npg_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return NPG_ENODEV;
    }
    npg_ctx *c = new npg_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    NPG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    NPG_HIP(hipEventCreate(&c->ev0));
    NPG_HIP(hipEventCreate(&c->ev1));
    c->scratch_doubles = 8192;
    NPG_HIP(hipMalloc((void **)&c->d_scratch, c->scratch_doubles * sizeof(double)));
    NPG_HIP(hipHostMalloc((void **)&c->h_scratch, c->scratch_doubles * sizeof(double), hipHostMallocDefault));
    *out = c;
    return NPG_OK;
}

NPG_API int npg_ctx_destroy(npg_ctx *ctx) {
    if (!ctx) return NPG_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->d_scratch) hipFree(ctx->d_scratch);
    if (ctx->h_scratch) hipHostFree(ctx->h_scratch);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return NPG_OK;
}

NPG_API int npg_ctx_sync(npg_ctx *ctx) {
    NPG_REQUIRE(ctx, "npg_ctx_sync: NULL context");
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    return NPG_OK;
}

NPG_API int npg_mem_status(npg_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    NPG_REQUIRE(ctx && free_bytes && total_bytes, "npg_mem_status: NULL argument");
    NPG_HIP(hipSetDevice(ctx->device));
    NPG_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return NPG_OK;
}

NPG_API int npg_device_name(npg_ctx *ctx, char *buf, size_t cap) {
    NPG_REQUIRE(ctx && buf && cap > 0, "npg_device_name: NULL argument");
    hipDeviceProp_t prop;
    NPG_HIP(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return NPG_OK;
}

NPG_API void *npg_ctx_stream(npg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

NPG_API int npg_timer_start(npg_ctx *ctx) {
    NPG_REQUIRE(ctx, "npg_timer_start: NULL context");
    NPG_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return NPG_OK;
}

NPG_API int npg_timer_stop(npg_ctx *ctx, double *ms) {
    NPG_REQUIRE(ctx && ms, "npg_timer_stop: NULL argument");
    NPG_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    NPG_HIP(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    NPG_HIP(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = f;
    return NPG_OK;
}

// ---- vectors ----------------------------------------------------------------------------------------------------------
NPG_API int npg_vec_create(npg_ctx *ctx, int64_t n, npg_vec **out) {
    NPG_REQUIRE(ctx && out && n >= 0, "npg_vec_create: bad argument");
    npg_vec *v = new npg_vec();
    v->ctx = ctx;
    v->n = n;
    NPG_HIP(hipSetDevice(ctx->device));
    NPG_HIP(hipMalloc((void **)&v->d, (size_t)(n > 0 ? n : 1) * sizeof(double)));
    NPG_HIP(hipMemsetAsync(v->d, 0, (size_t)(n > 0 ? n : 1) * sizeof(double), ctx->stream));
    *out = v;
    return NPG_OK;
}

NPG_API int npg_vec_destroy(npg_vec *v) {
    if (!v) return NPG_OK;
    if (v->owns && v->d) {
        hipStreamSynchronize(v->ctx->stream);
        hipFree(v->d);
    }
    delete v;
    return NPG_OK;
}

NPG_API int npg_vec_view(npg_vec *v, int64_t offset, int64_t n, npg_vec **out) {
    NPG_REQUIRE(v && out && offset >= 0 && n >= 0 && offset + n <= v->n, "npg_vec_view: window out of range");
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
    NPG_REQUIRE(v && host, "npg_vec_upload: NULL argument");
    NPG_HIP(hipMemcpyAsync(v->d, host, (size_t)v->n * sizeof(double), hipMemcpyHostToDevice, v->ctx->stream));
    NPG_HIP(hipStreamSynchronize(v->ctx->stream));
    return NPG_OK;
}

NPG_API int npg_vec_download(const npg_vec *v, double *host) {
    NPG_REQUIRE(v && host, "npg_vec_download: NULL argument");
    NPG_HIP(hipMemcpyAsync(host, v->d, (size_t)v->n * sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
    NPG_HIP(hipStreamSynchronize(v->ctx->stream));
    return NPG_OK;
}

NPG_API int npg_vec_upload_perm(npg_vec *v, const double *host, const int64_t *perm) {
    NPG_REQUIRE(v && host && perm, "npg_vec_upload_perm: NULL argument");
    int rc = ensure_stage(v->ctx, (size_t)v->n);
    if (rc) return rc;
    double *st = v->ctx->h_stage;
    for (int64_t i = 0; i < v->n; ++i) {
        NPG_REQUIRE(perm[i] >= 0, "npg_vec_upload_perm: negative index at %lld", (long long)i);
        st[i] = host[perm[i]];
    }
    NPG_HIP(hipMemcpyAsync(v->d, st, (size_t)v->n * sizeof(double), hipMemcpyHostToDevice, v->ctx->stream));
    NPG_HIP(hipStreamSynchronize(v->ctx->stream));
    return NPG_OK;
}

NPG_API int npg_vec_download_perm(const npg_vec *v, double *host, const int64_t *perm) {
    NPG_REQUIRE(v && host && perm, "npg_vec_download_perm: NULL argument");
    int rc = ensure_stage(v->ctx, (size_t)v->n);
    if (rc) return rc;
    double *st = v->ctx->h_stage;
    NPG_HIP(hipMemcpyAsync(st, v->d, (size_t)v->n * sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
    NPG_HIP(hipStreamSynchronize(v->ctx->stream));
    for (int64_t i = 0; i < v->n; ++i) {
        NPG_REQUIRE(perm[i] >= 0 && perm[i] < v->n, "npg_vec_download_perm: index out of range at %lld", (long long)i);
        host[i] = st[perm[i]];
    }
    return NPG_OK;
}

NPG_API int npg_vec_fill(npg_vec *v, double a) {
    NPG_REQUIRE(v, "npg_vec_fill: NULL vector");
    hipLaunchKernelGGL(k_fill, dim3(grid_for(v->n)), dim3(kBlock), 0, v->ctx->stream, v->d, a, v->n);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_vec_copy(npg_vec *dst, const npg_vec *src) {
    NPG_REQUIRE(dst && src && dst->n == src->n, "npg_vec_copy: length mismatch");
    NPG_HIP(hipMemcpyAsync(dst->d, src->d, (size_t)src->n * sizeof(double), hipMemcpyDeviceToDevice, dst->ctx->stream));
    return NPG_OK;
}

NPG_API int npg_vec_axpby(npg_vec *y, double a, const npg_vec *x, double b) {
    NPG_REQUIRE(y && x && y->n == x->n, "npg_vec_axpby: length mismatch");
    hipLaunchKernelGGL(k_axpby, dim3(grid_for(y->n)), dim3(kBlock), 0, y->ctx->stream, y->d, a, x->d, b, y->n);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_vec_mul(npg_vec *y, const npg_vec *d, const npg_vec *x) {
    NPG_REQUIRE(y && d && x && y->n == x->n && d->n == x->n, "npg_vec_mul: length mismatch");
    hipLaunchKernelGGL(k_mul, dim3(grid_for(y->n)), dim3(kBlock), 0, y->ctx->stream, y->d, d->d, x->d, y->n);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_vec_lincomb(npg_vec *y, int nterms, const double *coef, const npg_vec *const *xs) {
    NPG_REQUIRE(y && coef && xs && nterms >= 1 && nterms <= 8, "npg_vec_lincomb: need 1..8 terms");
    LinComb lc;
    lc.nterms = nterms;
    for (int k = 0; k < 8; ++k) {
        lc.coef[k] = 0.0;
        lc.x[k] = nullptr;
    }
    for (int k = 0; k < nterms; ++k) {
        NPG_REQUIRE(xs[k] && xs[k]->n == y->n, "npg_vec_lincomb: term %d length mismatch", k);
        lc.coef[k] = coef[k];
        lc.x[k] = xs[k]->d;
    }
    hipLaunchKernelGGL(k_lincomb, dim3(grid_for(y->n)), dim3(kBlock), 0, y->ctx->stream, y->d, lc, y->n);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

NPG_API int npg_vec_dot(const npg_vec *x, const npg_vec *y, double *out) {
    NPG_REQUIRE(x && y && out && x->n == y->n, "npg_vec_dot: length mismatch");
    return reduce_dot(x->ctx, x->d, y->d, x->n, out);
}

NPG_API int npg_vec_nrm2(const npg_vec *x, double *out) {
    NPG_REQUIRE(x && out, "npg_vec_nrm2: NULL argument");
    double s = 0.0;
    int rc = reduce_dot(x->ctx, x->d, x->d, x->n, &s);
    *out = std::sqrt(s);
    return rc;
}

NPG_API int npg_vec_is_constant(const npg_vec *x, double *value, int *is_constant) {
    NPG_REQUIRE(x && value && is_constant && x->n > 0, "npg_vec_is_constant: bad argument");
    npg_ctx *ctx = x->ctx;
    const int g = grid_for(x->n, 1024);
    hipLaunchKernelGGL(k_minmax_partial, dim3(g), dim3(kBlock), 0, ctx->stream, x->d, x->n, ctx->d_scratch + 8);
    NPG_HIP(hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch + 8, 2 * g * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < g; ++i) {
        lo = std::fmin(lo, ctx->h_scratch[2 * i]);
        hi = std::fmax(hi, ctx->h_scratch[2 * i + 1]);
    }
    *value = lo;
    *is_constant = lo == hi;
    return NPG_OK;
}

NPG_API int npg_vec_maxabs(const npg_vec *x, double *out, int *has_nan) {
    NPG_REQUIRE(x && out, "npg_vec_maxabs: NULL argument");
    return reduce_maxabs(x->ctx, x->d, x->n, out, has_nan);
}

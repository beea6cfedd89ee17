// General preconditioners for the saddle-point inversion and the flexible GMRES that drives them.
//
// The reference ships one experimental preconditioner, BlockDiagonalPreconditioner (src/preconditioners.jl:53-125: an inner
// CG per block - friction-only velocity block with ILU(0) / LU, pressure mass matrix / (alpha^2 eps^2) with Jacobi), which its
// author's own log shows to be slower in wall time than Diagonal(1/h^3) (scratch/inversion_log.md:107-191) because the
// friction block still needs O(1/h) inner iterations.  Two kinds live here behind one handle (npg_precond):
//
//   NPG_PC_BLOCKDIAG  the reference's construction: z[block k] = CG(A_k, r[block k]; Jacobi), warm-started from its previous
//                     output exactly as CgPreconditioner does (src/preconditioners.jl:24-37) - the device-resident CG of
//                     cg.hip is the inner solver;
//   NPG_PC_MG         geometric multigrid V-cycle on the WHOLE saddle-point system over the red-refinement hierarchy the big
//                     bowl meshes are made from (new work): rediscretised operators per level, P2 / P1
//                     nodal interpolation, and a Braess-Sarazin smoother built from node blocks:
//                         [ w Dh   G ] [du]   [r_u]      Dh = node-block diagonal of the velocity block (friction on the
//                         [ D      0 ] [dp] = [r_p]           diagonal, Coriolis coupling x and y of the same node),
//                     whose pressure system S dp = D Dh^-1 r_u - w r_p, S = D Dh^-1 G (explicit, assembled at set-up), is
//                     relaxed by a few damped-Jacobi sweeps.  Everything is SpMV + diagonal work: no triangular solves.
//
// Both are applied inexactly, so the outer solver is right-preconditioned FLEXIBLE GMRES(m) (npg_fgmres_*): classical
// Gram-Schmidt with a full second pass (two fused multi-dot reductions per step, one host synchronisation per iteration -
// an iteration costs a V-cycle, i.e. milliseconds on the meshes where this is used, so the host drives the loop).  The
// stopping rule is the reference's: scale * ||r|| <= atol + rtol * scale * ||r0|| with scale = 1/h^dim, the factor its
// Diagonal preconditioner puts on the residual (src/inversion.jl:42-54; right preconditioning leaves r the true residual).
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cmath>
#include <map>

#include "common.h"
#include "device_utils.h"

namespace npg {

static inline int grid_for(int64_t n, int cap = 2048) {
    int64_t g = (n + kBlock - 1) / kBlock;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// y = a x + b y (b == 0: y is not read).  x may BE y (in-place scaling by flexible GMRES): no __restrict__ here.
__global__ void k_mg_axpby(double *y, double a, const double *x, double b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

// y = w d x + b y
__global__ void k_mg_daxpby(double *__restrict__ y, double w, const double *__restrict__ d, const double *__restrict__ x,
                            double b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (b == 0.0) ? w * d[i] * x[i] : w * d[i] * x[i] + b * y[i];
}

// part[block][c] = sum_i V_c[i] w[i] for c < k (k <= 8 NG); V column-major with leading dimension ld
template <int NG>
__global__ void __launch_bounds__(kBlock) k_mdot(const double *__restrict__ V, int64_t ld, int k,
                                                 const double *__restrict__ w, int64_t n, double *__restrict__ part) {
    __shared__ double sh[4 * kPartStride];
    double acc[8 * NG];
#pragma unroll
    for (int c = 0; c < 8 * NG; ++c) acc[c] = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double wi = w[i];
#pragma unroll
        for (int c = 0; c < 8 * NG; ++c)
            if (c < k) acc[c] += V[(size_t)c * ld + i] * wi;
    }
    block_store_partials<8 * NG, 4>(acc, k, sh, part);
}

// w -= sum_c h[c] V_c ; part[block][0] = sum w_new^2
template <int NG>
__global__ void __launch_bounds__(kBlock) k_mupdate(const double *__restrict__ V, int64_t ld, int k,
                                                    const double *__restrict__ h, double *__restrict__ w, int64_t n,
                                                    double *__restrict__ part) {
    __shared__ double sh[4 * kPartStride];
    double hc[8 * NG];
#pragma unroll
    for (int c = 0; c < 8 * NG; ++c) hc[c] = c < k ? h[c] : 0.0;
    double acc[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double wi = w[i];
#pragma unroll
        for (int c = 0; c < 8 * NG; ++c)
            if (c < k) wi -= hc[c] * V[(size_t)c * ld + i];
        w[i] = wi;
        acc[0] += wi * wi;
    }
    block_store_partials<1, 4>(acc, 1, sh, part);
}

// out[c] = sum_b part[b][c]  (one block; fixed order)
__global__ void __launch_bounds__(kBlock) k_sum_partials(const double *__restrict__ part, int nblocks, int nvals,
                                                         double *__restrict__ out) {
    __shared__ double tmp[8 * kPartStride], res[kPartStride];
    reduce_partials<8, 128>(part, nblocks, nvals, tmp, res);
    if (threadIdx.x < nvals) out[threadIdx.x] = res[threadIdx.x];
}

// ---- explicit dense inverse (NPG_PC_DENSE) ----------------------------------------------------------------------------------
// dA (column-major n x n, zero-filled) <- CSR
__global__ void k_densify(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
                          int64_t n, double *__restrict__ dA) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) dA[r + (int64_t)col[k] * n] = val[k];
}

// part[s][i] = sum over the s-th chunk of columns j of M[i + j n] x[j]: thread = row i, consecutive lanes read consecutive
// doubles of one column (coalesced), eight columns in flight per thread; blockIdx.y = chunk
constexpr int kGemvChunkCols = 512;
template <typename T>
__global__ void __launch_bounds__(kBlock) k_dense_gemv_part(const T *__restrict__ M, int64_t n,
                                                            const double *__restrict__ x, double *__restrict__ part) {
    __shared__ double xs[kGemvChunkCols];
    const int64_t j0 = (int64_t)blockIdx.y * kGemvChunkCols;
    const int nj = (int)min((int64_t)kGemvChunkCols, n - j0);
    for (int j = threadIdx.x; j < nj; j += kBlock) xs[j] = x[j0 + j];
    __syncthreads();
    const int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (i >= n) return;
    const T *__restrict__ Mi = M + i + j0 * n;
    double acc = 0.0;
    int j = 0;
    for (; j + 8 <= nj; j += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(Mi + (int64_t)(j + u) * n);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)v[u] * xs[j + u];
    }
    for (; j < nj; ++j) acc += (double)Mi[(int64_t)j * n] * xs[j];
    part[(int64_t)blockIdx.y * n + i] = acc;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float *p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// fp32 inverse with a padded leading dimension (ld = n rounded up to 4 floats: every column starts 16-byte aligned): thread =
// FOUR adjacent rows (one 16-byte load per column), four columns per trip and the next trip's loads issued before this trip's
// arithmetic (the pattern that took the Krylov row kernels from 3.9 to 5.5 TB/s: profiles/r03_rows_bench.txt)
__global__ void __launch_bounds__(kBlock) k_dense_gemv_part4(const float *__restrict__ M, int64_t n, int64_t ld,
                                                             const double *__restrict__ x, double *__restrict__ part) {
    __shared__ double xs[kGemvChunkCols];
    const int64_t j0 = (int64_t)blockIdx.y * kGemvChunkCols;
    const int nj = (int)min((int64_t)kGemvChunkCols, n - j0);
    for (int j = threadIdx.x; j < nj; j += kBlock) xs[j] = x[j0 + j];
    __syncthreads();
    const int64_t i = 4 * (blockIdx.x * (int64_t)kBlock + threadIdx.x);
    if (i >= n) return;
    const float *__restrict__ Mi = M + i + j0 * ld;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    constexpr int U = 4;
    float4 v[U], vn[U];
    const int ntrip = nj / U;                        // whole trips; the tail columns one by one below
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (ntrip > 0) v[u] = nt_load4(Mi + (int64_t)u * ld);
    for (int tr = 0; tr < ntrip; ++tr) {
        const int j = tr * U;
        if (tr + 1 < ntrip) {
#pragma unroll
            for (int u = 0; u < U; ++u) vn[u] = nt_load4(Mi + (int64_t)(j + U + u) * ld);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double xv = xs[j + u];
            a0 += (double)v[u].x * xv;
            a1 += (double)v[u].y * xv;
            a2 += (double)v[u].z * xv;
            a3 += (double)v[u].w * xv;
        }
        if (tr + 1 < ntrip) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = vn[u];
        }
    }
    for (int j = ntrip * U; j < nj; ++j) {
        const float4 f = *reinterpret_cast<const float4 *>(Mi + (int64_t)j * ld);
        const double xv = xs[j];
        a0 += (double)f.x * xv;
        a1 += (double)f.y * xv;
        a2 += (double)f.z * xv;
        a3 += (double)f.w * xv;
    }
    double *out = part + (int64_t)blockIdx.y * n + i;          // (rows past n are padding of M: computed, not stored)
    out[0] = a0;
    if (i + 1 < n) out[1] = a1;
    if (i + 2 < n) out[2] = a2;
    if (i + 3 < n) out[3] = a3;
}

// fp16 inverse, every column scaled by its largest magnitude (cs[j]; the scale is folded into x when the chunk of x is staged):
// thread = EIGHT adjacent rows (one 16-byte load per column), the same four-columns-per-trip pipeline; products and the sums
// over a chunk's 512 columns in fp32 (the stored numbers carry 11 bits), the sum over the chunks in fp64 (k_dense_gemv_sum)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 nt_load8h(const _Float16 *p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return __builtin_bit_cast(f16x8, v);
}
__global__ void __launch_bounds__(kBlock) k_dense_gemv_part8h(const _Float16 *__restrict__ M, int64_t n, int64_t ld,
                                                              const double *__restrict__ cs, const double *__restrict__ x,
                                                              double *__restrict__ part) {
    __shared__ float xs[kGemvChunkCols];
    const int64_t j0 = (int64_t)blockIdx.y * kGemvChunkCols;
    const int nj = (int)min((int64_t)kGemvChunkCols, n - j0);
    for (int j = threadIdx.x; j < nj; j += kBlock) xs[j] = (float)(x[j0 + j] * cs[j0 + j]);
    __syncthreads();
    const int64_t i = 8 * (blockIdx.x * (int64_t)kBlock + threadIdx.x);
    if (i >= n) return;
    const _Float16 *__restrict__ Mi = M + i + j0 * ld;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    f16x8 v[U], vn[U];
    const int ntrip = nj / U;
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (ntrip > 0) v[u] = nt_load8h(Mi + (int64_t)u * ld);
    for (int tr = 0; tr < ntrip; ++tr) {
        const int j = tr * U;
        if (tr + 1 < ntrip) {
#pragma unroll
            for (int u = 0; u < U; ++u) vn[u] = nt_load8h(Mi + (int64_t)(j + U + u) * ld);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float xv = xs[j + u];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf((float)v[u][k], xv, a[k]);
        }
        if (tr + 1 < ntrip) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = vn[u];
        }
    }
    for (int j = ntrip * U; j < nj; ++j) {
        const f16x8 f = *reinterpret_cast<const f16x8 *>(Mi + (int64_t)j * ld);
        const float xv = xs[j];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_fmaf((float)f[k], xv, a[k]);
    }
    double *out = part + (int64_t)blockIdx.y * n + i;          // (rows past n are padding of M: computed, not stored)
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (i + k < n) out[k] = (double)a[k];
}

// cs[j] = max_i |M[i, j]| of a column-major fp64 n x n array (1 for an all-zero column): one workgroup per column
__global__ void __launch_bounds__(kBlock) k_col_absmax(const double *__restrict__ M, int64_t n, double *__restrict__ cs) {
    __shared__ double sh[kBlock];
    for (int64_t j = blockIdx.x; j < n; j += gridDim.x) {
        double m = 0.0;
        for (int64_t i = threadIdx.x; i < n; i += kBlock) m = fmax(m, fabs(M[j * n + i]));
        sh[threadIdx.x] = m;
        __syncthreads();
        for (int s = kBlock / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
            __syncthreads();
        }
        if (threadIdx.x == 0) cs[j] = sh[0] > 0.0 ? sh[0] : 1.0;
        __syncthreads();
    }
}

// fp64 column-major n x n -> fp16 with leading dimension ld >= n, column j divided by cs[j] (pad rows zeroed)
__global__ void k_to_half_ld(const double *__restrict__ src, const double *__restrict__ cs, _Float16 *__restrict__ dst, int64_t n,
                             int64_t ld) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < ld * n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = k / ld, i = k - j * ld;
        dst[k] = i < n ? (_Float16)(float)(src[j * n + i] / cs[j]) : (_Float16)0.f;
    }
}

// fp64 column-major n x n -> fp32 with leading dimension ld >= n (pad rows zeroed)
__global__ void k_to_float_ld(const double *__restrict__ src, float *__restrict__ dst, int64_t n, int64_t ld) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < ld * n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = k / ld, i = k - j * ld;
        dst[k] = i < n ? (float)src[j * n + i] : 0.f;
    }
}

// z = a (sum_s part[s]) + b z   (fixed order)
__global__ void __launch_bounds__(kBlock) k_dense_gemv_sum(const double *__restrict__ part, int nsplit, int64_t n, double a,
                                                           double b, double *__restrict__ z) {
    const int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += part[(int64_t)k * n + i];
    z[i] = (b == 0.0) ? a * s : a * s + b * z[i];
}

__global__ void k_to_float(const double *__restrict__ src, float *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

// x += sum_c y[c] Z_c
struct Coefs {
    double y[kMaxMem];
};
__global__ void __launch_bounds__(kBlock) k_combine_z(double *__restrict__ x, const double *__restrict__ Z, int64_t ld,
                                                      int k, Coefs cf, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double s = x[i];
        for (int c = 0; c < k; ++c) s += cf.y[c] * Z[(size_t)c * ld + i];
        x[i] = s;
    }
}

}  // namespace npg

using namespace npg;

struct MgLevel {
    const npg_csr *A = nullptr, *G = nullptr, *D = nullptr, *Dinv = nullptr, *S = nullptr, *P = nullptr, *R = nullptr;
    const npg_csr *Gh = nullptr;                     // Dinv G (npg_precond_mg_set_scaled_gradient): one Dinv product less per step
    int64_t n = 0, nu = 0, np = 0;
    double *sdinv = nullptr;                         // 1 / diag(S)
    double *r = nullptr, *t = nullptr, *rhs = nullptr, *dp = nullptr, *dp2 = nullptr, *res = nullptr, *x = nullptr,
           *b = nullptr;
    // distributed level (npg_precond_mg_set_level_dist): this rank's rows of every operator; n, nu, np count OWNED rows, the
    // operators' column spaces are [owned | ghosts] and the vectors that feed them carry the ghost entries behind the owned
    // ones, filled by these plans before each product - hx: whole vectors (A), hu: velocity parts (D), hp: pressure parts (G, S)
    npg_halo *hx = nullptr, *hu = nullptr, *hp = nullptr;
    bool dist = false;
    // transfers between TWO distributed levels (npg_precond_mg_set_transfer_dist; this level being the finer one): P holds this
    // rank's rows over the coarser level's [owned | ghosts of hP] columns, R the coarser level's owned rows over this level's
    // [owned | ghosts of hR] columns; the vectors that feed them are copied into xP / rR, whose ghost segments the plans fill
    npg_halo *hP = nullptr, *hR = nullptr;
    double *xP = nullptr, *rR = nullptr;
    // fp32 gather-layout copy of the iterate for the residuals r = b - A x of a large level whose A carries windowed tiles
    // (csr.hip: spmv_epi_gather32; mg_prepare)
    float *xg = nullptr;
    int64_t xg_n = 0;
    bool nb_ok = false;        // t = Dinv r_u may ride in the residual kernel (node-blocked A with Dinv's node counts; mg_prepare)
};

struct BlockPc {
    int64_t off = 0, n = 0;
    const npg_csr *A = nullptr;
    const npg_vec *jac = nullptr;
    npg_cg *cg = nullptr;
    npg_ilu0 *ilu = nullptr;                         // M of the block's CG: these factors instead of the Jacobi vector
    npg_vec *xk = nullptr, *rk = nullptr;            // the block's warm-started solution / right-hand side
    int64_t itmax = 0;
    double atol = 0.0, rtol = 0.0;
};

struct DenseInv {
    int64_t n = 0;
    double *M = nullptr;       // A^-1, column-major (fp64 storage) ...
    float *Mf = nullptr;       // ... or rounded to fp32 (half the bytes per application; a preconditioner may be inexact),
    int64_t ldf = 0;           //     leading dimension ldf = n rounded up to 4 (16-byte aligned columns)
    _Float16 *Mh = nullptr;    // ... or to fp16, column j divided by cs[j] = its largest magnitude (a quarter of the bytes),
    double *cs = nullptr;      //     leading dimension ldf = n rounded up to 8
    double *part = nullptr;    // [nsplit][n] partial products
    int nsplit = 0;
};

static void dense_free(DenseInv &d) {
    if (d.M) hipFree(d.M);
    if (d.Mf) hipFree(d.Mf);
    if (d.Mh) hipFree(d.Mh);
    if (d.cs) hipFree(d.cs);
    if (d.part) hipFree(d.part);
    d = DenseInv{};
}

struct npg_precond {
    npg_ctx *ctx = nullptr;
    int kind = 0;
    int64_t n = 0;
    DenseInv dense;            // NPG_PC_DENSE, or the multigrid's exact coarsest-level solve
    // multigrid
    std::vector<MgLevel> L;
    double omega = 2.5, jw = 0.7;
    int sweeps = 3, nu1 = 2, nu2 = 2, coarse = 20, gamma = 1;
    int mixed = 0;             // the cycle's SpMVs read fp32 copies of the level matrices' values (npg_precond_mg_set_mixed)
    std::vector<void *> allocs;
    // one instantiated hipGraph of the V-cycle per (input, output) address pair: flexible GMRES applies the preconditioner
    // to the same `memory` basis / Z column pairs in every restart cycle, so a cycle's ~300 launches (most of them
    // latency-bound coarse-level kernels) are enqueued by one hipGraphLaunch
    std::map<std::pair<const double *, double *>, hipGraphExec_t> graphs;
    uint64_t graphs_gen = 0;   // sum of the borrowed level matrices' generation counters when the graphs were captured
    bool use_graphs = true;
    // block diagonal
    std::vector<BlockPc> blocks;
    int64_t inner_iterations = 0, applications = 0;
    int64_t cycle_bytes = 0;   // bytes one application streams as its operators are laid out (counted once: byte_sink, common.h)
};

static void drop_graphs(npg_precond *pc) {
    for (auto &kv : pc->graphs) hipGraphExecDestroy(kv.second);
    pc->graphs.clear();
}

static inline void axpby(npg_ctx *c, double *y, double a, const double *x, double b, int64_t n) {
    if (byte_sink) *byte_sink += (b != 0.0 ? 24 : 16) * n;
    hipLaunchKernelGGL(k_mg_axpby, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, y, a, x, b, n);
}
static inline void daxpby(npg_ctx *c, double *y, double w, const double *d, const double *x, double b, int64_t n) {
    hipLaunchKernelGGL(k_mg_daxpby, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, y, w, d, x, b, n);
}

NPG_API int npg_precond_create(npg_ctx *ctx, int kind, int nparts, npg_precond **out) {
    NPG_REQUIRE(ctx && out, "npg_precond_create: NULL argument");
    NPG_REQUIRE(kind == NPG_PC_MG || kind == NPG_PC_BLOCKDIAG || kind == NPG_PC_DENSE, "npg_precond_create: unknown kind %d",
                kind);
    NPG_REQUIRE(nparts >= 1 && nparts <= 16, "npg_precond_create: need 1..16 levels / blocks");
    npg_precond *pc = new npg_precond();
    pc->ctx = ctx;
    pc->kind = kind;
    // relaunching hipGraphs under rocprofv3's kernel tracer can fault inside the tool (a batch of AQL packets that straddles
    // the end of the queue ring: profiles/r03_rocprofv3_graph_fault.txt): a traced process launches the cycle eagerly unless
    // NPG_MG_EAGER=0 says otherwise
    const bool traced = getenv("ROCPROFILER_LIBRARY_CTOR") || getenv("ROCPROF_OUTPUT_PATH") || getenv("ROCP_TOOL_LIBRARIES");
    pc->use_graphs = getenv("NPG_MG_EAGER") ? atoi(getenv("NPG_MG_EAGER")) == 0 : !traced;
    static bool said = false;
    if (traced && !getenv("NPG_MG_EAGER") && !said && (said = true))
        fprintf(stderr, "[npg] rocprofv3 detected: multigrid cycles are launched eagerly instead of replayed from hipGraphs "
                        "(NPG_MG_EAGER=0 overrides)\n");
    if (kind == NPG_PC_MG) pc->L.resize(nparts);
    if (kind == NPG_PC_BLOCKDIAG) pc->blocks.resize(nparts);
    *out = pc;
    return NPG_OK;
}

NPG_API int npg_precond_destroy(npg_precond *pc) {
    if (!pc) return NPG_OK;
    hipStreamSynchronize(pc->ctx->stream);
    drop_graphs(pc);
    dense_free(pc->dense);
    for (void *p : pc->allocs) hipFree(p);
    for (MgLevel &l : pc->L)
        if (l.xg) hipFree(l.xg);
    for (BlockPc &b : pc->blocks) {
        if (b.cg) npg_cg_destroy(b.cg);
        if (b.xk) npg_vec_destroy(b.xk);
        if (b.rk) npg_vec_destroy(b.rk);
    }
    delete pc;
    return NPG_OK;
}

// z = a A^-1 r + b z
static int dense_apply(npg_precond *pc, const double *r, double *z, double a, double b) {
    const DenseInv &d = pc->dense;
    hipStream_t st = pc->ctx->stream;
    if (byte_sink)          // the inverse in the storage that is read, + the split-column partial sums written and re-read, + r, z
        *byte_sink += (d.Mh ? 2 * d.n * d.ldf : d.Mf ? 4 * d.n * d.ldf : 8 * d.n * d.n) + 16 * (int64_t)d.nsplit * d.n + 24 * d.n;
    const int gx = (int)((d.n + kBlock - 1) / kBlock);
    if (d.Mh)
        hipLaunchKernelGGL(k_dense_gemv_part8h, dim3((unsigned)((d.n + 8 * kBlock - 1) / (8 * kBlock)), d.nsplit), dim3(kBlock), 0, st,
                           (const _Float16 *)d.Mh, d.n, d.ldf, (const double *)d.cs, r, d.part);
    else if (d.Mf)
        hipLaunchKernelGGL(k_dense_gemv_part4, dim3((unsigned)((d.n + 4 * kBlock - 1) / (4 * kBlock)), d.nsplit), dim3(kBlock), 0, st,
                           (const float *)d.Mf, d.n, d.ldf, r, d.part);
    else
        hipLaunchKernelGGL(k_dense_gemv_part<double>, dim3(gx, d.nsplit), dim3(kBlock), 0, st, d.M, d.n, r, d.part);
    hipLaunchKernelGGL(k_dense_gemv_sum, dim3(gx), dim3(kBlock), 0, st, d.part, d.nsplit, d.n, a, b, z);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

// A^-1 of a plain-CSR matrix as a dense fp64 array in HBM: densify, LU with partial pivoting and inversion by rocSOLVER
// (set-up; n^2 doubles - 2 GB at 16 k unknowns, 8 GB at 31 k), applied per solve by the hand-written GEMV above
static int dense_build(npg_precond *pc, const npg_csr *A, bool fp32, bool fp16 = false) {
    NPG_REQUIRE(A && A->m == A->n && A->nnode() == 0, "dense inverse: a square plain-CSR matrix is required");
    const int64_t n = A->m;
    // (46 340 = floor(sqrt(2^31)): rocSOLVER's getrf / getri address the n x n array with 32-bit element offsets - at 58 295 unknowns,
    //  the channel basin's coarsest level, they fault; round 5, profiles/r05_coarse_viscosity.txt)
    NPG_REQUIRE(n > 0 && n <= 46340, "dense inverse: %lld unknowns (limit 46 340: rocSOLVER's getrf / getri index the n x n array with 32-bit offsets)",
                (long long)n);
    hipStream_t st = pc->ctx->stream;
    DenseInv &d = pc->dense;
    dense_free(d);
    d.n = n;
    d.nsplit = (int)((n + kGemvChunkCols - 1) / kGemvChunkCols);
    NPG_HIP(hipMalloc((void **)&d.M, (size_t)n * n * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&d.part, (size_t)d.nsplit * n * sizeof(double)));
    NPG_HIP(hipMemsetAsync(d.M, 0, (size_t)n * n * sizeof(double), st));
    hipLaunchKernelGGL(k_densify, dim3(grid_for(n)), dim3(kBlock), 0, st, A->rowptr, A->col, A->val, n, d.M);
    NPG_HIP(hipGetLastError());
    rocblas_handle h = nullptr;
    rocblas_int *ipiv = nullptr, *info = nullptr;
    struct Guard {                       // every early return below releases the handle and the temporaries
        rocblas_handle &h;
        rocblas_int *&ipiv, *&info;
        ~Guard() {
            if (ipiv) hipFree(ipiv);
            if (info) hipFree(info);
            if (h) rocblas_destroy_handle(h);
        }
    } guard{h, ipiv, info};
    NPG_REQUIRE(rocblas_create_handle(&h) == rocblas_status_success, "dense inverse: rocblas_create_handle failed");
    rocblas_set_stream(h, st);
    NPG_HIP(hipMalloc((void **)&ipiv, (size_t)n * sizeof(rocblas_int)));
    NPG_HIP(hipMalloc((void **)&info, sizeof(rocblas_int)));
    rocblas_status s1 = rocsolver_dgetrf(h, (rocblas_int)n, (rocblas_int)n, d.M, (rocblas_int)n, ipiv, info);
    rocblas_int i1 = -1, i2 = -1;
    NPG_HIP(hipMemcpyAsync(&i1, info, sizeof i1, hipMemcpyDeviceToHost, st));
    NPG_HIP(hipStreamSynchronize(st));
    rocblas_status s2 = rocblas_status_success;
    if (s1 == rocblas_status_success && i1 == 0) {
        s2 = rocsolver_dgetri(h, (rocblas_int)n, d.M, (rocblas_int)n, ipiv, info);
        NPG_HIP(hipMemcpyAsync(&i2, info, sizeof i2, hipMemcpyDeviceToHost, st));
        NPG_HIP(hipStreamSynchronize(st));
    }
    NPG_REQUIRE(s1 == rocblas_status_success && i1 == 0 && s2 == rocblas_status_success && i2 == 0,
                "dense inverse: rocSOLVER getrf/getri failed (status %d/%d, info %d/%d: the matrix is singular to working "
                "precision, or out of memory)", (int)s1, (int)s2, (int)i1, (int)i2);
    if (fp16) {
        d.ldf = (n + 7) / 8 * 8;
        NPG_HIP(hipMalloc((void **)&d.cs, (size_t)n * sizeof(double)));
        NPG_HIP(hipMalloc((void **)&d.Mh, (size_t)d.ldf * n * sizeof(_Float16)));
        hipLaunchKernelGGL(k_col_absmax, dim3((unsigned)std::min<int64_t>(n, 4096)), dim3(kBlock), 0, st, d.M, n, d.cs);
        hipLaunchKernelGGL(k_to_half_ld, dim3(4096), dim3(kBlock), 0, st, d.M, d.cs, d.Mh, n, d.ldf);
        NPG_HIP(hipGetLastError());
        NPG_HIP(hipStreamSynchronize(st));
        // Column scaling keeps every column's largest entry exact, but rows of very different magnitude within one column (velocity
        // against pressure rows of a Stokes inverse differ by alpha^2 eps^2) may fall below fp16's range and flush to zero: what the
        // rounded inverse does to two probe vectors is compared, row by row, with the fp64 inverse while both are in memory; if the
        // typical row is off by more than fp16's rounding explains, or any sizeable row is lost, the fp32 storage is taken instead
        // (ADVICE round 4).  NPG_MG_COARSE_FP16_CHECK=0 skips the check.
        static const bool check16 = !getenv("NPG_MG_COARSE_FP16_CHECK") || atoi(getenv("NPG_MG_COARSE_FP16_CHECK")) != 0;
        bool keep16 = true;
        if (check16) {
            double *rv = nullptr, *y64 = nullptr, *y16 = nullptr;
            NPG_HIP(hipMalloc((void **)&rv, 3 * (size_t)n * sizeof(double)));
            y64 = rv + n;
            y16 = y64 + n;
            std::vector<double> h((size_t)n), a((size_t)n), b((size_t)n);
            for (int probe = 0; probe < 2 && keep16; ++probe) {
                for (int64_t i = 0; i < n; ++i)
                    h[(size_t)i] = probe == 0 ? std::sin(0.37 * (double)i) + 0.5 : ((i * 2654435761u) & 64 ? 1.0 : -1.0) * (1.0 + (double)(i % 7));
                NPG_HIP(hipMemcpy(rv, h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
                _Float16 *keep = d.Mh;
                d.Mh = nullptr;
                int rcd = dense_apply(pc, rv, y64, 1.0, 0.0);          // fp64 inverse
                d.Mh = keep;
                if (!rcd) rcd = dense_apply(pc, rv, y16, 1.0, 0.0);    // column-scaled fp16 inverse
                if (rcd) {
                    hipFree(rv);
                    return rcd;
                }
                NPG_HIP(hipMemcpyAsync(a.data(), y64, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
                NPG_HIP(hipMemcpyAsync(b.data(), y16, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
                NPG_HIP(hipStreamSynchronize(st));
                double amax = 0.0;
                for (double v : a) amax = std::max(amax, std::fabs(v));
                std::vector<double> rel;
                rel.reserve((size_t)n);
                for (int64_t i = 0; i < n; ++i)
                    if (std::fabs(a[(size_t)i]) > 1e-8 * amax) rel.push_back(std::fabs(b[(size_t)i] - a[(size_t)i]) / std::fabs(a[(size_t)i]));
                if (rel.empty()) continue;
                std::sort(rel.begin(), rel.end());
                const double med = rel[rel.size() / 2], p99 = rel[(size_t)(0.99 * (double)(rel.size() - 1))];
                if (!(med <= 5e-3 && p99 <= 0.2)) {
                    keep16 = false;
                    fprintf(stderr, "[npg mg] coarsest level: the column-scaled fp16 inverse differs from the fp64 one by %.1e (median row) / "
                                    "%.1e (99th percentile) on a probe vector - fp32 storage instead\n", med, p99);
                }
            }
            NPG_HIP(hipFree(rv));
        }
        if (!keep16) {
            NPG_HIP(hipFree(d.Mh));
            NPG_HIP(hipFree(d.cs));
            d.Mh = nullptr;
            d.cs = nullptr;
            d.ldf = (n + 3) / 4 * 4;
            NPG_HIP(hipMalloc((void **)&d.Mf, (size_t)d.ldf * n * sizeof(float)));
            hipLaunchKernelGGL(k_to_float_ld, dim3(4096), dim3(kBlock), 0, st, d.M, d.Mf, n, d.ldf);
            NPG_HIP(hipGetLastError());
            NPG_HIP(hipStreamSynchronize(st));
        }
        NPG_HIP(hipFree(d.M));
        d.M = nullptr;
    } else if (fp32) {
        d.ldf = (n + 3) / 4 * 4;
        NPG_HIP(hipMalloc((void **)&d.Mf, (size_t)d.ldf * n * sizeof(float)));
        hipLaunchKernelGGL(k_to_float_ld, dim3(4096), dim3(kBlock), 0, st, d.M, d.Mf, n, d.ldf);
        NPG_HIP(hipGetLastError());
        NPG_HIP(hipStreamSynchronize(st));
        NPG_HIP(hipFree(d.M));
        d.M = nullptr;
    }
    return NPG_OK;
}

NPG_API int npg_precond_dense_set(npg_precond *pc, const npg_csr *A, int fp32_storage) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_DENSE, "npg_precond_dense_set: not a dense-inverse preconditioner");
    int rc = dense_build(pc, A, fp32_storage != 0);
    if (rc) return rc;
    pc->n = A->m;
    return NPG_OK;
}

// the multigrid's coarsest level solved exactly by its explicit inverse instead of `coarse_sweeps` smoothing steps; call
// again after npg_precond_mg_update_level(level 0) to rebuild it
NPG_API int npg_precond_mg_set_coarse_dense(npg_precond *pc, int on) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG && !pc->L.empty() && pc->L[0].A, "npg_precond_mg_set_coarse_dense: level 0 is not set");
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    drop_graphs(pc);
    if (!on) {
        dense_free(pc->dense);
        return NPG_OK;
    }
    NPG_REQUIRE(on >= 1 && on <= 3, "npg_precond_mg_set_coarse_dense: mode %d (0 off, 1 fp64, 2 fp32, 3 scaled fp16 storage)", on);
    return dense_build(pc, pc->L[0].A, on == 2, on == 3);
}

// fp32 copies of one level's operators (mixed mode)
static int mg_refresh_fp32(const MgLevel &l) {
    int rc;
    for (const npg_csr *M : {l.A, l.G, l.D, l.Dinv, l.S, l.P, l.R, l.Gh})
        if (M && (rc = csr_refresh_fp32(M))) return rc;
    return NPG_OK;
}

static int mg_alloc(npg_precond *pc, double **p, int64_t n) {
    NPG_HIP(hipMalloc((void **)p, std::max<size_t>(1, (size_t)n) * sizeof(double)));
    pc->allocs.push_back(*p);
    NPG_HIP(hipMemsetAsync(*p, 0, std::max<size_t>(1, (size_t)n) * sizeof(double), pc->ctx->stream));
    return NPG_OK;
}

NPG_API int npg_precond_mg_set_level(npg_precond *pc, int level, const npg_csr *A, int64_t nu, const npg_csr *G,
                                     const npg_csr *D, const npg_csr *Dinv, const npg_csr *S, const npg_csr *P,
                                     const npg_csr *R) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG, "npg_precond_mg_set_level: not a multigrid preconditioner");
    NPG_REQUIRE(level >= 0 && level < (int)pc->L.size(), "npg_precond_mg_set_level: level %d out of range", level);
    NPG_REQUIRE(A && G && D && Dinv && S, "npg_precond_mg_set_level: NULL operator");
    const int64_t n = A->m, np = n - nu;
    NPG_REQUIRE(A->n == n && nu > 0 && np > 0, "npg_precond_mg_set_level: A must be square with 0 < nu < n");
    NPG_REQUIRE(G->m == nu && G->n == np && D->m == np && D->n == nu && Dinv->m == nu && Dinv->n == nu && S->m == np &&
                    S->n == np,
                "npg_precond_mg_set_level: block shapes do not match (n = %lld, nu = %lld)", (long long)n, (long long)nu);
    NPG_REQUIRE((level == 0) == (P == nullptr) && (P == nullptr) == (R == nullptr),
                "npg_precond_mg_set_level: level 0 takes no transfer operators, every other level needs P and R");
    MgLevel &l = pc->L[level];
    NPG_REQUIRE(l.A == nullptr, "npg_precond_mg_set_level: level %d is already set", level);
    if (P) {
        NPG_REQUIRE(pc->L[level - 1].A, "npg_precond_mg_set_level: set the levels coarse to fine");
        const int64_t nc = pc->L[level - 1].n;
        NPG_REQUIRE(P->m == n && P->n == nc && R->m == nc && R->n == n,
                    "npg_precond_mg_set_level: P must be %lld x %lld and R its transpose", (long long)n, (long long)nc);
    }
    l.A = A; l.G = G; l.D = D; l.Dinv = Dinv; l.S = S; l.P = P; l.R = R;
    l.n = n; l.nu = nu; l.np = np;
    int rc;
    if ((rc = mg_alloc(pc, &l.sdinv, np))) return rc;
    npg_vec sv;
    sv.ctx = pc->ctx; sv.n = np; sv.d = l.sdinv; sv.owns = false;
    if ((rc = npg_csr_inv_diag(S, &sv))) return rc;
    if ((rc = mg_alloc(pc, &l.r, n)) || (rc = mg_alloc(pc, &l.t, nu)) || (rc = mg_alloc(pc, &l.rhs, np)) ||
        (rc = mg_alloc(pc, &l.dp, np)) || (rc = mg_alloc(pc, &l.dp2, np)) || (rc = mg_alloc(pc, &l.res, np)))
        return rc;
    if (level + 1 < (int)pc->L.size())
        if ((rc = mg_alloc(pc, &l.x, n)) || (rc = mg_alloc(pc, &l.b, n))) return rc;
    pc->n = n;      // the finest level set so far
    if (pc->mixed && (rc = mg_refresh_fp32(l))) return rc;
    return NPG_OK;
}

// The finest level of a hierarchy, distributed: this rank's rows of every operator (column spaces [owned | ghosts]) and the
// three halo plans that fill the ghosts of the vectors feeding them; the levels below it are replicated (set with
// npg_precond_mg_set_level on every rank).  P: owned rows x all coarse columns, R: all coarse rows x owned columns.
NPG_API int npg_precond_mg_set_level_dist(npg_precond *pc, int level, const npg_csr *A, int64_t nu, const npg_csr *G,
                                          const npg_csr *D, const npg_csr *Dinv, const npg_csr *S, const npg_csr *P,
                                          const npg_csr *R, npg_halo *hx, npg_halo *hu, npg_halo *hp) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG, "npg_precond_mg_set_level_dist: not a multigrid preconditioner");
    NPG_REQUIRE(level >= 1 && level < (int)pc->L.size(), "npg_precond_mg_set_level_dist: level %d cannot be distributed", level);
    NPG_REQUIRE(A && G && D && Dinv && S && hx && hu && hp, "npg_precond_mg_set_level_dist: NULL argument");
    NPG_REQUIRE(level == (int)pc->L.size() - 1 || level == (int)pc->L.size() - 2,
                "npg_precond_mg_set_level_dist: the distributed levels are the finest one or two");
    const int64_t n = A->m, np = n - nu;
    NPG_REQUIRE(nu > 0 && np > 0 && hx->n_owned == n && A->n == n + hx->n_ghost && hu->n_owned == nu && D->m == np &&
                    D->n == nu + hu->n_ghost && hp->n_owned == np && G->m == nu && G->n == np + hp->n_ghost && S->m == np &&
                    S->n == G->n && Dinv->m == nu && Dinv->n == nu,
                "npg_precond_mg_set_level_dist: operator shapes do not match the halo plans (n = %lld, nu = %lld)", (long long)n,
                (long long)nu);
    MgLevel &l = pc->L[level];
    NPG_REQUIRE(l.A == nullptr && pc->L[level - 1].A, "npg_precond_mg_set_level_dist: set the coarser levels first, this one once");
    if (pc->L[level - 1].dist) {
        // the level below is distributed too: its transfers come with npg_precond_mg_set_transfer_dist
        NPG_REQUIRE(!P && !R, "npg_precond_mg_set_level_dist: above a distributed level P and R are set by npg_precond_mg_set_transfer_dist");
    } else {
        NPG_REQUIRE(P && R, "npg_precond_mg_set_level_dist: NULL transfer operators");
        const int64_t nc = pc->L[level - 1].n;
        NPG_REQUIRE(P->m == n && P->n == nc && R->m == nc && R->n == n, "npg_precond_mg_set_level_dist: P must be %lld x %lld and R its transpose",
                    (long long)n, (long long)nc);
    }
    l.A = A; l.G = G; l.D = D; l.Dinv = Dinv; l.S = S; l.P = P; l.R = R;
    l.n = n; l.nu = nu; l.np = np;
    l.hx = hx; l.hu = hu; l.hp = hp;
    l.dist = true;
    int rc;
    if ((rc = mg_alloc(pc, &l.sdinv, np))) return rc;
    npg_vec sv;
    sv.ctx = pc->ctx; sv.n = np; sv.d = l.sdinv; sv.owns = false;
    if ((rc = npg_csr_inv_diag(S, &sv))) return rc;
    if ((rc = mg_alloc(pc, &l.r, n)) || (rc = mg_alloc(pc, &l.t, D->n)) || (rc = mg_alloc(pc, &l.rhs, np)) ||
        (rc = mg_alloc(pc, &l.dp, G->n)) || (rc = mg_alloc(pc, &l.dp2, G->n)) || (rc = mg_alloc(pc, &l.res, np)) ||
        (rc = mg_alloc(pc, &l.x, A->n)))
        return rc;
    // (a distributed level that is not the finest receives a right-hand side from the level above; its iterate is l.x)
    if (level + 1 < (int)pc->L.size() && (rc = mg_alloc(pc, &l.b, n))) return rc;
    pc->n = n;
    pc->use_graphs = false;      // host barriers (rehearsal transports) and library collectives sit inside the cycle
    if (pc->mixed && (rc = mg_refresh_fp32(l))) return rc;
    return NPG_OK;
}

// Transfers between two DISTRIBUTED levels (`level` and the one below it, both set): P = this rank's rows of the prolongation over
// the coarser level's [owned | ghosts] columns (hP: the plan on the coarser iterate that fills those ghosts), R = the coarser
// level's owned rows of the restriction over this level's [owned | ghosts] columns (hR: the plan on this level's residual).
NPG_API int npg_precond_mg_set_transfer_dist(npg_precond *pc, int level, const npg_csr *P, const npg_csr *R, npg_halo *hP, npg_halo *hR) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG && level >= 1 && level < (int)pc->L.size(), "npg_precond_mg_set_transfer_dist: bad level");
    NPG_REQUIRE(P && R && hP && hR, "npg_precond_mg_set_transfer_dist: NULL argument");
    MgLevel &l = pc->L[level], &lc = pc->L[level - 1];
    NPG_REQUIRE(l.A && lc.A && l.dist && lc.dist && !l.P, "npg_precond_mg_set_transfer_dist: both levels must be distributed, the transfers set once");
    NPG_REQUIRE(hP->n_owned == lc.n && P->m == l.n && P->n == lc.n + hP->n_ghost && hR->n_owned == l.n && R->m == lc.n &&
                    R->n == l.n + hR->n_ghost,
                "npg_precond_mg_set_transfer_dist: operator shapes do not match the halo plans");
    l.P = P; l.R = R; l.hP = hP; l.hR = hR;
    int rc;
    if ((rc = mg_alloc(pc, &l.xP, P->n)) || (rc = mg_alloc(pc, &l.rR, R->n))) return rc;
    return NPG_OK;
}

// Swap in re-assembled operators of one level (same shapes): what the eddy closure's A refresh needs (src/model.jl:160-170)
NPG_API int npg_precond_mg_set_scaled_gradient(npg_precond *pc, int level, const npg_csr *Gh) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG && level >= 0 && level < (int)pc->L.size() && pc->L[level].A,
                "npg_precond_mg_set_scaled_gradient: the level is not set");
    MgLevel &l = pc->L[level];
    NPG_REQUIRE(!Gh || (Gh->m == l.G->m && Gh->n == l.G->n), "npg_precond_mg_set_scaled_gradient: Dinv G must have the shape of G (%lld x %lld)",
                (long long)l.G->m, (long long)l.G->n);
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    drop_graphs(pc);
    l.Gh = Gh;
    if (Gh && pc->mixed) return csr_refresh_fp32(Gh);
    return NPG_OK;
}

NPG_API int npg_precond_mg_update_level(npg_precond *pc, int level, const npg_csr *A, const npg_csr *G, const npg_csr *D,
                                        const npg_csr *Dinv, const npg_csr *S) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG, "npg_precond_mg_update_level: not a multigrid preconditioner");
    NPG_REQUIRE(level >= 0 && level < (int)pc->L.size() && pc->L[level].A, "npg_precond_mg_update_level: level %d is not set",
                level);
    NPG_REQUIRE(A && G && D && Dinv && S, "npg_precond_mg_update_level: NULL operator");
    MgLevel &l = pc->L[level];
    // (a distributed level's operators are row blocks with [owned | ghost] column spaces: compare with what the level holds)
    NPG_REQUIRE(A->m == l.n && A->n == l.A->n && G->m == l.nu && G->n == l.G->n && D->m == l.np && D->n == l.D->n &&
                    Dinv->m == l.nu && Dinv->n == l.nu && S->m == l.np && S->n == l.S->n,
                "npg_precond_mg_update_level: shapes differ from the level's");
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    drop_graphs(pc);
    l.A = A; l.G = G; l.D = D; l.Dinv = Dinv; l.S = S;
    npg_vec sv;
    sv.ctx = pc->ctx; sv.n = l.np; sv.d = l.sdinv; sv.owns = false;
    int rc = npg_csr_inv_diag(S, &sv);
    if (rc) return rc;
    return pc->mixed ? mg_refresh_fp32(l) : NPG_OK;
}

// Mixed precision: the cycle's SpMVs read fp32 copies of the level operators' values (8 instead of 12 bytes per entry, 12
// instead of 20 per node record); vectors, products and sums stay fp64, and so does the outer flexible GMRES with its
// own SpMV - a preconditioner may be inexact.  The copies are rebuilt by npg_precond_mg_update_level.
NPG_API int npg_precond_mg_set_mixed(npg_precond *pc, int on) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG, "npg_precond_mg_set_mixed: not a multigrid preconditioner");
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    drop_graphs(pc);
    pc->mixed = on ? 1 : 0;
    if (pc->mixed)
        for (const MgLevel &l : pc->L)
            if (l.A) {
                int rc = mg_refresh_fp32(l);
                if (rc) return rc;
            }
    return NPG_OK;
}

NPG_API int npg_precond_mg_set_params(npg_precond *pc, double omega, double jacobi_weight, int schur_sweeps, int nu1,
                                      int nu2, int coarse_sweeps) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG, "npg_precond_mg_set_params: not a multigrid preconditioner");
    NPG_REQUIRE(omega > 0 && jacobi_weight > 0 && schur_sweeps >= 1 && nu1 >= 0 && nu2 >= 0 && nu1 + nu2 >= 1 &&
                    coarse_sweeps >= 1,
                "npg_precond_mg_set_params: bad parameter");
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    drop_graphs(pc);
    pc->omega = omega; pc->jw = jacobi_weight; pc->sweeps = schur_sweeps;
    pc->nu1 = nu1; pc->nu2 = nu2; pc->coarse = coarse_sweeps;
    return NPG_OK;
}

NPG_API int npg_precond_mg_set_cycle(npg_precond *pc, int gamma) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_MG && (gamma == 1 || gamma == 2), "npg_precond_mg_set_cycle: gamma must be 1 (V) or 2 (W)");
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    drop_graphs(pc);
    pc->gamma = gamma;
    return NPG_OK;
}

// Levels of a million rows and more whose matrix carries a windowed tile set form their residuals from the fp32 gather-layout
// copy of the iterate (one fill kernel + the windowed product: 150 + 8 instead of 190 us at 2.15 M unknowns) - the rounding of x,
// 6e-8 |A| |x|, is far below what a smoothing step or the coarse-grid correction leaves; the outer flexible GMRES forms its own
// product from the fp64 vector.  Called before a cycle is enqueued or captured: buffers follow the matrices' current layouts.
constexpr int64_t kMgGatherMinRows = 1000000;
static int mg_prepare(npg_precond *pc) {
    static const bool on = !getenv("NPG_MG_GATHER32") || atoi(getenv("NPG_MG_GATHER32")) != 0;
    static const bool nb_on = !getenv("NPG_MG_NB_EPILOGUE") || atoi(getenv("NPG_MG_NB_EPILOGUE")) != 0;
    for (MgLevel &l : pc->L) {
        const bool nb = nb_on && l.A && l.Gh && !l.dist && !pc->mixed && nb_epilogue_ok(l.A, l.Dinv, l.nu);
        if (nb != l.nb_ok) {
            NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
            drop_graphs(pc);
            l.nb_ok = nb;
        }
        const int64_t need = (on && l.A && !l.dist && l.n >= kMgGatherMinRows) ? gather32_floats(l.A) : 0;
        if (need == l.xg_n) continue;
        NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
        drop_graphs(pc);
        if (l.xg) hipFree(l.xg);
        l.xg = nullptr;
        l.xg_n = 0;
        if (need) {
            NPG_HIP(hipMalloc((void **)&l.xg, (size_t)need * sizeof(float)));
            NPG_HIP(hipMemsetAsync(l.xg, 0, (size_t)need * sizeof(float), pc->ctx->stream));    // (the pad slots stay zero)
            l.xg_n = need;
        }
    }
    return NPG_OK;
}
// y = alpha A x + beta c (+ second output) on level l: the level's matrix in whichever form serves it fastest
static int mg_product(npg_precond *pc, MgLevel &l, const double *x, const SpmvEpi &e, const NbEpi *nb = nullptr) {
    return l.xg ? spmv_epi_gather32(l.A, x, l.xg, e, nb) : spmv_epi(l.A, x, e, nb);
}

// nsteps Braess-Sarazin steps on level l for A x = b.  Eight launches per step (seven with the scaled gradient): the vector updates ride in the epilogues of
// the SpMV kernels (SpmvEpi) - on the coarse levels, where every kernel is latency-bound, the launch count is the cost.
static int mg_smooth(npg_precond *pc, int lev, double *x, const double *b, int nsteps, bool x_is_zero) {
    MgLevel &l = pc->L[lev];
    npg_ctx *c = pc->ctx;
    const int64_t nu = l.nu, np = l.np;
    int rc;
    for (int s = 0; s < nsteps; ++s) {
        const bool zero = x_is_zero && s == 0;
        const double *r = l.r;
        bool have_t = false;                                                         // t (and x_u += t / w) formed by the residual kernel
        if (zero) {
            r = b;                                                                   // r = b - A 0
        } else {
            if (l.hx && (rc = halo_exchange_raw(l.hx, x))) return rc;
            SpmvEpi e{};                                                             // r = b - A x
            e.alpha = -1.0; e.beta = 1.0; e.c = b; e.y = l.r; e.f32 = pc->mixed;
            if (l.nb_ok) {
                // ... and t = Dh^-1 r_u in the same kernel: the node-block epilogue (NbEpi); x_u takes t / w further down, in the
                // kernel that subtracts (Dh^-1 G) dp / w (x is this product's input: it must not change under it)
                const npg_csr *Di = l.Dinv;
                NbEpi nb{Di->rowptr, Di->col, Di->val, (int32_t)nu, l.t, nullptr, 0.0, 0};
                if ((rc = mg_product(pc, l, x, e, &nb))) return rc;
                have_t = true;
            }
            if (!have_t && (rc = mg_product(pc, l, x, e))) return rc;
        }
        if (have_t) {
        } else if (l.Gh) {
            // x_u += Dh^-1 (r_u - G dp) / w  =  t / w - (Dh^-1 G) dp / w: the first part rides in the kernel that forms t
            SpmvEpi e{};                                                             // t = Dh^-1 r_u ; x_u (+)= t / w
            e.alpha = 1.0; e.beta = 0.0; e.y = l.t; e.f32 = pc->mixed;
            e.w = 1.0 / pc->omega; e.zin = zero ? nullptr : x; e.zc = 1.0; e.z = x;
            if ((rc = spmv_epi(l.Dinv, r, e))) return rc;
        } else if ((rc = spmv_raw(l.Dinv, r, l.t, 1.0, 0.0, pc->mixed))) return rc;  // t = Dh^-1 r_u
        double *dp = l.dp, *dq = l.dp2;
        if (l.hu && (rc = halo_exchange_raw(l.hu, l.t))) return rc;
        {
            SpmvEpi e{};                                                             // rhs = D t - w r_p ; dp = jw rhs / diag S
            e.alpha = 1.0; e.beta = -pc->omega; e.c = r + nu; e.y = l.rhs;
            e.w = pc->jw; e.dg = l.sdinv; e.z = dp; e.f32 = pc->mixed;
            if ((rc = spmv_epi(l.D, l.t, e))) return rc;
        }
        for (int k = 1; k < pc->sweeps; ++k) {                                       // damped Jacobi on S dp = rhs
            if (l.hp && (rc = halo_exchange_raw(l.hp, dp))) return rc;
            SpmvEpi e{};                                                             // res = rhs - S dp ; dq = dp + jw res / diag S
            e.alpha = -1.0; e.beta = 1.0; e.c = l.rhs; e.y = l.res;
            e.w = pc->jw; e.dg = l.sdinv; e.zin = dp; e.zc = 1.0; e.z = dq; e.f32 = pc->mixed;
            if ((rc = spmv_epi(l.S, dp, e))) return rc;
            std::swap(dp, dq);
        }
        if (l.hp && (rc = halo_exchange_raw(l.hp, dp))) return rc;
        if (l.Gh && have_t) {
            SpmvEpi e{};                                 // du = (t - (Dh^-1 G) dp) / w  (into t) ; x_u += du  (second output)
            e.alpha = -1.0 / pc->omega; e.beta = 1.0 / pc->omega; e.c = l.t; e.y = l.t; e.f32 = pc->mixed;
            e.w = 1.0; e.zin = x; e.zc = 1.0; e.z = x;
            if ((rc = spmv_epi(l.Gh, dp, e))) return rc;
        } else if (l.Gh) {
            if ((rc = spmv_raw(l.Gh, dp, x, -1.0 / pc->omega, 1.0, pc->mixed))) return rc;   // x_u -= (Dh^-1 G) dp / w
        } else {
            SpmvEpi e{};                                                             // t = r_u - G dp   (t is free again)
            e.alpha = -1.0; e.beta = 1.0; e.c = r; e.y = l.t; e.f32 = pc->mixed;
            if ((rc = spmv_epi(l.G, dp, e))) return rc;
            if ((rc = spmv_raw(l.Dinv, l.t, x, 1.0 / pc->omega, zero ? 0.0 : 1.0, pc->mixed))) return rc;   // x_u += Dh^-1 t / w
        }
        axpby(c, x + nu, 1.0, dp, zero ? 0.0 : 1.0, np);                             // x_p += dp
    }
    return NPG_OK;
}

// One multigrid cycle on level `lev` for A x = b: gamma = 1 is the V-cycle, gamma = 2 the W-cycle (the coarse problem of
// every level is visited gamma times, the second visit continuing from the first one's result).
static int mg_cycle(npg_precond *pc, int lev, double *x, const double *b, bool x_is_zero) {
    int rc;
    if (lev == 0 && (pc->dense.M || pc->dense.Mf || pc->dense.Mh)) {                 // direct coarsest-level solve
        if (x_is_zero) return dense_apply(pc, b, x, 1.0, 0.0);
        MgLevel &l0 = pc->L[0];
        SpmvEpi e{};
        e.alpha = -1.0; e.beta = 1.0; e.c = b; e.y = l0.r; e.f32 = pc->mixed;
        if ((rc = spmv_epi(l0.A, x, e))) return rc;
        return dense_apply(pc, l0.r, x, 1.0, 1.0);
    }
    if (lev == 0) return mg_smooth(pc, 0, x, b, pc->coarse, x_is_zero);
    MgLevel &l = pc->L[lev], &lc = pc->L[lev - 1];
    if (pc->nu1 > 0 && (rc = mg_smooth(pc, lev, x, b, pc->nu1, x_is_zero))) return rc;
    const bool still_zero = x_is_zero && pc->nu1 == 0;
    const double *rfine = b;                    // what is restricted: b itself while x = 0, else the residual
    if (!still_zero) {
        if (l.hx && (rc = halo_exchange_raw(l.hx, x))) return rc;
        SpmvEpi e{};
        e.alpha = -1.0; e.beta = 1.0; e.c = b; e.y = l.r; e.f32 = pc->mixed;
        if ((rc = mg_product(pc, l, x, e))) return rc;
        rfine = l.r;
    }
    if (l.hR) {
        // both levels distributed: the coarser level's OWNED rows of R reach this level's residual at a few rows of other ranks
        NPG_HIP(hipMemcpyAsync(l.rR, rfine, (size_t)l.n * sizeof(double), hipMemcpyDeviceToDevice, pc->ctx->stream));
        if ((rc = halo_exchange_raw(l.hR, l.rR))) return rc;
        rfine = l.rR;
    }
    if ((rc = spmv_raw(l.R, rfine, lc.b, 1.0, 0.0, pc->mixed))) return rc;
    // a distributed level above replicated ones: R holds this rank's columns of the restriction, the coarse right-hand side is
    // the sum of the ranks' parts - after it every rank runs the coarse levels redundantly on identical data - and the
    // prolongation back needs no communication (P holds this rank's rows)
    if (l.dist && !lc.dist && (rc = allreduce_big_device(pc->ctx, lc.b, lc.n))) return rc;
    for (int g = 0; g < pc->gamma; ++g)
        if ((rc = mg_cycle(pc, lev - 1, lc.x, lc.b, g == 0))) return rc;
    const double *xc = lc.x;
    if (l.hP) {             // both levels distributed: this rank's rows of P reach coarse entries of other ranks
        NPG_HIP(hipMemcpyAsync(l.xP, lc.x, (size_t)lc.n * sizeof(double), hipMemcpyDeviceToDevice, pc->ctx->stream));
        if ((rc = halo_exchange_raw(l.hP, l.xP))) return rc;
        xc = l.xP;
    }
    if ((rc = spmv_raw(l.P, xc, x, 1.0, still_zero ? 0.0 : 1.0, pc->mixed))) return rc;
    return mg_smooth(pc, lev, x, b, pc->nu2, false);
}

static int mg_vcycle(npg_precond *pc, int lev, double *x, const double *b) { return mg_cycle(pc, lev, x, b, true); }

NPG_API int npg_precond_blockdiag_set(npg_precond *pc, int k, int64_t offset, const npg_csr *A, const npg_vec *jacobi,
                                      int64_t itmax, double atol, double rtol) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_BLOCKDIAG, "npg_precond_blockdiag_set: not a block-diagonal preconditioner");
    NPG_REQUIRE(k >= 0 && k < (int)pc->blocks.size() && A && jacobi, "npg_precond_blockdiag_set: bad argument");
    NPG_REQUIRE(A->m == A->n && jacobi->n == A->m && offset >= 0, "npg_precond_blockdiag_set: block must be square");
    BlockPc &b = pc->blocks[k];
    NPG_REQUIRE(!b.A, "npg_precond_blockdiag_set: block %d is already set", k);
    b.off = offset; b.n = A->m; b.A = A; b.jac = jacobi; b.itmax = itmax; b.atol = atol; b.rtol = rtol;
    int rc;
    if ((rc = npg_cg_create(pc->ctx, b.n, &b.cg))) return rc;
    if ((rc = npg_vec_create(pc->ctx, b.n, &b.xk))) return rc;       // workspace.x .= 0 (src/preconditioners.jl:20)
    if ((rc = npg_vec_create(pc->ctx, b.n, &b.rk))) return rc;
    pc->n = std::max(pc->n, offset + b.n);
    return NPG_OK;
}

NPG_API int npg_precond_blockdiag_set_ilu0(npg_precond *pc, int k, npg_ilu0 *M) {
    NPG_REQUIRE(pc && pc->kind == NPG_PC_BLOCKDIAG, "npg_precond_blockdiag_set_ilu0: not a block-diagonal preconditioner");
    NPG_REQUIRE(k >= 0 && k < (int)pc->blocks.size() && pc->blocks[k].A, "npg_precond_blockdiag_set_ilu0: block %d is not set", k);
    int64_t nnz = 0;
    if (M) {
        int rc = npg_ilu0_info(M, nullptr, nullptr, &nnz);
        if (rc) return rc;
        NPG_REQUIRE(nnz == pc->blocks[k].A->nnz, "npg_precond_blockdiag_set_ilu0: the factors are not of this block's matrix");
    }
    NPG_HIP(hipStreamSynchronize(pc->ctx->stream));
    pc->blocks[k].ilu = M;
    return NPG_OK;
}

static int precond_apply_raw(npg_precond *pc, const double *r, double *z) {
    ++pc->applications;
    if (pc->kind == NPG_PC_DENSE) {
        NPG_REQUIRE(pc->dense.M || pc->dense.Mf, "npg_precond_apply: the dense inverse has not been set");
        return dense_apply(pc, r, z, 1.0, 0.0);
    }
    if (pc->kind == NPG_PC_MG) {
        NPG_REQUIRE(pc->L.back().A, "npg_precond_apply: multigrid levels are not all set");
        if (int rcp = mg_prepare(pc)) return rcp;
        const int top = (int)pc->L.size() - 1;
        if (pc->L[top].dist) {
            // the iterate is an SpMV input: it lives in the level's own buffer, which has room for the ghost entries
            MgLevel &lt = pc->L[top];
            const int rcv = mg_vcycle(pc, top, lt.x, r);
            if (rcv) return rcv;
            NPG_HIP(hipMemcpyAsync(z, lt.x, (size_t)lt.n * sizeof(double), hipMemcpyDeviceToDevice, pc->ctx->stream));
            return NPG_OK;
        }
        if (!pc->use_graphs) {
            int64_t count = 0;
            if (pc->cycle_bytes == 0) byte_sink = &count;
            const int rcv = mg_vcycle(pc, top, z, r);
            byte_sink = nullptr;
            if (count) pc->cycle_bytes = count;
            return rcv;
        }
        hipStream_t st = pc->ctx->stream;
        // the captured cycles bake in the borrowed matrices' tile tables and value arrays: a matrix whose layout was
        // rebuilt since (build_tiles, block_nodes, first fp32 copy) invalidates them.  (A borrowed matrix must outlive the
        // preconditioner or be replaced through the setters, which drop the graphs.)
        uint64_t gsum = 0;
        for (const MgLevel &lv : pc->L)
            for (const npg_csr *M : {lv.A, lv.G, lv.D, lv.Dinv, lv.S, lv.P, lv.R, lv.Gh})
                if (M) gsum += spmv_form(M)->gen;
        if (gsum != pc->graphs_gen) {
            drop_graphs(pc);
            pc->graphs_gen = gsum;
        }
        auto key = std::make_pair(r, z);
        auto it = pc->graphs.find(key);
        if (it == pc->graphs.end()) {
            if (pc->graphs.size() >= 64) drop_graphs(pc);                  // a caller that keeps changing buffers
            hipGraph_t g = nullptr;
            NPG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int64_t count = 0;
            byte_sink = &count;
            const int rcv = mg_vcycle(pc, top, z, r);
            byte_sink = nullptr;
            if (!rcv) pc->cycle_bytes = count;
            const hipError_t ec = hipStreamEndCapture(st, &g);
            if (rcv || ec != hipSuccess) {
                if (g) hipGraphDestroy(g);
                if (rcv) return rcv;
            }
            NPG_HIP(ec);
            hipGraphExec_t ex = nullptr;
            const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            NPG_HIP(ei);
            it = pc->graphs.emplace(key, ex).first;
        }
        NPG_HIP(hipGraphLaunch(it->second, st));
        return NPG_OK;
    }
    for (BlockPc &b : pc->blocks) {
        NPG_REQUIRE(b.A, "npg_precond_apply: a block has not been set");
        // mul!(yb, block.P^-1, xb): CG on the block, warm-started from its previous output (src/preconditioners.jl:24-37,118-125)
        NPG_HIP(hipMemcpyAsync(b.rk->d, r + b.off, (size_t)b.n * sizeof(double), hipMemcpyDeviceToDevice, pc->ctx->stream));
        npg_solve_stats st{};
        int rc = b.ilu ? ilu_pcg_raw(b.ilu, b.A, b.rk->d, b.xk->d, b.atol, b.rtol, b.itmax, &st)
                       : npg_cg_solve(b.cg, b.A, NPG_PRECOND_DIAG, 0.0, b.jac, b.rk, b.xk, b.atol, b.rtol, b.itmax, &st);
        if (rc) return rc;
        pc->inner_iterations += st.niter;
        // (a block solve that broke down - status 3: a non-finite or non-positive curvature, e.g. behind unusable ILU(0) factors - has
        //  NaNs in its iterate: handing them to the outer flexible GMRES would poison its basis)
        NPG_REQUIRE(st.status != 3, "npg_precond_apply: the inner CG of the block at offset %lld broke down after %d iterations; its output is not used",
                    (long long)b.off, st.niter);
        NPG_HIP(hipMemcpyAsync(z + b.off, b.xk->d, (size_t)b.n * sizeof(double), hipMemcpyDeviceToDevice, pc->ctx->stream));
    }
    return NPG_OK;
}

NPG_API int npg_precond_apply(npg_precond *pc, const npg_vec *r, npg_vec *z) {
    NPG_REQUIRE(pc && r && z, "npg_precond_apply: NULL argument");
    NPG_REQUIRE(pc->kind != NPG_PC_MG || pc->L.back().A, "npg_precond_apply: multigrid levels are not all set");
    NPG_REQUIRE(pc->kind != NPG_PC_DENSE || pc->dense.M || pc->dense.Mf, "npg_precond_apply: the dense inverse has not been set");
    NPG_REQUIRE(r->n == pc->n && z->n == pc->n && r->d != z->d, "npg_precond_apply: vectors must have %lld entries and not alias",
                (long long)pc->n);
    return precond_apply_raw(pc, r->d, z->d);
}

NPG_API int npg_precond_cycle_bytes(npg_precond *pc, int64_t *bytes) {
    NPG_REQUIRE(pc && bytes, "npg_precond_cycle_bytes: NULL argument");
    *bytes = pc->cycle_bytes;
    return NPG_OK;
}

NPG_API int npg_precond_counters(npg_precond *pc, int64_t *applications, int64_t *inner_iterations) {
    NPG_REQUIRE(pc, "npg_precond_counters: NULL handle");
    if (applications) *applications = pc->applications;
    if (inner_iterations) *inner_iterations = pc->inner_iterations;
    return NPG_OK;
}

// ---- flexible GMRES ----------------------------------------------------------------------------------------------------
struct npg_fgmres {
    npg_ctx *ctx = nullptr;
    int64_t n = 0, ld = 0;
    int mem = 20;
    double *V = nullptr, *Z = nullptr;       // (mem + 1) and mem columns of ld doubles
    double *part = nullptr;                  // partial rows of the reductions
    double *dsc = nullptr;                   // device scalars: h1 [0,32), h2 [32,64), norm^2 [64]
    double *hsc = nullptr;                   // pinned host copy
    std::vector<double> hist;
    // distributed (npg_fgmres_set_halo): n counts OWNED rows; the columns of Z (SpMV inputs) have room for the ghost entries
    npg_halo *halo = nullptr;
    int64_t n_ghost = 0;
};

constexpr int kFgBlocks = 1024;

NPG_API int npg_fgmres_create(npg_ctx *ctx, int64_t n, int memory, npg_fgmres **out) {
    NPG_REQUIRE(ctx && out && n > 0, "npg_fgmres_create: bad argument");
    NPG_REQUIRE(memory >= 1 && memory <= 23, "npg_fgmres_create: memory must be in 1..23");
    npg_fgmres *ws = new npg_fgmres();
    ws->ctx = ctx; ws->n = n; ws->mem = memory;
    ws->ld = (n + 31) / 32 * 32;
    NPG_HIP(hipSetDevice(ctx->device));
    NPG_HIP(hipMalloc((void **)&ws->V, (size_t)(memory + 1) * ws->ld * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&ws->Z, (size_t)memory * ws->ld * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&ws->part, (size_t)kFgBlocks * kPartStride * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&ws->dsc, 128 * sizeof(double)));          // (whole 32-double rows: the all-reduce moves rows)
    NPG_HIP(hipMemset(ws->dsc, 0, 128 * sizeof(double)));
    NPG_HIP(hipHostMalloc((void **)&ws->hsc, 128 * sizeof(double), hipHostMallocDefault));
    *out = ws;
    return NPG_OK;
}

// Row-block distributed solves: A is this rank's rows (columns [owned | ghosts]), x holds [owned | ghosts]; the ghosts of every
// SpMV input are filled through `h`, the Gram-Schmidt sums and norms are summed over the ranks (three 32-double all-reduces per
// iteration - an iteration is a whole V-cycle).  Collective: every rank sets its plan before the first solve.
NPG_API int npg_fgmres_set_halo(npg_fgmres *ws, npg_halo *h) {
    NPG_REQUIRE(ws && h && h->n_owned == ws->n, "npg_fgmres_set_halo: the plan must own the workspace's %lld rows", ws ? (long long)ws->n : 0LL);
    NPG_HIP(hipStreamSynchronize(ws->ctx->stream));
    ws->halo = h;
    ws->n_ghost = h->n_ghost;
    ws->ld = (ws->n + ws->n_ghost + 31) / 32 * 32;
    NPG_HIP(hipFree(ws->V));
    NPG_HIP(hipFree(ws->Z));
    NPG_HIP(hipMalloc((void **)&ws->V, (size_t)(ws->mem + 1) * ws->ld * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&ws->Z, (size_t)ws->mem * ws->ld * sizeof(double)));
    NPG_HIP(hipMemset(ws->Z, 0, (size_t)ws->mem * ws->ld * sizeof(double)));
    return NPG_OK;
}

NPG_API int npg_fgmres_destroy(npg_fgmres *ws) {
    if (!ws) return NPG_OK;
    hipStreamSynchronize(ws->ctx->stream);
    hipFree(ws->V); hipFree(ws->Z); hipFree(ws->part); hipFree(ws->dsc); hipHostFree(ws->hsc);
    delete ws;
    return NPG_OK;
}

NPG_API int64_t npg_fgmres_history(npg_fgmres *ws, double *buf, int64_t cap) {
    if (!ws || !buf) return -1;
    const int64_t k = std::min<int64_t>(cap, (int64_t)ws->hist.size());
    std::copy(ws->hist.begin(), ws->hist.begin() + k, buf);
    return k;
}

template <typename F>
static void by_groups(int k, F f) {
    if (k <= 8) f(std::integral_constant<int, 1>());
    else if (k <= 16) f(std::integral_constant<int, 2>());
    else f(std::integral_constant<int, 3>());
}

NPG_API int npg_fgmres_solve(npg_fgmres *ws, const npg_csr *A, npg_precond *pc, const npg_vec *y, npg_vec *x,
                             double scale, double atol, double rtol, int64_t itmax, npg_solve_stats *stats) {
    NPG_REQUIRE(ws && A && y && x && stats, "npg_fgmres_solve: NULL argument");
    NPG_REQUIRE(!A->uperm, "npg_fgmres_solve: the matrix carries an internal renumbering (npg_csr_block_nodes_dofs): npg_spmv and npg_gmres_solve only");
    const int64_t n = ws->n, ld = ws->ld;
    const int64_t nloc = n + ws->n_ghost;
    NPG_REQUIRE(A->m == n && A->n == nloc && y->n == n && x->n == nloc, "npg_fgmres_solve: system must be %lld x %lld (+%lld ghosts)",
                (long long)n, (long long)n, (long long)ws->n_ghost);
    npg_halo *dh = ws->halo;
    // sum over the ranks of a host scalar (distributed) - the host is synchronised at these points anyway
    auto all_sum = [&](double *v) -> int { return dh ? npg_comm_allreduce_sum(ws->ctx, v, 1) : NPG_OK; };
    NPG_REQUIRE(!pc || pc->n == n, "npg_fgmres_solve: the preconditioner is for %lld unknowns", (long long)(pc ? pc->n : 0));
    NPG_REQUIRE(scale > 0 && atol >= 0 && rtol >= 0, "npg_fgmres_solve: bad tolerance / scale");
    npg_ctx *c = ws->ctx;
    hipStream_t st = c->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct EvGuard {
        hipEvent_t &a, &b;
        ~EvGuard() {
            if (a) hipEventDestroy(a);
            if (b) hipEventDestroy(b);
        }
    } evguard{e0, e1};
    NPG_HIP(hipEventCreate(&e0));
    NPG_HIP(hipEventCreate(&e1));
    NPG_HIP(hipEventRecord(e0, st));
    if (itmax <= 0) itmax = 2 * n;
    const int mem = ws->mem;
    const int grid = std::min(kFgBlocks, grid_for(n, kFgBlocks));
    *stats = npg_solve_stats{};
    ws->hist.clear();
    int rc;
    double *V = ws->V, *Z = ws->Z;
    // r0 = y - A x
    {
        if (dh && (rc = halo_exchange_raw(dh, x->d))) return rc;
        SpmvEpi e{};
        e.alpha = -1.0; e.beta = 1.0; e.c = y->d; e.y = V;
        if ((rc = spmv_epi(A, x->d, e))) return rc;
    }
    double dd;
    if ((rc = reduce_dot(c, V, V, n, &dd)) || (rc = all_sum(&dd))) return rc;
    double beta = std::sqrt(dd);
    const double rnorm0 = scale * beta, eps = atol + rtol * rnorm0;
    stats->rnorm0 = rnorm0;
    stats->rnorm = rnorm0;
    ws->hist.push_back(rnorm0);
    int64_t it = 0;
    bool solved = rnorm0 <= eps, breakdown = false;
    if (beta == 0.0) stats->status = 4;
    std::vector<double> H((size_t)(mem + 1) * mem), g(mem + 1), cs(mem), sn(mem), yk(mem);
    while (!solved && !breakdown && it < itmax) {
        ++stats->npass;
        axpby(c, V, 1.0 / beta, V, 0.0, n);
        std::fill(g.begin(), g.end(), 0.0);
        g[0] = beta;
        int k = 0;
        for (int j = 0; j < mem && it < itmax; ++j) {
            double *vj = V + (size_t)j * ld, *zj = Z + (size_t)j * ld, *w = V + (size_t)(j + 1) * ld;
            if (pc) {
                if ((rc = precond_apply_raw(pc, vj, zj))) return rc;
            } else {
                axpby(c, zj, 1.0, vj, 0.0, n);
            }
            if (dh && (rc = halo_exchange_raw(dh, zj))) return rc;
            if ((rc = spmv_raw(A, zj, w, 1.0, 0.0))) return rc;
            const int kk = j + 1;
            // classical Gram-Schmidt, two full passes; everything the host needs arrives in one copy
            for (int pass = 0; pass < 2; ++pass) {
                double *hd = ws->dsc + 32 * pass;
                by_groups(kk, [&](auto ng) {
                    hipLaunchKernelGGL(k_mdot<decltype(ng)::value>, dim3(grid), dim3(kBlock), 0, st, V, ld, kk, w, n, ws->part);
                });
                hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, st, ws->part, grid, kk, hd);
                if (dh && (rc = allreduce_sum_device(c, hd, kPartStride))) return rc;
                by_groups(kk, [&](auto ng) {
                    hipLaunchKernelGGL(k_mupdate<decltype(ng)::value>, dim3(grid), dim3(kBlock), 0, st, V, ld, kk, hd, w, n,
                                       ws->part);
                });
            }
            hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(kBlock), 0, st, ws->part, grid, 1, ws->dsc + 64);
            if (dh && (rc = allreduce_sum_device(c, ws->dsc + 64, kPartStride))) return rc;
            NPG_HIP(hipMemcpyAsync(ws->hsc, ws->dsc, 65 * sizeof(double), hipMemcpyDeviceToHost, st));
            NPG_HIP(hipStreamSynchronize(st));
            double *Hj = &H[(size_t)j * (mem + 1)];          // column j
            for (int i = 0; i < kk; ++i) Hj[i] = ws->hsc[i] + ws->hsc[32 + i];
            const double hn = std::sqrt(std::max(0.0, ws->hsc[64]));
            NPG_REQUIRE(std::isfinite(hn), "npg_fgmres_solve: non-finite Arnoldi vector at iteration %lld", (long long)it);
            Hj[kk] = hn;
            for (int i = 0; i < j; ++i) {
                const double t = cs[i] * Hj[i] + sn[i] * Hj[i + 1];
                Hj[i + 1] = -sn[i] * Hj[i] + cs[i] * Hj[i + 1];
                Hj[i] = t;
            }
            const double d = std::hypot(Hj[j], Hj[j + 1]);
            cs[j] = d > 0 ? Hj[j] / d : 1.0;
            sn[j] = d > 0 ? Hj[j + 1] / d : 0.0;
            Hj[j] = d;
            Hj[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            ++it;
            k = j + 1;
            const double rn = scale * std::fabs(g[j + 1]);
            ws->hist.push_back(rn);
            stats->rnorm = rn;
            if (rn <= eps) break;
            if (hn <= 1e-300 || d == 0.0) { breakdown = true; break; }
            axpby(c, w, 1.0 / hn, w, 0.0, n);
        }
        // x += Z y,  H y = g (upper triangular, columns stored with stride mem + 1)
        for (int i = k - 1; i >= 0; --i) {
            double s = g[i];
            for (int jj = i + 1; jj < k; ++jj) s -= H[(size_t)jj * (mem + 1) + i] * yk[jj];
            const double piv = H[(size_t)i * (mem + 1) + i];
            yk[i] = piv != 0.0 ? s / piv : 0.0;
        }
        Coefs cf{};
        for (int i = 0; i < k; ++i) cf.y[i] = yk[i];
        if (k > 0) hipLaunchKernelGGL(k_combine_z, dim3(grid), dim3(kBlock), 0, st, x->d, Z, ld, k, cf, n);
        // true residual: the next pass starts from it, and a pass that met the estimate is confirmed by it
        {
            if (dh && (rc = halo_exchange_raw(dh, x->d))) return rc;
            SpmvEpi e{};
            e.alpha = -1.0; e.beta = 1.0; e.c = y->d; e.y = V;
            if ((rc = spmv_epi(A, x->d, e))) return rc;
        }
        if ((rc = reduce_dot(c, V, V, n, &dd)) || (rc = all_sum(&dd))) return rc;
        beta = std::sqrt(dd);
        stats->rnorm = scale * beta;
        solved = stats->rnorm <= eps;
        if (beta == 0.0) break;
    }
    NPG_HIP(hipGetLastError());
    NPG_HIP(hipEventRecord(e1, st));
    NPG_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    NPG_HIP(hipEventElapsedTime(&ms, e0, e1));
    stats->seconds = ms * 1e-3;
    stats->solved = solved ? 1 : 0;
    stats->niter = (int32_t)it;
    if (stats->status == 0) stats->status = solved ? 1 : breakdown ? 3 : 2;
    return NPG_OK;
}

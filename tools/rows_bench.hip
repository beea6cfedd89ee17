// Microbenchmark of the Gram-Schmidt "dots" pass over an fp32-stored, column-major Krylov basis (csrc/gmres.hip,
// k_gmres_dots_rows<NG, float>): what limits it to ~3.9 TB/s when the fp64-basis kernel streams 6 TB/s with the same loop?
//   hipcc --offload-arch=gfx950 -O3 tools/rows_bench.hip -o tools/rows_bench && tools/rows_bench [n] [j]
// Variants: A = the product kernel's loop (2 rows per thread, float2 loads, 256 threads, 3 workgroups per CU)
//           B = A with the multiply-adds of unused columns skipped (k <= j is wave-uniform)
//           C = 4 rows per thread (float4 loads)            D = A with 512-thread workgroups
//           E = A with 6 workgroups per CU                  F = fp64 basis, 1 row per thread (the round-2 kernel)
//           G = A with the next trip's loads issued before this trip's arithmetic
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

constexpr int NC = 16;      // columns handled by the instance (8 * NG with NG = 2)

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void finish(const double (&acc)[NC], double nrm, double *part) {
    double s = nrm;
#pragma unroll
    for (int k = 0; k < NC; ++k) s += wave_sum(acc[k]) * (k + 1);
    if ((threadIdx.x & 63) == 0) atomicAdd(part + (blockIdx.x & 255), s);
}

template <int NT, bool GUARD>
__global__ void __launch_bounds__(NT) k_A(const float *__restrict__ Vf, const double *__restrict__ w, int64_t n, int64_t ldv, int j,
                                        double *part) {
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    double nrm = 0.0;
    const int64_t nblk = (n + 2 * NT - 1) / (2 * NT);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = 2 * (blk * NT + threadIdx.x);
        if (r0 >= n) continue;
        float2 v[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float2 f = make_float2(0.f, 0.f);
            if (k <= j) f = *reinterpret_cast<const float2 *>(Vf + (size_t)k * ldv + r0);
            v[k] = f;
        }
        const double2 ww = *reinterpret_cast<const double2 *>(w + r0);
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (!GUARD || k <= j) acc[k] += (double)v[k].x * ww.x + (double)v[k].y * ww.y;
        nrm += ww.x * ww.x + ww.y * ww.y;
    }
    finish(acc, nrm, part);
}

__global__ void __launch_bounds__(256) k_C(const float *__restrict__ Vf, const double *__restrict__ w, int64_t n, int64_t ldv, int j,
                                         double *part) {
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    double nrm = 0.0;
    const int64_t nblk = (n + 4 * 256 - 1) / (4 * 256);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t r0 = 4 * (blk * 256 + threadIdx.x);
        if (r0 >= n) continue;
        float4 v[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k <= j) f = *reinterpret_cast<const float4 *>(Vf + (size_t)k * ldv + r0);
            v[k] = f;
        }
        const double2 wa = *reinterpret_cast<const double2 *>(w + r0), wb = *reinterpret_cast<const double2 *>(w + r0 + 2);
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k <= j) acc[k] += (double)v[k].x * wa.x + (double)v[k].y * wa.y + (double)v[k].z * wb.x + (double)v[k].w * wb.y;
        nrm += wa.x * wa.x + wa.y * wa.y + wb.x * wb.x + wb.y * wb.y;
    }
    finish(acc, nrm, part);
}

__global__ void __launch_bounds__(256) k_F(const double *__restrict__ V, const double *__restrict__ w, int64_t n, int64_t ldv, int j,
                                         double *part) {
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    double nrm = 0.0;
    for (int64_t row = blockIdx.x * 256LL + threadIdx.x; row < n; row += (int64_t)gridDim.x * 256) {
        double v[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) v[k] = (k <= j) ? V[(size_t)k * ldv + row] : 0.0;
        const double wv = w[row];
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] += v[k] * wv;
        nrm += wv * wv;
    }
    finish(acc, nrm, part);
}

__global__ void __launch_bounds__(256) k_G(const float *__restrict__ Vf, const double *__restrict__ w, int64_t n, int64_t ldv, int j,
                                         double *part) {
    double acc[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) acc[k] = 0.0;
    double nrm = 0.0;
    const int64_t nblk = (n + 2 * 256 - 1) / (2 * 256);
    float2 v[NC], vn[NC];
    double2 ww, wn;
    auto load = [&](int64_t blk, float2 (&o)[NC], double2 &ow) {
        const int64_t r0 = 2 * (blk * 256 + threadIdx.x);
        const bool ok = blk < nblk && r0 < n;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float2 f = make_float2(0.f, 0.f);
            if (k <= j && ok) f = *reinterpret_cast<const float2 *>(Vf + (size_t)k * ldv + r0);
            o[k] = f;
        }
        ow = ok ? *reinterpret_cast<const double2 *>(w + r0) : make_double2(0.0, 0.0);
    };
    load(blockIdx.x, v, ww);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        load(blk + gridDim.x, vn, wn);
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k <= j) acc[k] += (double)v[k].x * ww.x + (double)v[k].y * ww.y;
        nrm += ww.x * ww.x + ww.y * ww.y;
#pragma unroll
        for (int k = 0; k < NC; ++k) v[k] = vn[k];
        ww = wn;
    }
    finish(acc, nrm, part);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 2150791;
    const int j = argc > 2 ? atoi(argv[2]) : 12;
    const int64_t ldv = (n + 31) / 32 * 32 + 32;
    float *Vf;
    double *V, *w, *part;
    CK(hipMalloc((void **)&Vf, (size_t)NC * ldv * sizeof(float)));
    CK(hipMalloc((void **)&V, (size_t)NC * ldv * sizeof(double)));
    CK(hipMalloc((void **)&w, (size_t)(n + 8) * sizeof(double)));
    CK(hipMalloc((void **)&part, 256 * sizeof(double)));
    CK(hipMemset(Vf, 0, (size_t)NC * ldv * sizeof(float)));
    CK(hipMemset(V, 0, (size_t)NC * ldv * sizeof(double)));
    CK(hipMemset(w, 0, (size_t)(n + 8) * sizeof(double)));
    CK(hipMemset(part, 0, 256 * sizeof(double)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double b32 = (double)n * (4.0 * (j + 1) + 8.0), b64 = (double)n * (8.0 * (j + 1) + 8.0);
    auto timeit = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0));
        const int reps = 50;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e12);
    };
    printf("n = %lld rows, j = %d (%d columns in use)\n", (long long)n, j, j + 1);
    timeit("A 2 rows/thread, 256 thr, 768 wg", b32, [&] { hipLaunchKernelGGL((k_A<256, false>), dim3(768), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("B = A, unused columns skipped", b32, [&] { hipLaunchKernelGGL((k_A<256, true>), dim3(768), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("C 4 rows/thread (float4), 768 wg", b32, [&] { hipLaunchKernelGGL(k_C, dim3(768), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("D = B, 512 thr, 768 wg", b32, [&] { hipLaunchKernelGGL((k_A<512, true>), dim3(768), dim3(512), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("D2 = B, 1024 thr, 256 wg", b32, [&] { hipLaunchKernelGGL((k_A<1024, true>), dim3(256), dim3(1024), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("E = B, 256 thr, 1536 wg", b32, [&] { hipLaunchKernelGGL((k_A<256, true>), dim3(1536), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("E2 = B, 256 thr, 4096 wg", b32, [&] { hipLaunchKernelGGL((k_A<256, true>), dim3(4096), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("F fp64 basis, 1 row/thread, 768 wg", b64, [&] { hipLaunchKernelGGL(k_F, dim3(768), dim3(256), 0, 0, V, w, n, ldv, j, part); });
    timeit("G = B + next trip's loads first", b32, [&] { hipLaunchKernelGGL(k_G, dim3(768), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    timeit("C2 4 rows/thread (float4), 1536 wg", b32, [&] { hipLaunchKernelGGL(k_C, dim3(1536), dim3(256), 0, 0, Vf, w, n, ldv, j, part); });
    return 0;
}

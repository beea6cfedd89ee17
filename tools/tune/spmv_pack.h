// Tile-packed SpMV: the matrix stream of a tile is ONE contiguous blob in HBM, copied by LDS-DMA (global_load_lds_dwordx4,
// no register destination) into the LDS buffer of the tile AFTER the one being worked on.
//
// Why (profiles/r02_spmv_dma_phases.txt): spmv_device.h's tile function holds a dependent chain per tile - stream loads ->
// gathers -> products -> sums; 75 % of a tile's 24 k cycles are that chain and only the CU's other workgroups hide it.
// Prefetching the next tile into registers lost to register pressure; LDS-DMA from the seven separate CSR / record arrays
// worked (bit-identical) but spent 2 k cycles per tile on issuing it (seven short regions, each with its own alignment
// arithmetic).  With the blob the issue is one loop of 16-byte pieces, and the LDS image IS the blob:
//
//   int32  rp[nrows + 1]   CSR product-slot offsets (into the double section)      |
//   int32  prp[nnode + 1]  record offsets of the tile's nodes (node-block tiles)   |  each section padded to 16 bytes
//   int32  pcol[npe]       column node of a record                                  |
//   int32  col[n]          CSR columns                                              |
//   double K[npe], C[npe]  records (structure of arrays), each padded to even
//   double val[n]          CSR values, padded to even
//   (LDS only) double Z[npe]  the z-row products of the records
//
// Products overwrite the K / C / val slots in place (x-row products over K, y-row products over C).  The blob replaces
// rowptr / prow (8 B per row and node) by 4-byte tile-local offsets.
//
// Ordering (cdna_hip_programming.md section 5, "Read a staged buffer one phase AFTER the wait that retires it"):
//   A  s_waitcnt vmcnt(0) ; s_barrier            tile t's blob landed (every wave waited for its own pieces)
//   B  ds_read columns / values, gather x, ds_write products in place
//   C  s_waitcnt lgkmcnt(0) ; s_barrier
//   D  issue the DMA of tile t+1 into the other buffer (last read before barrier C of this tile)
//   E  segmented sums of tile t -> sw
//   F  s_waitcnt lgkmcnt(0) ; s_barrier ; epilogue (reads sw)
// C and F are raw s_barrier: a __syncthreads() would drain the DMA (vmcnt(0)).
#pragma once
#include "../../nupgcm_amd/csrc/spmv_device.h"

namespace npg {

// one packed tile: 32 bytes = one scalar load
struct PackDesc {
    int64_t off16;                     // blob start in 16-byte units
    int32_t r0, nrows, n, npe, nnode;  // rows [r0, r0 + nrows), n CSR entries, npe records of nnode nodes
    int32_t flags;                     // bit 0: node-block rows; bit 1: (x, y, z) nodes; bit 2: one long row, not packed
};

struct PackGeo {
    int i_prp, i_pcol, i_col, i_pdst, i_cdst, nints;   // int32 offsets of the sections (i_pdst, i_cdst: uint16 arrays)
    int npeA, nA, ndbl;                // doubles: K at 0, C at npeA, val at 2 npeA, (LDS) Z at 2 npeA + nA
};

template <bool PERM = false>
__host__ __device__ __forceinline__ PackGeo pack_geo(int nrows, int nnode, int npe, int n, bool blk) {
    PackGeo g;
    g.i_prp = (nrows + 1 + 3) & ~3;
    g.i_pcol = g.i_prp + (blk ? (nnode + 1 + 3) & ~3 : 0);
    g.i_col = g.i_pcol + ((npe + 3) & ~3);
    g.i_pdst = g.i_col + ((n + 3) & ~3);
    g.i_cdst = g.i_pdst + (PERM ? (((npe + 1) >> 1) + 3) & ~3 : 0);
    g.nints = g.i_cdst + (PERM ? (((n + 1) >> 1) + 3) & ~3 : 0);
    g.npeA = (npe + 1) & ~1;
    g.nA = (n + 1) & ~1;
    g.ndbl = 2 * g.npeA + g.nA;
    return g;
}

// LDS bytes a buffer needs for tiles of at most TNNZ product slots and TROWS rows
constexpr int pack_buf_bytes(int tnnz, int trows, bool perm = false) { return (perm ? 14 : 12) * tnnz + 6 * trows + (perm ? 160 : 128); }

__device__ __forceinline__ void wait_vm0_barrier() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one 16-byte piece per lane: the lanes of a wave write LDS at lds_wave_base + lane * 16 (lds_wave_base wave-uniform)
#ifndef NPG_GLDS_MODE
#define NPG_GLDS_MODE 0
#endif
template <bool NTL>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_wave_base) {
    unsigned keep;
    const uint32_t b = __builtin_amdgcn_readfirstlane(lds_wave_base);
#if NPG_GLDS_MODE == 1
    // experiment: M0 left at the piece's base, padded behind the instruction
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt\n\ts_nop 7" : : "v"(gsrc), "s"(b) : "memory");
    (void)keep;
#else
    if (NTL)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(b)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(b)
                     : "memory");
#endif
}

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)p;       // low half of a generic pointer into LDS = the LDS byte offset
}

// phase D: tile pd's blob -> the LDS bytes at buf
template <int NT, bool NTL, bool PERM = false>
__device__ __forceinline__ void pack_issue(const char *__restrict__ blob, const PackDesc &pd, const char *buf) {
    if (pd.flags & 4) return;
    const PackGeo g = pack_geo<PERM>(pd.nrows, pd.nnode, pd.npe, pd.n, pd.flags & 1);
    const int bytes = 4 * g.nints + 8 * g.ndbl;
    const char *src = blob + pd.off16 * 16;
    const uint32_t dst = lds_addr(buf);
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i * 16 < bytes; i += NT) glds16<NTL>(src + (size_t)i * 16, dst + (uint32_t)(i - lane) * 16);
}

// phases B and C for the tile staged at buf
template <int NT, class XF, int TNNZ, bool PERM = false>
__device__ __forceinline__ void pack_products(const CsrDev &A, const XF x, const PackDesc &pd, char *buf) {
    const bool blk = pd.flags & 1, full = pd.flags & 2;
    const PackGeo g = pack_geo<PERM>(pd.nrows, pd.nnode, pd.npe, pd.n, blk);
    const uint16_t *PD = reinterpret_cast<const uint16_t *>(buf) + 2 * g.i_pdst;
    const uint16_t *CD = reinterpret_cast<const uint16_t *>(buf) + 2 * g.i_cdst;
    int pd_[(TNNZ / 2 + NT - 1) / NT], cd_[(TNNZ + NT - 1) / NT];
    const int32_t *I = reinterpret_cast<const int32_t *>(buf);
    double *D = reinterpret_cast<double *>(buf + 4 * g.nints);
    const int n = pd.n, npe = pd.npe;
    constexpr int UP = (TNNZ / 2 + NT - 1) / NT;       // surface-node tiles: two product slots per record
    constexpr int UC = (TNNZ + NT - 1) / NT;
    int32_t rc[UP], cc[UC];
    double rK[UP], rC[UP], zz[UP], cv[UC], xa[UC];
    double2 xx[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int e = threadIdx.x + u * NT;
        const bool ok = e < npe;
        rc[u] = ok ? I[g.i_pcol + e] : 0;
        pd_[u] = (PERM && ok) ? PD[e] : e;
        rK[u] = ok ? D[e] : 0.0;
        rC[u] = ok ? D[g.npeA + e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
        const int k = threadIdx.x + u * NT;
        const bool ok = k < n;
        cc[u] = ok ? I[g.i_col + k] : 0;
        cd_[u] = (PERM && ok) ? CD[k] : k;
        cv[u] = ok ? D[2 * g.npeA + k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int cf = rc[u] < A.nfull ? rc[u] : A.nfull;
        const int xo = 2 * rc[u] + cf;                          // first DoF of node c
        xx[u] = x.two(xo);
        zz[u] = (full && rc[u] < A.nfull) ? x.third(xo + 2) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) xa[u] = x(cc[u]);
    const int zoff = 2 * g.npeA + g.nA;
    if (PERM) wait_lds_barrier();          // a product lands in ANOTHER stream position's slot: every value is in registers first
#pragma unroll
    for (int u = 0; u < UP; ++u) {
        const int e = threadIdx.x + u * NT;
        if (e < npe) {
            D[pd_[u]] = rK[u] * xx[u].x + rC[u] * xx[u].y;
            D[g.npeA + pd_[u]] = rK[u] * xx[u].y - rC[u] * xx[u].x;
            if (full) D[zoff + pd_[u]] = rK[u] * zz[u];
        }
    }
#pragma unroll
    for (int u = 0; u < UC; ++u) {
        const int k = threadIdx.x + u * NT;
        if (k < n) D[2 * g.npeA + cd_[u]] = cv[u] * xa[u];
    }
    wait_lds_barrier();
}

// phase E: sw[r] = sum of row r's products (the caller fences: wait_lds_barrier() before reading sw)
template <int NT, int L, bool PERM = false>
__device__ __forceinline__ void pack_sums(const PackDesc &pd, const char *buf, double *__restrict__ sw) {
    const bool blk = pd.flags & 1, full = pd.flags & 2;
    const PackGeo g = pack_geo<PERM>(pd.nrows, pd.nnode, pd.npe, pd.n, blk);
    const int32_t *I = reinterpret_cast<const int32_t *>(buf);
    const double *D = reinterpret_cast<const double *>(buf + 4 * g.nints);
    const int ncomp = full ? 3 : 2, qmul = full ? 21846 : 32768;
    const int zoff = 2 * g.npeA + g.nA;
    const int gq = threadIdx.x / L, l = threadIdx.x % L;
    for (int r = gq; r < pd.nrows; r += NT / L) {
        double s = 0.0;
        const int e = I[r + 1];
        for (int k = I[r] + 2 * l; k < e; k += 2 * L) {
            const double a = D[k], b = D[k + 1];
            s += a + (k + 1 < e ? b : 0.0);
        }
        if (blk) {
            const int q = (r * qmul) >> 16;                      // r / 3 or r / 2 (r < 2^15)
            const int comp = r - q * ncomp;
            const int pb = comp == 0 ? 0 : comp == 1 ? g.npeA : zoff;
            const int pe = pb + I[g.i_prp + q + 1];
            for (int k = pb + I[g.i_prp + q] + 2 * l; k < pe; k += 2 * L) {
                const double a = D[k], b = D[k + 1];
                s += a + (k + 1 < pe ? b : 0.0);
            }
        }
        s = group_sum_dpp<L>(s);
        if (l == 0) sw[r] = s;
    }
}

// one very long row (flags & 4): straight from the CSR arrays, by the whole workgroup; result in sw[0] (unfenced)
template <int NT, class XF>
__device__ __forceinline__ void pack_long_row(const CsrDev &A, const XF x, const PackDesc &pd, char *buf, double *sw) {
    const int64_t base = A.rowptr[pd.r0];
    double s = 0.0;
    for (int k = threadIdx.x; k < pd.n; k += NT) s += A.val[base + k] * x(A.col[base + k]);
    s = wave_sum(s);
    double *D = reinterpret_cast<double *>(buf);
    if ((threadIdx.x & 63) == 0) D[threadIdx.x >> 6] = s;
    wait_lds_barrier();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < NT / 64; ++w) tot += D[w];
        sw[0] = tot;
    }
}

// ---- gathers one tile ahead: the x values of tile t+1 are requested (into registers) before tile t's segmented sums and
// used after them
template <int NT, int TNNZ>
struct PackX {
    static constexpr int UP = (TNNZ / 2 + NT - 1) / NT;
    static constexpr int UC = (TNNZ + NT - 1) / NT;
    double2 xx[UP];
    double zz[UP], xa[UC];
};

template <int NT, class XF, int TNNZ>
__device__ __forceinline__ void pack_gather(const CsrDev &A, const XF x, const PackDesc &pd, const char *buf,
                                            PackX<NT, TNNZ> &G) {
    const bool blk = pd.flags & 1, full = pd.flags & 2;
    const PackGeo g = pack_geo(pd.nrows, pd.nnode, pd.npe, pd.n, blk);
    const int32_t *I = reinterpret_cast<const int32_t *>(buf);
    int32_t rc[PackX<NT, TNNZ>::UP], cc[PackX<NT, TNNZ>::UC];
#pragma unroll
    for (int u = 0; u < PackX<NT, TNNZ>::UP; ++u) {
        const int e = threadIdx.x + u * NT;
        rc[u] = e < pd.npe ? I[g.i_pcol + e] : 0;
    }
#pragma unroll
    for (int u = 0; u < PackX<NT, TNNZ>::UC; ++u) {
        const int k = threadIdx.x + u * NT;
        cc[u] = k < pd.n ? I[g.i_col + k] : 0;
    }
#pragma unroll
    for (int u = 0; u < PackX<NT, TNNZ>::UP; ++u) {
        const int cf = rc[u] < A.nfull ? rc[u] : A.nfull;
        const int xo = 2 * rc[u] + cf;
        G.xx[u] = x.two(xo);
        G.zz[u] = (full && rc[u] < A.nfull) ? x.third(xo + 2) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < PackX<NT, TNNZ>::UC; ++u) G.xa[u] = x(cc[u]);
}

// products of the tile staged at buf from the x values gathered earlier (in place; unfenced)
template <int NT, int TNNZ>
__device__ __forceinline__ void pack_multiply(const PackDesc &pd, char *buf, const PackX<NT, TNNZ> &G) {
    const bool blk = pd.flags & 1, full = pd.flags & 2;
    const PackGeo g = pack_geo(pd.nrows, pd.nnode, pd.npe, pd.n, blk);
    double *D = reinterpret_cast<double *>(buf + 4 * g.nints);
    const int zoff = 2 * g.npeA + g.nA;
#pragma unroll
    for (int u = 0; u < PackX<NT, TNNZ>::UP; ++u) {
        const int e = threadIdx.x + u * NT;
        if (e < pd.npe) {
            const double K = D[e], C = D[g.npeA + e];
            D[e] = K * G.xx[u].x + C * G.xx[u].y;
            D[g.npeA + e] = K * G.xx[u].y - C * G.xx[u].x;
            if (full) D[zoff + e] = K * G.zz[u];
        }
    }
#pragma unroll
    for (int u = 0; u < PackX<NT, TNNZ>::UC; ++u) {
        const int k = threadIdx.x + u * NT;
        if (k < pd.n) D[2 * g.npeA + k] *= G.xa[u];
    }
}

}  // namespace npg

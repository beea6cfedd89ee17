// CSR SpMV building blocks shared by the stand-alone SpMV and by the fused Krylov kernels.
//
// Layout in HBM: rowptr int64[m+1], col int32[nnz], val fp64[nnz] - 12 bytes streamed per stored entry - plus, for the
// inversion matrix with constant viscosity, a "node-block" part (below).
//
// "CSR-stream" tiles.  The host groups consecutive whole rows into tiles of at most TNNZ LDS product slots and kTileRows
// rows.  A workgroup handles a tile in two phases:
//   1. every thread streams PAIRS of adjacent entries of the tile's contiguous [rowptr[r0], rowptr[r1]) range: one
//      16-byte `val` load and one 8-byte `col` load per lane (the tile start is rounded down to an even entry so the
//      16-byte loads are aligned), U2 independent pairs in flight per lane, issued non-temporally - the matrix is read
//      exactly once per SpMV and must not evict the x vector from L2; then gathers x[col] (served by L2 / Infinity
//      Cache: the RCM ordering keeps a tile's columns close) and puts the products into LDS;
//   2. a sub-group of L lanes per row sums that row's segment of the LDS products (in-register DPP reduction) -
//      the wavefront-segmented sum.
// Streaming is therefore independent of the row-length distribution: short velocity rows and long pressure rows cost the
// same per stored entry.  A row longer than TNNZ forms a tile of its own and is handled by the whole workgroup.
//
// Node-block part.  With constant viscosity the velocity block of A_inversion couples node q to node c through ONE friction
// number K_qc (the same for the x-x, y-y and z-z entries) and ONE Coriolis number C_qc (x-y entry, minus it for y-x)
// (/root/reference/src/inversion.jl:183-192: same-component friction + f z-cross-u).  The DoF ordering of nupgcm_amd.fe
// puts the components of a node next to each other: first the nodes with all three components free ("full", rows
// 3q .. 3q+2), then the nodes with free x and y only ("surface": w = 0 at z = 0, rows 3 nfull + 2 (q - nfull) + {0, 1}).
// Those five (four) CSR entries - 60 (48) bytes and five (four) 8-byte gathers - are stored ONCE as a record {c, K, C} of
// 20 bytes and cost one 16-byte gather of (x, y)_c plus one 8-byte gather of z_c from the same cache line.
//
// Coupling records (the rows BEHIND the block rows: the divergence rows of A_inversion).  Such a row p holds, for every node c
// near it, the three numbers (d_x, d_y, d_z) that multiply (x, y, z)_c.  Those three (two) CSR entries - 36 (24) bytes and
// three (two) gathers - are stored as one 28-byte record {c, d_x, d_y, d_z}: the same two gathers as a node record, the three
// products summed in registers into ONE LDS slot.  These tiles are a tile class of their own (no node records in them), so
// the record loop below is the only stream beside the CSR remainder.
//
// Column records (the block rows' entries OUTSIDE the block columns: the gradient entries of the velocity rows, a rank's ghost
// columns).  For node q every distinct column m its rows touch there is stored as one 28-byte record {m, a_x, a_y, a_z} - the
// coefficients of x_m in the rows x_q, y_q, z_q: one gather of x_m and three products instead of up to three CSR entries with a
// gather each.  With them the block rows hold no CSR entries at all, so a block tile runs the node-record loop and this loop,
// a pressure-row tile the coupling-record loop and the CSR loop: never more than two streams per tile.
//
// Configuration (512 threads, 5824 product slots and at most 256 rows per tile, 4 pairs per lane, 3 workgroups per CU: fp64
// + 64-bit addressing needs ~80 VGPRs, which rules out two 1024-thread workgroups per CU) chosen from the sweep in
// profiles/r01_spmv_variants.txt (tools/spmv_tune.py) and from whole-timestep runs on bowl3D h = 0.02: 4096 / 4608 / 5120 /
// 5376 slots with 512-row tiles 3818 / 3677 / 3606 / 3601 ms per step (5632: 4703, only two workgroups of the Arnoldi kernel
// fit a CU); with 256-row tiles (3.5 KiB less row bookkeeping) 5824 slots 3514 ms, 6000 slots 3585 ms (more than 3 x 512
// records in many tiles: a second dependent trip).
#pragma once
#include "device_utils.h"

namespace npg {

constexpr int kTileNnz = 5824;   // LDS product slots per tile: 45.5 KiB of fp64 (three workgroups of the Arnoldi kernel
                                 // = 3 x 51 KiB of the CU's 160 KiB); tiles are latency-bound, so the largest that keeps three fits best

// device view of a matrix: the (remainder) CSR arrays + the optional node-block part
struct CsrDev {
    const int64_t *rowptr;
    const int32_t *col;
    const double *val;
    int64_t nnz;              // entries in col/val
    const int64_t *prow;      // [nfull + nsurf + 1] offsets of node q's records into pcol / pkc; null without node blocks
    const int32_t *pcol;      // column node c
    const double2 *pkc;       // {K, C}: A[x_q,x_c] = A[y_q,y_c] = A[z_q,z_c] = K ; A[x_q,y_c] = C ; A[y_q,x_c] = -C
    int nfull, nsurf;         // nodes with (x, y, z) rows / with (x, y) rows; block rows = 3 nfull + 2 nsurf
    const float *val32;       // fp32 copies of val / pkc (tile functions instantiated with F32 read these; null otherwise)
    const float2 *pkc32;
    const int64_t *drow;      // [m - block rows + 1] offsets of a non-block row's coupling records; null without them
    const int32_t *dcol;      // column node c
    const double2 *dxy;       // (d_x, d_y): A[p, x_c], A[p, y_c]
    const double *dz;         // d_z: A[p, z_c] (0 for a surface node)
    const float2 *dxy32;      // fp32 copies
    const float *dz32;
    const int64_t *grow;      // [nfull + nsurf + 1] offsets of node q's column records; null without them (block rows then
                              // keep their remaining entries in rowptr / col / val)
    const int32_t *gcol;      // column m (any column outside the block columns)
    const double2 *gxy;       // (a_x, a_y): A[x_q, m], A[y_q, m]
    const double *gz;         // a_z: A[z_q, m] (0 for an (x, y)-only node)
    const float2 *gxy32;      // fp32 copies
    const float *gz32;
    // full node records (function-valued viscosity: all nine component pairs of a node pair; npg_csr_pack_nodes): record e of
    // the prow / pcol index holds a_i = A[row component i / 3 of q, column component i % 3 of c]; a_0 .. a_7 are stored in PAIRS -
    // (a_2k, a_2k+1) at pk9[2 k npk9 + 2 e + {0, 1}], four 16-byte-per-record streams - and a_8 at pk9[8 npk9 + e]
    const double *pk9;
    const float *pk9_32;
    int64_t npk9;
};

__device__ __forceinline__ int block_rows(const CsrDev &A) { return 3 * A.nfull + 2 * A.nsurf; }
// node of a block row that starts a node
__device__ __forceinline__ int node_of_row(const CsrDev &A, int r) {
    return r < 3 * A.nfull ? r / 3 : A.nfull + ((r - 3 * A.nfull) >> 1);
}

// SpMV input accessor: a plain contiguous vector (the Krylov kernels also plug in an on-the-fly corrected input)
struct PlainX {
    const double *x;
    __device__ __forceinline__ double operator()(int c) const { return x[c]; }
    // (x[i], x[i+1]); i is any element offset (8-byte aligned address)
    __device__ __forceinline__ double2 two(int i) const {
        double2 r;
        __builtin_memcpy(&r, x + i, sizeof r);
        return r;
    }
    // the z component of a node whose (x, y) were fetched with two(): separate hook so that diagnostics can price it
    __device__ __forceinline__ double third(int i) const { return x[i]; }
};

// An accessor may serve a node's components with ONE 16-byte gather (kNode4 = true, node4(c) -> (x, y, z, pad)); the tile
// function then skips the (x, y) + z pair of gathers.  Detected by name so that plain accessors need not mention it.
template <class XF, class = void>
struct node4_of {
    static constexpr bool value = false;
};
template <class XF>
struct node4_of<XF, decltype((void)XF::kNode4)> {
    static constexpr bool value = XF::kNode4;
};

// fp32 copy of a vector in GATHER layout: the components of block node c in floats [4c, 4c + 3) (the fourth, and the third of
// an (x, y)-only node, stay zero), every other entry i (pressure rows, a rank's ghosts) at 4 nnode + (i - block rows).
// The Krylov kernels keep their SpMV input in this form beside the fp64 vector (gmres.hip): a node record then costs one
// 16-byte gather instead of a 16-byte and an 8-byte one, a coupling record likewise, a CSR entry a 4-byte gather.
struct GatherMap {
    float *p;
    int nf3, nf, nbr, off;        // 3 nfull, nfull, block rows, 4 nnode - block rows
    __host__ __device__ __forceinline__ int pos(int i) const {
        if (i >= nbr) return i + off;
        if (i < nf3) return i + (int)(((unsigned long long)(unsigned)i * 0xAAAAAAABull) >> 33);      // i + i / 3
        const int r = i - nf3;
        return 4 * nf + 2 * r - (r & 1);
    }
};
struct PaddedX {
    static constexpr bool kNode4 = true;
    GatherMap g;
    __device__ __forceinline__ double operator()(int c) const { return (double)g.p[g.pos(c)]; }
    __device__ __forceinline__ float4 node4(int c) const { return *reinterpret_cast<const float4 *>(g.p + 4 * (size_t)c); }
    // entry i behind the block rows (pressure rows, a rank's ghosts) as stored
    __device__ __forceinline__ float behind(int i) const { return g.p[i + g.off]; }
    __device__ __forceinline__ double2 two(int i) const { return make_double2((*this)(i), (*this)(i + 1)); }
    __device__ __forceinline__ double third(int i) const { return (*this)(i); }
};

// optional phase profiling of spmv_tile (tuning harness): the default does nothing
struct NoProf {
    __device__ __forceinline__ void stamp(int) const {}
};

template <int TNNZ>
struct TileLdsT {
    alignas(16) double prod[TNNZ + 2];           // +2: the CSR part of a tile may start on an odd entry
    int32_t rp[kTileRows + 1];       // CSR row offsets into prod
    int32_t prp[kTileRows / 2 + 1];  // node offsets into the block products / row offsets into the coupling-record sums
};                                   // (a tile of rows with coupling records holds at most kTileRows / 2 rows)
using TileLds = TileLdsT<kTileNnz>;

// Phase 1 + 2 for one tile of rows [r0, r1).  On return (after the trailing barrier) out[r - r0] holds (A x)[r].
// NT = threads in the workgroup, L = lanes per row, U2 = independent entry pairs per lane and trip.
// Inside the block rows a tile holds whole nodes of one kind (full or surface).
// F32: the matrix values come from the fp32 copies (8 instead of 12 bytes per CSR entry, 12 instead of 20 per record);
// products and sums stay fp64.
// REMAT: see the comment at `tid` below.
// N9: the instance knows the FULL node records (nine values per node pair) - compiled only into the kernels that serve such
// matrices, so that the loop does not weigh on the register budget of the others.
template <int NT, int L, class XF, int TNNZ = kTileNnz, int U2 = 4, class PROF = NoProf, bool F32 = false, bool REMAT = false,
          bool N9 = false>
__device__ __forceinline__ void spmv_tile(const CsrDev &A, const XF x, const TileDesc &td, TileLdsT<TNNZ> &t,
                                          double *__restrict__ out, PROF prof = PROF()) {
    const int r0 = td.r0, nrows = td.nrows, r1 = r0 + nrows;
    const int64_t base = td.base;
    const int n = td.n;
    // REMAT: the lane's stream offsets are re-made per tile.  Left to itself hipcc hoists the per-lane base addresses of every
    // stream out of the caller's tile loop; in the Arnoldi kernel, at its 80-register cap, they were then spilled and reloaded
    // tile by tile (three dependent scratch round trips at the head of a record loop).  The stand-alone SpMV has the
    // registers and is 2 % faster with the hoisted addresses.
    int tid = threadIdx.x;
    if constexpr (REMAT) asm volatile("" : "+v"(tid));
    const bool blk = r0 < block_rows(A);
    const bool full = r0 < 3 * A.nfull;
    const int ncomp = full ? 3 : 2;
    const bool drec = !blk && A.drow != nullptr;
    const bool grec = blk && A.grow != nullptr;      // the tile's (base, n) then describe column records, not CSR entries
    const int npe = td.npe;           // records of this tile
    const int64_t pbase = td.pbase;
    int nnode = 0, q0 = 0;
    if (blk) {
        q0 = node_of_row(A, r0);
        nnode = node_of_row(A, r1) - q0;
    }
    const int64_t abase = base & ~1LL;
    const int off = grec ? 0 : (int)(base - abase);
    const int total = grec ? ncomp * n : n + off;             // product slots behind the record products
    const int slot0 = blk ? ncomp * npe : (drec ? npe : 0);   // CSR / column-record products live behind the record products
    if (grec)       // (rp is free in these tiles: it holds the nodes' column-record offsets)
        for (int q = threadIdx.x; q <= nnode; q += NT) t.rp[q] = (int32_t)(A.grow[q0 + q] - base);
    else
        for (int r = threadIdx.x; r <= nrows; r += NT) t.rp[r] = (int32_t)(A.rowptr[r0 + r] - base) + off + slot0;
    if (blk)
        for (int q = threadIdx.x; q <= nnode; q += NT) t.prp[q] = (int32_t)(A.prow[q0 + q] - pbase);
    else if (drec)
        for (int q = threadIdx.x; q <= nrows; q += NT) t.prp[q] = (int32_t)(A.drow[r0 - block_rows(A) + q] - pbase);
    if (slot0 + total <= TNNZ + 2) {
        // ---- record stream: {c, K, C} -> products for the x row (first npe slots), the y row (next npe) and the z row.
        // A full tile holds at most TNNZ / 3 records: three per lane cover it in one trip.
        constexpr int UP = U2 > 3 ? 3 : U2;
        if (drec) {
            // records in flight per lane: with the (x, y) + z pair of fp64 gathers a third one spills at the Arnoldi kernel's
            // 80-VGPR cap; the single 16-byte gather of the padded accessor leaves room for it
            constexpr int UD = node4_of<XF>::value ? 3 : 2;
            // ---- coupling records {c, d_x, d_y, d_z}: one product slot per record
            for (int e0 = tid; e0 < npe; e0 += UD * NT) {
                int32_t c[UD];
                double2 dd[UD], xx[UD];
                double dz[UD], zz[UD];
#pragma unroll
                for (int u = 0; u < UD; ++u) {
                    const int e = e0 + u * NT;
                    if (e < npe) {
                        c[u] = __builtin_nontemporal_load(A.dcol + pbase + e);
                        if (F32) {
                            const float *p = reinterpret_cast<const float *>(A.dxy32 + pbase + e);
                            dd[u].x = (double)__builtin_nontemporal_load(p);
                            dd[u].y = (double)__builtin_nontemporal_load(p + 1);
                            dz[u] = (double)__builtin_nontemporal_load(A.dz32 + pbase + e);
                        } else {
                            const double *p = reinterpret_cast<const double *>(A.dxy + pbase + e);
                            dd[u].x = __builtin_nontemporal_load(p);
                            dd[u].y = __builtin_nontemporal_load(p + 1);
                            dz[u] = __builtin_nontemporal_load(A.dz + pbase + e);
                        }
                    } else {
                        c[u] = 0;
                        dd[u] = make_double2(0.0, 0.0);
                        dz[u] = 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < UD; ++u) {
                    if constexpr (node4_of<XF>::value) {
                        const float4 f = x.node4(c[u]);               // (an (x, y)-only node's third float is its zero pad)
                        xx[u] = make_double2((double)f.x, (double)f.y);
                        zz[u] = (double)f.z;
                    } else {
                        const int cf = c[u] < A.nfull ? c[u] : A.nfull;
                        const int xo = 2 * c[u] + cf;
                        xx[u] = x.two(xo);
                        zz[u] = c[u] < A.nfull ? x.third(xo + 2) : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < UD; ++u) {
                    const int e = e0 + u * NT;
                    if (e < npe) t.prod[e] = dd[u].x * xx[u].x + dd[u].y * xx[u].y + dz[u] * zz[u];
                }
            }
        }
        if constexpr (N9) {
            // ---- full node records {c, a_00 .. a_22}: the same three product slots per record as the {c, K, C} form
            if (blk && A.pk9) {
                constexpr int U9 = 1;
                for (int e0 = tid; e0 < npe; e0 += U9 * NT) {
                    int32_t c[U9];
                    double a[U9][9];
                    double2 xx[U9];
                    double zz[U9];
#pragma unroll
                    for (int u = 0; u < U9; ++u) {
                        const int e = e0 + u * NT;
                        const bool in = e < npe;
                        c[u] = in ? __builtin_nontemporal_load(A.pcol + pbase + e) : 0;
                        // values in PAIRS: (a_0, a_1) .. (a_6, a_7) as four 16-byte streams, a_8 as an 8-byte one (pk9 layout below) -
                        // five loads per record instead of nine: the tile is bound by the vector-memory instructions it issues
                        const double *__restrict__ b9 = A.pk9 + 2 * (pbase + e);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if ((!full && k == 3) || !in) {       // (x, y)-only row nodes have no z row (but their rows do reach z_c)
                                a[u][2 * k] = a[u][2 * k + 1] = 0.0;
                            } else {
                                a[u][2 * k] = __builtin_nontemporal_load(b9 + (int64_t)(2 * k) * A.npk9);
                                a[u][2 * k + 1] = __builtin_nontemporal_load(b9 + (int64_t)(2 * k) * A.npk9 + 1);
                            }
                        }
                        a[u][8] = (full && in) ? __builtin_nontemporal_load(A.pk9 + 8 * A.npk9 + pbase + e) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < U9; ++u) {
                        if constexpr (node4_of<XF>::value) {
                            const float4 f = x.node4(c[u]);
                            xx[u] = make_double2((double)f.x, (double)f.y);
                            zz[u] = (double)f.z;                          // (zero pad for an (x, y)-only column node)
                        } else {
                            const int cf = c[u] < A.nfull ? c[u] : A.nfull;
                            const int xo = 2 * c[u] + cf;
                            xx[u] = x.two(xo);
                            zz[u] = c[u] < A.nfull ? x.third(xo + 2) : 0.0;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U9; ++u) {
                        const int e = e0 + u * NT;
                        if (e < npe) {
                            t.prod[e] = a[u][0] * xx[u].x + a[u][1] * xx[u].y + a[u][2] * zz[u];
                            t.prod[npe + e] = a[u][3] * xx[u].x + a[u][4] * xx[u].y + a[u][5] * zz[u];
                            if (full) t.prod[2 * npe + e] = a[u][6] * xx[u].x + a[u][7] * xx[u].y + a[u][8] * zz[u];
                        }
                    }
                }
            }
        }
        for (int e0 = tid; e0 < ((blk && !(N9 && A.pk9)) ? npe : 0); e0 += UP * NT) {
            int32_t c[UP];
            double2 kc[UP], xx[UP];
            double zz[UP];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int e = e0 + u * NT;
                if (e < npe) {
                    c[u] = __builtin_nontemporal_load(A.pcol + pbase + e);
                    if (F32) {
                        const float *p = reinterpret_cast<const float *>(A.pkc32 + pbase + e);
                        kc[u].x = (double)__builtin_nontemporal_load(p);
                        kc[u].y = (double)__builtin_nontemporal_load(p + 1);
                    } else {
                        const double *p = reinterpret_cast<const double *>(A.pkc + pbase + e);
                        kc[u].x = __builtin_nontemporal_load(p);
                        kc[u].y = __builtin_nontemporal_load(p + 1);
                    }
                } else {
                    c[u] = 0;
                    kc[u] = make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                if constexpr (node4_of<XF>::value) {
                    const float4 f = x.node4(c[u]);
                    xx[u] = make_double2((double)f.x, (double)f.y);
                    zz[u] = full ? (double)f.z : 0.0;
                } else {
                    const int cf = c[u] < A.nfull ? c[u] : A.nfull;
                    const int xo = 2 * c[u] + cf;                       // first DoF of node c
                    xx[u] = x.two(xo);
                    zz[u] = (full && c[u] < A.nfull) ? x.third(xo + 2) : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const int e = e0 + u * NT;
                if (e < npe) {
                    t.prod[e] = kc[u].x * xx[u].x + kc[u].y * xx[u].y;
                    t.prod[npe + e] = kc[u].x * xx[u].y - kc[u].y * xx[u].x;
                    if (full) t.prod[2 * npe + e] = kc[u].x * zz[u];
                }
            }
        }
        // ---- column records {m, a_x, a_y, a_z}: products for the x row (n slots behind the node-record products), y, z
        if (grec) {
            constexpr int UG = UP;
            for (int e0 = tid; e0 < n; e0 += UG * NT) {
                int32_t c[UG];
                double2 aa[UG];
                double az[UG], xv[UG];
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    const int e = e0 + u * NT;
                    if (e < n) {
                        c[u] = __builtin_nontemporal_load(A.gcol + base + e);
                        if (F32) {
                            const float *p = reinterpret_cast<const float *>(A.gxy32 + base + e);
                            aa[u].x = (double)__builtin_nontemporal_load(p);
                            aa[u].y = (double)__builtin_nontemporal_load(p + 1);
                            az[u] = full ? (double)__builtin_nontemporal_load(A.gz32 + base + e) : 0.0;
                        } else {
                            const double *p = reinterpret_cast<const double *>(A.gxy + base + e);
                            aa[u].x = __builtin_nontemporal_load(p);
                            aa[u].y = __builtin_nontemporal_load(p + 1);
                            az[u] = full ? __builtin_nontemporal_load(A.gz + base + e) : 0.0;
                        }
                    } else {
                        c[u] = 0;
                        aa[u] = make_double2(0.0, 0.0);
                        az[u] = 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < UG; ++u) xv[u] = x(c[u]);
#pragma unroll
                for (int u = 0; u < UG; ++u) {
                    const int e = e0 + u * NT;
                    if (e < n) {
                        t.prod[slot0 + e] = aa[u].x * xv[u];
                        t.prod[slot0 + n + e] = aa[u].y * xv[u];
                        if (full) t.prod[slot0 + 2 * n + e] = az[u] * xv[u];
                    }
                }
            }
        }
        // ---- CSR stream
        for (int k0 = 2 * tid; k0 < (grec ? 0 : total); k0 += 2 * NT * U2) {
            int2 c[U2];
            double2 v[U2];
            double xa[U2], xb[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                if (k < total && abase + k + 1 < A.nnz) {
                    const long long cc = __builtin_nontemporal_load(reinterpret_cast<const long long *>(A.col + abase + k));
                    c[u] = make_int2((int)(cc & 0xffffffffLL), (int)(cc >> 32));
                    if (F32) {
                        v[u].x = (double)__builtin_nontemporal_load(A.val32 + abase + k);
                        v[u].y = (double)__builtin_nontemporal_load(A.val32 + abase + k + 1);
                    } else {
                        v[u].x = __builtin_nontemporal_load(A.val + abase + k);
                        v[u].y = __builtin_nontemporal_load(A.val + abase + k + 1);
                    }
                } else if (k < total && abase + k < A.nnz) {
                    c[u] = make_int2(A.col[abase + k], 0);
                    v[u] = make_double2(F32 ? (double)A.val32[abase + k] : A.val[abase + k], 0.0);
                } else {
                    c[u] = make_int2(0, 0);
                    v[u] = make_double2(0.0, 0.0);
                }
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                xa[u] = x(c[u].x);
                xb[u] = x(c[u].y);
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int k = k0 + u * 2 * NT;
                if (k < total) t.prod[slot0 + k] = (k >= off) ? v[u].x * xa[u] : 0.0;
                if (k + 1 < total) t.prod[slot0 + k + 1] = v[u].y * xb[u];
            }
        }
        prof.stamp(0);
        __syncthreads();
        prof.stamp(1);
        // segmented sums: L lanes per row, two adjacent products per lane and trip (one ds_read2_b64): this phase is bound
        // by instruction issue (every wave of the CU is in it at once), not by LDS latency - an unrolled
        // several-rows-in-flight version measured 1.5x slower
        const int g = threadIdx.x / L, l = threadIdx.x % L;
        if (grec) {
            // block tile in all-record form: a lane group sums the two or three rows of a NODE together - their segments have
            // the same shape, so the offsets, the loop control and the latency of the LDS reads are paid once for three
            // independent accumulations (per row the same terms in the same order as the row-wise loop below)
            for (int q = g; q < nnode; q += NT / L) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
                const int cb = t.rp[q], ce = t.rp[q + 1];
                for (int k = cb + 2 * l; k < ce; k += 2 * L) {
                    const bool two = k + 1 < ce;
                    const double a0 = t.prod[slot0 + k], b0 = t.prod[slot0 + k + 1];
                    const double a1 = t.prod[slot0 + n + k], b1 = t.prod[slot0 + n + k + 1];
                    s0 += a0 + (two ? b0 : 0.0);
                    s1 += a1 + (two ? b1 : 0.0);
                    if (full) {
                        const double a2 = t.prod[slot0 + 2 * n + k], b2 = t.prod[slot0 + 2 * n + k + 1];
                        s2 += a2 + (two ? b2 : 0.0);
                    }
                }
                const int pb = t.prp[q], pe = t.prp[q + 1];
                for (int k = pb + 2 * l; k < pe; k += 2 * L) {
                    const bool two = k + 1 < pe;
                    const double a0 = t.prod[k], b0 = t.prod[k + 1];
                    const double a1 = t.prod[npe + k], b1 = t.prod[npe + k + 1];
                    s0 += a0 + (two ? b0 : 0.0);
                    s1 += a1 + (two ? b1 : 0.0);
                    if (full) {
                        const double a2 = t.prod[2 * npe + k], b2 = t.prod[2 * npe + k + 1];
                        s2 += a2 + (two ? b2 : 0.0);
                    }
                }
                s0 = group_sum_dpp<L>(s0);
                s1 = group_sum_dpp<L>(s1);
                if (full) s2 = group_sum_dpp<L>(s2);
                if (l == 0) {
                    out[q * ncomp] = s0;
                    out[q * ncomp + 1] = s1;
                    if (full) out[q * ncomp + 2] = s2;
                }
            }
        }
        for (int r = g; r < (grec ? 0 : nrows); r += NT / L) {
            double s = 0.0;
            const int q = !blk ? r : full ? (r * 21846) >> 16 : r >> 1;     // node of a block row: r / 3 for r < 2^15
            // second segment: the row's CSR products, or (column records) its component's share of the node's records
            const int sb = grec ? slot0 + (r - q * ncomp) * n : 0;
            const int e = grec ? sb + t.rp[q + 1] : t.rp[r + 1];
            for (int k = (grec ? sb + t.rp[q] : t.rp[r]) + 2 * l; k < e; k += 2 * L) {
                const double a = t.prod[k], b = t.prod[k + 1];       // k + 1 <= TNNZ + 2 stays inside the tile struct
                s += a + (k + 1 < e ? b : 0.0);
            }
            if (blk || drec) {
                const int pb = blk ? (r - q * ncomp) * npe : 0, pe = pb + t.prp[q + 1];
                for (int k = pb + t.prp[q] + 2 * l; k < pe; k += 2 * L) {
                    const double a = t.prod[k], b = t.prod[k + 1];
                    s += a + (k + 1 < pe ? b : 0.0);
                }
            }
            s = group_sum_dpp<L>(s);
            if (l == 0) out[r] = s;
        }
        prof.stamp(2);
    } else {
        // one very long (plain CSR) row: the whole workgroup strides over it, tree-reduce through LDS
        double s = 0.0;
        for (int k = threadIdx.x; k < n; k += NT) s += (F32 ? (double)A.val32[base + k] : A.val[base + k]) * x(A.col[base + k]);
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) t.prod[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int w = 0; w < NT / 64; ++w) tot += t.prod[w];
            out[0] = tot;
        }
    }
    __syncthreads();
}

// boundaries-only entry (tuning harness): builds the descriptor with the dependent loads the product kernels avoid
template <int NT, int L, class XF, int TNNZ = kTileNnz, int U2 = 4>
__device__ __forceinline__ void spmv_tile(const CsrDev &A, const XF x, int r0, int r1, TileLdsT<TNNZ> &t,
                                          double *__restrict__ out) {
    TileDesc td;
    td.r0 = r0;
    td.nrows = r1 - r0;
    td.base = A.rowptr[r0];
    td.n = (int)(A.rowptr[r1] - td.base);
    td.pbase = 0;
    td.npe = 0;
    if (r0 < block_rows(A)) {
        td.pbase = A.prow[node_of_row(A, r0)];
        td.npe = (int)(A.prow[node_of_row(A, r1)] - td.pbase);
        if (A.grow) {
            td.base = A.grow[node_of_row(A, r0)];
            td.n = (int)(A.grow[node_of_row(A, r1)] - td.base);
        }
    } else if (A.drow) {
        td.pbase = A.drow[r0 - block_rows(A)];
        td.npe = (int)(A.drow[r1 - block_rows(A)] - td.pbase);
    }
    spmv_tile<NT, L, XF, TNNZ, U2>(A, x, td, t, out);
}

}  // namespace npg

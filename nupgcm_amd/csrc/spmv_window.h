// Windowed tiles: the rows of a node-blocked matrix ({c, K, C} node records + {m, a_x, a_y, a_z} column records in the block
// rows, {c, d_x, d_y, d_z} coupling records behind them; spmv_device.h) with every gather of the input vector served from LDS.
//
// Why.  The record-form SpMV is bound by the rate at which a CU's texture-address / L1 path accepts vector-memory lanes, not by
// HBM (profiles/r02_spmv_experiments.txt section 6: with EVERY gather an L1 hit the kernel is 3 % faster,
// section 7: gathers served by LDS reads cost 9 us per SpMV instead of 55).  A tile of ~64 row nodes holds ~1 800 node records
// but only ~430 DISTINCT column nodes (RCM keeps a tile's columns close; bowl3D h = 0.02: 4.2 records per distinct node) and
// ~500 column records on ~90 distinct pressure columns.  So the host stores, per tile, the ascending list of distinct column
// nodes (`wlist`) and of distinct other columns (`vlist`); the tile gathers each of them ONCE - adjacent lanes gather ascending
// nodes - into an LDS window, and the records address the window through 16-bit indices (2 bytes per record instead of the
// 4-byte column: the lists cost less than the indices save).  Divergent gather lanes per SpMV of that matrix: 30 M -> 8.3 M.
//
// Pair sums.  Every node's (every row's) record list is padded to an even count with a zero record, and a lane takes two ADJACENT
// records of the same row node: their products are summed in registers before they reach LDS - half the product slots, half the
// LDS traffic and half the segmented-sum trips; what the slots save pays for the window's LDS.  The values of a pair are stored
// SPLIT by position in the pair (pkc2 / dxy2: all first records, then all second records), so that each of a lane's two 16-byte
// loads belongs to a fully contiguous stream: with the interleaved array (32-byte lane stride, every 128-byte line touched by
// two instructions) the same kernel took 164 us instead of 143 (profiles/r04_windowed_tiles.txt).
//
// One dependent chain per tile, and the window is not on it: a tile requests the NEXT tile's lists before its own records,
// gathers the next tile's columns once the lists are in (in flight during its own products and segmented sums; hipcc's
// __syncthreads waits for LDS only, loads stay in flight across barriers) and hands the gathered window on in registers
// (WinPre); the node bookkeeping of the segmented sums is precomputed per node (`wbk`) and requested first, because vmcnt counts
// in order and a wait for two words must not be a wait for the record stream.  What a tile waits for is its own record stream:
// at most kWinPairs record pairs and kWinCols column records per lane, all requested at once (more per lane - tiles as large as
// the LDS would allow - spill at the Arnoldi kernel's 80-register cap: 3 + 2 spills 25 VGPRs, 2 + 1 needs 76).
//
// The input accessor must serve a node's components as one float4 (`node4`) and an entry behind the block rows as a float
// (`behind`): the fp32 gather-layout copy of the Krylov vector (PaddedX).  Products and sums are fp64.
#pragma once
#include "spmv_device.h"

namespace npg {

// device view of a matrix's windowed tile set (npg_csr, build_window_tiles); passed beside CsrDev to the kernels that use it
struct WinDev {
    const uint16_t *widx, *gidx;   // 16-bit window indices of the node records / column records (indexed like pcol / gcol)
    int64_t ngrec;                 // column records in all (a tile WITHOUT any clamps its idle loads to the last one)
    const double2 *gxy;            // the column records' values: the matrix's own (CsrDev::gxy / gz) or, when ghost nodes turned some
    const double *gz;              // of them into node records, the windowed set's own arrays
    const int32_t *wlist, *vlist;  // per-tile lists of distinct column nodes / other columns (WTileDesc::woff, voff index them)
    const double2 *pkc2;           // {K, C} split by position in the record pair: [npairs] first records, [npairs] second records
    int64_t npairs;
    const int32_t *wbk;            // per block node q: {end of its column records, end of its record PAIRS} relative to its tile
    const uint16_t *dwidx;         // windowed tiles of the rows behind the block rows: window indices of the coupling records,
    const int32_t *dbk;            // per such row the tile-local end of its coupling record pairs (null: those rows keep
    const double2 *dxy2;           // ordinary tiles), (d_x, d_y) split like pkc2
    const double *dz;              // d_z of those coupling records (the matrix's own array, or the windowed set's when ghost nodes added records)
    int64_t ndpairs;
};

__device__ __forceinline__ TileDesc ordinary(const WTileDesc &w) { return TileDesc{w.base, w.pbase, w.r0, w.nrows, w.n, w.npe}; }

constexpr int kWinPairs = 2;      // record pairs per lane a windowed tile may hold (npe <= 2 * kWinPairs * threads)
constexpr int kWinCols = 1;       // column records per lane
constexpr int kWinNodes = 2;      // distinct column nodes per lane (nw); distinct other columns: one per lane (nv)

// The window of a tile as gathered: held in registers from the middle of the PREVIOUS tile (below) to the start of its own.
struct WinPre {
    float4 f[kWinNodes];
    float v;
};

// the two halves of a window prefetch: request the next tile's lists / gather their columns
// (most tiles have at most NT distinct column nodes: the second list load and gather are skipped for them, wave-uniformly - the
//  tile is bound by the vector-memory instructions its waves issue)
template <int NT>
__device__ __forceinline__ void win_lists(const WinDev &A, const WTileDesc &next, int tid, int32_t (&wc)[kWinNodes], int32_t &vc) {
    wc[0] = __builtin_nontemporal_load(A.wlist + next.woff + min(tid, next.nw - 1));
#pragma unroll
    for (int u = 1; u < kWinNodes; ++u) {
        wc[u] = 0;
        if (next.nw > u * NT) wc[u] = __builtin_nontemporal_load(A.wlist + next.woff + min(tid + u * NT, next.nw - 1));
    }
    vc = __builtin_nontemporal_load(A.vlist + next.voff + min(tid, max(next.nv - 1, 0)));
}
template <int NT, class XF, int DIAG = 0>
__device__ __forceinline__ void win_gather(const XF &x, const WTileDesc &next, const int32_t (&wc)[kWinNodes], int32_t vc, WinPre &w) {
    w.f[0] = (DIAG & 1) ? make_float4((float)wc[0], 1.f, 2.f, 0.f) : x.node4(wc[0]);
#pragma unroll
    for (int u = 1; u < kWinNodes; ++u)
        if (next.nw > u * NT) w.f[u] = (DIAG & 1) ? make_float4((float)wc[u], 1.f, 2.f, 0.f) : x.node4(wc[u]);
    w.v = (DIAG & 1) ? (float)vc : x.behind(vc);
}

// lists -> gathers for a tile nobody prefetched (a workgroup's first windowed tile): two dependent round trips, once
template <int NT, class XF>
__device__ __forceinline__ void win_first(const WinDev &W, const XF &x, const WTileDesc &td, WinPre &w) {
    const int tid = threadIdx.x;
    int32_t wc[kWinNodes], vc;
    win_lists<NT>(W, td, tid, wc, vc);
    win_gather<NT>(x, td, wc, vc, w);
}

// One windowed BLOCK tile.  `w`: on entry this tile's window (win_first, or the previous call), on return - if `pre_next` - the window
// of tile `next`: its lists are requested before this tile's records, its gathers are issued once the lists are in and are in
// flight during this tile's products and segmented sums, so that the only global round trip a tile waits for is its own
// record stream (hipcc's __syncthreads waits for LDS only: loads stay in flight across the barriers).
// On return (after the trailing barrier) out[r - r0] holds (A x)[r] for the tile's rows.
// DIAG (timing diagnostics of tools/window_ab.py only; results are then meaningless): bit 0 = the window is filled without
// gathering, bit 1 = no segmented sums, bit 2 = no record loads (products of constants), bit 5 = the products never reach LDS: no
// product slots, no segmented sums and one barrier less - an upper bound on what a register segmented scan could buy.
template <int NT, int L, class XF, int TNNZ, class PROF = NoProf, int DIAG = 0>
__device__ __forceinline__ void spmv_tile_win(const CsrDev &A, const WinDev &W, const XF x, const WTileDesc &td, const WTileDesc &next, bool pre_next,
                                              WinPre &w, TileLdsT<TNNZ> &t, double *__restrict__ out, PROF prof = PROF()) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // per-tile re-made lane offsets (spmv_tile's REMAT): keeps the stream bases out of the caller's loop
    prof.stamp(4);
    const bool full = td.r0 < 3 * A.nfull;
    const int ncomp = full ? 3 : 2;
    const int q0 = node_of_row(A, td.r0);
    const int nnode = full ? (td.nrows * 21846) >> 16 : td.nrows >> 1;       // nrows / 3 for nrows < 2^15
    const int npair = td.npe >> 1, n = td.n, nw = td.nw, nv = td.nv;
    const int64_t pbase = td.pbase, base = td.base;
    const int slot0 = ncomp * npair;                         // column-record products live behind the pair products
    const int slots = slot0 + ncomp * n;
    float4 *__restrict__ win = reinterpret_cast<float4 *>(t.prod + ((slots + 1) & ~1));
    float *__restrict__ vwin = reinterpret_cast<float *>(win + nw);
    // node bookkeeping of the segmented sums, tile-local and ready for LDS (requested FIRST: vmcnt counts in order, and a wait
    // for these two words must not be a wait for the record stream behind them)
    int2 bk = make_int2(0, 0);          // (only the waves that hold nodes issue the load)
    if (tid < nnode) bk = *reinterpret_cast<const int2 *>(W.wbk + 2 * (size_t)(q0 + tid));
    // ---- (1) this tile's window, gathered a tile ago
    if (tid < nw) win[tid] = w.f[0];
#pragma unroll
    for (int u = 1; u < kWinNodes; ++u)
        if (nw > u * NT && tid + u * NT < nw) win[tid + u * NT] = w.f[u];
    if (tid < nv) vwin[tid] = w.v;
    prof.stamp(8);
    // Every load below is unconditional, from an index clamped into the tile's range (a lane without work re-reads the last
    // element and never uses it): no control flow between the loads, so they are all in flight together.
    // ---- (2) the NEXT tile's distinct columns
    int32_t wc[kWinNodes], vc = 0;
    if (pre_next) win_lists<NT>(W, next, tid, wc, vc);
    // ---- (3) every record of this tile
    uint32_t ip[kWinPairs];
    double k0[kWinPairs], c0[kWinPairs], k1[kWinPairs], c1[kWinPairs];
    {
        const uint32_t *__restrict__ wp = reinterpret_cast<const uint32_t *>(W.widx + pbase);
        const double *__restrict__ ka = reinterpret_cast<const double *>(W.pkc2 + (pbase >> 1));
        const double *__restrict__ kb = reinterpret_cast<const double *>(W.pkc2 + W.npairs + (pbase >> 1));
#pragma unroll
        for (int u = 0; u < kWinPairs; ++u) {
            const int p = min(tid + u * NT, npair - 1);
            if (DIAG & 4) {
                ip[u] = (uint32_t)p & 0x00ff00ffu;
                k0[u] = c0[u] = k1[u] = c1[u] = (double)p;
                continue;
            }
            ip[u] = __builtin_nontemporal_load(wp + p);
            k0[u] = __builtin_nontemporal_load(ka + 2 * p);
            c0[u] = __builtin_nontemporal_load(ka + 2 * p + 1);
            k1[u] = __builtin_nontemporal_load(kb + 2 * p);
            c1[u] = __builtin_nontemporal_load(kb + 2 * p + 1);
        }
    }
    uint32_t gi[kWinCols];
    double ax[kWinCols], ay[kWinCols], az[kWinCols];
    {
#pragma unroll
        for (int u = 0; u < kWinCols; ++u) {
            const int e = min(tid + u * NT, max(n - 1, 0));
            if (DIAG & 4) {
                gi[u] = (uint32_t)e & 31u;
                ax[u] = ay[u] = az[u] = (double)e;
                continue;
            }
            // (a tile without column records - n = 0 - still issues these loads: keep them inside the arrays)
            const int64_t ge = min(base + e, W.ngrec - 1);
            const double *__restrict__ gq = reinterpret_cast<const double *>(W.gxy + ge);
            gi[u] = __builtin_nontemporal_load(W.gidx + ge);
            ax[u] = __builtin_nontemporal_load(gq);
            ay[u] = __builtin_nontemporal_load(gq + 1);
            az[u] = __builtin_nontemporal_load(W.gz + ge);
        }
    }
    prof.stamp(9);
    if (tid < nnode) {
        t.rp[tid + 1] = bk.x;
        t.prp[tid + 1] = bk.y;
    }
    if (tid == 0) t.rp[0] = t.prp[0] = 0;
    prof.stamp(3);
    __syncthreads();
    prof.stamp(5);
    // ---- (4) ONE gather per distinct column of the NEXT tile, ascending along the lanes: in flight until that tile starts
    if (pre_next) win_gather<NT, XF, DIAG>(x, next, wc, vc, w);
    // ---- (5) products: two adjacent records of one row node per lane, summed before they reach LDS
    double dacc = 0.0;          // (DIAG bit 5: the products stay in the lane - no product slots, no segmented sums, one barrier less)
#pragma unroll
    for (int u = 0; u < kWinPairs; ++u) {
        const int p = tid + u * NT;
        if (p < npair) {
            const float4 fa = win[ip[u] & 0xffffu], fb = win[ip[u] >> 16];
            const double xa = (double)fa.x, ya = (double)fa.y, xb = (double)fb.x, yb = (double)fb.y;
            if constexpr ((DIAG & 32) != 0) {
                dacc += ((k0[u] * xa + c0[u] * ya) + (k1[u] * xb + c1[u] * yb)) + ((k0[u] * ya - c0[u] * xa) + (k1[u] * yb - c1[u] * xb)) +
                        (k0[u] * (double)fa.z + k1[u] * (double)fb.z);
                continue;
            }
            t.prod[p] = (k0[u] * xa + c0[u] * ya) + (k1[u] * xb + c1[u] * yb);
            t.prod[npair + p] = (k0[u] * ya - c0[u] * xa) + (k1[u] * yb - c1[u] * xb);
            if (full) t.prod[2 * npair + p] = k0[u] * (double)fa.z + k1[u] * (double)fb.z;
        }
    }
#pragma unroll
    for (int u = 0; u < kWinCols; ++u) {
        const int e = tid + u * NT;
        if (e < n) {
            const double xv = (double)vwin[gi[u]];
            if constexpr ((DIAG & 32) != 0) {
                dacc += (ax[u] + ay[u] + az[u]) * xv;
                continue;
            }
            t.prod[slot0 + e] = ax[u] * xv;
            t.prod[slot0 + n + e] = ay[u] * xv;
            if (full) t.prod[slot0 + 2 * n + e] = az[u] * xv;
        }
    }
    prof.stamp(0);
    if constexpr ((DIAG & 32) != 0) {
        out[tid & (kTileRows - 1)] = dacc;
        __syncthreads();
        return;
    }
    __syncthreads();
    prof.stamp(1);
    // ---- (6) segmented sums: a lane group sums the two or three rows of a NODE together (spmv_tile's node-wise loop)
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    for (int q = g; q < ((DIAG & 2) ? 0 : nnode); q += NT / L) {
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
            const double a1 = t.prod[npair + k], b1 = t.prod[npair + k + 1];
            s0 += a0 + (two ? b0 : 0.0);
            s1 += a1 + (two ? b1 : 0.0);
            if (full) {
                const double a2 = t.prod[2 * npair + k], b2 = t.prod[2 * npair + k + 1];
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
    prof.stamp(2);
    __syncthreads();
}

// One windowed tile of rows BEHIND the block rows (the divergence rows): coupling records {c, d_x, d_y, d_z} in pairs - two
// adjacent records of one row per lane, their six products summed into ONE LDS slot - with every distinct column node gathered
// once into the window, as in the block tiles; same prefetch protocol (`w`, `next`, `pre_next`).  These tiles hold no other
// entries (the host keeps a matrix whose rows behind the block hold CSR entries on ordinary tiles).
template <int NT, int L, class XF, int TNNZ>
__device__ __forceinline__ void spmv_tile_winrows(const CsrDev &A, const WinDev &W, const XF x, const WTileDesc &td, const WTileDesc &next, bool pre_next,
                                                  WinPre &w, TileLdsT<TNNZ> &t, double *__restrict__ out) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int nrows = td.nrows, npair = td.npe >> 1, nw = td.nw;
    const int64_t pbase = td.pbase;
    float4 *__restrict__ win = reinterpret_cast<float4 *>(t.prod + ((npair + 1) & ~1));
    int32_t bk = 0;
    if (tid < nrows) bk = W.dbk[(td.r0 - block_rows(A)) + tid];
    if (tid < nw) win[tid] = w.f[0];
#pragma unroll
    for (int u = 1; u < kWinNodes; ++u)
        if (nw > u * NT && tid + u * NT < nw) win[tid + u * NT] = w.f[u];
    int32_t wc[kWinNodes], vc = 0;
    if (pre_next) win_lists<NT>(W, next, tid, wc, vc);
    uint32_t ip[kWinPairs];
    double dx0[kWinPairs], dy0[kWinPairs], dx1[kWinPairs], dy1[kWinPairs], dz0[kWinPairs], dz1[kWinPairs];
    {
        const uint32_t *__restrict__ wp = reinterpret_cast<const uint32_t *>(W.dwidx + pbase);
        const double *__restrict__ xa = reinterpret_cast<const double *>(W.dxy2 + (pbase >> 1));
        const double *__restrict__ xb = reinterpret_cast<const double *>(W.dxy2 + W.ndpairs + (pbase >> 1));
        const double *__restrict__ zz = W.dz + pbase;
#pragma unroll
        for (int u = 0; u < kWinPairs; ++u) {
            const int p = min(tid + u * NT, npair - 1);
            ip[u] = __builtin_nontemporal_load(wp + p);
            dx0[u] = __builtin_nontemporal_load(xa + 2 * p);
            dy0[u] = __builtin_nontemporal_load(xa + 2 * p + 1);
            dx1[u] = __builtin_nontemporal_load(xb + 2 * p);
            dy1[u] = __builtin_nontemporal_load(xb + 2 * p + 1);
            dz0[u] = __builtin_nontemporal_load(zz + 2 * p);
            dz1[u] = __builtin_nontemporal_load(zz + 2 * p + 1);
        }
    }
    if (tid < nrows) t.prp[tid + 1] = bk;
    if (tid == 0) t.prp[0] = 0;
    __syncthreads();
    if (pre_next) win_gather<NT>(x, next, wc, vc, w);
#pragma unroll
    for (int u = 0; u < kWinPairs; ++u) {
        const int p = tid + u * NT;
        if (p < npair) {
            const float4 fa = win[ip[u] & 0xffffu], fb = win[ip[u] >> 16];
            t.prod[p] = (dx0[u] * (double)fa.x + dy0[u] * (double)fa.y + dz0[u] * (double)fa.z) +
                        (dx1[u] * (double)fb.x + dy1[u] * (double)fb.y + dz1[u] * (double)fb.z);
        }
    }
    __syncthreads();
    const int g = threadIdx.x / L, l = threadIdx.x % L;
    for (int r = g; r < nrows; r += NT / L) {
        double s = 0.0;
        const int pe = t.prp[r + 1];
        for (int k = t.prp[r] + 2 * l; k < pe; k += 2 * L) {
            const double a = t.prod[k], b = t.prod[k + 1];
            s += a + (k + 1 < pe ? b : 0.0);
        }
        s = group_sum_dpp<L>(s);
        if (l == 0) out[r] = s;
    }
    __syncthreads();
}

}  // namespace npg

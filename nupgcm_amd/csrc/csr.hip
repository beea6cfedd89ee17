// CSR matrices on the device: construction from Julia's CSC, round trip, same-pattern combination, SpMV.
#include <algorithm>

#include "common.h"
#include "spmv_device.h"
#include "spmv_window.h"

namespace npg {

// Row tiles for the CSR-stream kernels: consecutive whole rows, at most kTileNnz stored entries and kTileRows rows per
// tile; small matrices get about one tile per CU.  A row longer than kTileNnz becomes a tile of its own.
int tile_boundaries(const npg_csr *A, int tile_slots, std::vector<int32_t> &tp, int max_rows) {
    const int64_t m = A->m;
    const int64_t *rp = A->h_rowptr.data();
    const int64_t *pp = A->nnode() ? A->h_prow.data() : nullptr;
    const int64_t nf3 = 3 * (int64_t)A->nfull, nbr = A->block_rows();
    auto node = [&](int64_t r) { return r < nf3 ? r / 3 : A->nfull + (r - nf3) / 2; };      // r starts a node
    const int64_t *dp = A->drow ? A->h_drow.data() : nullptr;          // coupling records of the rows behind the block rows
    const int64_t *gp = A->grow ? A->h_grow.data() : nullptr;          // column records of the block rows
    const int64_t slots_total = rp[m] + (pp ? 3 * pp[A->nfull] + 2 * (pp[A->nnode()] - pp[A->nfull]) : 0) + (dp ? dp[m - nbr] : 0) +
                                (gp ? 3 * gp[A->nfull] + 2 * (gp[A->nnode()] - gp[A->nfull]) : 0);
    int64_t target = slots_total / (int64_t)A->ctx->num_cu;
    target = std::min<int64_t>(tile_slots, std::max<int64_t>(1024, target)); // >= 1 tile per CU on small matrices
    // LDS product slots of rows [a, b): their CSR entries + one per component per record (a, b on node boundaries of
    // one kind inside the block rows)
    auto slots = [&](int64_t a, int64_t b) {
        int64_t s = rp[b] - rp[a];
        if (pp && a < nbr) s += (a < nf3 ? 3 : 2) * (pp[node(b)] - pp[node(a)]);
        if (gp && a < nbr) s += (a < nf3 ? 3 : 2) * (gp[node(b)] - gp[node(a)]);
        if (dp && a >= nbr) s += dp[b - nbr] - dp[a - nbr];
        return s;
    };
    const int64_t max_rows_rec = std::min<int64_t>(max_rows, kTileRows / 2);      // TileLds::prp
    tp.clear();
    tp.push_back(0);
    int64_t r = 0;
    while (r < m) {
        const bool inblk = r < nbr;
        const int64_t step = !inblk ? 1 : (r < nf3 ? 3 : 2), lim = !inblk ? m : (r < nf3 ? nf3 : nbr);
        int64_t r1 = r + step;
        const int64_t mr = (!inblk && dp) ? max_rows_rec : max_rows;
        while (r1 < lim && r1 + step - r <= mr && slots(r, r1 + step) <= target) r1 += step;
        NPG_REQUIRE(!inblk || slots(r, r1) <= tile_slots, "build_tiles: the rows of one node do not fit one tile");
        NPG_REQUIRE(inblk || !dp || slots(r, r1) <= tile_slots, "build_tiles: a row with coupling records does not fit one tile");
        tp.push_back((int32_t)r1);
        r = r1;
    }
    return NPG_OK;
}

// flag[t] != 0: tile t reads a column >= first_ghost (one 64-thread workgroup per tile)
__global__ void k_tile_ghost_flags(const TileDesc *__restrict__ td, int ntiles, const int32_t *__restrict__ col,
                                   const int32_t *__restrict__ gcol, int32_t block_rows, int32_t first_ghost,
                                   int32_t *__restrict__ flag) {
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const TileDesc q = td[t];
        const int32_t *__restrict__ cc = (gcol && q.r0 < block_rows) ? gcol : col;     // column records or CSR entries
        int f = 0;
        for (int k = threadIdx.x; k < q.n; k += blockDim.x) f |= cc[q.base + k] >= first_ghost;
        if (f) flag[t] = 1;
    }
}

int build_tiles(npg_csr *A) {
    std::vector<int32_t> tp;
    int rc = tile_boundaries(A, kTileNnz, tp);
    if (rc) return rc;
    const int64_t m = A->m;
    A->ntiles = (int32_t)tp.size() - 1;
    // lanes per row in the segmented sums, from the LDS product slots per row (a lane takes two products per trip):
    // measured on bowl3D h = 0.02 (42 slots per row with node blocks) 8 lanes beat 16 by 2-3 % and 4 by 1 %
    const int64_t nrec = A->nnode() ? A->h_prow[A->nnode()] : 0;
    const double mean = m > 0 ? (double)(A->rnnz + 3 * nrec + A->ndrec + 3 * A->ngrec) / (double)m : 0.0;
    A->lanes_default = mean <= 12 ? 4 : mean <= 64 ? 8 : mean <= 256 ? 16 : 32;
    if (getenv("NPG_SPMV_LANES")) A->lanes_default = atoi(getenv("NPG_SPMV_LANES"));      // tuning override: 4, 8, 16 or 32
    if (!A->lanes_set) A->lanes = A->lanes_default;
    const int64_t *rp = A->h_rowptr.data();
    const int64_t nf3 = 3 * (int64_t)A->nfull, nbr = A->block_rows();
    auto node = [&](int64_t r) { return r < nf3 ? r / 3 : A->nfull + (r - nf3) / 2; };
    std::vector<TileDesc> td((size_t)A->ntiles);
    for (int t = 0; t < A->ntiles; ++t) {
        const int64_t r0 = tp[t], r1 = tp[t + 1];
        TileDesc &q = td[t];
        q.r0 = (int32_t)r0;
        q.nrows = (int32_t)(r1 - r0);
        q.base = rp[r0];
        q.n = (int32_t)(rp[r1] - rp[r0]);
        q.pbase = 0;
        q.npe = 0;
        if (r0 < nbr) {
            q.pbase = A->h_prow[node(r0)];
            q.npe = (int32_t)(A->h_prow[node(r1)] - q.pbase);
            if (A->grow) {          // (base, n) of a block tile describe its column records
                q.base = A->h_grow[node(r0)];
                q.n = (int32_t)(A->h_grow[node(r1)] - q.base);
            }
        } else if (A->drow) {
            q.pbase = A->h_drow[r0 - nbr];
            q.npe = (int32_t)(A->h_drow[r1 - nbr] - q.pbase);
        }
    }
    if (A->tile_ptr) NPG_HIP(hipFree(A->tile_ptr));
    A->gen++;
    NPG_HIP(hipMalloc((void **)&A->tile_ptr, std::max<size_t>(1, td.size()) * sizeof(TileDesc)));
    NPG_HIP(hipMemcpy(A->tile_ptr, td.data(), td.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    A->ntiles_interior = A->ntiles;
    A->h_tiles = td;
    if (A->n > A->m && A->ntiles > 0 && (A->rnnz > 0 || A->ngrec > 0)) {
        // row block of a distributed matrix: tiles that read no ghost column come first, so that a solver can run them
        // while the halo exchange is in flight (record columns are owned nodes by construction: only the CSR part counts)
        int32_t *dflag;
        std::vector<int32_t> flag((size_t)A->ntiles, 0);
        NPG_HIP(hipMalloc((void **)&dflag, flag.size() * sizeof(int32_t)));
        NPG_HIP(hipMemsetAsync(dflag, 0, flag.size() * sizeof(int32_t), A->ctx->stream));
        hipLaunchKernelGGL(k_tile_ghost_flags, dim3(std::min<int>(A->ntiles, 4096)), dim3(64), 0, A->ctx->stream, A->tile_ptr,
                           A->ntiles, A->col, (const int32_t *)A->gcol, (int32_t)A->block_rows(), (int32_t)A->m, dflag);
        NPG_HIP(hipMemcpyAsync(flag.data(), dflag, flag.size() * sizeof(int32_t), hipMemcpyDeviceToHost, A->ctx->stream));
        NPG_HIP(hipStreamSynchronize(A->ctx->stream));
        NPG_HIP(hipFree(dflag));
        std::vector<TileDesc> ord;
        ord.reserve(td.size());
        for (int pass = 0; pass < 2; ++pass) {
            for (int t = 0; t < A->ntiles; ++t)
                if ((flag[t] != 0) == (pass == 1)) ord.push_back(td[t]);
            if (pass == 0) A->ntiles_interior = (int32_t)ord.size();
        }
        NPG_HIP(hipMemcpy(A->tile_ptr, ord.data(), ord.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
        A->h_tiles = ord;
    }
    return NPG_OK;
}

static void free_window_tiles(npg_csr *A) {
    for (void *p : {(void *)A->wtile_ptr, (void *)A->widx, (void *)A->gidx, (void *)A->wlist, (void *)A->vlist, (void *)A->wbk, (void *)A->dwidx, (void *)A->dbk, (void *)A->pkc2, (void *)A->dxy2,
                    (void *)A->gslot, (void *)A->wgval, (void *)A->wdz})
        if (p) hipFree(p);
    A->gslot = nullptr;
    A->wgval = nullptr;
    A->wdz = nullptr;
    A->nw_grec = A->nw_rec = A->nw_drec = 0;
    A->wtile_ptr = nullptr;
    A->widx = A->gidx = nullptr;
    A->wlist = A->vlist = nullptr;
    A->wbk = A->dbk = nullptr;
    A->pkc2 = A->dxy2 = nullptr;
    A->dwidx = nullptr;
    A->nwrow_tiles = 0;
    A->nwtiles = A->nwtiles_interior = 0;
    A->nwlist = A->nvlist = 0;
}

// The windowed tile set of a node-blocked matrix in all-record form (spmv_window.h): a second tiling of the block rows whose
// tiles are sized by what they need in LDS - pair-summed products, column-record products, the window of distinct column
// nodes (16 B each) and of distinct other columns (4 B each) - followed by the ordinary tiles of the rows behind the block
// rows.  pcol / gcol: host copies of the record columns (every node's list already padded to an even count).
// Leaves the matrix without a windowed set (no error) when some node's rows would not fit a tile.
static int build_window_tiles_scaled(npg_csr *A, const std::vector<int32_t> &pcol_in, const std::vector<int32_t> &gcol_in,
                                     const std::vector<int32_t> &dcol_in, const std::vector<double> &pkc_in, const std::vector<double> &dxy_in,
                                     const std::vector<double> &gxy_in, const std::vector<double> &gz_in, const std::vector<double> &dz_in,
                                     const std::vector<int32_t> &rcol, const std::vector<double> &rval, double scale) {
    free_window_tiles(A);
    const int64_t nnode = A->nnode(), nfull = A->nfull, nbr = A->block_rows();
    if (nnode == 0 || !A->grow || A->pk9) return NPG_OK;
    // ---- ghost nodes (npg_csr_set_ghost_nodes): the column records of a row node that reference the components of ONE ghost node
    // and carry the {K, C} structure (A[x_q,x_c] = A[y_q,y_c] = A[z_q,z_c] = K, A[x_q,y_c] = -A[y_q,x_c] = C, nothing else) become a
    // node record with column node nnode + g in THIS tile set's own record arrays; everything else is carried over unchanged
    const int64_t ngn = A->ngn();
    std::vector<int64_t> wprow, wgrow;
    std::vector<int32_t> wpcol, wgcol;
    std::vector<double> wpkc, wgxy, wgz;
    int64_t converted = 0;
    std::vector<int32_t> gnode((size_t)std::max<int64_t>(A->n - A->m, 0), -1), gcomp(gnode.size(), 0);
    if (ngn > 0) {
        for (int64_t g = 0; g < ngn; ++g)
            for (int a = 0; a < A->gn_ncomp[(size_t)g]; ++a) {
                gnode[(size_t)(A->gn_col[(size_t)g] - A->m + a)] = (int32_t)g;
                gcomp[(size_t)(A->gn_col[(size_t)g] - A->m + a)] = a;
            }
        const std::vector<int64_t> &prow0 = A->h_prow, &grow0 = A->h_grow;
        wprow.assign((size_t)nnode + 1, 0);
        wgrow.assign((size_t)nnode + 1, 0);
        std::vector<int64_t> slot;           // per ghost node touched by the current row node: index of its records by component
        for (int64_t q = 0; q < nnode; ++q) {
            const bool qfull = q < nfull;
            // the node's own records, without the zero record that padded them (it is re-made below)
            int64_t pe = prow0[q + 1];
            if (pe - prow0[q] >= 2 && pkc_in[2 * (pe - 1)] == 0.0 && pkc_in[2 * (pe - 1) + 1] == 0.0 && pcol_in[pe - 1] == pcol_in[pe - 2]) --pe;
            for (int64_t e = prow0[q]; e < pe; ++e) {
                wpcol.push_back(pcol_in[e]);
                wpkc.push_back(pkc_in[2 * e]);
                wpkc.push_back(pkc_in[2 * e + 1]);
            }
            double scale_q = 0.0;
            for (int64_t e = prow0[q]; e < pe; ++e) scale_q = std::max(scale_q, std::fabs(pkc_in[2 * e]));
            const double tol = 1e-12 * scale_q;
            // column records, ascending columns: the components of a ghost node are adjacent
            for (int64_t e = grow0[q]; e < grow0[q + 1];) {
                const int64_t m0 = gcol_in[e];
                const int32_t g = m0 >= A->m ? gnode[(size_t)(m0 - A->m)] : -1;
                if (g < 0 || gcomp[(size_t)(m0 - A->m)] != 0) {
                    wgcol.push_back(gcol_in[e]);
                    wgxy.push_back(gxy_in[2 * e]);
                    wgxy.push_back(gxy_in[2 * e + 1]);
                    wgz.push_back(gz_in[e]);
                    ++e;
                    continue;
                }
                // records of this ghost node: columns m0 (x), m0 + 1 (y)[, m0 + 2 (z)], any of them possibly absent
                const int nc = A->gn_ncomp[(size_t)g];
                int64_t rec[3] = {-1, -1, -1}, e1 = e;
                while (e1 < grow0[q + 1] && gcol_in[e1] < m0 + nc) {
                    rec[gcol_in[e1] - m0] = e1;
                    ++e1;
                }
                auto val = [&](int c, int a) { return rec[c] < 0 ? 0.0 : (a == 2 ? gz_in[rec[c]] : gxy_in[2 * rec[c] + a]); };
                const double K = val(0, 0), Cc = val(1, 0);
                bool ok = std::fabs(val(1, 1) - K) <= tol && std::fabs(val(0, 1) + Cc) <= tol && std::fabs(val(0, 2)) <= tol &&
                          std::fabs(val(1, 2)) <= tol;
                if (nc == 3) ok = ok && std::fabs(val(2, 0)) <= tol && std::fabs(val(2, 1)) <= tol && (!qfull || std::fabs(val(2, 2) - K) <= tol);
                if (ok) {
                    wpcol.push_back((int32_t)(nnode + g));
                    wpkc.push_back(K);
                    wpkc.push_back(Cc);
                    ++converted;
                } else {
                    for (int64_t k = e; k < e1; ++k) {
                        wgcol.push_back(gcol_in[k]);
                        wgxy.push_back(gxy_in[2 * k]);
                        wgxy.push_back(gxy_in[2 * k + 1]);
                        wgz.push_back(gz_in[k]);
                    }
                }
                e = e1;
            }
            if (((int64_t)wpcol.size() - wprow[q]) & 1) {            // zero record on the node's last column node
                wpcol.push_back(wpcol.back());
                wpkc.push_back(0.0);
                wpkc.push_back(0.0);
            }
            wprow[q + 1] = (int64_t)wpcol.size();
            wgrow[q + 1] = (int64_t)wgcol.size();
        }
    }
    const bool own = ngn > 0 && converted > 0;
    const std::vector<int64_t> &prow = own ? wprow : A->h_prow, &grow = own ? wgrow : A->h_grow;
    const std::vector<int32_t> &pcol = own ? wpcol : pcol_in, &gcol = own ? wgcol : gcol_in;
    const std::vector<double> &pkc = own ? wpkc : pkc_in;
    const int64_t nwin_nodes = nnode + (own ? ngn : 0);
    constexpr int NT = 512;
    int64_t hard = 8 * (int64_t)kTileNnz;
    if (getenv("NPG_WIN_BYTES")) hard = std::min<int64_t>(hard, std::max<int64_t>(8192, atoll(getenv("NPG_WIN_BYTES"))));    // tuning: smaller tiles
    scale = std::min(1.0, std::max(0.25, scale));       // all caps of a tile scaled (build_window_tiles; NPG_WIN_SCALE)
    hard = (int64_t)(scale * (double)hard);
    const int64_t cap_p = (int64_t)(scale * kWinPairs * NT), cap_c = (int64_t)(scale * kWinCols * NT);
    // small matrices: about one tile per CU (as tile_boundaries does)
    const int64_t total = 8 * (3 * (prow[nfull] / 2 + grow[nfull]) + 2 * ((prow[nnode] - prow[nfull]) / 2 + grow[nnode] - grow[nfull]));
    const int64_t soft = std::min<int64_t>(hard, std::max<int64_t>(8 * 1024, total / std::max(1, A->ctx->num_cu)));
    std::vector<int32_t> stampW((size_t)nwin_nodes, -1), stampV((size_t)A->n, -1), posW((size_t)nwin_nodes, 0), posV((size_t)A->n, 0);
    std::vector<uint16_t> widx(pcol.size()), gidx(gcol.size());
    std::vector<int32_t> wlist, vlist, tw, tv, nwl, nvl, wbk((size_t)2 * nnode);
    std::vector<WTileDesc> blk;
    std::vector<char> ghost;
    auto bytes_of = [](int ncomp, int64_t np, int64_t ng, int64_t nw, int64_t nv) {
        const int64_t slots = ncomp * (np + ng);
        return 8 * ((slots + 1) & ~(int64_t)1) + 16 * nw + 4 * nv;
    };
    for (int kind = 0; kind < 2; ++kind) {
        const int64_t lo = kind ? nfull : 0, hi = kind ? nnode : nfull;
        const int ncomp = kind ? 2 : 3;
        int64_t q = lo;
        while (q < hi) {
            const int32_t T = (int32_t)blk.size();
            tw.clear();
            tv.clear();
            int64_t np = 0, ng = 0, qe = q;
            while (qe < hi) {
                nwl.clear();
                nvl.clear();
                for (int64_t e = prow[qe]; e < prow[qe + 1]; ++e)
                    if (stampW[pcol[e]] != T) {
                        stampW[pcol[e]] = T;
                        nwl.push_back(pcol[e]);
                    }
                for (int64_t e = grow[qe]; e < grow[qe + 1]; ++e)
                    if (stampV[gcol[e]] != T) {
                        stampV[gcol[e]] = T;
                        nvl.push_back(gcol[e]);
                    }
                const int64_t np2 = np + (prow[qe + 1] - prow[qe]) / 2, ng2 = ng + (grow[qe + 1] - grow[qe]);
                const int64_t nw2 = (int64_t)(tw.size() + nwl.size()), nv2 = (int64_t)(tv.size() + nvl.size());
                const int64_t by = bytes_of(ncomp, np2, ng2, nw2, nv2);
                const bool shape = (qe + 1 - q) * ncomp <= kTileRows && np2 <= cap_p && ng2 <= cap_c && nw2 <= kWinNodes * NT && nv2 <= NT;
                if (qe == q && !(shape && by <= hard)) return NPG_OK;       // one node's rows do not fit: no windowed set
                if (qe > q && !(shape && by <= soft)) {
                    for (int32_t c : nwl) stampW[c] = -1;
                    for (int32_t c : nvl) stampV[c] = -1;
                    break;
                }
                tw.insert(tw.end(), nwl.begin(), nwl.end());
                tv.insert(tv.end(), nvl.begin(), nvl.end());
                np = np2;
                ng = ng2;
                ++qe;
            }
            std::sort(tw.begin(), tw.end());
            std::sort(tv.begin(), tv.end());
            for (size_t i = 0; i < tw.size(); ++i) posW[tw[i]] = (int32_t)i;
            for (size_t i = 0; i < tv.size(); ++i) posV[tv[i]] = (int32_t)i;
            for (int64_t e = prow[q]; e < prow[qe]; ++e) widx[e] = (uint16_t)posW[pcol[e]];
            for (int64_t e = grow[q]; e < grow[qe]; ++e) gidx[e] = (uint16_t)posV[gcol[e]];
            for (int64_t k = q; k < qe; ++k) {
                wbk[2 * k] = (int32_t)(grow[k + 1] - grow[q]);
                wbk[2 * k + 1] = (int32_t)((prow[k + 1] - prow[q]) >> 1);
            }
            WTileDesc d;
            d.r0 = (int32_t)(kind ? 3 * nfull + 2 * (q - nfull) : 3 * q);
            d.nrows = (int32_t)((qe - q) * ncomp);
            d.base = grow[q];
            d.n = (int32_t)ng;
            d.pbase = prow[q];
            d.npe = (int32_t)(2 * np);
            d.woff = (int32_t)wlist.size();
            d.nw = (int32_t)tw.size();
            d.voff = (int32_t)vlist.size();
            d.nv = (int32_t)tv.size();
            NPG_REQUIRE(d.nw > 0, "build_window_tiles: a block tile without column nodes");
            wlist.insert(wlist.end(), tw.begin(), tw.end());
            vlist.insert(vlist.end(), tv.begin(), tv.end());
            blk.push_back(d);
            ghost.push_back((!tv.empty() && tv.back() >= A->m) || tw.back() >= nnode);   // (ghost columns; ghost nodes as record columns)
            q = qe;
        }
    }
    // The rows behind the block rows (the divergence rows) as windowed tiles too, when all they hold is coupling records: two
    // adjacent records of a row per lane (every row's list is padded to an even count), at most kWinPairs pairs per lane and
    // kWinNodes distinct column nodes per lane - small tiles, one dependent chain each (the ordinary tile function walks such
    // a row block in four trips of two round trips).  NPG_WIN_ROWS=0: ordinary tiles for these rows.
    std::vector<WTileDesc> rowt;
    std::vector<char> rghost;
    std::vector<uint16_t> dwidx;
    std::vector<int32_t> dbk;
    const int64_t nbehind = A->m - nbr;
    const bool rows_env = !(getenv("NPG_WIN_ROWS") && atoi(getenv("NPG_WIN_ROWS")) == 0);
    // A rank's row block with ghost nodes: what the rows behind the block rows (the divergence rows) hold on ghost columns is CSR
    // remainder for the ordinary tiles; for THIS tile set the entries on a ghost node's components become one coupling record
    // {nnode + g, d_x, d_y, d_z} behind the row's owned-node records - if that accounts for every remaining entry of those rows,
    // they are windowed like a one-GPU matrix's (and the Arnoldi kernel runs its instance without ordinary tile code)
    std::vector<int64_t> wdrow;
    std::vector<int32_t> wdcol;
    std::vector<double> wdxy, wdz;
    bool rows_own = false;
    if (own && rows_env && A->drow && nbehind > 0 && A->n > A->m) {
        rows_own = true;
        wdrow.assign((size_t)nbehind + 1, 0);
        const std::vector<int64_t> &d0 = A->h_drow, &rp0 = A->h_rowptr;
        for (int64_t r = 0; r < nbehind && rows_own; ++r) {
            int64_t e1 = d0[r + 1];
            if (e1 - d0[r] >= 2 && dxy_in[2 * (e1 - 1)] == 0.0 && dxy_in[2 * (e1 - 1) + 1] == 0.0 && dz_in[e1 - 1] == 0.0 &&
                dcol_in[e1 - 1] == dcol_in[e1 - 2])
                --e1;                                                    // (the zero record that padded the list: re-made below)
            for (int64_t e = d0[r]; e < e1; ++e) {
                wdcol.push_back(dcol_in[e]);
                wdxy.push_back(dxy_in[2 * e]);
                wdxy.push_back(dxy_in[2 * e + 1]);
                wdz.push_back(dz_in[e]);
            }
            int32_t cur = -1;
            for (int64_t k = rp0[nbr + r]; k < rp0[nbr + r + 1]; ++k) {
                const int64_t m0 = rcol[k];
                const int32_t g = m0 >= A->m ? gnode[(size_t)(m0 - A->m)] : -1;
                if (g < 0) {
                    rows_own = false;                                    // an entry that is no node component: ordinary tiles for these rows
                    break;
                }
                if (g != cur) {
                    wdcol.push_back((int32_t)(nnode + g));
                    wdxy.push_back(0.0);
                    wdxy.push_back(0.0);
                    wdz.push_back(0.0);
                    cur = g;
                }
                const int a = gcomp[(size_t)(m0 - A->m)];
                if (a == 2) wdz.back() = rval[k];
                else wdxy[wdxy.size() - 2 + a] = rval[k];
            }
            if (((int64_t)wdcol.size() - wdrow[r]) & 1) {
                wdcol.push_back(wdcol.back());
                wdxy.push_back(0.0);
                wdxy.push_back(0.0);
                wdz.push_back(0.0);
            }
            wdrow[r + 1] = (int64_t)wdcol.size();
        }
    }
    const std::vector<int32_t> &dcol = rows_own ? wdcol : dcol_in;
    const std::vector<double> &dxy = rows_own ? wdxy : dxy_in;
    if (rows_env && A->drow && nbehind > 0 && ((A->h_rowptr[A->m] - A->h_rowptr[nbr] == 0 && A->n == A->m) || rows_own)) {
        const std::vector<int64_t> &drow = rows_own ? wdrow : A->h_drow;
        dwidx.assign(dcol.size(), 0);
        dbk.assign((size_t)nbehind, 0);
        int64_t cap_pairs = cap_p;
        {       // small matrices: about as many tiles as the block rows got per byte
            const int64_t want = std::max<int64_t>(1, (int64_t)blk.size() * (drow[nbehind] * 28) / std::max<int64_t>(1, total * 3));
            cap_pairs = std::min<int64_t>(cap_pairs, std::max<int64_t>(64, drow[nbehind] / 2 / want));
        }
        int64_t r = 0;
        bool ok = true;
        while (r < nbehind && ok) {
            const int32_t T = (int32_t)(blk.size() + rowt.size());
            tw.clear();
            int64_t np = 0, re = r;
            while (re < nbehind) {
                nwl.clear();
                for (int64_t e = drow[re]; e < drow[re + 1]; ++e)
                    if (stampW[dcol[e]] != T) {
                        stampW[dcol[e]] = T;
                        nwl.push_back(dcol[e]);
                    }
                const int64_t np2 = np + (drow[re + 1] - drow[re]) / 2, nw2 = (int64_t)(tw.size() + nwl.size());
                const bool hardfit = re + 1 - r <= kTileRows / 2 && np2 <= kWinPairs * NT && nw2 <= kWinNodes * NT &&
                                     8 * ((np2 + 1) & ~(int64_t)1) + 16 * nw2 <= hard;
                if (re == r && !hardfit) {
                    ok = false;
                    break;
                }
                if (re > r && !(hardfit && np2 <= cap_pairs)) {
                    for (int32_t c : nwl) stampW[c] = -1;
                    break;
                }
                tw.insert(tw.end(), nwl.begin(), nwl.end());
                np = np2;
                ++re;
            }
            if (!ok) break;
            std::sort(tw.begin(), tw.end());
            for (size_t i = 0; i < tw.size(); ++i) posW[tw[i]] = (int32_t)i;
            for (int64_t e = drow[r]; e < drow[re]; ++e) dwidx[e] = (uint16_t)posW[dcol[e]];
            for (int64_t k = r; k < re; ++k) dbk[k] = (int32_t)((drow[k + 1] - drow[r]) >> 1);
            WTileDesc d;
            d.r0 = (int32_t)(nbr + r);
            d.nrows = (int32_t)(re - r);
            d.base = 0;
            d.n = 0;
            d.pbase = drow[r];
            d.npe = (int32_t)(2 * np);
            d.woff = (int32_t)wlist.size();
            d.nw = (int32_t)tw.size();
            d.voff = (int32_t)vlist.size();
            d.nv = 0;
            if (d.nw == 0) {            // (a block of empty rows: nothing to window)
                ok = false;
                break;
            }
            wlist.insert(wlist.end(), tw.begin(), tw.end());
            rowt.push_back(d);
            rghost.push_back(tw.back() >= nnode);
            r = re;
        }
        if (!ok) rowt.clear();
    }
    vlist.push_back((int32_t)nbr);           // sentinel: a tile without such columns still reads one entry at its offset
    NPG_REQUIRE(wlist.size() < (size_t)INT32_MAX && vlist.size() < (size_t)INT32_MAX, "build_window_tiles: window lists exceed int32 offsets");
    // the tiles of the rows behind the block rows: windowed (above), or as build_tiles made (and ordered) them
    std::vector<WTileDesc> ord;
    ord.reserve(blk.size() + A->h_tiles.size() + rowt.size());
    auto widen = [](const TileDesc &t) { return WTileDesc{t.base, t.pbase, t.r0, t.nrows, t.n, t.npe, 0, 0, 0, 0}; };
    int32_t nint = 0;
    // order within a pass (tuning, NPG_WIN_ORDER): 0 = block tiles, then the rows behind them; 1 = the other way round;
    // 2 = the tiles of the rows behind the block rows spread evenly among the block tiles
    const int order = getenv("NPG_WIN_ORDER") ? atoi(getenv("NPG_WIN_ORDER")) : 0;
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<WTileDesc> a, b;
        for (size_t t = 0; t < blk.size(); ++t)
            if ((ghost[t] != 0) == (pass == 1)) a.push_back(blk[t]);
        if (!rowt.empty()) {
            for (size_t t = 0; t < rowt.size(); ++t)
                if ((rghost[t] != 0) == (pass == 1)) b.push_back(rowt[t]);
        } else {
            for (int32_t t = 0; t < A->ntiles; ++t)
                if (A->h_tiles[t].r0 >= nbr && (t >= A->ntiles_interior) == (pass == 1)) b.push_back(widen(A->h_tiles[t]));
        }
        if (order == 1) {
            ord.insert(ord.end(), b.begin(), b.end());
            ord.insert(ord.end(), a.begin(), a.end());
        } else if (order == 2 && !b.empty()) {
            size_t ib = 0;
            for (size_t ia = 0; ia < a.size(); ++ia) {
                while (ib < b.size() && ib * a.size() <= ia * b.size()) ord.push_back(b[ib++]);
                ord.push_back(a[ia]);
            }
            while (ib < b.size()) ord.push_back(b[ib++]);
        } else {
            ord.insert(ord.end(), a.begin(), a.end());
            ord.insert(ord.end(), b.begin(), b.end());
        }
        if (pass == 0) nint = (int32_t)ord.size();
    }
    // NPG_WIN_UNCACHED=1 (experiment): the once-read record streams in uncached device memory, so that they do not displace the
    // Krylov basis and the gather-layout copy from L2 / the Infinity Cache
    static const bool uncached = getenv("NPG_WIN_UNCACHED") && atoi(getenv("NPG_WIN_UNCACHED")) != 0;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        if (uncached && bytes >= (1u << 20))
            NPG_HIP(hipExtMallocWithFlags(dst, bytes, hipDeviceMallocUncached));
        else
            NPG_HIP(hipMalloc(dst, std::max<size_t>(bytes, 16)));
        if (bytes) NPG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return NPG_OK;
    };
    int rc = NPG_OK;
    auto chk = [&](int r) { if (rc == NPG_OK) rc = r; };
    chk(up((void **)&A->wtile_ptr, ord.data(), ord.size() * sizeof(WTileDesc)));
    chk(up((void **)&A->widx, widx.data(), widx.size() * sizeof(uint16_t)));
    chk(up((void **)&A->gidx, gidx.data(), gidx.size() * sizeof(uint16_t)));
    chk(up((void **)&A->wlist, wlist.data(), wlist.size() * sizeof(int32_t)));
    chk(up((void **)&A->vlist, vlist.data(), vlist.size() * sizeof(int32_t)));
    chk(up((void **)&A->wbk, wbk.data(), wbk.size() * sizeof(int32_t)));
    {       // the {K, C} values once more, split by position in the pair: [pairs] first records, then [pairs] second records -
            // a lane's two 16-byte loads are then each part of a fully contiguous stream (the interleaved pkc gives 32-byte strides)
        const size_t P = pcol.size() / 2;
        std::vector<double> k2(4 * P);
        for (size_t pr = 0; pr < P; ++pr) {
            k2[2 * pr] = pkc[4 * pr];
            k2[2 * pr + 1] = pkc[4 * pr + 1];
            k2[2 * P + 2 * pr] = pkc[4 * pr + 2];
            k2[2 * P + 2 * pr + 1] = pkc[4 * pr + 3];
        }
        chk(up((void **)&A->pkc2, k2.data(), k2.size() * sizeof(double)));
        A->npairs = (int64_t)P;
    }
    if (!rowt.empty()) {
        chk(up((void **)&A->dwidx, dwidx.data(), dwidx.size() * sizeof(uint16_t)));
        chk(up((void **)&A->dbk, dbk.data(), dbk.size() * sizeof(int32_t)));
        const size_t P = dcol.size() / 2;      // (d_x, d_y) split by position in the record pair, like pkc2
        std::vector<double> d2(4 * P);
        for (size_t pr = 0; pr < P; ++pr) {
            d2[2 * pr] = dxy[4 * pr];
            d2[2 * pr + 1] = dxy[4 * pr + 1];
            d2[2 * P + 2 * pr] = dxy[4 * pr + 2];
            d2[2 * P + 2 * pr + 1] = dxy[4 * pr + 3];
        }
        chk(up((void **)&A->dxy2, d2.data(), d2.size() * sizeof(double)));
        A->ndpairs = (int64_t)P;
        if (rows_own) chk(up((void **)&A->wdz, wdz.data(), wdz.size() * sizeof(double)));
        A->nw_drec = (int64_t)dcol.size();
    }
    if (own) {
        std::vector<double> gv(3 * wgz.size());
        std::copy(wgxy.begin(), wgxy.end(), gv.begin());
        std::copy(wgz.begin(), wgz.end(), gv.begin() + (ptrdiff_t)wgxy.size());
        chk(up((void **)&A->wgval, gv.data(), gv.size() * sizeof(double)));
        // where the halo unpack stores a ghost column's float beside its place behind the owned entries: its node's 4-float slot
        std::vector<int32_t> gs((size_t)(A->n - A->m), -1);
        for (int64_t g = 0; g < ngn; ++g)
            for (int a = 0; a < A->gn_ncomp[(size_t)g]; ++a) gs[(size_t)(A->gn_col[(size_t)g] - A->m + a)] = (int32_t)(4 * (nnode + g) + a);
        chk(up((void **)&A->gslot, gs.data(), gs.size() * sizeof(int32_t)));
    }
    A->nw_grec = (int64_t)gcol.size();
    A->nw_rec = prow[nnode];
    A->nwrow_tiles = (int32_t)rowt.size();
    if (rc != NPG_OK) {
        free_window_tiles(A);
        return rc;
    }
    A->nwtiles = (int32_t)ord.size();
    A->nwtiles_interior = nint;
    A->nwlist = (int64_t)wlist.size();
    A->nvlist = (int64_t)vlist.size();
    // lanes per node in the segmented sums of a windowed tile (a lane takes two slots per trip): measured on bowl3D h = 0.02
    // (14 pair slots + 8 column-record slots per row, 64 nodes per tile) 8 lanes 136.9 us per product against 142.6 with 4
    const double mean = (double)(prow[nnode] / 2 + grow[nnode]) / (double)nnode;
    A->wlanes = mean <= 12 ? 4 : 8;
    if (getenv("NPG_SPMV_WLANES")) A->wlanes = atoi(getenv("NPG_SPMV_WLANES")) == 8 ? 8 : 4;
    A->gen++;
    return NPG_OK;
}

// The windowed tile set at full tile size (NPG_WIN_SCALE: all caps of a tile scaled, tuning).  A rule that shrinks the tiles of
// small matrices until their number fills a whole number of rounds of the 3 x num_cu persistent workgroups was measured and
// dropped: bowl3D h = 0.04 (2.2 -> 2.9 rounds) 27.5 -> 29.0 us per Arnoldi launch, h = 0.05 (1.1 -> 1.9 rounds) 18.8 -> 20.6 us -
// the workgroups do not move in rounds, and smaller tiles cost more than a partly filled last round (profiles/r04_windowed_tiles.txt).
static int build_window_tiles(npg_csr *A, const std::vector<int32_t> &pcol, const std::vector<int32_t> &gcol,
                              const std::vector<int32_t> &dcol, const std::vector<double> &pkc, const std::vector<double> &dxy,
                              const std::vector<double> &gxy, const std::vector<double> &gz, const std::vector<double> &dz,
                              const std::vector<int32_t> &rcol, const std::vector<double> &rval) {
    const double s_env = getenv("NPG_WIN_SCALE") ? atof(getenv("NPG_WIN_SCALE")) : -1.0;
    return build_window_tiles_scaled(A, pcol, gcol, dcol, pkc, dxy, gxy, gz, dz, rcol, rval, s_env > 0 ? s_env : 1.0);
}

constexpr int kSpmvThreads = 512;

// (see NbEpi, common.h) after the tile's y has been written back to LDS
template <int NT>
__device__ __forceinline__ void nb_epilogue(const NbEpi &nb, int r0, int nrows, const double *sw) {
    __syncthreads();
    for (int r = threadIdx.x; r < nrows; r += NT) {
        const int row = r0 + r;
        if (row < nb.rows) {
            double s = 0.0;
            for (int64_t k = nb.drp[row]; k < nb.drp[row + 1]; ++k) s += nb.dval[k] * sw[nb.dcol[k] - r0];
            nb.t[row] = s;
            if (nb.xu) nb.xu[row] = (nb.zero ? 0.0 : nb.xu[row]) + nb.w * s;
        }
    }
    __syncthreads();               // (the next tile writes sw)
}

template <int L, bool F32, bool N9 = false, bool NB = false>
__global__ void __launch_bounds__(kSpmvThreads, 6) k_spmv(CsrDev A, const TileDesc *__restrict__ tile_ptr, int ntiles,
                                                        const double *__restrict__ x, SpmvEpi e, NbEpi nb = NbEpi{}) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    TileDesc td = tile_ptr[t];
    while (true) {
        const int tn = t + gridDim.x;
        TileDesc nd = td;
        if (tn < ntiles) nd = tile_ptr[tn];              // in flight during this tile
        spmv_tile<kSpmvThreads, L, PlainX, kTileNnz, 4, NoProf, F32, false, N9>(A, PlainX{x}, td, tl, sw);
        for (int r = threadIdx.x; r < td.nrows; r += kSpmvThreads) {
            const int row = td.r0 + r;
            double v = e.alpha * sw[r];
            if (e.beta != 0.0) v += e.beta * e.c[row];
            e.y[row] = v;
            if (e.z) {                                   // second output: z = zc zin + w dg .* y  (smoother relaxations, mg.hip)
                double t = e.w * (e.dg ? e.dg[row] : 1.0) * v;
                if (e.zin) t += e.zc * e.zin[row];
                e.z[row] = t;
            }
            if constexpr (NB) sw[r] = v;
        }
        if constexpr (NB) nb_epilogue<kSpmvThreads>(nb, td.r0, td.nrows, sw);
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
}

__global__ void k_combine(double *out, double a, const double *X, double b, const double *Y, const double *Z,
                          int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * X[i] + b * (Y[i] + Z[i]);
}

__global__ void k_gather_values(double *__restrict__ dst, const double *__restrict__ src, const int64_t *__restrict__ idx,
                                int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[idx[i]];
}

__global__ void k_inv_diag(const int64_t *rowptr, const int32_t *col, const double *val, double *d, int64_t m) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < m; r += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0;
        for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
            if (col[k] == r) a += val[k];
        d[r] = 1.0 / a;
    }
}

static int upload_csr(npg_ctx *ctx, int64_t m, int64_t n, std::vector<int64_t> &&rowptr, const std::vector<int32_t> &col,
                      const std::vector<double> &val, npg_csr **out) {
    npg_csr *A = new npg_csr();
    A->ctx = ctx;
    A->m = m;
    A->n = n;
    A->nnz = (int64_t)col.size();
    A->rnnz = A->nnz;
    NPG_HIP(hipSetDevice(ctx->device));
    NPG_HIP(hipMalloc((void **)&A->rowptr, (size_t)(m + 1) * sizeof(int64_t)));
    NPG_HIP(hipMalloc((void **)&A->col, std::max<size_t>(1, col.size()) * sizeof(int32_t)));
    NPG_HIP(hipMalloc((void **)&A->val, std::max<size_t>(1, val.size()) * sizeof(double)));
    NPG_HIP(hipMemcpy(A->rowptr, rowptr.data(), (size_t)(m + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (!col.empty()) {
        NPG_HIP(hipMemcpy(A->col, col.data(), col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->val, val.data(), val.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    A->h_rowptr = std::move(rowptr);
    int rc = build_tiles(A);
    if (rc) return rc;
    *out = A;
    return NPG_OK;
}

template <int L>
static void launch_spmv(const npg_csr *A, const double *x, const SpmvEpi &e, const NbEpi *nb = nullptr) {
    const int grid = std::min<int>(A->ntiles, 3 * A->ctx->num_cu);      // 3 x 38 KiB of LDS per CU
    if (nb) {                   // (node-blocked {K, C} matrices only: nb_epilogue_ok)
        hipLaunchKernelGGL((k_spmv<L, false, false, true>), dim3(std::max(grid, 1)), dim3(kSpmvThreads), 0, A->ctx->stream, csr_view(A),
                           A->tile_ptr, A->ntiles, x, e, *nb);
        return;
    }
    if (A->pk9) {               // full node records (their fp32 copies are not kept: the values change with the closures)
        hipLaunchKernelGGL((k_spmv<L, false, true>), dim3(std::max(grid, 1)), dim3(kSpmvThreads), 0, A->ctx->stream, csr_view(A),
                           A->tile_ptr, A->ntiles, x, e, NbEpi{});
        return;
    }
    if (e.f32 && A->val32 && (A->nnode() == 0 || A->pkc32))
        hipLaunchKernelGGL((k_spmv<L, true>), dim3(std::max(grid, 1)), dim3(kSpmvThreads), 0, A->ctx->stream, csr_view(A),
                           A->tile_ptr, A->ntiles, x, e, NbEpi{});
    else
        hipLaunchKernelGGL((k_spmv<L, false>), dim3(std::max(grid, 1)), dim3(kSpmvThreads), 0, A->ctx->stream, csr_view(A),
                           A->tile_ptr, A->ntiles, x, e, NbEpi{});
}

__global__ void k_to_float32(const double *__restrict__ src, float *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

int csr_refresh_fp32(const npg_csr *Ac) {
    npg_csr *A = const_cast<npg_csr *>(spmv_form(Ac));       // the copies are a cache of val / pkc, not part of the matrix's value
    if (A->pk9) return NPG_OK;           // full node records: no fp32 copies (launch_spmv reads the fp64 values)
    NPG_HIP(hipSetDevice(A->ctx->device));
    const int64_t nz = A->rnnz, nrec = A->nnode() ? A->h_prow[A->nnode()] : 0;
    if (!A->val32) {
        NPG_HIP(hipMalloc((void **)&A->val32, std::max<size_t>(4, (size_t)nz + 2) * sizeof(float)));
        A->gen++;
    }
    if (nz)
        hipLaunchKernelGGL(k_to_float32, dim3((unsigned)std::min<int64_t>(4096, (nz + 255) / 256)), dim3(256), 0, A->ctx->stream,
                           A->val, A->val32, nz);
    if (nrec) {
        if (!A->pkc32) NPG_HIP(hipMalloc((void **)&A->pkc32, (size_t)2 * nrec * sizeof(float)));
        hipLaunchKernelGGL(k_to_float32, dim3((unsigned)std::min<int64_t>(4096, (2 * nrec + 255) / 256)), dim3(256), 0,
                           A->ctx->stream, A->pkc, A->pkc32, 2 * nrec);
    }
    if (A->ngrec) {
        if (!A->gval32) NPG_HIP(hipMalloc((void **)&A->gval32, (size_t)3 * A->ngrec * sizeof(float)));
        hipLaunchKernelGGL(k_to_float32, dim3((unsigned)std::min<int64_t>(4096, (3 * A->ngrec + 255) / 256)), dim3(256), 0,
                           A->ctx->stream, A->gval, A->gval32, 3 * A->ngrec);
    }
    if (A->ndrec) {
        if (!A->dval32) NPG_HIP(hipMalloc((void **)&A->dval32, (size_t)3 * A->ndrec * sizeof(float)));
        hipLaunchKernelGGL(k_to_float32, dim3((unsigned)std::min<int64_t>(4096, (3 * A->ndrec + 255) / 256)), dim3(256), 0,
                           A->ctx->stream, A->dval, A->dval32, 3 * A->ndrec);
    }
    return NPG_OK;
}

CsrDev csr_view(const npg_csr *A) {
    CsrDev v;
    v.rowptr = A->rowptr;
    v.col = A->col;
    v.val = A->val;
    v.nnz = A->rnnz;
    v.prow = A->prow;
    v.pcol = A->pcol;
    v.pkc = reinterpret_cast<const double2 *>(A->pkc);
    v.nfull = A->nfull;
    v.nsurf = A->nsurf;
    v.val32 = A->val32;
    v.pkc32 = reinterpret_cast<const float2 *>(A->pkc32);
    v.drow = A->drow;
    v.dcol = A->dcol;
    v.dxy = reinterpret_cast<const double2 *>(A->dval);
    v.dz = A->dval ? A->dval + 2 * A->ndrec : nullptr;
    v.dxy32 = reinterpret_cast<const float2 *>(A->dval32);
    v.dz32 = A->dval32 ? A->dval32 + 2 * A->ndrec : nullptr;
    v.grow = A->grow;
    v.gcol = A->gcol;
    v.gxy = reinterpret_cast<const double2 *>(A->gval);
    v.gz = A->gval ? A->gval + 2 * A->ngrec : nullptr;
    v.gxy32 = reinterpret_cast<const float2 *>(A->gval32);
    v.gz32 = A->gval32 ? A->gval32 + 2 * A->ngrec : nullptr;
    v.pk9 = A->pk9;
    v.pk9_32 = A->pk9_32;
    v.npk9 = A->npk9;
    return v;
}

WinDev win_view(const npg_csr *A) {
    WinDev w;
    w.widx = A->widx;
    w.gidx = A->gidx;
    w.ngrec = A->wgval ? A->nw_grec : A->ngrec;
    w.gxy = reinterpret_cast<const double2 *>(A->wgval ? A->wgval : A->gval);
    w.gz = A->wgval ? A->wgval + 2 * A->nw_grec : (A->gval ? A->gval + 2 * A->ngrec : nullptr);
    w.dz = A->wdz ? A->wdz : (A->dval ? A->dval + 2 * A->ndrec : nullptr);
    w.wlist = A->wlist;
    w.vlist = A->vlist;
    w.pkc2 = reinterpret_cast<const double2 *>(A->pkc2);
    w.npairs = A->npairs;
    w.wbk = A->wbk;
    w.dwidx = A->dwidx;
    w.dbk = A->dbk;
    w.dxy2 = reinterpret_cast<const double2 *>(A->dxy2);
    w.ndpairs = A->ndpairs;
    return w;
}

}  // namespace npg

using namespace npg;

// A block-diagonal inverse made by npg_csr_line_block_inverse also holds its blocks as DENSE packs (lb_val, lb_val32, lb_val16), which
// products read instead of the CSR values; only that routine refreshes them.  Any other writer of the CSR values of such a handle
// would leave the packs stale and products would silently use the old inverse: refused.
#define NPG_REQUIRE_NO_PACKS(A, who)                                                                                           \
    NPG_REQUIRE((A)->lb_nblocks == 0, who ": the matrix holds dense line-block packs (npg_csr_line_block_inverse); rewriting its CSR " \
                "values would leave them stale - build the values through npg_csr_line_block_inverse")

// Store the velocity block of A_inversion node by node (see spmv_device.h): one record {c, K, C} per coupled node pair.
// The structure is verified entry by entry on the host (K_xx = K_yy = K_zz, C_xy = -C_yx within rtol * row scale, nothing
// else in the block); if anything does not match the matrix is left untouched and *blocked = 0.
NPG_API int npg_csr_block_nodes(npg_csr *A, int64_t nfull, int64_t nsurf, double rtol, int *blocked) {
    NPG_REQUIRE(A && blocked && nfull >= 0 && nsurf >= 0, "npg_csr_block_nodes: bad argument");
    const int64_t nf3 = 3 * nfull, nbr = nf3 + 2 * nsurf, nnode = nfull + nsurf;
    NPG_REQUIRE(nbr <= A->m && nbr <= A->n, "npg_csr_block_nodes: more block rows than the matrix has");
    NPG_REQUIRE(A->nnode() == 0 && !A->packed, "npg_csr_block_nodes: matrix is already stored by node blocks (or has a record-form companion)");
    *blocked = 0;
    if (nnode == 0) return NPG_OK;
    NPG_HIP(hipStreamSynchronize(A->ctx->stream));
    std::vector<int32_t> col((size_t)A->nnz);
    std::vector<double> val((size_t)A->nnz);
    NPG_HIP(hipMemcpy(col.data(), A->col, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    NPG_HIP(hipMemcpy(val.data(), A->val, val.size() * sizeof(double), hipMemcpyDeviceToHost));
    const std::vector<int64_t> &rp = A->h_rowptr;
    std::vector<int64_t> prow((size_t)nnode + 1, 0), nrp((size_t)A->m + 1, 0);
    std::vector<int32_t> pcol, ncol, gcolv;
    std::vector<double> pkc, nval, gxy, gzv;
    std::vector<int64_t> growv((size_t)nnode + 1, 0);
    pcol.reserve((size_t)A->nnz / 5);
    pkc.reserve((size_t)A->nnz / 5 * 2);
    ncol.reserve((size_t)A->nnz / 2);
    nval.reserve((size_t)A->nnz / 2);
    // first DoF of column node c / node and component of a block column
    auto first = [&](int64_t c) { return c < nfull ? 3 * c : nf3 + 2 * (c - nfull); };
    // windowed tile set (spmv_window.h; NPG_SPMV_WINDOW=0: none): every node's record list is padded to an even count
    const char *we = getenv("NPG_SPMV_WINDOW");
    const bool want_win = !(we && atoi(we) == 0);
    int64_t nrec_real = 0;
    for (int64_t q = 0; q < nnode; ++q) {
        const bool qfull = q < nfull;
        const int64_t rx = first(q), ry = rx + 1, rz = rx + 2;
        int64_t ia = rp[rx], ib = rp[ry], iz = qfull ? rp[rz] : 0;
        const int64_t a1 = rp[rx + 1], b1 = rp[ry + 1], z1 = qfull ? rp[rz + 1] : 0;
        double scale = 0.0;
        for (int64_t k = ia; k < a1 && col[k] < nbr; ++k) scale = std::max(scale, std::fabs(val[k]));
        const double tol = rtol * scale;
        // sorted rows: the block columns come first.  x row: (x_c, K) (y_c, C); y row: (x_c, -C) (y_c, K); z row: (z_c, K)
        while (ia < a1 && col[ia] < nbr) {
            if (ia + 1 >= a1 || ib + 1 >= b1) return NPG_OK;
            const int64_t c0 = col[ia];
            const int64_t c = c0 < nf3 ? c0 / 3 : nfull + (c0 - nf3) / 2;
            if (c0 != first(c) || col[ia + 1] != c0 + 1 || col[ib] != c0 || col[ib + 1] != c0 + 1) return NPG_OK;
            const double K = val[ia], Cc = val[ia + 1];
            if (std::fabs(val[ib + 1] - K) > tol || std::fabs(val[ib] + Cc) > tol) return NPG_OK;
            if (qfull && c < nfull) {
                if (iz >= z1 || col[iz] != c0 + 2 || std::fabs(val[iz] - K) > tol) return NPG_OK;
                ++iz;
            }
            pcol.push_back((int32_t)c);
            pkc.push_back(K);
            pkc.push_back(Cc);
            ia += 2;
            ib += 2;
        }
        // nothing of the block may be left in the y and z rows
        if (ib < b1 && col[ib] < nbr) return NPG_OK;
        if (qfull && iz < z1 && col[iz] < nbr) return NPG_OK;
        nrec_real += (int64_t)pcol.size() - prow[q];
        if (want_win && (((int64_t)pcol.size() - prow[q]) & 1)) {        // zero record on the node's last column node
            pcol.push_back(pcol.back());
            pkc.push_back(0.0);
            pkc.push_back(0.0);
        }
        prow[q + 1] = (int64_t)pcol.size();
        const int64_t lo[3] = {ia, ib, iz}, hi[3] = {a1, b1, z1};
        // what is left of the node's rows (columns outside the block): as column records {m, a_x, a_y, a_z} - a three-way
        // merge of the sorted rows - and, in case they are not used, as plain CSR entries
        {
            int64_t it[3] = {lo[0], lo[1], qfull ? lo[2] : 0};
            const int nc = qfull ? 3 : 2;
            for (;;) {
                int64_t mcol = INT64_MAX;
                for (int a = 0; a < nc; ++a)
                    if (it[a] < hi[a]) mcol = std::min<int64_t>(mcol, col[it[a]]);
                if (mcol == INT64_MAX) break;
                double a3[3] = {0.0, 0.0, 0.0};
                for (int a = 0; a < nc; ++a)
                    if (it[a] < hi[a] && col[it[a]] == mcol) a3[a] = val[it[a]++];
                gcolv.push_back((int32_t)mcol);
                gxy.push_back(a3[0]);
                gxy.push_back(a3[1]);
                gzv.push_back(a3[2]);
            }
            growv[q + 1] = (int64_t)gcolv.size();
        }
        for (int a = 0; a < (qfull ? 3 : 2); ++a) {
            for (int64_t k = lo[a]; k < hi[a]; ++k) {
                ncol.push_back(col[k]);
                nval.push_back(val[k]);
            }
            nrp[rx + a + 1] = (int64_t)ncol.size();
        }
    }
    // column records pay when they replace more than 28 / 12 entries each (NPG_SPMV_COLUMN_RECORDS=0: never)
    const char *ge = getenv("NPG_SPMV_COLUMN_RECORDS");
    const bool colrec = !(ge && atoi(ge) == 0) && !gcolv.empty() && 28 * gcolv.size() <= 12 * ncol.size();
    if (colrec) {           // the block rows keep no CSR entries
        ncol.clear();
        nval.clear();
        for (int64_t r = 0; r < nbr; ++r) nrp[r + 1] = 0;
    }
    // rows behind the block rows: their entries in the block columns become coupling records {c, d_x, d_y, d_z} (spmv_device.h)
    // unless NPG_SPMV_COUPLING=0 or a row would not fit a tile that way
    const char *ce = getenv("NPG_SPMV_COUPLING");
    bool coupling = !(ce && atoi(ce) == 0) && A->m > nbr;
    std::vector<int64_t> drow;
    std::vector<int32_t> dcol;
    std::vector<double> dxy, dz;
    int64_t ndrec_real = 0;
    if (coupling) {
        drow.assign((size_t)(A->m - nbr) + 1, 0);
        for (int64_t r = nbr; r < A->m && coupling; ++r) {
            int64_t k = rp[r], nrec = 0;
            while (k < rp[r + 1] && col[k] < nbr) {
                const int64_t c0 = col[k];
                const int64_t c = c0 < nf3 ? c0 / 3 : nfull + (c0 - nf3) / 2, f = first(c);
                double d[3] = {0.0, 0.0, 0.0};
                for (; k < rp[r + 1] && col[k] < nbr && col[k] - f < (c < nfull ? 3 : 2) && col[k] >= f; ++k) d[col[k] - f] = val[k];
                dcol.push_back((int32_t)c);
                dxy.push_back(d[0]);
                dxy.push_back(d[1]);
                dz.push_back(d[2]);
                ++nrec;
            }
            ndrec_real += nrec;
            if (want_win && (nrec & 1)) {          // zero record on the row's last column node (windowed tiles take records in pairs)
                dcol.push_back(dcol.back());
                dxy.push_back(0.0);
                dxy.push_back(0.0);
                dz.push_back(0.0);
                ++nrec;
            }
            drow[r - nbr + 1] = (int64_t)dcol.size();
            if (nrec + (rp[r + 1] - k) > kTileNnz) coupling = false;
        }
    }
    for (int64_t r = nbr; r < A->m; ++r) {
        for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
            if (coupling && col[k] < nbr) continue;
            ncol.push_back(col[k]);
            nval.push_back(val[k]);
        }
        nrp[r + 1] = (int64_t)ncol.size();
    }
    // swap the device arrays; a clone shares rowptr / col / tiles with its pattern owner and must detach from them
    if (A->owns_pattern) {
        NPG_HIP(hipFree(A->col));
        NPG_HIP(hipFree(A->rowptr));
        if (A->tile_ptr) NPG_HIP(hipFree(A->tile_ptr));
    }
    NPG_HIP(hipFree(A->val));
    A->col = nullptr;
    A->val = nullptr;
    A->rowptr = nullptr;
    A->tile_ptr = nullptr;
    A->owns_pattern = true;
    NPG_HIP(hipMalloc((void **)&A->rowptr, nrp.size() * sizeof(int64_t)));
    NPG_HIP(hipMalloc((void **)&A->col, std::max<size_t>(1, ncol.size()) * sizeof(int32_t)));
    NPG_HIP(hipMalloc((void **)&A->val, std::max<size_t>(1, nval.size()) * sizeof(double)));
    NPG_HIP(hipMalloc((void **)&A->prow, prow.size() * sizeof(int64_t)));
    NPG_HIP(hipMalloc((void **)&A->pcol, std::max<size_t>(1, pcol.size()) * sizeof(int32_t)));
    NPG_HIP(hipMalloc((void **)&A->pkc, std::max<size_t>(2, pkc.size()) * sizeof(double)));
    NPG_HIP(hipMemcpy(A->rowptr, nrp.data(), nrp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    if (!ncol.empty()) {
        NPG_HIP(hipMemcpy(A->col, ncol.data(), ncol.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->val, nval.data(), nval.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    NPG_HIP(hipMemcpy(A->prow, prow.data(), prow.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    if (!pcol.empty()) {
        NPG_HIP(hipMemcpy(A->pcol, pcol.data(), pcol.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->pkc, pkc.data(), pkc.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (colrec) {
        const size_t ng = gcolv.size();
        NPG_HIP(hipMalloc((void **)&A->grow, growv.size() * sizeof(int64_t)));
        NPG_HIP(hipMalloc((void **)&A->gcol, ng * sizeof(int32_t)));
        NPG_HIP(hipMalloc((void **)&A->gval, 3 * ng * sizeof(double)));
        NPG_HIP(hipMemcpy(A->grow, growv.data(), growv.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->gcol, gcolv.data(), ng * sizeof(int32_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->gval, gxy.data(), 2 * ng * sizeof(double), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->gval + 2 * ng, gzv.data(), ng * sizeof(double), hipMemcpyHostToDevice));
        A->ngrec = (int64_t)ng;
        A->h_grow = std::move(growv);
    }
    if (coupling && !dcol.empty()) {
        const size_t nd = dcol.size();
        NPG_HIP(hipMalloc((void **)&A->drow, drow.size() * sizeof(int64_t)));
        NPG_HIP(hipMalloc((void **)&A->dcol, nd * sizeof(int32_t)));
        NPG_HIP(hipMalloc((void **)&A->dval, 3 * nd * sizeof(double)));
        NPG_HIP(hipMemcpy(A->drow, drow.data(), drow.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->dcol, dcol.data(), nd * sizeof(int32_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->dval, dxy.data(), 2 * nd * sizeof(double), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(A->dval + 2 * nd, dz.data(), nd * sizeof(double), hipMemcpyHostToDevice));
        A->ndrec = (int64_t)nd;
        A->ndrec_real = ndrec_real;
        A->h_drow = std::move(drow);
    }
    A->h_rowptr = std::move(nrp);
    A->h_prow = std::move(prow);
    A->rnnz = (int64_t)ncol.size();
    A->nfull = (int32_t)nfull;
    A->nsurf = (int32_t)nsurf;
    A->nrec_real = nrec_real;
    int rc = build_tiles(A);
    if (rc) return rc;
    if (want_win && colrec && (rc = build_window_tiles(A, pcol, gcolv, dcol, pkc, dxy, gxy, gzv, dz, ncol, nval))) return rc;
    *blocked = 1;
    return NPG_OK;
}

namespace npg {
__global__ void k_gather_map(double *__restrict__ dst, const double *__restrict__ src, const int64_t *__restrict__ map, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = map[i];
        dst[i] = k >= 0 ? src[k] : 0.0;
    }
}

int csr_repack(const npg_csr *A) {
    if (!A || !A->packed) return NPG_OK;
    npg_csr *P = A->packed;
    auto go = [&](double *dst, const int64_t *map, int64_t n) {
        if (n > 0)
            hipLaunchKernelGGL(k_gather_map, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, A->ctx->stream, dst,
                               (const double *)A->val, map, n);
    };
    go(P->pk9, P->map9, 9 * P->npk9);
    go(P->dval, P->mapd, 3 * P->ndrec);
    go(P->gval, P->mapg, 3 * P->ngrec);
    go(P->val, P->maprem, P->rnnz);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}
}  // namespace npg

// Record-form companion of a PLAIN matrix whose velocity block has no {K, C} structure (function-valued viscosity: the
// full-stress form couples all nine component pairs of a node pair, src/inversion.jl:172-181): FULL node records {c, a_00 .. a_22}
// (one gather of node c's components for nine entries), coupling records for the rows behind the block rows and column records
// for what the block rows hold outside the block columns - nothing is left as CSR entries but what the rows behind the block
// hold outside it.  The plain matrix stays what the element kernels assemble into; the companion remembers for every value it
// stores where in the plain `val` it comes from and is refreshed on the device after every change (csr_repack: assembly,
// combine, gather_values, zero_values).  Products, solves and preconditioner applications read the companion (spmv_form).
// *packed = 0 and nothing changed if the rows of some node would not fit a tile.
NPG_API int npg_csr_pack_nodes(npg_csr *A, int64_t nfull, int64_t nsurf, int *packed) {
    NPG_REQUIRE(A && packed && nfull >= 0 && nsurf >= 0, "npg_csr_pack_nodes: bad argument");
    const int64_t nf3 = 3 * nfull, nbr = nf3 + 2 * nsurf, nnode = nfull + nsurf;
    NPG_REQUIRE(nbr <= A->m && nbr <= A->n, "npg_csr_pack_nodes: more block rows than the matrix has");
    NPG_REQUIRE(A->nnode() == 0 && !A->packed, "npg_csr_pack_nodes: the matrix is in record form already");
    *packed = 0;
    if (nnode == 0) return NPG_OK;
    NPG_HIP(hipSetDevice(A->ctx->device));
    NPG_HIP(hipStreamSynchronize(A->ctx->stream));
    std::vector<int32_t> col((size_t)A->nnz);
    NPG_HIP(hipMemcpy(col.data(), A->col, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    const std::vector<int64_t> &rp = A->h_rowptr;
    auto first = [&](int64_t c) { return c < nfull ? 3 * c : nf3 + 2 * (c - nfull); };
    auto node_of = [&](int64_t c0) { return c0 < nf3 ? c0 / 3 : nfull + (c0 - nf3) / 2; };
    std::vector<int64_t> prow((size_t)nnode + 1, 0), grow((size_t)nnode + 1, 0), nrp((size_t)A->m + 1, 0);
    std::vector<int32_t> pcol, gcol, dcol, ncol;
    std::vector<int64_t> m9, mg3, md3, mrem;          // source positions, record-major (re-laid below)
    for (int64_t q = 0; q < nnode; ++q) {
        const int nc = q < nfull ? 3 : 2;
        const int64_t rx = first(q);
        int64_t it[3] = {0, 0, 0}, en[3] = {0, 0, 0};
        for (int a = 0; a < nc; ++a) {
            it[a] = rp[rx + a];
            en[a] = rp[rx + a + 1];
        }
        for (;;) {                                    // block columns, node by node (rows are sorted: block columns come first)
            int64_t cn = INT64_MAX;
            for (int a = 0; a < nc; ++a)
                if (it[a] < en[a] && col[it[a]] < nbr) cn = std::min(cn, node_of(col[it[a]]));
            if (cn == INT64_MAX) break;
            int64_t src[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
            const int64_t f = first(cn);
            for (int a = 0; a < nc; ++a)
                while (it[a] < en[a] && col[it[a]] < nbr && node_of(col[it[a]]) == cn) {
                    src[3 * a + (int)(col[it[a]] - f)] = it[a];
                    ++it[a];
                }
            pcol.push_back((int32_t)cn);
            m9.insert(m9.end(), src, src + 9);
        }
        prow[q + 1] = (int64_t)pcol.size();
        for (;;) {                                    // what is left: columns outside the block, one record per column
            int64_t mc = INT64_MAX;
            for (int a = 0; a < nc; ++a)
                if (it[a] < en[a]) mc = std::min<int64_t>(mc, col[it[a]]);
            if (mc == INT64_MAX) break;
            int64_t src[3] = {-1, -1, -1};
            for (int a = 0; a < nc; ++a)
                if (it[a] < en[a] && col[it[a]] == mc) src[a] = it[a]++;
            gcol.push_back((int32_t)mc);
            mg3.insert(mg3.end(), src, src + 3);
        }
        grow[q + 1] = (int64_t)gcol.size();
    }
    std::vector<int64_t> drow((size_t)(A->m - nbr) + 1, 0);
    for (int64_t r = nbr; r < A->m; ++r) {
        int64_t k = rp[r], nrec = 0;
        while (k < rp[r + 1] && col[k] < nbr) {
            const int64_t cn = node_of(col[k]), f = first(cn);
            int64_t src[3] = {-1, -1, -1};
            for (; k < rp[r + 1] && col[k] < nbr && node_of(col[k]) == cn; ++k) src[col[k] - f] = k;
            dcol.push_back((int32_t)cn);
            md3.insert(md3.end(), src, src + 3);
            ++nrec;
        }
        drow[r - nbr + 1] = (int64_t)dcol.size();
        for (; k < rp[r + 1]; ++k) {
            ncol.push_back(col[k]);
            mrem.push_back(k);
        }
        nrp[r + 1] = (int64_t)ncol.size();
        if (nrec + (nrp[r + 1] - nrp[r]) > kTileNnz) return NPG_OK;          // (a row that would not fit a tile: stay plain)
    }
    for (int64_t q = 0; q < nnode; ++q)
        if ((q < nfull ? 3 : 2) * ((prow[q + 1] - prow[q]) + (grow[q + 1] - grow[q])) > kTileNnz) return NPG_OK;
    // the companion object
    npg_csr *P = new npg_csr();
    P->ctx = A->ctx;
    P->m = A->m;
    P->n = A->n;
    P->nnz = A->nnz;
    P->rnnz = (int64_t)ncol.size();
    P->nfull = (int32_t)nfull;
    P->nsurf = (int32_t)nsurf;
    const int64_t n9 = (int64_t)pcol.size(), ng = (int64_t)gcol.size(), nd = (int64_t)dcol.size();
    P->npk9 = n9;
    P->ngrec = ng;
    P->ndrec = nd;
    std::vector<int64_t> map9((size_t)9 * n9), mapg((size_t)3 * ng), mapd((size_t)3 * nd);
    for (int64_t e = 0; e < n9; ++e) {           // (a_2k, a_2k+1) pairs as four 16-byte streams, a_8 as one 8-byte stream (spmv_device.h)
        for (int s9 = 0; s9 < 8; ++s9) map9[(size_t)(s9 & ~1) * n9 + 2 * e + (s9 & 1)] = m9[(size_t)9 * e + s9];
        map9[(size_t)8 * n9 + e] = m9[(size_t)9 * e + 8];
    }
    for (int64_t e = 0; e < ng; ++e) {               // [2 ng] (a_x, a_y) pairs, then [ng] a_z
        mapg[2 * e] = mg3[3 * e];
        mapg[2 * e + 1] = mg3[3 * e + 1];
        mapg[2 * ng + e] = mg3[3 * e + 2];
    }
    for (int64_t e = 0; e < nd; ++e) {
        mapd[2 * e] = md3[3 * e];
        mapd[2 * e + 1] = md3[3 * e + 1];
        mapd[2 * nd + e] = md3[3 * e + 2];
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        NPG_HIP(hipMalloc(dst, std::max<size_t>(bytes, 16)));
        if (bytes) NPG_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return NPG_OK;
    };
    int rc = NPG_OK;
    auto chk = [&](int r) { if (rc == NPG_OK) rc = r; };
    chk(up((void **)&P->rowptr, nrp.data(), nrp.size() * 8));
    chk(up((void **)&P->col, ncol.data(), ncol.size() * 4));
    chk(up((void **)&P->prow, prow.data(), prow.size() * 8));
    chk(up((void **)&P->pcol, pcol.data(), pcol.size() * 4));
    chk(up((void **)&P->grow, grow.data(), grow.size() * 8));
    chk(up((void **)&P->gcol, gcol.data(), gcol.size() * 4));
    chk(up((void **)&P->drow, drow.data(), drow.size() * 8));
    chk(up((void **)&P->dcol, dcol.data(), dcol.size() * 4));
    chk(up((void **)&P->map9, map9.data(), map9.size() * 8));
    chk(up((void **)&P->mapg, mapg.data(), mapg.size() * 8));
    chk(up((void **)&P->mapd, mapd.data(), mapd.size() * 8));
    chk(up((void **)&P->maprem, mrem.data(), mrem.size() * 8));
    if (rc == NPG_OK) {
        hipError_t e1 = hipMalloc((void **)&P->val, std::max<size_t>(16, ncol.size() * 8));
        hipError_t e2 = hipMalloc((void **)&P->pk9, std::max<size_t>(16, (size_t)9 * n9 * 8));
        hipError_t e3 = hipMalloc((void **)&P->gval, std::max<size_t>(16, (size_t)3 * ng * 8));
        hipError_t e4 = hipMalloc((void **)&P->dval, std::max<size_t>(16, (size_t)3 * nd * 8));
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
            set_error("npg_csr_pack_nodes: out of device memory");
            rc = NPG_ENOMEM;
        }
    }
    if (rc != NPG_OK) {
        npg_csr_destroy(P);
        return rc;
    }
    if (nd == 0) {                 // (no coupling records: the tile code keys on drow)
        hipFree(P->drow);
        P->drow = nullptr;
    }
    P->h_rowptr = std::move(nrp);
    P->h_prow = std::move(prow);
    P->h_grow = std::move(grow);
    P->h_drow = std::move(drow);
    P->owns_pattern = true;
    A->packed = P;
    rc = csr_repack(A);
    if (rc == NPG_OK) rc = build_tiles(P);
    if (rc != NPG_OK) {
        A->packed = nullptr;
        npg_csr_destroy(P);
        return rc;
    }
    *packed = 1;
    return NPG_OK;
}

namespace npg {
__global__ void k_perm_gather(double *__restrict__ dst, const double *__restrict__ src, const int32_t *__restrict__ perm, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[perm[i]];
}
__global__ void k_perm_scatter(double *__restrict__ dst, const double *__restrict__ src, const int32_t *__restrict__ perm, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[perm[i]] = src[i];
}
void perm_gather(const npg_csr *A, double *dst, const double *src) {
    hipLaunchKernelGGL(k_perm_gather, dim3((unsigned)std::min<int64_t>(4096, (A->m + 255) / 256)), dim3(256), 0, A->ctx->stream, dst, src,
                       (const int32_t *)A->uperm, A->m);
}
void perm_scatter(const npg_csr *A, double *dst, const double *src) {
    hipLaunchKernelGGL(k_perm_scatter, dim3((unsigned)std::min<int64_t>(4096, (A->m + 255) / 256)), dim3(256), 0, A->ctx->stream, dst, src,
                       (const int32_t *)A->uperm, A->m);
}
}  // namespace npg

// npg_csr_block_nodes for a matrix in ANY DoF order.  node_of_dof[i] >= 0: DoF i is component comp_of_dof[i] (0, 1, 2) of the
// velocity node with that label (any non-negative labels: Gridap's node ids do); < 0: not a velocity DoF (pressure).  The
// library renumbers internally - [x, y, z of every node with three components | x, y of every node with exactly those two | the
// other velocity DoFs | the rest], nodes in the order of their first DoF in the caller's numbering (so a caller's RCM locality is
// kept) - permutes the matrix on the host, blocks it, and keeps the permutation in the handle.
NPG_API int npg_csr_block_nodes_dofs(npg_csr *A, const int64_t *node_of_dof, const int32_t *comp_of_dof, double rtol, int *blocked) {
    NPG_REQUIRE(A && node_of_dof && comp_of_dof && blocked, "npg_csr_block_nodes_dofs: NULL argument");
    NPG_REQUIRE(A->m == A->n, "npg_csr_block_nodes_dofs: the matrix must be square (%lld x %lld)", (long long)A->m, (long long)A->n);
    NPG_REQUIRE(A->nnode() == 0 && !A->packed && !A->uperm, "npg_csr_block_nodes_dofs: the matrix is in record form already");
    NPG_REQUIRE(A->owns_pattern, "npg_csr_block_nodes_dofs: the matrix shares its pattern with another (npg_csr_clone)");
    *blocked = 0;
    const int64_t N = A->m;
    // nodes in the order of their first DoF; per node the DoF of each component
    std::vector<int64_t> labels;
    labels.reserve((size_t)N);
    for (int64_t i = 0; i < N; ++i)
        if (node_of_dof[i] >= 0) {
            NPG_REQUIRE(comp_of_dof[i] >= 0 && comp_of_dof[i] < 3, "npg_csr_block_nodes_dofs: component %d of DoF %lld", (int)comp_of_dof[i], (long long)i);
            labels.push_back(node_of_dof[i]);
        }
    if (labels.empty()) return NPG_OK;
    std::vector<int64_t> uniq(labels);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const int64_t nn = (int64_t)uniq.size();
    std::vector<int64_t> dof((size_t)3 * nn, -1), first((size_t)nn, INT64_MAX);
    for (int64_t i = 0; i < N; ++i)
        if (node_of_dof[i] >= 0) {
            const int64_t q = std::lower_bound(uniq.begin(), uniq.end(), node_of_dof[i]) - uniq.begin();
            NPG_REQUIRE(dof[3 * q + comp_of_dof[i]] < 0, "npg_csr_block_nodes_dofs: node %lld has two DoFs for component %d",
                        (long long)node_of_dof[i], (int)comp_of_dof[i]);
            dof[3 * q + comp_of_dof[i]] = i;
            first[q] = std::min(first[q], i);
        }
    std::vector<int64_t> order((size_t)nn);
    for (int64_t q = 0; q < nn; ++q) order[q] = q;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return first[a] < first[b]; });
    std::vector<int32_t> perm;          // internal -> caller
    perm.reserve((size_t)N);
    std::vector<char> used((size_t)N, 0);
    int64_t nfull = 0, nsurf = 0;
    for (int64_t q : order)
        if (dof[3 * q] >= 0 && dof[3 * q + 1] >= 0 && dof[3 * q + 2] >= 0) {
            for (int a = 0; a < 3; ++a) {
                perm.push_back((int32_t)dof[3 * q + a]);
                used[dof[3 * q + a]] = 1;
            }
            ++nfull;
        }
    for (int64_t q : order)
        if (dof[3 * q] >= 0 && dof[3 * q + 1] >= 0 && dof[3 * q + 2] < 0) {
            for (int a = 0; a < 2; ++a) {
                perm.push_back((int32_t)dof[3 * q + a]);
                used[dof[3 * q + a]] = 1;
            }
            ++nsurf;
        }
    if (nfull + nsurf == 0) return NPG_OK;
    for (int64_t i = 0; i < N; ++i)
        if (!used[i]) perm.push_back((int32_t)i);
    std::vector<int32_t> iperm((size_t)N);
    for (int64_t i = 0; i < N; ++i) iperm[perm[i]] = (int32_t)i;
    // A' = A[perm, perm] on the host, rows sorted
    NPG_HIP(hipSetDevice(A->ctx->device));
    NPG_HIP(hipStreamSynchronize(A->ctx->stream));
    // (what the handle will carry is allocated first: a failure here leaves the caller's matrix as it was)
    int32_t *d_perm = nullptr;
    double *d_vec[3] = {nullptr, nullptr, nullptr};
    auto release = [&]() {
        if (d_perm) hipFree(d_perm);
        for (double *q : d_vec)
            if (q) hipFree(q);
    };
    {
        hipError_t e0 = hipMalloc((void **)&d_perm, (size_t)N * sizeof(int32_t));
        for (int k = 0; k < 3 && e0 == hipSuccess; ++k) e0 = hipMalloc((void **)&d_vec[k], (size_t)N * sizeof(double));
        if (e0 == hipSuccess) e0 = hipMemcpy(d_perm, perm.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e0 != hipSuccess) {
            release();
            set_error("npg_csr_block_nodes_dofs: out of device memory");
            return NPG_ENOMEM;
        }
    }
    std::vector<int32_t> col((size_t)A->nnz), ncol((size_t)A->nnz);
    std::vector<double> val((size_t)A->nnz), nval((size_t)A->nnz);
    NPG_HIP(hipMemcpy(col.data(), A->col, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    NPG_HIP(hipMemcpy(val.data(), A->val, val.size() * sizeof(double), hipMemcpyDeviceToHost));
    const std::vector<int64_t> orp = A->h_rowptr;
    std::vector<int64_t> nrp((size_t)N + 1, 0);
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t i = 0; i < N; ++i) {
        const int64_t r = perm[i];
        row.clear();
        // (exact zeros are not carried over: Gridap stores structural zeros - 32 % of a constant-viscosity A_inversion - and a
        //  caller that uploads as the reference does, CuSparseMatrixCSR(a) of ext/nuPGCMCUDAExt.jl:27, hands them over; the record
        //  form has no use for them)
        for (int64_t k = orp[r]; k < orp[r + 1]; ++k)
            if (val[k] != 0.0) row.emplace_back(iperm[col[k]], val[k]);
        std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) { return a.first < b.first; });
        int64_t o = nrp[i];
        for (const auto &e : row) {
            ncol[o] = e.first;
            nval[o++] = e.second;
        }
        nrp[i + 1] = o;
    }
    auto put = [&](const std::vector<int64_t> &rp, const std::vector<int32_t> &c, const std::vector<double> &v) -> int {
        NPG_HIP(hipMemcpy(A->rowptr, rp.data(), rp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        if (!c.empty()) {
            NPG_HIP(hipMemcpy(A->col, c.data(), c.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            NPG_HIP(hipMemcpy(A->val, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        A->h_rowptr = rp;
        return build_tiles(A);
    };
    const int64_t nnz_in = A->nnz;
    A->nnz = nrp[(size_t)N];                      // (<= nnz_in: the arrays keep their size)
    ncol.resize((size_t)A->nnz);
    nval.resize((size_t)A->nnz);
    int rc = put(nrp, ncol, nval);
    if (rc == NPG_OK) rc = npg_csr_block_nodes(A, nfull, nsurf, rtol, blocked);
    if (rc != NPG_OK || !*blocked) {              // not the structure (or an error): the caller's matrix again, untouched
        if (A->nnode() == 0) {
            A->nnz = nnz_in;
            const int rc2 = put(orp, col, val);
            if (rc == NPG_OK) rc = rc2;
        }
        release();
        return rc;
    }
    A->uperm = d_perm;
    for (int k = 0; k < 3; ++k) A->uvec[k] = d_vec[k];
    return NPG_OK;
}

// Ghost nodes of a rank's row block (columns [m, n) are the ghosts): node g's components are the ghost columns
// [first_col[g], first_col[g] + ncomp[g]), ncomp 3 or 2.  Call BEFORE npg_csr_block_nodes: its windowed tile set then stores
// owned-ghost node couplings as node records (common.h, npg_csr::gn_col).  The caller orders a rank's ghosts so that a node's
// components are adjacent (ascending global ids within an owner do that in the node-block numbering).
NPG_API int npg_csr_set_ghost_nodes(npg_csr *A, int64_t n_nodes, const int32_t *first_col, const int32_t *ncomp) {
    NPG_REQUIRE(A && n_nodes >= 0 && (n_nodes == 0 || (first_col && ncomp)), "npg_csr_set_ghost_nodes: bad argument");
    NPG_REQUIRE(A->nnode() == 0 && !A->packed, "npg_csr_set_ghost_nodes: call before npg_csr_block_nodes");
    std::vector<char> seen((size_t)std::max<int64_t>(A->n - A->m, 0), 0);
    for (int64_t g = 0; g < n_nodes; ++g) {
        NPG_REQUIRE(ncomp[g] == 2 || ncomp[g] == 3, "npg_csr_set_ghost_nodes: node %lld has %d components", (long long)g, (int)ncomp[g]);
        NPG_REQUIRE(first_col[g] >= A->m && (int64_t)first_col[g] + ncomp[g] <= A->n, "npg_csr_set_ghost_nodes: node %lld is not in the ghost columns",
                    (long long)g);
        for (int a = 0; a < ncomp[g]; ++a) {
            NPG_REQUIRE(!seen[(size_t)(first_col[g] - A->m + a)], "npg_csr_set_ghost_nodes: ghost column %d named twice", first_col[g] + a);
            seen[(size_t)(first_col[g] - A->m + a)] = 1;
        }
    }
    A->gn_col.assign(first_col, first_col + n_nodes);
    A->gn_ncomp.assign(ncomp, ncomp + n_nodes);
    return NPG_OK;
}

// the (x, y)-only special case kept for callers that interleave two components: rows 2q, 2q+1 for q < npairs
NPG_API int npg_csr_pair_xy(npg_csr *A, int64_t npairs, double rtol, int *paired) {
    return npg_csr_block_nodes(A, 0, npairs, rtol, paired);
}

NPG_API int npg_csr_create_from_csc(npg_ctx *ctx, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                                    const double *nzval, int drop_zeros, npg_csr **out) {
    NPG_REQUIRE(ctx && colptr && out && m >= 0 && n >= 0, "npg_csr_create_from_csc: bad argument");
    NPG_REQUIRE(m < INT32_MAX && n < INT32_MAX, "npg_csr_create_from_csc: dimensions exceed int32 indices");
    const int64_t nnz_in = colptr[n];
    NPG_REQUIRE(nnz_in == 0 || (rowval && nzval), "npg_csr_create_from_csc: NULL arrays");
    std::vector<int64_t> rowptr((size_t)m + 1, 0);
    for (int64_t j = 0; j < n; ++j) {
        NPG_REQUIRE(colptr[j + 1] >= colptr[j], "npg_csr_create_from_csc: colptr not monotone at %lld", (long long)j);
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            NPG_REQUIRE(rowval[k] >= 0 && rowval[k] < m, "npg_csr_create_from_csc: row index out of range");
            if (drop_zeros && nzval[k] == 0.0) continue;
            ++rowptr[(size_t)rowval[k] + 1];
        }
    }
    for (int64_t i = 0; i < m; ++i) rowptr[i + 1] += rowptr[i];
    const int64_t nnz = rowptr[m];
    std::vector<int32_t> col((size_t)nnz);
    std::vector<double> val((size_t)nnz);
    std::vector<int64_t> next(rowptr.begin(), rowptr.end() - 1);
    for (int64_t j = 0; j < n; ++j)        // columns ascending => each CSR row ends up sorted by column
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            if (drop_zeros && nzval[k] == 0.0) continue;
            const int64_t p = next[rowval[k]]++;
            col[p] = (int32_t)j;
            val[p] = nzval[k];
        }
    return upload_csr(ctx, m, n, std::move(rowptr), col, val, out);
}

NPG_API int npg_csr_create(npg_ctx *ctx, int64_t m, int64_t n, const int64_t *rowptr, const int32_t *colind,
                           const double *val, npg_csr **out) {
    NPG_REQUIRE(ctx && rowptr && out && m >= 0 && n >= 0, "npg_csr_create: bad argument");
    NPG_REQUIRE(m < INT32_MAX && n < INT32_MAX, "npg_csr_create: dimensions exceed int32 indices");
    NPG_REQUIRE(rowptr[0] == 0, "npg_csr_create: rowptr must be 0-based");
    const int64_t nnz = rowptr[m];
    for (int64_t i = 0; i < m; ++i)
        NPG_REQUIRE(rowptr[i + 1] >= rowptr[i], "npg_csr_create: rowptr not monotone at %lld", (long long)i);
    for (int64_t k = 0; k < nnz; ++k)
        NPG_REQUIRE(colind[k] >= 0 && colind[k] < n, "npg_csr_create: column index out of range at %lld", (long long)k);
    std::vector<int64_t> rp(rowptr, rowptr + m + 1);
    std::vector<int32_t> col(colind, colind + nnz);
    std::vector<double> v((size_t)nnz, 0.0);
    if (val) std::copy(val, val + nnz, v.begin());
    return upload_csr(ctx, m, n, std::move(rp), col, v, out);
}

NPG_API int npg_csr_destroy(npg_csr *A) {
    if (!A) return NPG_OK;
    hipStreamSynchronize(A->ctx->stream);
    if (A->owns_pattern) {
        if (A->rowptr) hipFree(A->rowptr);
        if (A->col) hipFree(A->col);
        if (A->tile_ptr) hipFree(A->tile_ptr);
    }
    if (A->val) hipFree(A->val);
    if (A->prow) hipFree(A->prow);
    if (A->pcol) hipFree(A->pcol);
    if (A->pkc) hipFree(A->pkc);
    if (A->val32) hipFree(A->val32);
    if (A->pkc32) hipFree(A->pkc32);
    if (A->packed) npg_csr_destroy(A->packed);
    if (A->pk9) hipFree(A->pk9);
    if (A->pk9_32) hipFree(A->pk9_32);
    for (int64_t *mp : {A->map9, A->mapd, A->mapg, A->maprem})
        if (mp) hipFree(mp);
    if (A->grow) hipFree(A->grow);
    if (A->gcol) hipFree(A->gcol);
    if (A->gval) hipFree(A->gval);
    if (A->gval32) hipFree(A->gval32);
    if (A->drow) hipFree(A->drow);
    if (A->dcol) hipFree(A->dcol);
    if (A->dval) hipFree(A->dval);
    if (A->dval32) hipFree(A->dval32);
    free_window_tiles(A);
    for (void *q : {(void *)A->lb_ptr, (void *)A->lb_dofs, (void *)A->lb_off, (void *)A->lb_val, (void *)A->lb_val32, (void *)A->lb_val16, (void *)A->lb_scale})
        if (q) hipFree(q);
    if (A->uperm) hipFree(A->uperm);
    for (double *p : A->uvec)
        if (p) hipFree(p);
    delete A;
    return NPG_OK;
}

NPG_API int npg_csr_shape(const npg_csr *A, int64_t *m, int64_t *n, int64_t *nnz) {
    NPG_REQUIRE(A, "npg_csr_shape: NULL matrix");
    if (m) *m = A->m;
    if (n) *n = A->n;
    if (nnz) *nnz = A->nnz;
    return NPG_OK;
}

NPG_API int npg_csr_storage(const npg_csr *A, int64_t *npairs, int64_t *paired_records, int64_t *csr_entries) {
    NPG_REQUIRE(A, "npg_csr_storage: NULL matrix");
    if (npairs) *npairs = A->nnode();
    if (paired_records) *paired_records = A->nnode() ? (A->pk9 ? A->h_prow[A->nnode()] : A->nrec_real) : 0;      // (without zero-record padding)
    if (csr_entries) *csr_entries = A->rnnz;
    return NPG_OK;
}

NPG_API int npg_csr_spmv_bytes(const npg_csr *Ap, int64_t *matrix_bytes) {
    NPG_REQUIRE(Ap && matrix_bytes, "npg_csr_spmv_bytes: NULL argument");
    const npg_csr *A = spmv_form(Ap);
    const int64_t nnode = A->nnode(), nrec = nnode ? A->h_prow[nnode] : 0;
    int64_t b = 8 * (A->m + 1) + 12 * A->rnnz;
    if (nnode) b += 8 * (nnode + 1) + 4 * nrec + (A->pk9 ? 72 : 16) * nrec;
    if (A->grow) b += 8 * (nnode + 1) + 28 * A->ngrec;
    if (A->drow) b += 8 * (A->m - A->block_rows() + 1) + 28 * A->ndrec;
    *matrix_bytes = b;
    return NPG_OK;
}

NPG_API int npg_csr_window_info(const npg_csr *Ap, int64_t *tiles, int64_t *block_tiles, int64_t *distinct, int64_t *matrix_bytes) {
    NPG_REQUIRE(Ap, "npg_csr_window_info: NULL matrix");
    const npg_csr *A = spmv_form(Ap);
    int64_t nblk = 0;
    if (A->wtile_ptr) {
        // (block tiles are the descriptors with a window: count them from the lists' owner - every block tile has nw > 0)
        const int64_t nbr = A->block_rows();
        int64_t behind = A->nwrow_tiles;
        if (!behind)
            for (const npg::TileDesc &t : A->h_tiles) behind += t.r0 >= nbr;
        nblk = A->nwtiles - behind;
    }
    if (tiles) *tiles = A->nwtiles;
    if (block_tiles) *block_tiles = nblk;
    if (distinct) *distinct = A->nwlist + A->nvlist;
    if (matrix_bytes) {
        int64_t b = 0;
        if (A->wtile_ptr) {
            const int64_t nnode = A->nnode(), nrec = A->nw_rec ? A->nw_rec : A->h_prow[nnode], ngr = A->nw_rec ? A->nw_grec : A->ngrec;
            b = 48 * (int64_t)A->nwtiles + 18 * nrec + 26 * ngr + 4 * (A->nwlist + A->nvlist) + 8 * nnode +
                12 * A->rnnz + 8 * (A->m - A->block_rows() + 1);
            if (A->drow) b += A->nwrow_tiles ? 4 * (A->m - A->block_rows()) + 26 * (A->nw_drec ? A->nw_drec : A->ndrec) : 8 * (A->m - A->block_rows() + 1) + 28 * A->ndrec;
        }
        *matrix_bytes = b;
    }
    return NPG_OK;
}

NPG_API int npg_csr_coupling_records(const npg_csr *A, int64_t *records) {
    NPG_REQUIRE(A && records, "npg_csr_coupling_records: NULL argument");
    *records = (A->pk9 || !A->drow ? A->ndrec : A->ndrec_real) + A->ngrec;      // (without zero-record padding)
    return NPG_OK;
}

NPG_API int npg_csr_download(const npg_csr *A, int64_t *rowptr, int32_t *colind, double *val) {
    NPG_REQUIRE(A, "npg_csr_download: NULL matrix");
    NPG_REQUIRE(A->nnode() == 0, "npg_csr_download: the matrix is stored by node blocks; download it before npg_csr_block_nodes");
    NPG_HIP(hipStreamSynchronize(A->ctx->stream));
    if (rowptr) std::copy(A->h_rowptr.begin(), A->h_rowptr.end(), rowptr);
    if (colind && A->nnz) NPG_HIP(hipMemcpy(colind, A->col, (size_t)A->nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (val && A->nnz) NPG_HIP(hipMemcpy(val, A->val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost));
    return NPG_OK;
}

NPG_API int npg_csr_to_csc(const npg_csr *A, int64_t *colptr, int64_t *rowval, double *nzval) {
    NPG_REQUIRE(A && colptr && rowval && nzval, "npg_csr_to_csc: NULL argument");
    std::vector<int32_t> col((size_t)A->nnz);
    std::vector<double> val((size_t)A->nnz);
    int rc = npg_csr_download(A, nullptr, col.data(), val.data());
    if (rc) return rc;
    std::fill(colptr, colptr + A->n + 1, 0);
    for (int64_t k = 0; k < A->nnz; ++k) ++colptr[col[k] + 1];
    for (int64_t j = 0; j < A->n; ++j) colptr[j + 1] += colptr[j];
    std::vector<int64_t> next(colptr, colptr + A->n);
    for (int64_t i = 0; i < A->m; ++i)
        for (int64_t k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; ++k) {
            const int64_t p = next[col[k]]++;
            rowval[p] = i;
            nzval[p] = val[k];
        }
    return NPG_OK;
}

NPG_API int npg_csr_clone(const npg_csr *A, npg_csr **out) {
    NPG_REQUIRE(A && out, "npg_csr_clone: NULL argument");
    NPG_REQUIRE(A->nnode() == 0, "npg_csr_clone: the matrix is stored by node blocks");
    npg_csr *B = new npg_csr();
    B->ctx = A->ctx;
    B->m = A->m;
    B->n = A->n;
    B->nnz = A->nnz;
    B->rnnz = A->rnnz;
    B->rowptr = A->rowptr;        // pattern is shared; the original must outlive the clone
    B->col = A->col;
    B->tile_ptr = A->tile_ptr;
    B->ntiles = A->ntiles;
    B->ntiles_interior = A->ntiles_interior;
    B->lanes = A->lanes;
    B->owns_pattern = false;
    B->h_rowptr = A->h_rowptr;
    NPG_HIP(hipMalloc((void **)&B->val, std::max<size_t>(1, (size_t)A->nnz) * sizeof(double)));
    NPG_HIP(hipMemcpyAsync(B->val, A->val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToDevice, A->ctx->stream));
    *out = B;
    return NPG_OK;
}

// After npg_csr_block_nodes(_dofs) `val` holds only the CSR remainder of the record form (rnnz <= nnz entries) and, with _dofs, rows
// and columns are in the library's internal order: entry points that treat `val` as nnz plain CSR entries refuse such a matrix.
#define NPG_REQUIRE_PLAIN(A, who)                                                                                              \
    NPG_REQUIRE((A)->nnode() == 0 && !(A)->uperm, who ": the matrix is stored by node records (npg_csr_block_nodes%s): its value " \
                "array is not nnz plain CSR entries", (A)->uperm ? "_dofs, internal renumbering" : "")

NPG_API int npg_csr_zero_values(npg_csr *A) {
    NPG_REQUIRE(A, "npg_csr_zero_values: NULL matrix");
    NPG_REQUIRE_PLAIN(A, "npg_csr_zero_values");
    NPG_REQUIRE_NO_PACKS(A, "npg_csr_zero_values");
    NPG_HIP(hipMemsetAsync(A->val, 0, (size_t)A->nnz * sizeof(double), A->ctx->stream));
    return csr_repack(A);
}

NPG_API int npg_csr_combine(npg_csr *out, double a, const npg_csr *X, double b, const npg_csr *Y, const npg_csr *Z) {
    NPG_REQUIRE(out && X && Y && Z, "npg_csr_combine: NULL argument");
    NPG_REQUIRE(out->nnz == X->nnz && X->nnz == Y->nnz && Y->nnz == Z->nnz && out->m == X->m && X->m == Y->m &&
                    Y->m == Z->m,
                "npg_csr_combine: operands must share one sparsity pattern");
    NPG_REQUIRE_PLAIN(out, "npg_csr_combine");
    NPG_REQUIRE_NO_PACKS(out, "npg_csr_combine");
    NPG_REQUIRE_PLAIN(X, "npg_csr_combine");
    NPG_REQUIRE_PLAIN(Y, "npg_csr_combine");
    NPG_REQUIRE_PLAIN(Z, "npg_csr_combine");
    const int grid = (int)std::min<int64_t>(2048, (out->nnz + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_combine, dim3(std::max(grid, 1)), dim3(kBlock), 0, out->ctx->stream, out->val, a, X->val, b,
                       Y->val, Z->val, out->nnz);
    NPG_HIP(hipGetLastError());
    return csr_repack(out);
}

// ---- device-side pieces of the multigrid set-up (mg.hip): node-block inverse and S = D Dinv G on fixed patterns --------------
__device__ __forceinline__ double csr_entry(const int64_t *rowptr, const int32_t *col, const double *val, int64_t row, int32_t c) {
    int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
    while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = col[mid];
        if (v == c) return val[mid];
        if (v < c) lo = mid + 1; else hi = mid - 1;
    }
    return 0.0;
}

// Dinv = inverse of the node-block diagonal of A[0:nu, 0:nu]: 3 x 3 blocks for the first n_full nodes, 2 x 2 for the next
// n_surf, 1 x 1 for the rest.  Dinv's pattern holds exactly those blocks (sorted rows), so entry (r, c) of node block q sits
// at rowptr[r] + (c - first(q)).  One thread per node.
__global__ void k_node_block_inverse(const int64_t *__restrict__ arp, const int32_t *__restrict__ acol,
                                     const double *__restrict__ aval, int64_t n_full, int64_t n_surf, int64_t nu,
                                     const int64_t *__restrict__ drp, double *__restrict__ dval) {
    const int64_t nnode = n_full + n_surf + (nu - 3 * n_full - 2 * n_surf);
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnode; q += (int64_t)gridDim.x * blockDim.x) {
        int sz;
        int64_t r0;
        if (q < n_full) { sz = 3; r0 = 3 * q; }
        else if (q < n_full + n_surf) { sz = 2; r0 = 3 * n_full + 2 * (q - n_full); }
        else { sz = 1; r0 = 3 * n_full + 2 * n_surf + (q - n_full - n_surf); }
        double a[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int i = 0; i < sz; ++i)
            for (int j = 0; j < sz; ++j) a[i][j] = csr_entry(arp, acol, aval, r0 + i, (int32_t)(r0 + j));
        double inv[3][3];
        if (sz == 1) {
            inv[0][0] = 1.0 / a[0][0];
        } else if (sz == 2) {
            const double det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
            inv[0][0] = a[1][1] / det; inv[0][1] = -a[0][1] / det;
            inv[1][0] = -a[1][0] / det; inv[1][1] = a[0][0] / det;
        } else {
            const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                         c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
            const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
            inv[0][0] = c00 / det; inv[1][0] = c01 / det; inv[2][0] = c02 / det;
            inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
            inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det;
            inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
            inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det;
            inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
            inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
        }
        for (int i = 0; i < sz; ++i)
            for (int j = 0; j < sz; ++j) dval[drp[r0 + i] + j] = inv[i][j];
    }
}

// Dinv = inverse of a block diagonal of A[0:nu, 0:nu] whose blocks are arbitrary index sets (the velocity unknowns of a vertical
// line of nodes: the z-line smoother of mg.hip): block b = dofs[bp[b] .. bp[b+1]), ascending; Dinv's pattern holds exactly the blocks
// (row i: all unknowns of its block, ascending), so entry (dofs[bp[b] + i], dofs[bp[b] + j]) sits at drp[row] + j.  One workgroup
// per block: the block is collected from A's rows into LDS, inverted in place by Gauss-Jordan elimination WITHOUT pivoting (the
// velocity block's symmetric part is positive definite - friction - and the Coriolis part skew: every leading minor is regular),
// and written out.  *bad counts blocks that met a zero or non-finite pivot.
__global__ void __launch_bounds__(256) k_line_block_inverse(const int64_t *__restrict__ arp, const int32_t *__restrict__ acol,
                                                            const double *__restrict__ aval, const int64_t *__restrict__ bp,
                                                            const int64_t *__restrict__ dofs, int64_t nblocks,
                                                            const int64_t *__restrict__ drp, double *__restrict__ dval,
                                                            const int64_t *__restrict__ boff, double *__restrict__ dense,
                                                            float *__restrict__ dense32, _Float16 *__restrict__ dense16,
                                                            double *__restrict__ scale, int *bad) {
    extern __shared__ double lds[];
    __shared__ int flag;
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int64_t b0 = bp[b];
        const int n = (int)(bp[b + 1] - b0);
        double *M = lds;                                         // n x n, row-major
        int32_t *ids = reinterpret_cast<int32_t *>(lds + (size_t)n * n);
        __syncthreads();
        for (int e = threadIdx.x; e < n * n; e += blockDim.x) M[e] = 0.0;
        for (int e = threadIdx.x; e < n; e += blockDim.x) ids[e] = (int32_t)dofs[b0 + e];
        if (threadIdx.x == 0) flag = 0;
        __syncthreads();
        // row i of the block <- the entries of A's row ids[i] whose column is in the block (16 lanes per row)
        for (int i = threadIdx.x / 16; i < n; i += blockDim.x / 16) {
            const int64_t r = ids[i];
            for (int64_t k = arp[r] + (threadIdx.x & 15); k < arp[r + 1]; k += 16) {
                const int32_t c = acol[k];
                int lo = 0, hi = n - 1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int32_t v = ids[mid];
                    if (v == c) { M[i * n + mid] = aval[k]; break; }
                    if (v < c) lo = mid + 1; else hi = mid - 1;
                }
            }
        }
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            const double piv = M[k * n + k];
            if (threadIdx.x == 0 && !(fabs(piv) > 0.0 && fabs(piv) < 1e300)) flag = 1;
            __syncthreads();
            const double ip = 1.0 / piv;
            // row k <- row k / pivot with the pivot's own column replaced by the identity's (in-place Gauss-Jordan)
            for (int j = threadIdx.x; j < n; j += blockDim.x) M[k * n + j] = (j == k) ? ip : M[k * n + j] * ip;
            __syncthreads();
            for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
                const int i = e / n, j = e - i * n;
                if (i == k) continue;
                const double f = M[i * n + k];               // (column k of the other rows is read before any of it is rewritten:
                if (j == k) continue;                        //  those entries are finished in the pass below)
                M[e] -= f * M[k * n + j];
            }
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += blockDim.x)
                if (i != k) M[i * n + k] = -M[i * n + k] * ip;
            __syncthreads();
        }
        const int64_t o = boff[b];
        // column scales of the fp16 pack: the largest magnitude of every column of the inverse
        for (int c = threadIdx.x; c < n; c += blockDim.x) {
            double mx = 0.0;
            for (int r = 0; r < n; ++r) mx = fmax(mx, fabs(M[r * n + c]));
            scale[ids[c]] = mx > 0.0 ? mx : 1.0;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
            const int i = e / n, j = e - i * n;
            dval[drp[ids[i]] + j] = M[e];
            const double t = M[j * n + i];                    // dense packs: column-major, entry (row j, column i) at o + i n + j
            dense[o + e] = t;
            dense32[o + e] = (float)t;
            dense16[o + e] = (_Float16)(float)(t / scale[ids[i]]);
        }
        if (threadIdx.x == 0 && flag) atomicAdd(bad, 1);
    }
}

// S = D Dinv G into S's fixed pattern: one thread per row, the row's slots found by binary search; products outside the
// pattern are counted in *missing
__global__ void k_triple_product(const int64_t *__restrict__ drp, const int32_t *__restrict__ dcol, const double *__restrict__ dval,
                                 const int64_t *__restrict__ irp, const int32_t *__restrict__ icol, const double *__restrict__ ival,
                                 const int64_t *__restrict__ grp, const int32_t *__restrict__ gcol, const double *__restrict__ gval,
                                 int64_t m, const int64_t *__restrict__ srp, const int32_t *__restrict__ scol,
                                 double *__restrict__ sval, int *missing) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s0 = srp[i], s1 = srp[i + 1];
        for (int64_t k = s0; k < s1; ++k) sval[k] = 0.0;
        for (int64_t kd = drp[i]; kd < drp[i + 1]; ++kd) {
            const int32_t a = dcol[kd];
            const double dv = dval[kd];
            for (int64_t ki = irp[a]; ki < irp[a + 1]; ++ki) {
                const int32_t b = icol[ki];
                const double w = dv * ival[ki];
                for (int64_t kg = grp[b]; kg < grp[b + 1]; ++kg) {
                    const int32_t j = gcol[kg];
                    int64_t lo = s0, hi = s1 - 1, slot = -1;
                    while (lo <= hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        const int32_t v = scol[mid];
                        if (v == j) { slot = mid; break; }
                        if (v < j) lo = mid + 1; else hi = mid - 1;
                    }
                    if (slot >= 0) sval[slot] += w * gval[kg];
                    else atomicAdd(missing, 1);
                }
            }
        }
    }
}

// C = A B into C's fixed pattern (one thread per row; products outside the pattern are counted in *missing)
__global__ void k_fixed_product(const int64_t *__restrict__ arp, const int32_t *__restrict__ acol, const double *__restrict__ aval,
                                const int64_t *__restrict__ brp, const int32_t *__restrict__ bcol, const double *__restrict__ bval,
                                int64_t m, const int64_t *__restrict__ crp, const int32_t *__restrict__ ccol,
                                double *__restrict__ cval, int *missing) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s0 = crp[i], s1 = crp[i + 1];
        for (int64_t k = s0; k < s1; ++k) cval[k] = 0.0;
        for (int64_t ka = arp[i]; ka < arp[i + 1]; ++ka) {
            const int32_t a = acol[ka];
            const double w = aval[ka];
            for (int64_t kb = brp[a]; kb < brp[a + 1]; ++kb) {
                const int32_t j = bcol[kb];
                int64_t lo = s0, hi = s1 - 1, slot = -1;
                while (lo <= hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    const int32_t v = ccol[mid];
                    if (v == j) { slot = mid; break; }
                    if (v < j) lo = mid + 1; else hi = mid - 1;
                }
                if (slot >= 0) cval[slot] += w * bval[kb];
                else atomicAdd(missing, 1);
            }
        }
    }
}

struct npg_index {
    npg_ctx *ctx = nullptr;
    int64_t n = 0, bound = 0;
    int64_t *d = nullptr;
};

NPG_API int npg_index_create(npg_ctx *ctx, int64_t n, const int64_t *host, int64_t bound, npg_index **out) {
    NPG_REQUIRE(ctx && out && n >= 0 && (n == 0 || host), "npg_index_create: bad argument");
    for (int64_t i = 0; i < n; ++i)
        NPG_REQUIRE(host[i] >= 0 && host[i] < bound, "npg_index_create: entry %lld = %lld outside [0, %lld)", (long long)i,
                    (long long)host[i], (long long)bound);
    npg_index *ix = new npg_index();
    ix->ctx = ctx;
    ix->n = n;
    ix->bound = bound;
    NPG_HIP(hipSetDevice(ctx->device));
    NPG_HIP(hipMalloc((void **)&ix->d, std::max<size_t>(1, (size_t)n) * sizeof(int64_t)));
    if (n) NPG_HIP(hipMemcpy(ix->d, host, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    *out = ix;
    return NPG_OK;
}

NPG_API int npg_index_destroy(npg_index *ix) {
    if (!ix) return NPG_OK;
    hipStreamSynchronize(ix->ctx->stream);
    hipFree(ix->d);
    delete ix;
    return NPG_OK;
}

NPG_API int npg_csr_gather_values(npg_csr *dst, const npg_csr *src, const npg_index *map) {
    NPG_REQUIRE(dst && src && map, "npg_csr_gather_values: NULL argument");
    NPG_REQUIRE_NO_PACKS(dst, "npg_csr_gather_values");
    NPG_REQUIRE(dst->nnode() == 0 && src->nnode() == 0, "npg_csr_gather_values: node-blocked matrices are not supported");
    NPG_REQUIRE(map->n == dst->nnz && map->bound == src->nnz,
                "npg_csr_gather_values: the map has %lld entries into %lld values, the matrices have %lld and %lld",
                (long long)map->n, (long long)map->bound, (long long)dst->nnz, (long long)src->nnz);
    const int grid = (int)std::min<int64_t>(2048, (dst->nnz + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_gather_values, dim3(std::max(grid, 1)), dim3(kBlock), 0, dst->ctx->stream, dst->val, src->val,
                       map->d, dst->nnz);
    NPG_HIP(hipGetLastError());
    return csr_repack(dst);
}

// the stored values of a plain-CSR matrix as a vector and back: what lets an ordinary halo plan move matrix VALUES between ranks
// (the distributed multigrid's rows of T = Dinv G that belong to a neighbour's ghost unknowns, partition.py)
__global__ void k_scatter_from_vec(double *__restrict__ dst, const double *__restrict__ v, const int64_t *__restrict__ map, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v[map[i]];
}
NPG_API int npg_csr_values_to_vec(const npg_csr *A, npg_vec *v) {
    NPG_REQUIRE(A && v, "npg_csr_values_to_vec: NULL argument");
    NPG_REQUIRE(A->nnode() == 0 && !A->uperm, "npg_csr_values_to_vec: a plain-CSR matrix is required");
    NPG_REQUIRE(v->n >= A->nnz, "npg_csr_values_to_vec: the vector has %lld entries, the matrix %lld values", (long long)v->n, (long long)A->nnz);
    NPG_HIP(hipMemcpyAsync(v->d, A->val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToDevice, A->ctx->stream));
    return NPG_OK;
}
NPG_API int npg_csr_values_from_vec(npg_csr *A, const npg_vec *v, const npg_index *map) {
    NPG_REQUIRE(A && v && map, "npg_csr_values_from_vec: NULL argument");
    NPG_REQUIRE_NO_PACKS(A, "npg_csr_values_from_vec");
    NPG_REQUIRE(A->nnode() == 0 && !A->uperm, "npg_csr_values_from_vec: a plain-CSR matrix is required");
    NPG_REQUIRE(map->n == A->nnz && map->bound <= v->n, "npg_csr_values_from_vec: the map has %lld entries below %lld, the matrix %lld values, the vector %lld entries",
                (long long)map->n, (long long)map->bound, (long long)A->nnz, (long long)v->n);
    const int grid = (int)std::min<int64_t>(2048, (A->nnz + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_scatter_from_vec, dim3(std::max(grid, 1)), dim3(kBlock), 0, A->ctx->stream, A->val, (const double *)v->d, (const int64_t *)map->d, A->nnz);
    NPG_HIP(hipGetLastError());
    return csr_repack(A);
}

NPG_API int npg_csr_node_block_inverse(npg_csr *Dinv, const npg_csr *A, int64_t n_full, int64_t n_surf) {
    NPG_REQUIRE(Dinv && A && n_full >= 0 && n_surf >= 0, "npg_csr_node_block_inverse: bad argument");
    NPG_REQUIRE_NO_PACKS(Dinv, "npg_csr_node_block_inverse");
    NPG_REQUIRE(A->nnode() == 0 && Dinv->nnode() == 0, "npg_csr_node_block_inverse: node-blocked matrices are not supported");
    const int64_t nu = Dinv->m;
    NPG_REQUIRE(Dinv->n == nu && A->m >= nu && A->n >= nu && 3 * n_full + 2 * n_surf <= nu,
                "npg_csr_node_block_inverse: Dinv must be nu x nu with nu <= size(A) and 3 n_full + 2 n_surf <= nu");
    NPG_REQUIRE(Dinv->nnz == 9 * n_full + 4 * n_surf + (nu - 3 * n_full - 2 * n_surf),
                "npg_csr_node_block_inverse: Dinv's pattern must hold exactly the node blocks (%lld entries expected, %lld found)",
                (long long)(9 * n_full + 4 * n_surf + (nu - 3 * n_full - 2 * n_surf)), (long long)Dinv->nnz);
    const int64_t nnode = n_full + n_surf + (nu - 3 * n_full - 2 * n_surf);
    const int grid = (int)std::min<int64_t>(2048, (nnode + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_node_block_inverse, dim3(std::max(grid, 1)), dim3(kBlock), 0, A->ctx->stream, A->rowptr, A->col, A->val,
                       n_full, n_surf, nu, Dinv->rowptr, Dinv->val);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

// S = D Dinv G for a LINE-block Dinv (npg_csr_line_block_inverse) in two passes that never touch a product twice:
//   W_l = B_l G[l, :]   per line l: its dense inverse block times its rows of G, as a dense n_l x m_l array over the line's distinct
//                        pressure columns wcols[wptr[l] ..) (one workgroup per line, W_l accumulated in LDS);
//   S[p, :] = sum over the lines l that row p of D touches of D[p, l] W_l   (one wavefront per row: lanes over W_l's columns, the
//                        row's entries of D pre-sorted by line on the host; sums accumulated in LDS in a fixed order).
// The generic triple product (k_triple_product) costs n_l times more on such a Dinv: 1.7 s against milliseconds at 3.9 M unknowns.
__global__ void __launch_bounds__(256) k_line_schur_w(const int64_t *__restrict__ bp, const int64_t *__restrict__ dofs,
                                                      const int64_t *__restrict__ boff, const double *__restrict__ B, int64_t nlines,
                                                      const int64_t *__restrict__ grp, const int32_t *__restrict__ gcol,
                                                      const double *__restrict__ gval, const int64_t *__restrict__ wptr,
                                                      const int64_t *__restrict__ wcols, const int64_t *__restrict__ woff,
                                                      double *__restrict__ W, int *missing) {
    extern __shared__ double lds[];
    for (int64_t l = blockIdx.x; l < nlines; l += gridDim.x) {
        const int64_t b0 = bp[l];
        const int n = (int)(bp[l + 1] - b0), m = (int)(wptr[l + 1] - wptr[l]);
        const double *__restrict__ Bb = B + boff[l];
        const int64_t *__restrict__ wc = wcols + wptr[l];
        __syncthreads();
        for (int e = threadIdx.x; e < n * m; e += blockDim.x) lds[e] = 0.0;
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            const int64_t r = dofs[b0 + k], g0 = grp[r];
            const int ng = (int)(grp[r + 1] - g0);
            // (i, e) pairs: row i of the block, entry e of G's row r - distinct entries, distinct columns of W: no two threads meet
            for (int q = threadIdx.x; q < n * ng; q += blockDim.x) {
                const int e = q / n, i = q - e * n;
                const int64_t c = gcol[g0 + e];
                int lo = 0, hi = m - 1, pos = -1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int64_t v = wc[mid];
                    if (v == c) { pos = mid; break; }
                    if (v < c) lo = mid + 1; else hi = mid - 1;
                }
                if (pos >= 0) lds[i * m + pos] += Bb[(size_t)k * n + i] * gval[g0 + e];
                else if (i == 0) atomicAdd(missing, 1);
            }
            __syncthreads();
        }
        double *__restrict__ Wl = W + woff[l];
        for (int e = threadIdx.x; e < n * m; e += blockDim.x) Wl[e] = lds[e];
    }
}

constexpr int kMaxSchurRow = 1024;          // entries in one row of S that k_line_schur_s accumulates in LDS
__global__ void __launch_bounds__(256) k_line_schur_s(int64_t np, const double *__restrict__ dval, const int64_t *__restrict__ dperm,
                                                      const int64_t *__restrict__ dpos, const int64_t *__restrict__ seg_ptr,
                                                      const int64_t *__restrict__ seg_line, const int64_t *__restrict__ seg_start,
                                                      const int64_t *__restrict__ wptr, const int64_t *__restrict__ wcols,
                                                      const int64_t *__restrict__ woff, const double *__restrict__ W,
                                                      const int64_t *__restrict__ srp, const int32_t *__restrict__ scol,
                                                      double *__restrict__ sval, int *missing) {
    __shared__ double acc[4][kMaxSchurRow];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + w; p < np; p += (int64_t)gridDim.x * 4) {
        const int64_t s0 = srp[p];
        const int len = (int)(srp[p + 1] - s0);
        for (int e = lane; e < len; e += 64) acc[w][e] = 0.0;
        __builtin_amdgcn_wave_barrier();
        for (int64_t sg = seg_ptr[p]; sg < seg_ptr[p + 1]; ++sg) {
            const int64_t l = seg_line[sg], e0 = seg_start[sg], e1 = seg_start[sg + 1];
            const int m = (int)(wptr[l + 1] - wptr[l]);
            const double *__restrict__ Wl = W + woff[l];
            for (int c = lane; c < m; c += 64) {
                double a = 0.0;
                for (int64_t e = e0; e < e1; ++e) a += dval[dperm[e]] * Wl[dpos[e] * m + c];
                const int32_t col = (int32_t)wcols[wptr[l] + c];
                int lo = 0, hi = len - 1, pos = -1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int32_t v = scol[s0 + mid];
                    if (v == col) { pos = mid; break; }
                    if (v < col) lo = mid + 1; else hi = mid - 1;
                }
                if (pos >= 0) acc[w][pos] += a;          // (the columns of one line are distinct: no two lanes meet)
                else atomicAdd(missing, 1);
            }
            __builtin_amdgcn_wave_barrier();             // (a wave's LDS operations complete in order: the next line sees these sums)
        }
        for (int e = lane; e < len; e += 64) sval[s0 + e] = acc[w][e];
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr int kMaxLineBlock = 136;          // unknowns in one block: 136^2 doubles + ids = 148.5 KB of the CU's 160 KB of LDS
NPG_API int npg_csr_line_block_inverse(npg_csr *Dinv, const npg_csr *A, const npg_index *block_ptr, const npg_index *block_dofs) {
    NPG_REQUIRE(Dinv && A && block_ptr && block_dofs, "npg_csr_line_block_inverse: NULL argument");
    NPG_REQUIRE(A->nnode() == 0 && Dinv->nnode() == 0, "npg_csr_line_block_inverse: node-blocked matrices are not supported");
    const int64_t nu = Dinv->m, nb = block_ptr->n - 1;
    NPG_REQUIRE(Dinv->n == nu && A->m >= nu && A->n >= nu && nb >= 1 && block_dofs->n == nu && block_dofs->bound == nu &&
                    block_ptr->bound == nu + 1,
                "npg_csr_line_block_inverse: Dinv must be nu x nu, the blocks a partition of [0, nu) (block_dofs: nu entries below nu; "
                "block_ptr: offsets up to nu)");
    // the blocks against Dinv's pattern (host: index arrays only)
    std::vector<int64_t> bp((size_t)nb + 1), dofs((size_t)nu);
    NPG_HIP(hipSetDevice(A->ctx->device));
    NPG_HIP(hipStreamSynchronize(A->ctx->stream));
    NPG_HIP(hipMemcpy(bp.data(), block_ptr->d, bp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    NPG_HIP(hipMemcpy(dofs.data(), block_dofs->d, dofs.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    NPG_REQUIRE(bp[0] == 0 && bp[(size_t)nb] == nu, "npg_csr_line_block_inverse: block_ptr must run from 0 to nu");
    int64_t nmax = 0, total = 0;
    const int64_t *drp = Dinv->h_rowptr.data();
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t n = bp[(size_t)b + 1] - bp[(size_t)b];
        NPG_REQUIRE(n >= 1 && n <= kMaxLineBlock, "npg_csr_line_block_inverse: block %lld has %lld unknowns (1..%d)", (long long)b,
                    (long long)n, kMaxLineBlock);
        for (int64_t e = bp[(size_t)b]; e < bp[(size_t)b + 1]; ++e) {
            NPG_REQUIRE(e == bp[(size_t)b] || dofs[(size_t)e] > dofs[(size_t)e - 1], "npg_csr_line_block_inverse: block %lld is not ascending", (long long)b);
            NPG_REQUIRE(drp[dofs[(size_t)e] + 1] - drp[dofs[(size_t)e]] == n, "npg_csr_line_block_inverse: row %lld of Dinv does not hold its block (%lld entries, block of %lld)",
                        (long long)dofs[(size_t)e], (long long)(drp[dofs[(size_t)e] + 1] - drp[dofs[(size_t)e]]), (long long)n);
        }
        nmax = std::max(nmax, n);
        total += n * n;
    }
    NPG_REQUIRE(total == Dinv->nnz, "npg_csr_line_block_inverse: Dinv's pattern must hold exactly the blocks (%lld entries expected, %lld found)",
                (long long)total, (long long)Dinv->nnz);
    const size_t lds = (size_t)nmax * nmax * sizeof(double) + (size_t)nmax * sizeof(int32_t) + 8;
    NPG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_line_block_inverse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    npg_ctx *ctx = A->ctx;
    if (Dinv->lb_nblocks != nb || !Dinv->lb_val) {         // first call: the dense pack and the handle's own copy of the blocks
        for (void *q : {(void *)Dinv->lb_ptr, (void *)Dinv->lb_dofs, (void *)Dinv->lb_off, (void *)Dinv->lb_val, (void *)Dinv->lb_val32,
                        (void *)Dinv->lb_val16, (void *)Dinv->lb_scale})
            if (q) hipFree(q);
        Dinv->lb_ptr = Dinv->lb_dofs = Dinv->lb_off = nullptr;
        Dinv->lb_val = nullptr;
        Dinv->lb_val32 = nullptr;
        Dinv->lb_val16 = nullptr;
        Dinv->lb_scale = nullptr;
        Dinv->lb_nblocks = 0;
        std::vector<int64_t> off((size_t)nb + 1, 0);
        for (int64_t b = 0; b < nb; ++b) {
            const int64_t n = bp[(size_t)b + 1] - bp[(size_t)b];
            off[(size_t)b + 1] = off[(size_t)b] + n * n;
        }
        NPG_HIP(hipMalloc((void **)&Dinv->lb_ptr, bp.size() * sizeof(int64_t)));
        NPG_HIP(hipMalloc((void **)&Dinv->lb_dofs, dofs.size() * sizeof(int64_t)));
        NPG_HIP(hipMalloc((void **)&Dinv->lb_off, off.size() * sizeof(int64_t)));
        NPG_HIP(hipMalloc((void **)&Dinv->lb_val, (size_t)total * sizeof(double)));
        NPG_HIP(hipMalloc((void **)&Dinv->lb_val32, (size_t)total * sizeof(float)));
        NPG_HIP(hipMalloc((void **)&Dinv->lb_val16, (size_t)total * sizeof(_Float16)));
        NPG_HIP(hipMalloc((void **)&Dinv->lb_scale, (size_t)nu * sizeof(double)));
        NPG_HIP(hipMemcpy(Dinv->lb_ptr, bp.data(), bp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(Dinv->lb_dofs, dofs.data(), dofs.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        NPG_HIP(hipMemcpy(Dinv->lb_off, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        Dinv->lb_nblocks = nb;
        Dinv->gen++;
    }
    int *bad = reinterpret_cast<int *>(ctx->d_scratch);
    NPG_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    const int grid = (int)std::min<int64_t>(nb, 8 * ctx->num_cu);
    hipLaunchKernelGGL(k_line_block_inverse, dim3(grid), dim3(256), lds, ctx->stream, A->rowptr, A->col, A->val, block_ptr->d, block_dofs->d,
                       nb, Dinv->rowptr, Dinv->val, (const int64_t *)Dinv->lb_off, Dinv->lb_val, Dinv->lb_val32, Dinv->lb_val16, Dinv->lb_scale, bad);
    NPG_HIP(hipGetLastError());
    int nbad = 0;
    NPG_HIP(hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_REQUIRE(nbad == 0, "npg_csr_line_block_inverse: %d blocks are singular to working precision (zero or non-finite pivot)", nbad);
    return csr_repack(Dinv);
}

NPG_API int npg_csr_line_schur(npg_csr *S, const npg_csr *D, const npg_csr *Dinv, const npg_csr *G, const npg_index *wptr,
                               const npg_index *wcols, const npg_index *woff, const npg_index *dperm, const npg_index *dpos,
                               const npg_index *seg_ptr, const npg_index *seg_line, const npg_index *seg_start) {
    NPG_REQUIRE(S && D && Dinv && G && wptr && wcols && woff && dperm && dpos && seg_ptr && seg_line && seg_start,
                "npg_csr_line_schur: NULL argument");
    NPG_REQUIRE(S->nnode() == 0 && D->nnode() == 0 && G->nnode() == 0, "npg_csr_line_schur: node-blocked matrices are not supported");
    NPG_REQUIRE(Dinv->lb_nblocks > 0 && Dinv->lb_val, "npg_csr_line_schur: Dinv is not a line-block inverse (npg_csr_line_block_inverse)");
    const int64_t nl = Dinv->lb_nblocks, np = S->m, nu = Dinv->m;
    NPG_REQUIRE(D->n == nu && G->m == nu && D->m == np && S->n == G->n, "npg_csr_line_schur: shapes do not chain");
    NPG_REQUIRE(wptr->n == nl + 1 && woff->n == nl + 1 && wcols->bound == G->n && dperm->n == D->nnz && dperm->bound == D->nnz &&
                    dpos->n == D->nnz && seg_ptr->n == np + 1 && seg_start->n == seg_line->n + 1 && seg_line->bound == nl &&
                    seg_start->bound == D->nnz + 1 && seg_ptr->bound == seg_line->n + 1,
                "npg_csr_line_schur: the index arrays do not match the matrices (lines %lld, rows %lld, entries of D %lld)", (long long)nl,
                (long long)np, (long long)D->nnz);
    npg_ctx *ctx = S->ctx;
    NPG_HIP(hipSetDevice(ctx->device));
    std::vector<int64_t> h_woff((size_t)nl + 1);
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_HIP(hipMemcpy(h_woff.data(), woff->d, h_woff.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    const int64_t wtotal = h_woff[(size_t)nl];
    NPG_REQUIRE(h_woff[0] == 0 && wtotal == woff->bound - 1, "npg_csr_line_schur: woff must run from 0 to its bound - 1 = the size of the W arrays");
    int64_t wmax = 0;
    for (int64_t l = 0; l < nl; ++l) {
        NPG_REQUIRE(h_woff[(size_t)l + 1] >= h_woff[(size_t)l], "npg_csr_line_schur: woff must ascend");
        wmax = std::max(wmax, h_woff[(size_t)l + 1] - h_woff[(size_t)l]);
    }
    constexpr int64_t kMaxW = 150 * 1024 / 8;          // doubles of one W_l that fit the CU's LDS beside nothing else
    NPG_REQUIRE(wmax <= kMaxW, "npg_csr_line_schur: a line's W block has %lld entries (limit %lld)", (long long)wmax, (long long)kMaxW);
    int64_t max_row = 0;
    for (int64_t r = 0; r < np; ++r) max_row = std::max(max_row, S->h_rowptr[(size_t)r + 1] - S->h_rowptr[(size_t)r]);
    NPG_REQUIRE(max_row <= kMaxSchurRow, "npg_csr_line_schur: a row of S has %lld entries (limit %d)", (long long)max_row, kMaxSchurRow);
    double *W = nullptr;
    NPG_HIP(hipMalloc((void **)&W, std::max<size_t>(1, (size_t)wtotal) * sizeof(double)));
    int *missing = reinterpret_cast<int *>(ctx->d_scratch);
    NPG_HIP(hipMemsetAsync(missing, 0, sizeof(int), ctx->stream));
    const size_t lds = std::max<size_t>(1024, (size_t)wmax * sizeof(double));          // LDS of the W pass: the largest n_l m_l
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(k_line_schur_w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) { hipFree(W); NPG_HIP(ea); }
    hipLaunchKernelGGL(k_line_schur_w, dim3((unsigned)std::min<int64_t>(nl, 4 * ctx->num_cu)), dim3(256), lds, ctx->stream,
                       (const int64_t *)Dinv->lb_ptr, (const int64_t *)Dinv->lb_dofs, (const int64_t *)Dinv->lb_off, (const double *)Dinv->lb_val, nl,
                       G->rowptr, G->col, G->val, (const int64_t *)wptr->d, (const int64_t *)wcols->d, (const int64_t *)woff->d, W, missing);
    hipLaunchKernelGGL(k_line_schur_s, dim3((unsigned)std::min<int64_t>((np + 3) / 4, 16 * ctx->num_cu)), dim3(256), 0, ctx->stream, np,
                       (const double *)D->val, (const int64_t *)dperm->d, (const int64_t *)dpos->d, (const int64_t *)seg_ptr->d,
                       (const int64_t *)seg_line->d, (const int64_t *)seg_start->d, (const int64_t *)wptr->d, (const int64_t *)wcols->d,
                       (const int64_t *)woff->d, (const double *)W, S->rowptr, S->col, S->val, missing);
    hipError_t el = hipGetLastError();
    int miss = 0;
    hipError_t ec = hipMemcpyAsync(&miss, missing, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t es = hipStreamSynchronize(ctx->stream);
    hipFree(W);
    NPG_HIP(el);
    NPG_HIP(ec);
    NPG_HIP(es);
    NPG_REQUIRE(miss == 0, "npg_csr_line_schur: %d products fall outside the patterns handed in", miss);
    return NPG_OK;
}

NPG_API int npg_csr_triple_product(npg_csr *S, const npg_csr *D, const npg_csr *Dinv, const npg_csr *G) {
    NPG_REQUIRE(S && D && Dinv && G, "npg_csr_triple_product: NULL argument");
    NPG_REQUIRE_NO_PACKS(S, "npg_csr_triple_product");
    NPG_REQUIRE(S->nnode() == 0 && D->nnode() == 0 && Dinv->nnode() == 0 && G->nnode() == 0,
                "npg_csr_triple_product: node-blocked matrices are not supported");
    NPG_REQUIRE(D->n == Dinv->m && Dinv->n == G->m && S->m == D->m && S->n == G->n, "npg_csr_triple_product: shapes do not chain");
    npg_ctx *ctx = S->ctx;
    int *missing = reinterpret_cast<int *>(ctx->d_scratch);
    NPG_HIP(hipMemsetAsync(missing, 0, sizeof(int), ctx->stream));
    const int grid = (int)std::min<int64_t>(4096, (S->m + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_triple_product, dim3(std::max(grid, 1)), dim3(kBlock), 0, ctx->stream, D->rowptr, D->col, D->val,
                       Dinv->rowptr, Dinv->col, Dinv->val, G->rowptr, G->col, G->val, S->m, S->rowptr, S->col, S->val, missing);
    NPG_HIP(hipGetLastError());
    int miss = 0;
    NPG_HIP(hipMemcpyAsync(&miss, missing, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_REQUIRE(miss == 0, "npg_csr_triple_product: %d products fall outside S's pattern", miss);
    return NPG_OK;
}

NPG_API int npg_csr_product(npg_csr *Cm, const npg_csr *A, const npg_csr *B) {
    NPG_REQUIRE(Cm && A && B, "npg_csr_product: NULL argument");
    NPG_REQUIRE_NO_PACKS(Cm, "npg_csr_product");
    NPG_REQUIRE(Cm->nnode() == 0 && A->nnode() == 0 && B->nnode() == 0, "npg_csr_product: node-blocked matrices are not supported");
    NPG_REQUIRE(A->n == B->m && Cm->m == A->m && Cm->n == B->n, "npg_csr_product: shapes do not chain");
    npg_ctx *ctx = Cm->ctx;
    int *missing = reinterpret_cast<int *>(ctx->d_scratch);
    NPG_HIP(hipMemsetAsync(missing, 0, sizeof(int), ctx->stream));
    const int grid = (int)std::min<int64_t>(4096, (Cm->m + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_fixed_product, dim3(std::max(grid, 1)), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, A->val, B->rowptr,
                       B->col, B->val, Cm->m, Cm->rowptr, Cm->col, Cm->val, missing);
    NPG_HIP(hipGetLastError());
    int miss = 0;
    NPG_HIP(hipMemcpyAsync(&miss, missing, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NPG_HIP(hipStreamSynchronize(ctx->stream));
    NPG_REQUIRE(miss == 0, "npg_csr_product: %d products fall outside the result's pattern", miss);
    return csr_repack(Cm);
}

NPG_API int npg_csr_inv_diag(const npg_csr *A, npg_vec *d) {
    NPG_REQUIRE(A && d && A->m <= A->n && d->n == A->m, "npg_csr_inv_diag: shape mismatch");   // m < n: a rank's row block
    NPG_REQUIRE_PLAIN(A, "npg_csr_inv_diag");
    const int grid = (int)std::min<int64_t>(2048, (A->m + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_inv_diag, dim3(std::max(grid, 1)), dim3(kBlock), 0, A->ctx->stream, A->rowptr, A->col, A->val,
                       d->d, A->m);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

namespace npg {
// y = alpha B x + beta c for a block-diagonal B held as dense column-major blocks (npg_csr_line_block_inverse): one wavefront per
// block, lane = row (two rows per lane from 65 unknowns on), the block's x staged in LDS; a column is one contiguous read
template <typename T>
__global__ void __launch_bounds__(256) k_line_apply(const int64_t *__restrict__ bp, const int64_t *__restrict__ dofs,
                                                    const int64_t *__restrict__ boff, const T *__restrict__ B, int64_t nblocks,
                                                    const double *__restrict__ x, double alpha, double beta, const double *c,
                                                    double *y, const double *__restrict__ cs = nullptr) {
    constexpr int kMaxN = 136;
    __shared__ double xs[4][kMaxN];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t b = (int64_t)blockIdx.x * 4 + w; b < nblocks; b += (int64_t)gridDim.x * 4) {
        const int64_t b0 = bp[b];
        const int n = (int)(bp[b + 1] - b0);
        const T *__restrict__ Bb = B + boff[b];
        int64_t r0 = -1, r1 = -1, r2 = -1;
        if (lane < n) r0 = dofs[b0 + lane];
        if (lane + 64 < n) r1 = dofs[b0 + lane + 64];
        if (lane + 128 < n) r2 = dofs[b0 + lane + 128];
        // (cs: the column scales of the fp16 pack, folded into x as it is staged)
        if (r0 >= 0) xs[w][lane] = cs ? x[r0] * cs[r0] : x[r0];
        if (r1 >= 0) xs[w][lane + 64] = cs ? x[r1] * cs[r1] : x[r1];
        if (r2 >= 0) xs[w][lane + 128] = cs ? x[r2] * cs[r2] : x[r2];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's LDS stores have landed (a wave is in lock-step)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        // Every load is unconditional, from a row index clamped into the block (a lane without a row re-reads the last one and its
        // sum is never stored): no control flow between the loads of a trip, so all of them are in flight together - with one
        // column per trip the kernel ran at the memory LATENCY (2 loads in flight per wave: 3.0 TB/s on 1.1 GB).
        const int i0 = min(lane, n - 1), i1 = min(lane + 64, n - 1), i2 = min(lane + 128, n - 1);
        constexpr int U = 8;
        int j = 0;
        if (n <= 64) {
            for (; j + U <= n; j += U) {
                T v[U];
#pragma unroll
                for (int q = 0; q < U; ++q) v[q] = Bb[(size_t)(j + q) * n + i0];
#pragma unroll
                for (int q = 0; q < U; ++q) a0 += (double)v[q] * xs[w][j + q];
            }
        } else if (n <= 128) {
            for (; j + U <= n; j += U) {
                T v[U], u[U];
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const T *__restrict__ col = Bb + (size_t)(j + q) * n;
                    v[q] = col[i0];
                    u[q] = col[i1];
                }
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const double xv = xs[w][j + q];
                    a0 += (double)v[q] * xv;
                    a1 += (double)u[q] * xv;
                }
            }
        }
        for (; j < n; ++j) {                     // the last columns, and blocks of more than 128 unknowns column by column
            const double xv = xs[w][j];
            const T *__restrict__ col = Bb + (size_t)j * n;
            const T c0 = col[i0], c1 = col[i1], c2 = col[i2];
            a0 += (double)c0 * xv;
            a1 += (double)c1 * xv;
            a2 += (double)c2 * xv;
        }
        if (r0 >= 0) y[r0] = alpha * a0 + (beta != 0.0 ? beta * c[r0] : 0.0);
        if (r1 >= 0) y[r1] = alpha * a1 + (beta != 0.0 ? beta * c[r1] : 0.0);
        if (r2 >= 0) y[r2] = alpha * a2 + (beta != 0.0 ? beta * c[r2] : 0.0);
        __builtin_amdgcn_wave_barrier();
    }
}

static int line_apply(const npg_csr *A, const double *x, const SpmvEpi &e) {
    const int grid = (int)std::min<int64_t>((A->lb_nblocks + 3) / 4, 16 * (int64_t)A->ctx->num_cu);
    static const bool fp16 = !getenv("NPG_LINE_FP16") || atoi(getenv("NPG_LINE_FP16")) != 0;
    if (e.f32 && fp16)          // (products that accept rounded operator values: the column-scaled fp16 pack, half of fp32's bytes)
        hipLaunchKernelGGL(k_line_apply<_Float16>, dim3(std::max(grid, 1)), dim3(256), 0, A->ctx->stream, (const int64_t *)A->lb_ptr,
                           (const int64_t *)A->lb_dofs, (const int64_t *)A->lb_off, (const _Float16 *)A->lb_val16, A->lb_nblocks, x, e.alpha, e.beta,
                           e.c, e.y, (const double *)A->lb_scale);
    else if (e.f32)
        hipLaunchKernelGGL(k_line_apply<float>, dim3(std::max(grid, 1)), dim3(256), 0, A->ctx->stream, (const int64_t *)A->lb_ptr,
                           (const int64_t *)A->lb_dofs, (const int64_t *)A->lb_off, (const float *)A->lb_val32, A->lb_nblocks, x, e.alpha, e.beta, e.c,
                           e.y);
    else
        hipLaunchKernelGGL(k_line_apply<double>, dim3(std::max(grid, 1)), dim3(256), 0, A->ctx->stream, (const int64_t *)A->lb_ptr,
                           (const int64_t *)A->lb_dofs, (const int64_t *)A->lb_off, (const double *)A->lb_val, A->lb_nblocks, x, e.alpha, e.beta, e.c,
                           e.y);
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

// y[0, m) = alpha A x[0, n) + beta y on raw device pointers (x and y may be windows into larger vectors; they must not overlap)
bool nb_epilogue_ok(const npg_csr *Ap, const npg_csr *Dinv, int64_t nu) {
    const npg_csr *A = spmv_form(Ap);
    if (!A || !Dinv || A->nnode() == 0 || A->pk9 || A->uperm || Dinv->lb_nblocks || Dinv->nnode() || A->n != A->m) return false;
    const int64_t nf = A->nfull, ns = A->nsurf, rest = nu - 3 * nf - 2 * ns;
    return rest >= 0 && Dinv->m == nu && Dinv->nnz == 9 * nf + 4 * ns + rest && A->block_rows() <= nu;
}

thread_local int64_t *byte_sink = nullptr;

int64_t spmv_stream_bytes(const npg_csr *Ap, const SpmvEpi &e, bool windowed) {
    const npg_csr *A = spmv_form(Ap);
    static const bool dense_blocks = !getenv("NPG_LINE_DENSE") || atoi(getenv("NPG_LINE_DENSE")) != 0;
    static const bool fp16 = !getenv("NPG_LINE_FP16") || atoi(getenv("NPG_LINE_FP16")) != 0;
    int64_t b = 0;
    if (A->lb_nblocks && !e.z && dense_blocks) {
        // dense column-major block pack (k_line_apply): sum of n_b^2 values = the CSR form's nnz, + block offsets and DoF lists
        b = A->nnz * (e.f32 ? (fp16 ? 2 : 4) : 8) + 16 * A->lb_nblocks + 8 * A->m + (e.f32 && fp16 ? 8 * A->m : 0);
    } else if (windowed) {
        int64_t wb = 0;
        npg_csr_window_info(A, nullptr, nullptr, nullptr, &wb);
        b = wb;
    } else {
        npg_csr_spmv_bytes(A, &b);
        if (e.f32 && A->val32) b -= 4 * A->rnnz + (A->pkc32 && A->nnode() ? 8 * A->h_prow[A->nnode()] : 0);      // fp32 value copies
    }
    // vectors: input once (gathers beyond that are cache hits by design), outputs and epilogue operands per row
    b += (windowed ? 4 : 8) * A->n + 8 * A->m * (1 + (e.beta != 0.0 && e.c != e.y ? 1 : 0) + (e.beta != 0.0 && e.c == e.y ? 1 : 0) +
                                                 (e.z ? 1 + (e.zin ? 1 : 0) + (e.dg ? 1 : 0) : 0));
    return b;
}

int spmv_epi(const npg_csr *Ap, const double *x, const SpmvEpi &e, const NbEpi *nb) {
    const npg_csr *A = spmv_form(Ap);
    static const bool dense_blocks = !getenv("NPG_LINE_DENSE") || atoi(getenv("NPG_LINE_DENSE")) != 0;
    if (byte_sink) *byte_sink += spmv_stream_bytes(A, e, false) + (nb ? 8 * (int64_t)nb->rows * 4 : 0);
    if (A->lb_nblocks && !e.z && dense_blocks) return line_apply(A, x, e);
    if (int rc = check_record_view(A, true, "spmv")) return rc;
    NPG_REQUIRE(!nb || (A->nnode() > 0 && !A->pk9), "spmv: the node-block epilogue needs a matrix stored by {c, K, C} node blocks");
    switch (A->lanes) {
        case 4: launch_spmv<4>(A, x, e, nb); break;
        case 8: launch_spmv<8>(A, x, e, nb); break;
        case 16: launch_spmv<16>(A, x, e, nb); break;
        default: launch_spmv<32>(A, x, e, nb); break;
    }
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}

int spmv_raw(const npg_csr *A, const double *x, double *y, double alpha, double beta, int f32) {
    SpmvEpi e{};
    e.alpha = alpha;
    e.beta = beta;
    e.c = y;
    e.y = y;
    e.f32 = f32;
    return spmv_epi(A, x, e);
}
}  // namespace npg

namespace npg {
__global__ void k_fill_gather32(const double *__restrict__ x, GatherMap g, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        g.p[g.pos((int)i)] = (float)x[i];
}

// y = A x with x read from its fp32 gather-layout copy: the SpMV of the Krylov kernels' gather-layout instance, stand-alone.
// WL > 0: the matrix's windowed tile set (block tiles gather every distinct column once into LDS); 0: its ordinary tiles.
// phase stamps of the windowed block tiles (NPG_WIN_DIAG=128, tools/window_ab.py): s_memtime at the marks of spmv_tile_win,
// taken by thread 0's wave without waiting for memory
struct WinProf {
    unsigned long long *c;
    __device__ __forceinline__ void stamp(int i) const {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        c[i] = t;
    }
};

template <int L, int WL>
__global__ void __launch_bounds__(kSpmvThreads, 6) k_spmv_g32_timed(CsrDev A, WinDev W, const WTileDesc *__restrict__ tiles, int ntiles, GatherMap g,
                                                                  double *__restrict__ y, unsigned long long *acc) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    unsigned long long a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nt = 0, st[10];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    WTileDesc td = tiles[t];
    WinPre pre;
    bool have = false;
    while (true) {
        const int tn = t + gridDim.x;
        WTileDesc nd = td;
        if (tn < ntiles) nd = tiles[tn];
        if (td.nw && td.r0 < block_rows(A)) {
            if (!have) win_first<kSpmvThreads>(W, PaddedX{g}, td, pre);
            have = tn < ntiles && nd.nw != 0 && nd.r0 < block_rows(A);
            spmv_tile_win<kSpmvThreads, WL, PaddedX, kTileNnz, WinProf>(A, W, PaddedX{g}, td, nd, have, pre, tl, sw, WinProf{st});
            WinProf{st}.stamp(6);
            for (int r = threadIdx.x; r < td.nrows; r += kSpmvThreads) y[td.r0 + r] = sw[r];
            WinProf{st}.stamp(7);
            a[0] += st[3] - st[9];      // bookkeeping waited for and written
            a[8] += st[8] - st[4];      // window (gathered a tile ago) written
            a[9] += st[9] - st[8];      // loads issued
            a[1] += st[5] - st[3];      // barrier 1
            a[2] += st[0] - st[5];      // gathers issued, records waited for, products
            a[3] += st[1] - st[0];      // barrier 2
            a[4] += st[2] - st[1];      // segmented sums
            a[5] += st[6] - st[2];      // barrier 3
            a[6] += st[7] - st[6];      // output stores issued
            ++nt;
        } else {
            have = false;
            pre = WinPre{};
        }
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
    if (threadIdx.x == 0) {
        for (int i = 0; i < 10; ++i)
            if (i != 7) atomicAdd(acc + i, a[i]);
        atomicAdd(acc + 7, nt);
    }
}

template <int L, int WL, int DIAG = 0>
__global__ void __launch_bounds__(kSpmvThreads, 6) k_spmv_g32(CsrDev A, WinDev W, const void *__restrict__ tiles_, int ntiles, GatherMap g,
                                                            double *__restrict__ y) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    if (DIAG & 64) {          // timing diagnostic: the workgroups of a CU start a third of a tile apart
        const int k = blockIdx.x / 256;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(110);
    }
    if constexpr (WL == 0) {          // the matrix's ordinary tiles
        const TileDesc *__restrict__ tiles = static_cast<const TileDesc *>(tiles_);
        TileDesc td = tiles[t];
        while (true) {
            const int tn = t + gridDim.x;
            TileDesc nd = td;
            if (tn < ntiles) nd = tiles[tn];
            spmv_tile<kSpmvThreads, L, PaddedX, kTileNnz, 2, NoProf, false, true, false>(A, PaddedX{g}, td, tl, sw);
            for (int r = threadIdx.x; r < td.nrows; r += kSpmvThreads) y[td.r0 + r] = sw[r];
            if (tn >= ntiles) break;
            t = tn;
            td = nd;
        }
    } else {
        const WTileDesc *__restrict__ tiles = static_cast<const WTileDesc *>(tiles_);
        WTileDesc td = tiles[t];
        WinPre pre;
        bool have = false;           // `pre` holds td's window
        while (true) {
            const int tn = t + gridDim.x;
            WTileDesc nd = td;
            if (tn < ntiles) nd = tiles[tn];
            if (td.nw) {
                if (!(DIAG & 16)) {
                    if (!have) win_first<kSpmvThreads>(W, PaddedX{g}, td, pre);
                    have = tn < ntiles && nd.nw != 0;
                    if (td.r0 < block_rows(A))
                        spmv_tile_win<kSpmvThreads, WL, PaddedX, kTileNnz, NoProf, (DIAG & 39)>(A, W, PaddedX{g}, td, nd, have, pre, tl, sw);
                    else if (!(DIAG & 8))
                        spmv_tile_winrows<kSpmvThreads, L>(A, W, PaddedX{g}, td, nd, have, pre, tl, sw);
                }
            } else if (!(DIAG & 8)) {
                have = false;
                spmv_tile<kSpmvThreads, L, PaddedX, kTileNnz, 2, NoProf, false, true, false>(A, PaddedX{g}, ordinary(td), tl, sw);
                pre = WinPre{};              // (dead across the call above: nothing to keep in registers)
            }
            for (int r = threadIdx.x; r < td.nrows; r += kSpmvThreads) y[td.r0 + r] = sw[r];
            if (tn >= ntiles) break;
            t = tn;
            td = nd;
        }
    }
}

// the same product with the tiled SpMV's epilogue (SpmvEpi: y = alpha A x + beta c, second output z) on the windowed tile set:
// the residuals of the multigrid cycle's finest level (mg.hip), whose input is the fp32 gather-layout copy of the iterate
template <int L, int WL, bool NB = false>
__global__ void __launch_bounds__(kSpmvThreads, 6) k_spmv_g32e(CsrDev A, WinDev W, const WTileDesc *__restrict__ tiles, int ntiles, GatherMap g,
                                                             SpmvEpi e, NbEpi nb = NbEpi{}) {
    __shared__ TileLds tl;
    __shared__ double sw[kTileRows];
    int t = blockIdx.x;
    if (t >= ntiles) return;
    WTileDesc td = tiles[t];
    WinPre pre;
    bool have = false;           // `pre` holds td's window
    while (true) {
        const int tn = t + gridDim.x;
        WTileDesc nd = td;
        if (tn < ntiles) nd = tiles[tn];
        if (td.nw) {
            if (!have) win_first<kSpmvThreads>(W, PaddedX{g}, td, pre);
            have = tn < ntiles && nd.nw != 0;
            if (td.r0 < block_rows(A))
                spmv_tile_win<kSpmvThreads, WL, PaddedX, kTileNnz, NoProf, 0>(A, W, PaddedX{g}, td, nd, have, pre, tl, sw);
            else
                spmv_tile_winrows<kSpmvThreads, L>(A, W, PaddedX{g}, td, nd, have, pre, tl, sw);
        } else {
            have = false;
            spmv_tile<kSpmvThreads, L, PaddedX, kTileNnz, 2, NoProf, false, true, false>(A, PaddedX{g}, ordinary(td), tl, sw);
            pre = WinPre{};
        }
        for (int r = threadIdx.x; r < td.nrows; r += kSpmvThreads) {
            const int row = td.r0 + r;
            double v = e.alpha * sw[r];
            if (e.beta != 0.0) v += e.beta * e.c[row];
            e.y[row] = v;
            if (e.z) {
                double q = e.w * (e.dg ? e.dg[row] : 1.0) * v;
                if (e.zin) q += e.zc * e.zin[row];
                e.z[row] = q;
            }
            if constexpr (NB) sw[r] = v;
        }
        if constexpr (NB) nb_epilogue<kSpmvThreads>(nb, td.r0, td.nrows, sw);
        if (tn >= ntiles) break;
        t = tn;
        td = nd;
    }
}

int64_t gather32_floats(const npg_csr *Ap);
// node slots (4 floats each) at the head of a gather-layout copy of one of A's input vectors: the owned block nodes and, behind
// them, the ghost nodes the windowed tile set addresses as record columns
int64_t gather32_nodes(const npg_csr *Ap) {
    const npg_csr *A = spmv_form(Ap);
    return A->nnode() + (A->gslot ? A->ngn() : 0);
}
// x (n entries, fp64) -> its fp32 gather-layout copy; then y = alpha A fl32(x) + beta c ... on A's windowed tiles
int spmv_epi_gather32(const npg_csr *Ap, const double *x, float *xg, const SpmvEpi &e, const NbEpi *nb) {
    const npg_csr *A = spmv_form(Ap);
    NPG_REQUIRE(gather32_floats(A) > 0 && xg, "spmv_epi_gather32: the matrix has no windowed tile set");
    if (byte_sink) *byte_sink += spmv_stream_bytes(A, e, true) + 12 * A->n + (nb ? 8 * (int64_t)nb->rows * 4 : 0);    // (+ the fill kernel: 8 B in, 4 B out)
    if (int rc = check_record_view(A, false, "spmv_epi_gather32")) return rc;
    const int64_t nbr = A->block_rows();
    const GatherMap g{xg, 3 * A->nfull, A->nfull, (int)nbr, (int)(gather32_nodes(A) * 4 - nbr)};
    hipLaunchKernelGGL(k_fill_gather32, dim3((unsigned)std::min<int64_t>(2048, (A->n + 255) / 256)), dim3(256), 0, A->ctx->stream, x, g, A->n);
    const dim3 grid(std::max(1, std::min<int>(A->nwtiles, 3 * A->ctx->num_cu))), blk(kSpmvThreads);
    const WinDev W = win_view(A);
#define NPG_G32E(LL)                                                                                                                     \
    if (nb && A->wlanes == 8)                                                                                                            \
        hipLaunchKernelGGL((k_spmv_g32e<LL, 8, true>), grid, blk, 0, A->ctx->stream, csr_view(A), W, A->wtile_ptr, A->nwtiles, g, e, *nb); \
    else if (nb)                                                                                                                         \
        hipLaunchKernelGGL((k_spmv_g32e<LL, 4, true>), grid, blk, 0, A->ctx->stream, csr_view(A), W, A->wtile_ptr, A->nwtiles, g, e, *nb); \
    else if (A->wlanes == 8)                                                                                                             \
        hipLaunchKernelGGL((k_spmv_g32e<LL, 8>), grid, blk, 0, A->ctx->stream, csr_view(A), W, A->wtile_ptr, A->nwtiles, g, e, NbEpi{}); \
    else                                                                                                                                 \
        hipLaunchKernelGGL((k_spmv_g32e<LL, 4>), grid, blk, 0, A->ctx->stream, csr_view(A), W, A->wtile_ptr, A->nwtiles, g, e, NbEpi{})
    switch (A->lanes) {
        case 4: NPG_G32E(4); break;
        case 8: NPG_G32E(8); break;
        case 16: NPG_G32E(16); break;
        default: NPG_G32E(32); break;
    }
#undef NPG_G32E
    NPG_HIP(hipGetLastError());
    return NPG_OK;
}
// floats the gather-layout copy of one of A's input vectors takes (0: A has no windowed tile set to multiply it with)
int64_t gather32_floats(const npg_csr *Ap) {
    const npg_csr *A = spmv_form(Ap);
    if (!(A->nnode() > 0 && !A->pk9 && !A->uperm && A->wtile_ptr && A->n == A->m)) return 0;
    return 4 * gather32_nodes(A) + (A->n - A->block_rows()) + 8;
}

template <int L>
static void launch_spmv_g32(const npg_csr *A, const GatherMap &g, double *y, bool win) {
    const void *tiles = win ? (const void *)A->wtile_ptr : (const void *)A->tile_ptr;
    const int nt = win ? A->nwtiles : A->ntiles;
    const WinDev W = win_view(A);
    const dim3 grid(std::max(1, std::min<int>(nt, 3 * A->ctx->num_cu))), blk(kSpmvThreads);
    static const int diag = getenv("NPG_WIN_DIAG") ? atoi(getenv("NPG_WIN_DIAG")) : 0;      // tools/window_ab.py: timing diagnostics
    if (win && diag == 128) {
        unsigned long long *acc = nullptr, h[10];
        if (hipMalloc((void **)&acc, sizeof h) != hipSuccess) return;
        hipMemsetAsync(acc, 0, sizeof h, A->ctx->stream);
        hipLaunchKernelGGL((k_spmv_g32_timed<L, 4>), grid, blk, 0, A->ctx->stream, csr_view(A), W, A->wtile_ptr, nt, g, y, acc);
        hipMemcpyAsync(h, acc, sizeof h, hipMemcpyDeviceToHost, A->ctx->stream);
        hipStreamSynchronize(A->ctx->stream);
        hipFree(acc);
        static int said = 0;
        if (said++ % 64 == 3)
            fprintf(stderr, "windowed block tiles, s_memtime ticks per tile (thread 0's wave, %llu tiles): window write %.0f | loads issued %.0f | bookkeeping %.0f | barrier1 %.0f | gather+wait+products %.0f | "
                    "barrier2 %.0f | sums %.0f | barrier3 %.0f | stores %.0f\n", h[7], (double)h[8] / h[7], (double)h[9] / h[7], (double)h[0] / h[7], (double)h[1] / h[7], (double)h[2] / h[7],
                    (double)h[3] / h[7], (double)h[4] / h[7], (double)h[5] / h[7], (double)h[6] / h[7]);
        return;
    }
    if (win && diag) {
        switch (diag) {
            case 1: hipLaunchKernelGGL((k_spmv_g32<L, 4, 1>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 2: hipLaunchKernelGGL((k_spmv_g32<L, 4, 2>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 3: hipLaunchKernelGGL((k_spmv_g32<L, 4, 3>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 4: hipLaunchKernelGGL((k_spmv_g32<L, 4, 4>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 8: hipLaunchKernelGGL((k_spmv_g32<L, 4, 8>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 16: hipLaunchKernelGGL((k_spmv_g32<L, 4, 16>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 9: hipLaunchKernelGGL((k_spmv_g32<L, 4, 9>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 10: hipLaunchKernelGGL((k_spmv_g32<L, 4, 10>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 11: hipLaunchKernelGGL((k_spmv_g32<L, 4, 11>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 12: hipLaunchKernelGGL((k_spmv_g32<L, 4, 12>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 15: hipLaunchKernelGGL((k_spmv_g32<L, 4, 15>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 256: hipLaunchKernelGGL((k_spmv_g32<L, 4, 0>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;      // (the diagnostics' own baseline: 4 lanes per node)
            case 32: hipLaunchKernelGGL((k_spmv_g32<L, 4, 32>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 40: hipLaunchKernelGGL((k_spmv_g32<L, 4, 40>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 64: hipLaunchKernelGGL((k_spmv_g32<L, 4, 64>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            case 72: hipLaunchKernelGGL((k_spmv_g32<L, 4, 72>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
            default: hipLaunchKernelGGL((k_spmv_g32<L, 4, 7>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y); break;
        }
        return;
    }
    if (win && A->wlanes == 8)
        hipLaunchKernelGGL((k_spmv_g32<L, 8>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y);
    else if (win)
        hipLaunchKernelGGL((k_spmv_g32<L, 4>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y);
    else
        hipLaunchKernelGGL((k_spmv_g32<L, 0>), grid, blk, 0, A->ctx->stream, csr_view(A), W, tiles, nt, g, y);
}
}  // namespace npg

// y = A fl32(x): the product the Krylov kernels' gather-layout instance forms (npg_gmres_set_gather) as a call of its own - x
// is first copied, rounded to fp32, into the gather layout of spmv_device.h (a node's components padded to 16 bytes), then
// multiplied in fp64.  windowed != 0: on the matrix's windowed tile set (npg_csr_window_info; an error if it has none).
// `reps` > 1 repeats the product (timing loops: the copy is made once).
NPG_API int npg_spmv_gather32(const npg_csr *A, const npg_vec *x, npg_vec *y, int windowed, int reps) {
    NPG_REQUIRE(A && x && y, "npg_spmv_gather32: NULL argument");
    NPG_REQUIRE(x->n == A->n && y->n == A->m, "npg_spmv_gather32: A is %lld x %lld but x has %lld and y has %lld entries",
                (long long)A->m, (long long)A->n, (long long)x->n, (long long)y->n);
    NPG_REQUIRE(A->nnode() > 0 && !A->pk9 && !A->packed, "npg_spmv_gather32: the matrix is not stored by {c, K, C} node blocks");
    NPG_REQUIRE(!A->uperm, "npg_spmv_gather32: the matrix carries an internal renumbering (npg_csr_block_nodes_dofs)");
    NPG_REQUIRE(!windowed || A->wtile_ptr, "npg_spmv_gather32: the matrix has no windowed tile set");
    NPG_HIP(hipSetDevice(A->ctx->device));
    const int64_t nbr = A->block_rows(), need = 4 * gather32_nodes(A) + (A->n - nbr) + 8;
    float *buf = nullptr;
    NPG_HIP(hipMalloc((void **)&buf, (size_t)need * sizeof(float)));
    NPG_HIP(hipMemsetAsync(buf, 0, (size_t)need * sizeof(float), A->ctx->stream));
    const GatherMap g{buf, 3 * A->nfull, A->nfull, (int)nbr, (int)(4 * gather32_nodes(A) - nbr)};
    hipLaunchKernelGGL(k_fill_gather32, dim3((unsigned)std::min<int64_t>(4096, (A->n + 255) / 256)), dim3(256), 0, A->ctx->stream, x->d, g,
                       A->n);
    for (int r = 0; r < std::max(1, reps); ++r) switch (A->lanes) {
            case 4: launch_spmv_g32<4>(A, g, y->d, windowed != 0); break;
            case 8: launch_spmv_g32<8>(A, g, y->d, windowed != 0); break;
            case 16: launch_spmv_g32<16>(A, g, y->d, windowed != 0); break;
            default: launch_spmv_g32<32>(A, g, y->d, windowed != 0); break;
        }
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(A->ctx->stream);
    hipFree(buf);
    NPG_HIP(e);
    NPG_HIP(e2);
    return NPG_OK;
}

NPG_API int npg_csr_set_lanes(npg_csr *A, int lanes) {
    NPG_REQUIRE(A, "npg_csr_set_lanes: NULL handle");
    NPG_REQUIRE(lanes == 0 || lanes == 4 || lanes == 8 || lanes == 16 || lanes == 32, "npg_csr_set_lanes: %d lanes (4, 8, 16, 32 or 0)", lanes);
    npg_csr *S = const_cast<npg_csr *>(spmv_form(A));
    NPG_HIP(hipStreamSynchronize(A->ctx->stream));
    S->lanes_set = lanes != 0;
    S->lanes = lanes ? lanes : S->lanes_default;
    S->gen++;                    // captured graphs that launch this matrix's products bake the instance in
    return NPG_OK;
}

NPG_API int npg_spmv(const npg_csr *A, const npg_vec *x, npg_vec *y, double alpha, double beta) {
    NPG_REQUIRE(A && x && y, "npg_spmv: NULL argument");
    NPG_REQUIRE(x->n == A->n && y->n == A->m, "npg_spmv: A is %lld x %lld but x has %lld and y has %lld entries",
                (long long)A->m, (long long)A->n, (long long)x->n, (long long)y->n);
    NPG_REQUIRE(x->d != y->d, "npg_spmv: x and y must not alias");
    if (A->uperm) {         // npg_csr_block_nodes_dofs: vectors come and go in the caller's DoF order
        perm_gather(A, A->uvec[0], x->d);
        if (beta != 0.0) perm_gather(A, A->uvec[1], y->d);
        const int rc = spmv_raw(A, A->uvec[0], A->uvec[1], alpha, beta);
        if (rc) return rc;
        perm_scatter(A, y->d, A->uvec[1]);
        NPG_HIP(hipGetLastError());
        return NPG_OK;
    }
    return spmv_raw(A, x->d, y->d, alpha, beta);
}

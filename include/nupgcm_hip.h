/* nupgcm_hip.h - C ABI of libnupgcm_hip.so: the MI355X (gfx950) device layer behind nuPGCM's
 * Architecture / IterativeSolverToolkit / InversionToolkit / EvolutionToolkit surface.
 *
 * Each entry point names the reference interface it replaces (file:line under /root/reference/).  The reference reaches
 * its device through Julia dispatch on CuArray / CuSparseMatrixCSR (ext/nuPGCMCUDAExt.jl:24-33) and through Krylov.jl
 * (src/iterative_solvers.jl:58); a Julia package extension binds these symbols with `ccall`, and the
 * Python package `nupgcm_amd` binds the same symbols with ctypes.
 *
 * Conventions: plain C, opaque handles, every call returns NPG_OK (0) or a negative NPG_E* code and leaves a message in
 * npg_last_error().  All indices crossing the boundary are 0-based.  All floating point is fp64; device column indices
 * are int32.  Calls enqueue on the context's HIP stream and are host-synchronous only where they return host data.
 * A context is thread-compatible, not thread-safe.
 */
#ifndef NUPGCM_HIP_H
#define NUPGCM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPG_OK 0
#define NPG_EINVAL (-1)   /* bad argument / shape mismatch                                    */
#define NPG_EHIP (-2)     /* a HIP runtime call failed                                        */
#define NPG_ENOMEM (-3)   /* device or host allocation failed                                 */
#define NPG_ENODEV (-4)   /* no usable gfx950 device                                          */
#define NPG_ECOMM (-5)    /* RCCL failure                                                     */
#define NPG_EBLOWUP (-6)  /* blow-up guard tripped (src/model.jl:149-153)                     */

typedef struct npg_ctx npg_ctx;
typedef struct npg_vec npg_vec;
typedef struct npg_csr npg_csr;
typedef struct npg_gmres npg_gmres;
typedef struct npg_cg npg_cg;
typedef struct npg_fe npg_fe;
typedef struct npg_halo npg_halo;
typedef struct npg_precond npg_precond;
typedef struct npg_ilu0 npg_ilu0;
typedef struct npg_index npg_index;
typedef struct npg_fgmres npg_fgmres;

/* ---- context: replaces the implicit CUDA.jl device/stream (ext/nuPGCMCUDAExt.jl:8-16) ------------------------- */
int npg_ctx_create(int device, npg_ctx **out);
int npg_ctx_destroy(npg_ctx *ctx);
const char *npg_last_error(void);
int npg_ctx_sync(npg_ctx *ctx);
/* print_memory_status(::GPU)  (src/architectures.jl:20, ext/nuPGCMCUDAExt.jl:33) */
int npg_mem_status(npg_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
int npg_device_name(npg_ctx *ctx, char *buf, size_t cap);
/* the hipStream_t all work is enqueued on (so a caller can bracket it with HIP events) */
void *npg_ctx_stream(npg_ctx *ctx);
/* HIP-event timer on the context's stream: start ... stop -> elapsed milliseconds (host-synchronous at stop) */
int npg_timer_start(npg_ctx *ctx);
int npg_timer_stop(npg_ctx *ctx, double *ms);

/* ---- dense vectors: on_architecture(::GPU, ::Array) / (::CPU, ::CuArray), vector_type(::GPU, T)
 *      (ext/nuPGCMCUDAExt.jl:24-26,31) and the BLAS-1 / broadcast calls of SURVEY.md section 2a ------------------ */
int npg_vec_create(npg_ctx *ctx, int64_t n, npg_vec **out);              /* zero-filled */
int npg_vec_destroy(npg_vec *v);
/* non-owning window [offset, offset+n) of v (e.g. the velocity part x[1:nu] of [u; p], src/model.jl:314); the parent must
 * outlive the view */
int npg_vec_view(npg_vec *v, int64_t offset, int64_t n, npg_vec **out);
int64_t npg_vec_len(const npg_vec *v);
int npg_vec_upload(npg_vec *v, const double *host);
int npg_vec_download(const npg_vec *v, double *host);
/* v[i] = host[perm[i]]   -- `rhs[perm]` then H2D   (src/model.jl:274-275, src/evolution.jl:91-106) */
int npg_vec_upload_perm(npg_vec *v, const double *host, const int64_t *perm);
/* host[i] = v[perm[i]]   -- `solver.x[inv_perm]` then D2H   (src/model.jl:282,312) */
int npg_vec_download_perm(const npg_vec *v, double *host, const int64_t *perm);
int npg_vec_fill(npg_vec *v, double a);
int npg_vec_copy(npg_vec *dst, const npg_vec *src);
int npg_vec_axpby(npg_vec *y, double a, const npg_vec *x, double b);    /* y = a x + b y */
int npg_vec_dot(const npg_vec *x, const npg_vec *y, double *out);
int npg_vec_nrm2(const npg_vec *x, double *out);
/* max |x_i| and whether any entry is NaN: the blow-up guard's reductions (src/model.jl:149-153) */
int npg_vec_maxabs(const npg_vec *x, double *out, int *has_nan);
/* *is_constant = every entry equals *value (min == max).  Lets a binding recognise `Diagonal(s * ones(n))` - the inversion
 * preconditioner of src/inversion.jl:54 - without indexing a device vector, and pass it as NPG_PRECOND_SCALAR */
int npg_vec_is_constant(const npg_vec *x, double *value, int *is_constant);
/* y = sum_k coef[k] * xs[k]   (k < nterms <= 8): the fused broadcasts at src/inversion.jl:104, src/model.jl:278 */
int npg_vec_lincomb(npg_vec *y, int nterms, const double *coef, const npg_vec *const *xs);
/* y = d .* x   -- mul!(y, ::Diagonal, x) */
int npg_vec_mul(npg_vec *y, const npg_vec *d, const npg_vec *x);

/* ---- CSR matrices: on_architecture(::GPU, ::SparseMatrixCSC) = CuSparseMatrixCSR(a) and its inverse
 *      (ext/nuPGCMCUDAExt.jl:27-29) ------------------------------------------------------------------------------- */
/* CSC (Julia SparseMatrixCSC, made 0-based by the binding) -> device CSR.  drop_zeros != 0 removes entries that are
 * exactly 0.0 (Gridap stores structural zeros; a correct SpMV need not stream them). */
int npg_csr_create_from_csc(npg_ctx *ctx, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                            const double *nzval, int drop_zeros, npg_csr **out);
int npg_csr_create(npg_ctx *ctx, int64_t m, int64_t n, const int64_t *rowptr, const int32_t *colind,
                   const double *val, npg_csr **out);
int npg_csr_destroy(npg_csr *A);
int npg_csr_shape(const npg_csr *A, int64_t *m, int64_t *n, int64_t *nnz);
/* device CSR -> host CSC arrays (caller allocates n+1 / nnz / nnz) : on_architecture(::CPU, ::CuSparseMatrixCSR) */
int npg_csr_to_csc(const npg_csr *A, int64_t *colptr, int64_t *rowval, double *nzval);
/* host CSR copy (rowptr m+1 int64, colind nnz int32, val nnz) */
int npg_csr_download(const npg_csr *A, int64_t *rowptr, int32_t *colind, double *val);
/* new matrix with the pattern of A and values copied from it */
int npg_csr_clone(const npg_csr *A, npg_csr **out);
int npg_csr_zero_values(npg_csr *A);
/* out = a*X + b*(Y + Z) on identical patterns: `A = M + theta*(Kh + Kv)` (src/evolution.jl:144,162; src/model.jl:254)
 * done on the device instead of a host SparseMatrixCSC add + re-upload */
int npg_csr_combine(npg_csr *out, double a, const npg_csr *X, double b, const npg_csr *Y, const npg_csr *Z);
/* Store the velocity block of A_inversion node by node.  With constant viscosity node q couples to node c through one
 * friction number K (the x-x, y-y and z-z entries) and one Coriolis number C (x-y entry, -C for y-x)
 * (src/inversion.jl:183-192).  The ordering of nupgcm_amd.fe puts first the n_full nodes with all three components free
 * (rows 3q..3q+2), then the n_surf nodes with free x and y only (w = 0 at the surface; rows 3 n_full + 2 (q - n_full) + {0,1}).
 * Five (four) 12-byte CSR entries and five (four) 8-byte gathers become one 20-byte record {c, K, C} and one gather of the
 * node's contiguous components.  The structure is verified entry by entry (relative tolerance rtol); *blocked = 0 and the
 * matrix is untouched if it does not hold (e.g. function-valued nu).  A node-blocked matrix can be multiplied and solved
 * with, but not downloaded, cloned or re-assembled. */
int npg_csr_block_nodes(npg_csr *A, int64_t n_full, int64_t n_surf, double rtol, int *blocked);
/* (round 5; multi-GPU) Ghost NODES of a rank's row block - the matrix's columns [m, n) are the rank's ghosts: node g's (x, y[, z])
 * components are the ghost columns [first_col[g], first_col[g] + ncomp[g]), ncomp = 3 or 2.  Call before npg_csr_block_nodes:
 * the windowed tile set then stores the coupling of an owned node to a ghost node as ONE {c, K, C} node record instead of three
 * column records (a rank's boundary tiles keep their size), the gather-layout copy of the Krylov vector gives
 * ghost nodes 4-float slots behind the owned nodes' and the halo unpack fills them. */
int npg_csr_set_ghost_nodes(npg_csr *A, int64_t n_nodes, const int32_t *first_col, const int32_t *ncomp);
/* The same for a matrix in ANY DoF order - the reference's own RCM of the velocity mass-matrix graph comes out component by
 * component (src/dofs.jl:27-41,70-100), not node by node.  node_of_dof[i] >= 0: DoF i is component comp_of_dof[i] (0, 1, 2) of
 * the velocity node with that label (any non-negative labels: Gridap's node ids); < 0: not a velocity DoF.  The library
 * renumbers internally ([x, y, z of every node with three free components | x, y of every node with those two | other velocity
 * DoFs | the rest], nodes in the order of their first DoF in the caller's numbering), permutes the matrix, blocks it as
 * npg_csr_block_nodes does (windowed tile set included) and KEEPS THE PERMUTATION IN THE HANDLE: npg_spmv and npg_gmres_solve
 * go on taking and returning vectors in the caller's order (right-hand side, warm start / solution and a vector preconditioner
 * are gathered / scattered on the device, four vector passes per solve), so npg_vec_upload_perm / npg_vec_download_perm with
 * the caller's own permutations stay what they were.  Other solvers refuse such a matrix.  *blocked = 0 and the matrix
 * untouched if the structure does not hold. */
int npg_csr_block_nodes_dofs(npg_csr *A, const int64_t *node_of_dof, const int32_t *comp_of_dof, double rtol, int *blocked);
/* the two-component special case: npg_csr_block_nodes(A, 0, npairs, ...) */
int npg_csr_pair_xy(npg_csr *A, int64_t npairs, double rtol, int *paired);
/* how the matrix is laid out in HBM: number of block nodes, {c, K, C} records (20 bytes each, standing for 4 or 5 CSR
 * entries) and plain CSR entries (12 bytes each).  Plain matrix: 0, 0, nnz. */
int npg_csr_storage(const npg_csr *A, int64_t *nodes, int64_t *records, int64_t *csr_entries);
/* npg_csr_block_nodes also packs (a) what the rows BEHIND the block rows (the divergence rows of A_inversion, the last block row
 * of src/inversion.jl:183-192) hold in the block columns: one 28-byte record {c, d_x, d_y, d_z} per (row, column node) in
 * place of three (two) CSR entries (NPG_SPMV_COUPLING=0 keeps them as CSR entries); (b) what the block rows hold OUTSIDE the
 * block columns (the gradient entries, a rank's ghost columns): one 28-byte record {m, a_x, a_y, a_z} per (node, column) - the
 * coefficients of column m in the node's x, y, z rows - whenever that is fewer bytes than the entries (NPG_SPMV_COLUMN_RECORDS=0:
 * never).  *records = how many 28-byte records of both kinds (0: none). */
int npg_csr_coupling_records(const npg_csr *A, int64_t *records);
/* Record form for a matrix whose velocity block has NO {K, C} structure - function-valued viscosity, where the full-stress form
 * (src/inversion.jl:172-181) couples all nine component pairs of a node pair, and the eddy closure re-assembles the values
 * (src/model.jl:160-170).  The plain matrix A stays what it is (assembly target, download, clone ...); a companion in record form
 * is attached to it: FULL node records {c, a_00 .. a_22} (76 bytes and one gather of node c's components for nine entries of
 * 12 bytes and nine gathers), coupling records and column records as npg_csr_block_nodes builds them.  The companion knows where
 * in A's value array each of its values comes from and follows every change of A on the device (npg_fe_assemble_matrix,
 * npg_csr_combine, npg_csr_gather_values, npg_csr_zero_values); npg_spmv, npg_gmres_solve and the preconditioners read it.
 * *packed = 0 and nothing attached if the rows of some node would not fit an SpMV tile. */
int npg_csr_pack_nodes(npg_csr *A, int64_t n_full, int64_t n_surf, int *packed);
/* bytes of the matrix arrays one SpMV streams from HBM in the form the kernels read (records and their offset arrays, remaining
 * CSR entries, row offsets): plain CSR 12 nnz + 8 (m + 1); record forms as laid out by npg_csr_block_nodes / npg_csr_pack_nodes */
int npg_csr_spmv_bytes(const npg_csr *A, int64_t *matrix_bytes);
/* npg_csr_block_nodes also builds (NPG_SPMV_WINDOW=0: not) a WINDOWED tile set of the block rows for the Krylov kernels that
 * gather their SpMV input from its fp32 gather-layout copy (npg_gmres_set_gather): every tile lists its distinct column nodes
 * and distinct other columns, gathers each of them ONCE into an LDS window, and the node / column records address the window by
 * 16-bit indices; a node's record list is padded to an even count with a zero record so that two adjacent records of one row
 * node are summed before they reach LDS (csrc/spmv_window.h).  *tiles = tiles in that set (0: none), *block_tiles = how many of
 * them are windowed block tiles, *distinct = entries of both window lists (= gathers of the input one product issues in the
 * block rows), *matrix_bytes = what one product streams from HBM in that form (records with 2-byte indices, window lists,
 * offsets, descriptors, and the coupling records / CSR entries of the rows behind the block rows). */
int npg_csr_window_info(const npg_csr *A, int64_t *tiles, int64_t *block_tiles, int64_t *distinct, int64_t *matrix_bytes);
/* lanes per row in the tiled SpMV's segmented sums (4, 8, 16 or 32; 0 returns to the rule of thumb from the mean row length that
 * every matrix starts with).  A tuning knob: results change only in the order of a row's partial sums. */
int npg_csr_set_lanes(npg_csr *A, int lanes);
/* y = A fl32(x): the product of the Krylov kernels' gather-layout instance as a call of its own - x is copied, rounded to fp32,
 * into the gather layout (a node's components padded to 16 bytes) and multiplied in fp64 (synchronous).  windowed != 0: on the
 * windowed tile set (an error without one), 0: on the ordinary tiles.  reps > 1 repeats the product (timing).  Only for matrices
 * stored by {c, K, C} node blocks (npg_csr_block_nodes). */
int npg_spmv_gather32(const npg_csr *A, const npg_vec *x, npg_vec *y, int windowed, int reps);
/* dst.val[k] = src.val[map[k]]: a matrix whose entries are a fixed subset / rearrangement of another's - a rank's row block
 * of a replicated, re-assembled matrix (closure refreshes of K_v and A, src/model.jl:160-170,229-261, when the solve is
 * distributed).  npg_index = a device-resident int64 index array, every entry checked against `bound` at creation. */
int npg_index_create(npg_ctx *ctx, int64_t n, const int64_t *host, int64_t bound, npg_index **out);
int npg_index_destroy(npg_index *ix);
int npg_csr_gather_values(npg_csr *dst, const npg_csr *src, const npg_index *map);
/* Device-side pieces of the multigrid set-up / refresh (csrc/mg.hip), all on FIXED patterns supplied by the host:
 * Dinv <- inverse of the node-block diagonal of A[0:nu, 0:nu] (3 x 3 blocks for the first n_full nodes, 2 x 2 for the next
 * n_surf, 1 x 1 for the rest; Dinv's pattern holds exactly these blocks), and S <- D Dinv G (products outside S's pattern
 * are an error).  With npg_csr_gather_values for G = A[u, p] and D = A[p, u] a re-assembled A (eddy closure) refreshes a
 * level's smoother without leaving the device. */
int npg_csr_node_block_inverse(npg_csr *Dinv, const npg_csr *A, int64_t n_full, int64_t n_surf);
/* Dinv = inverse of a block diagonal of A[0:nu, 0:nu] with arbitrary blocks: block b = block_dofs[block_ptr[b] .. block_ptr[b+1])
 * (ascending; the blocks partition [0, nu); at most 136 unknowns each), Dinv's pattern = exactly the blocks.  The z-line smoother
 * of the multigrid cycle: a block = the velocity unknowns of the nodes above one another (new work, csrc/mg.hip). */
int npg_csr_line_block_inverse(npg_csr *Dinv, const npg_csr *A, const npg_index *block_ptr, const npg_index *block_dofs);
/* S = D Dinv G for a line-block Dinv (npg_csr_line_block_inverse) without the generic triple product's n_line-fold work: per
 * line l the dense W_l = B_l G[l, :] over the line's distinct pressure columns wcols[wptr[l] .. wptr[l+1]) (ascending), stored at
 * woff[l] (n_l x m_l, row-major; woff's bound = total + 1); then row p of S sums D[p, l] W_l over the lines it touches.  dperm:
 * D's entry positions sorted by (row, line, column); dpos: the position of each such entry's column inside its line; segment s =
 * entries [seg_start[s], seg_start[s+1]) of that order = one (row, line) pair, line seg_line[s]; row p owns segments
 * [seg_ptr[p], seg_ptr[p+1]).  All index arrays are the caller's (host logic on patterns: nupgcm_amd/multigrid.py). */
int npg_csr_line_schur(npg_csr *S, const npg_csr *D, const npg_csr *Dinv, const npg_csr *G, const npg_index *wptr,
                       const npg_index *wcols, const npg_index *woff, const npg_index *dperm, const npg_index *dpos,
                       const npg_index *seg_ptr, const npg_index *seg_line, const npg_index *seg_start);
/* the stored values of a plain-CSR matrix into the head of a vector, and values[k] = v[map[k]] back: with an ordinary halo plan on
 * such a vector, matrix VALUES travel between ranks (distributed multigrid: the rows of Dinv G that belong to a neighbour's ghost
 * unknowns are refreshed without leaving the device) */
int npg_csr_values_to_vec(const npg_csr *A, npg_vec *v);
int npg_csr_values_from_vec(npg_csr *A, const npg_vec *v, const npg_index *map);
/* C = A B on C's FIXED pattern (plain CSR; an error if a product falls outside it) */
int npg_csr_product(npg_csr *C, const npg_csr *A, const npg_csr *B);
int npg_csr_triple_product(npg_csr *S, const npg_csr *D, const npg_csr *Dinv, const npg_csr *G);
/* d[i] = 1 / A[i,i]   -- `Diagonal(1 ./ diag(A))` (src/evolution.jl:149,167; src/model.jl:256) */
int npg_csr_inv_diag(const npg_csr *A, npg_vec *d);
/* y = alpha * A x + beta * y   -- mul!(y, A, x) / A*x  (cuSPARSE SpMV in the reference) */
int npg_spmv(const npg_csr *A, const npg_vec *x, npg_vec *y, double alpha, double beta);

/* ---- Krylov solvers: Krylov.krylov_solve!(workspace, A, y, x; M=P, kwargs...) (src/iterative_solvers.jl:58) ---- */
#define NPG_PRECOND_NONE 0
#define NPG_PRECOND_SCALAR 1   /* P = Diagonal(s * ones(n))   (src/inversion.jl:54)          */
#define NPG_PRECOND_DIAG 2     /* P = Diagonal(d)             (src/evolution.jl:149,167)     */

typedef struct npg_solve_stats {
    int32_t solved;       /* workspace.stats.solved                                             */
    int32_t niter;        /* workspace.stats.niter (cumulative inner iterations)                */
    int32_t npass;        /* GMRES restart cycles started                                       */
    int32_t status;       /* 1 solved, 2 itmax reached, 3 breakdown, 4 zero residual at start   */
    int32_t nreorth;      /* GMRES: second Gram-Schmidt passes taken                            */
    int32_t nflagged;   /* GMRES: Arnoldi steps flagged by the device - a Pythagorean norm that lost > 4 digits (distributed), or a
                         * column that was due a second Gram-Schmidt pass while the fast kernels ran (the next solves use the
                         * full kernels); 0 for CG */
    double rnorm0;        /* || M r0 ||                                                         */
    double rnorm;         /* last residual estimate                                             */
    double seconds;       /* host wall time of the call                                         */
} npg_solve_stats;

/* GmresWorkspace(n, n, VT; memory)  (src/inversion.jl:84) */
int npg_gmres_create(npg_ctx *ctx, int64_t n, int memory, npg_gmres **out);
int npg_gmres_destroy(npg_gmres *ws);
/* Left-preconditioned restarted GMRES(memory).  x is in/out: the incoming x is the warm start (x aliases workspace.x in
 * the reference, src/iterative_solvers.jl:26-29).  itmax == 0 means 2 n.  Orthogonalisation is classical Gram-Schmidt
 * with a selective second pass (reorth_eta: take it when ||w'|| < eta ||w||; eta <= 0 never, eta >= 1 always) instead
 * of Krylov.jl's sequential modified Gram-Schmidt: one reduction per pass instead of j. */
int npg_gmres_solve(npg_gmres *ws, const npg_csr *A, int precond_kind, double precond_scalar,
                    const npg_vec *precond_diag, const npg_vec *y, npg_vec *x, double atol, double rtol,
                    int64_t itmax, double reorth_eta, npg_solve_stats *stats);
/* Profile mode: the next solves launch eagerly (no hipGraph) with HIP events on the context's stream around every
 * Arnoldi kernel - the fused SpMV + Gram-Schmidt-dots kernel that dominates the solve - and accumulate their durations
 * over complete restart cycles.  npg_gmres_get_profile returns the accumulated milliseconds and launch count. */
int npg_gmres_set_profile(npg_gmres *ws, int on);
/* kernel organisation of the Arnoldi step: 0 = fused (SpMV + Gram-Schmidt dots in one kernel, group-interleaved basis:
 * latency-bound sizes), 1 = split (SpMV kernel + row-streaming dots/orthogonalisation kernels, column-major basis:
 * bandwidth-bound sizes), -1 = by size (split from 8192 rows; default).  Same arithmetic either way. */
/* Stored Krylov basis of the split kernel organisation (n >= 8192): 64 = fp64, 32 = fp32 ("compressed basis": only the
 * stored copy used by the Gram-Schmidt sums and x += V y is rounded; sums, products and the restart residual stay fp64, and so
 * does the SpMV input unless its fp32 gather-layout copy is in use - npg_gmres_set_gather below, the default wherever it applies),
 * 0 = by tolerance (fp32 when rtol >= 1e-7, the reference's 1e-6 included).  NPG_GMRES_BASIS=32|64 overrides the default. */
int npg_gmres_set_basis(npg_gmres *ws, int bits);
/* Where the basis is stored in fp32 (above), with a node-blocked matrix, the Arnoldi kernel gathers its SpMV input
 * from an fp32 copy of the Krylov vector laid out for gathering - a node's components padded to 16 bytes: one gather per node
 * record instead of two (the kernel is bound by its gather instructions).  The copy carries the rounding the
 * stored basis column has anyway; products and sums stay fp64.  mode: -1 = default (on where it applies; NPG_GMRES_XG=0 turns
 * the default off), 0 = off, 1 = on where it applies, 2 = on, but on the matrix's ordinary tiles even when it has a windowed
 * tile set (npg_csr_window_info; NPG_GMRES_WINDOW=0 makes that the default). */
int npg_gmres_set_gather(npg_gmres *ws, int mode);
int npg_gmres_set_split(npg_gmres *ws, int mode);
int npg_gmres_get_profile(npg_gmres *ws, double *ms_total, int64_t *launches);
/* residual history of the last solve (workspace.stats.residuals with history=true): returns entries written */
int64_t npg_gmres_history(npg_gmres *ws, double *buf, int64_t cap);

/* CgWorkspace(n, n, VT)  (src/evolution.jl:120) */
int npg_cg_create(npg_ctx *ctx, int64_t n, npg_cg **out);
int npg_cg_destroy(npg_cg *ws);
int npg_cg_solve(npg_cg *ws, const npg_csr *A, int precond_kind, double precond_scalar, const npg_vec *precond_diag,
                 const npg_vec *y, npg_vec *x, double atol, double rtol, int64_t itmax, npg_solve_stats *stats);
int64_t npg_cg_history(npg_cg *ws, double *buf, int64_t cap);

/* ---- general preconditioners + flexible GMRES: src/preconditioners.jl:1-125 and SURVEY 8f rank 1 --------------------------
 * `P` of the IterativeSolverToolkit is a tagged union: NPG_PRECOND_NONE / _SCALAR / _DIAG are linear diagonal actions that
 * npg_gmres_solve / npg_cg_solve fold into their SpMV epilogue; an npg_precond is an INEXACT operator application
 * (inner iterations), which only a flexible Krylov method may use - iterative_solve! hands such a P to npg_fgmres_solve. */
#define NPG_PC_BLOCKDIAG 1   /* BlockDiagonalPreconditioner([Block(CgPreconditioner(A_k, Diagonal), indices_k)...])
                                (src/preconditioners.jl:53-125): nparts blocks                                        */
#define NPG_PC_MG 2          /* geometric multigrid V-cycle on the saddle-point system (new work): nparts levels      */
#define NPG_PC_DENSE 3       /* explicit dense inverse in HBM (new work; small systems, <= 46 340 unknowns: rocSOLVER indexes n x n with 32 bits): nparts = 1 */
int npg_precond_create(npg_ctx *ctx, int kind, int nparts, npg_precond **out);
int npg_precond_destroy(npg_precond *pc);
/* Block k acts on x[offset, offset + n_k): CG on A_k with M = Diagonal(jacobi) (ldiv = false), itmax / atol / rtol as given
 * (itmax 0 = 2 n_k), warm-started from the block's previous output - CgPreconditioner, src/preconditioners.jl:5-37.  The
 * matrices and vectors are borrowed and must outlive the preconditioner. */
int npg_precond_blockdiag_set(npg_precond *pc, int k, int64_t offset, const npg_csr *A_k, const npg_vec *jacobi,
                              int64_t itmax, double atol, double rtol);
/* ILU(0) of a plain-CSR square matrix with a full diagonal and ascending columns - KrylovPreconditioners.kp_ilu0(P) of the
 * reference's GPU P-block (src/preconditioners.jl:101-107; csrilu02 + two csrsv2 there): level-scheduled IKJ factorisation in
 * A's pattern and level-scheduled triangular solves, csrc/ilu.hip.  The handle keeps its own copy of the pattern. */
int npg_ilu0_create(npg_ctx *ctx, const npg_csr *A, npg_ilu0 **out);
int npg_ilu0_destroy(npg_ilu0 *M);
int npg_ilu0_refactor(npg_ilu0 *M, const npg_csr *A);                 /* A's values changed, same pattern */
int npg_ilu0_apply(npg_ilu0 *M, const npg_vec *r, npg_vec *z);        /* z = U^-1 L^-1 r  (ldiv!) */
int npg_ilu0_info(const npg_ilu0 *M, int64_t *levels_lower, int64_t *levels_upper, int64_t *nnz);
int npg_ilu0_factors(const npg_ilu0 *M, double *host_values);         /* L (strictly lower, unit diagonal implied) and U, A's pattern */
/* CG on A x = b with M = the factors, ldiv = true, warm start x: what the reference's CgPreconditioner runs for the P-block
 * (src/preconditioners.jl:24-37; Krylov.jl cg: gamma = r'z, stop at sqrt(gamma) <= atol + rtol sqrt(gamma_0); itmax 0 = 2 n) */
int npg_cg_ilu0_solve(npg_ilu0 *M, const npg_csr *A, const npg_vec *b, npg_vec *x, double atol, double rtol, int64_t itmax,
                      npg_solve_stats *stats);
/* block k of a block-diagonal preconditioner (already set with npg_precond_blockdiag_set) runs its CG with M = these factors
 * instead of the Jacobi vector (NULL: back to Jacobi).  Borrowed. */
int npg_precond_blockdiag_set_ilu0(npg_precond *pc, int k, npg_ilu0 *M);
/* Multigrid level `level` (0 = coarsest; set them coarse to fine).  A = the level's saddle-point matrix ordered [u; p]
 * with nu velocity unknowns, G = A[0:nu, nu:], D = A[nu:, 0:nu], Dinv = inverse of the node-block diagonal of A[0:nu, 0:nu]
 * (a node's components are adjacent; <= 3 entries per row), S = D Dinv G.  P (n_level x n_{level-1}) interpolates from the
 * next coarser level, R = P' restricts; both NULL at level 0.  All handles are borrowed. */
int npg_precond_mg_set_level(npg_precond *pc, int level, const npg_csr *A, int64_t nu, const npg_csr *G, const npg_csr *D,
                             const npg_csr *Dinv, const npg_csr *S, const npg_csr *P, const npg_csr *R);
/* replace the operators of a level by re-assembled ones of the same shapes (the eddy closure's refresh of A,
 * src/model.jl:160-170); the previous handles are no longer referenced afterwards */
/* Gh = Dinv G (same shape as G; borrowed, values kept current by the caller: npg_csr_product after every change of Dinv / G).
 * With it a smoothing step applies Dinv once instead of twice: x_u += Dinv (r_u - G dp) / w is formed as t / w (in the kernel that
 * forms t = Dinv r_u) minus (Dinv G) dp / w.  NULL returns to the two-product form. */
int npg_precond_mg_set_scaled_gradient(npg_precond *pc, int level, const npg_csr *Gh);
int npg_precond_mg_update_level(npg_precond *pc, int level, const npg_csr *A, const npg_csr *G, const npg_csr *D,
                                const npg_csr *Dinv, const npg_csr *S);
/* omega: scaling of the node-block diagonal in the Braess-Sarazin smoother (> largest eigenvalue of Dinv F; default 2.5),
 * jacobi_weight / schur_sweeps: damped-Jacobi relaxation of the pressure system (0.7, 3), nu1 / nu2: pre- / post-smoothing
 * steps (2, 2), coarse_sweeps: smoothing steps that stand in for the coarsest-level solve (20). */
int npg_precond_mg_set_params(npg_precond *pc, double omega, double jacobi_weight, int schur_sweeps, int nu1, int nu2,
                              int coarse_sweeps);
/* gamma = 1: V-cycle (default); gamma = 2: W-cycle - every level's coarse problem is visited twice, which keeps the
 * convergence independent of the number of levels where the V-cycle's degrades (4 levels: DESIGN.md section 4.5) */
int npg_precond_mg_set_cycle(npg_precond *pc, int gamma);
/* Mixed precision inside the cycle: its SpMVs read fp32 copies of the level operators' values (8 instead of 12 bytes per
 * CSR entry, 12 instead of 20 per node record); vectors, products, sums and the outer flexible GMRES (with its own fp64
 * SpMV) stay fp64.  npg_precond_mg_update_level rebuilds the copies. */
int npg_precond_mg_set_mixed(npg_precond *pc, int on);
/* NPG_PC_DENSE: A^-1 as n^2 doubles in HBM (2 GB at 16 k unknowns, 8 GB at 31 k: what 288 GB buy) - densified, factorised
 * and inverted once by rocSOLVER (getrf + getri, set-up), applied per solve by a hand-written split-column GEMV at HBM speed.
 * For the reference's small meshes, where the scalar-preconditioned GMRES is bound by kernel latency (19 us x 600 iterations),
 * this is the device counterpart of its CPU() path's `lu(A)` + `ldiv!` (src/inversion.jl:55-58, src/iterative_solvers.jl:42-47);
 * behind flexible GMRES it needs 1-2 iterations.  A must be plain CSR.  Call again to follow a re-assembled A. */
int npg_precond_dense_set(npg_precond *pc, const npg_csr *A, int fp32_storage);
/* fp32_storage != 0 keeps the inverse rounded to fp32 (half the bytes per application; products still accumulate in fp64):
 * one application then carries ~1e-7 relative error and flexible GMRES takes 2-3 iterations at tight tolerances.
 * multigrid: solve the coarsest level with its dense inverse instead of smoothing steps; on = 1: fp64 storage, on = 2: fp32
 * (a coarse-grid correction inside a preconditioner does not need more), on = 3: fp16 storage, every column divided by its
 * largest magnitude (a quarter of the bytes; products and chunk sums in fp32), on = 0: back to smoothing steps */
int npg_precond_mg_set_coarse_dense(npg_precond *pc, int on);
/* z = M^-1 r (one application: one V-cycle / one round of inner CG solves) */
int npg_precond_apply(npg_precond *pc, const npg_vec *r, npg_vec *z);
int npg_precond_counters(npg_precond *pc, int64_t *applications, int64_t *inner_iterations);
/* Bytes ONE application of the preconditioner streams as its operators are laid out in HBM (multigrid: every product of the
 * V-cycle with the matrix in the form its kernel reads - records, windowed tiles, fp32 / fp16 value copies - plus its vectors;
 * dense inverse: n^2 values in their storage type).  Counted on the host when the cycle is first enqueued (0 before that); the
 * numerator of the multigrid roofline in bench.py.  Distributed levels are not counted (0). */
int npg_precond_cycle_bytes(npg_precond *pc, int64_t *bytes);

/* Right-preconditioned flexible GMRES(memory), restarted.  x in/out (warm start, as npg_gmres_solve).  pc may be NULL.
 * Stopping rule of the reference with its Diagonal(scale) preconditioner made explicit: scale ||y - A x|| <= atol + rtol
 * scale ||y - A x0||  (src/inversion.jl:42-54,76; Krylov.jl gmres!).  stats->rnorm0 / rnorm are scaled residual norms; the
 * returned rnorm is the TRUE residual recomputed after the last pass.  itmax == 0 means 2 n. */
int npg_fgmres_create(npg_ctx *ctx, int64_t n, int memory, npg_fgmres **out);
int npg_fgmres_destroy(npg_fgmres *ws);
/* Distributed flexible GMRES: A = this rank's rows (columns [owned | ghosts]), x = [owned | ghosts]; SpMV inputs get their
 * ghosts through `h`, reductions are summed over the ranks.  With npg_precond_mg_set_level_dist: the multigrid whose finest
 * level is row-partitioned like the system and whose coarser levels are replicated (one coarse-vector all-reduce per cycle). */
int npg_fgmres_set_halo(npg_fgmres *ws, npg_halo *h);
int npg_precond_mg_set_level_dist(npg_precond *pc, int level, const npg_csr *A, int64_t nu_owned, const npg_csr *G,
                                  const npg_csr *D, const npg_csr *Dinv, const npg_csr *S, const npg_csr *P, const npg_csr *R,
                                  npg_halo *hx, npg_halo *hu, npg_halo *hp);
/* Two distributed levels: the level below a distributed level may be distributed as well (the finest one or two levels of a
 * hierarchy).  Set the coarser one first - npg_precond_mg_set_level_dist with its transfers to the replicated level below it -,
 * then the finer one with P = R = NULL, then the transfers between the two: P = this rank's rows of the prolongation over the
 * coarser level's [owned | ghosts of hP] columns, R = the coarser level's owned rows of the restriction over the finer level's
 * [owned | ghosts of hR] columns; hP / hR are halo plans on the coarser iterate / the finer residual. */
int npg_precond_mg_set_transfer_dist(npg_precond *pc, int level, const npg_csr *P, const npg_csr *R, npg_halo *hP, npg_halo *hR);
int npg_fgmres_solve(npg_fgmres *ws, const npg_csr *A, npg_precond *pc, const npg_vec *y, npg_vec *x, double scale,
                     double atol, double rtol, int64_t itmax, npg_solve_stats *stats);
int64_t npg_fgmres_history(npg_fgmres *ws, double *buf, int64_t cap);

/* ---- element-local finite-element kernels: Gridap.assemble_vector / assemble_matrix call sites ------------------
 * The host supplies what Gridap holds: cell geometry, per-cell DoF tables and the quadrature / shape tables, so the
 * device uses the same rule as `Measure(Omega, 4)` (src/meshes.jl:33).  A DoF table entry >= 0 is an index into the
 * *device vector the field lives in* (i.e. already composed with the RCM permutation the solvers use,
 * src/dofs.jl:27-41); an entry < 0 is Dirichlet value number (-1 - entry) of the `diri` array. */
typedef struct npg_fe_desc {
    int64_t ncell;
    int32_t nq;               /* quadrature points per cell                                                    */
    int32_t nloc_b;           /* buoyancy nodes per cell: 10 (P2) or 4 (P1)                                    */
    const double *grad_lambda;/* [ncell][4][3] physical gradients of the barycentric coordinates              */
    const double *wdet;       /* [ncell] |det J|                                                               */
    const double *qw;         /* [nq] quadrature weights on the reference cell                                 */
    const double *N2;         /* [nq][10] P2 shape values                                                      */
    const double *dN2;        /* [nq][10][4] P2 shape derivatives w.r.t. barycentric coordinates               */
    const double *Nb;         /* [nq][nloc_b] buoyancy shape values (== N2 when nloc_b == 10)                  */
    const double *dNb;        /* [nq][nloc_b][4]                                                               */
    const double *N1;         /* [nq][4] P1 shape values (pressure)                                            */
    const int32_t *cell_u;    /* [ncell][10][3] velocity DoF table (into the inversion solution vector)        */
    const int32_t *cell_p;    /* [ncell][4] pressure DoF table (into the inversion solution vector)            */
    const int32_t *cell_b;    /* [ncell][nloc_b] buoyancy DoF table (into the evolution solution vector)       */
    const double *u_diri;     /* Dirichlet values referenced by cell_u                                          */
    int64_t n_u_diri;
    const double *b_diri;     /* Dirichlet values referenced by cell_b                                          */
    int64_t n_b_diri;
    int64_t n_inv;            /* length of the inversion vector [u; p]                                          */
    int64_t n_b;              /* length of the buoyancy vector                                                  */
} npg_fe_desc;

int npg_fe_create(npg_ctx *ctx, const npg_fe_desc *desc, npg_fe **out);
int npg_fe_destroy(npg_fe *fe);
/* per-cell, per-quadrature-point coefficient tables [ncell][nq], pre-evaluated by the host from the user's closures
 * (nu, kappa_h, kappa_v, f): the device never runs user code.  name in {"nu","kappa_h","kappa_v","f"} */
int npg_fe_set_coeff(npg_fe *fe, const char *name, const double *values);

/* Arithmetic of the element-LOCAL work of every assembly kernel below (shape tables, geometry, nodal values, the integrand
 * at a quadrature point).  NPG_FE_FP64 (default) matches Gridap.  NPG_FE_FP32 is the mixed mode of BASELINE.json configs[4]
 * ("mixed fp32 assembly / fp64 solve"; new work - the reference assembles in Float64 only): local products in fp32, every
 * sum over quadrature points / cells, all stored matrices and vectors, and the solvers in fp64.  Entries then carry a
 * relative error of a few 1e-7 of the largest local entry. */
#define NPG_FE_FP64 0
#define NPG_FE_FP32 1
int npg_fe_set_precision(npg_fe *fe, int precision);
int npg_fe_get_precision(const npg_fe *fe);

#define NPG_BDF1 1
#define NPG_BDF2 2
/* y = assemble_vector(d -> advection_lform(d, b, b_prev, u, u_prev, ...), B_test)[perm]
 *       + theta*rhs_diff + dt*rhs_flux - (rhs_M + theta*(rhs_h + rhs_v))
 * i.e. src/model.jl:269-278 in one call, entirely on the device.  x_inv / x_inv_prev are the inversion solution vectors
 * [u; p] (current and previous step), b / b_prev the evolution solution vectors.  Any of the five rhs_* may be NULL. */
int npg_fe_evolution_rhs(npg_fe *fe, int scheme, double dt, double N2, double theta, const npg_vec *b,
                         const npg_vec *b_prev, const npg_vec *x_inv, const npg_vec *x_inv_prev,
                         const npg_vec *rhs_diff, const npg_vec *rhs_flux, const npg_vec *rhs_M, const npg_vec *rhs_h,
                         const npg_vec *rhs_v, npg_vec *y);
/* advection part alone (for parity tests of src/model.jl:292-300) */
int npg_fe_advection_rhs(npg_fe *fe, int scheme, double dt, double N2, const npg_vec *b, const npg_vec *b_prev,
                         const npg_vec *x_inv, const npg_vec *x_inv_prev, npg_vec *out);

/* Matrix (re)assembly into an existing CSR pattern (values are overwritten).
 * which: */
#define NPG_MAT_M 1        /* build_M        src/evolution.jl:209-212 ; lift = rhs_M                         */
#define NPG_MAT_KH 2       /* build_Kh       src/evolution.jl:225-228 ; coefficient "kappa_h" ; lift = rhs_h */
#define NPG_MAT_KV 3       /* build_Kv       src/evolution.jl:240-246 ; coefficient "kappa_v" ; lift = rhs_v */
#define NPG_MAT_A 4        /* build_A_inversion(!) src/inversion.jl:133-192, coefficients "nu", "f"          */
#define NPG_MAT_B 5        /* build_B_inversion    src/inversion.jl:199-219 ; lift = Dirichlet part of b0    */
/* a2e2 = alpha^2 eps^2 (A), scale = 1/alpha (B); full_stress != 0 selects the sigma(u):sigma(v) form used when nu is
 * a function (src/inversion.jl:172-181).  lift (may be NULL) receives sum_j a(phi_j^D, phi_i) b_D,j, the Dirichlet
 * correction vector of build_matrix_vector (src/evolution.jl:256-260) / build_b_inversion (src/inversion.jl:241). */
int npg_fe_assemble_matrix(npg_fe *fe, int which, double scale, int full_stress, npg_csr *A, npg_vec *lift);
/* rhs_diff = -N2 int kappa_v d_z(d)   (src/evolution.jl:269-278) */
int npg_fe_assemble_rhs_diff(npg_fe *fe, double N2, npg_vec *out);
/* kappa_v <- kappa_v0 + kappa_c (1 + tanh(-alpha (N2 + d_z b) / N2min)) / 2 at every quadrature point
 * (kappa_v_convection, src/inputs.jl:87-91, called at src/model.jl:229-232) */
int npg_fe_update_kappa_convection(npg_fe *fe, const double *kappa_v0_host_or_null, double kappa_c, double N2min,
                                   double alpha, double N2, const npg_vec *b);
/* nu <- LogSumExp(nu_min, f^2 / sqrt(N2min^2 + (alpha (N2 + d_z b))^2)) (nu_eddy, src/inputs.jl:130-137) */
int npg_fe_update_nu_eddy(npg_fe *fe, double N2min, double alpha, double N2, double smoothing, double nu_min,
                          const npg_vec *b);
/* Coefficient `name` of `coarse` <- the volume-weighted average, over each coarse cell's eight children (cells 8c .. 8c+7 of the
 * uniformly refined mesh `fine` lives on), of the quadrature mean of `fine`'s table: the multigrid hierarchy's coarse levels follow
 * a closure (nu_eddy, src/inputs.jl:130-137; refreshed at src/model.jl:160-170) through the FINE level's viscosity instead of
 * re-evaluating the non-linear closure on an injected buoyancy.  No counterpart in the reference (it has no multigrid). */
int npg_fe_restrict_coeff(npg_fe *coarse, const npg_fe *fine, const char *name);
/* out[c] = quadrature mean of coefficient `name` over cell c (out: one entry per cell of the engine's mesh): what a rank of a
 * partitioned multigrid level contributes to that average - its cells' children live on other ranks too, so the sums over children
 * are formed on the host from the ranks' cell means (nupgcm_amd/partition.py). */
int npg_fe_coeff_cell_mean(const npg_fe *fe, const char *name, npg_vec *out);
/* CFL:  min_K h_K / max(max_q |u|, u_min)   (update_dt!, src/timesteppers.jl:108-119) */
int npg_fe_cfl_ratio(npg_fe *fe, const double *h_cells_host, double u_min, const npg_vec *x_inv, double *out);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (new work: the reference is single-device) ----------------- */
#define NPG_UNIQUE_ID_BYTES 128
int npg_comm_unique_id(void *id128);                                  /* rank 0 makes it, the launcher broadcasts it */
int npg_comm_init(npg_ctx *ctx, const void *id128, int rank, int nranks);
/* one line of JSON about the communicator: rank, nranks, rccl_ranks (ncclCommCount), device, in_cycle_transport */
int npg_comm_info(npg_ctx *ctx, char *buf, size_t cap);
/* auto transport only: drop the peer windows, RCCL carries the in-cycle traffic from here on (no live halo plans) */
int npg_comm_disable_peer(npg_ctx *ctx);
int npg_comm_allreduce_sum(npg_ctx *ctx, double *host_inout, int n);  /* tiny host-side helper for tests/bench */
int npg_comm_allreduce_vec(npg_ctx *ctx, npg_vec *v);                 /* <= 32 doubles, in place, on the context's stream */
/* Replicate a row-block distributed vector on every rank: segment s of `full` ([global_off, global_off + len)) is owned
 * by rank seg_rank[s], who holds it at local[local_off ..].  One grouped ncclBroadcast per segment over xGMI. */
int npg_comm_allgather_segments(npg_ctx *ctx, const npg_vec *local, int nseg, const int32_t *seg_rank,
                                const int64_t *seg_local_off, const int64_t *seg_global_off, const int64_t *seg_len,
                                npg_vec *full);
/* Halo plan for a row-block distributed CSR: this rank owns n_owned rows; columns >= n_owned of the local matrix are
 * ghosts filled from neighbours.  send_idx lists owned entries to ship to each peer, recv goes to consecutive ghost
 * slots. */
int npg_halo_create(npg_ctx *ctx, int64_t n_owned, int64_t n_ghost, int npeers, const int32_t *peer_rank,
                    const int64_t *send_ptr, const int32_t *send_idx, const int64_t *recv_ptr, npg_halo **out);
int npg_halo_destroy(npg_halo *h);
int npg_halo_exchange(npg_halo *h, npg_vec *x_with_ghosts);
/* Attach a plan to a workspace created for n = n_owned: npg_gmres_solve / npg_cg_solve then take A as the rank's
 * n_owned x (n_owned + n_ghost) row block, y with n_owned and x with n_owned + n_ghost entries, exchange the interface
 * before every SpMV and all-reduce the inner products (two 32-double messages per GMRES iteration). */
int npg_gmres_set_halo(npg_gmres *ws, npg_halo *h);
/* Distributed cycles (new work, no reference counterpart).  overlap (default 1): the split cycle (>= 8192 rows per rank)
 * runs the tiles of the row block that read no ghost column while the halo exchange of the Arnoldi vector is in flight
 * on the plan's own stream, the others behind it; 0 exchanges first.  graph (default 0): replay the cycle - RCCL calls
 * included - from a hipGraph instead of launching it eagerly (RCCL transport only; exercised on a one-rank communicator
 * so far).  Same iterates in every combination (tests/rccl_selftest_worker.py). */
int npg_gmres_set_dist_options(npg_gmres *ws, int overlap, int graph);
int npg_cg_set_halo(npg_cg *ws, npg_halo *h);

#ifdef __cplusplus
}
#endif
#endif /* NUPGCM_HIP_H */

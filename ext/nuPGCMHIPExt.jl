# nuPGCMHIPExt.jl - the package extension a nuPGCM maintainer adds in place of ext/nuPGCMCUDAExt.jl (load ONE of the two:
# both define the `GPU()` methods of on_architecture / vector_type).  It binds the C ABI of include/nupgcm_hip.h.
#
# STATUS: never executed - the build image has no Julia.  Every method below has a line-for-line twin in
# tests/test_gpu_julia_mirror.py which drives the SAME entry points in the SAME order with the SAME argument conventions
# (1-based -> 0-based shifts, CSC input, native-order vectors + host permutations) from reference-shaped inputs, on the GPU.
#
# How it hooks in without touching the reference's toolkits (dispatch points, reference file:line):
#   * array conversion ........ the ten methods of ext/nuPGCMCUDAExt.jl:24-33 for HIPVector / HIPSparseMatrixCSR
#   * Krylov workspaces ....... src/inversion.jl:84, src/evolution.jl:120 and src/preconditioners.jl:19 call
#                               `Krylov.GmresWorkspace(N, N, VT; memory)` / `Krylov.CgWorkspace(N, N, VT)` with
#                               VT = vector_type(arch, T): methods of those two constructors for VT = HIPVector{Float64}
#                               return the device-resident workspaces of libnupgcm_hip (`workspace.x .= 0` then lowers to
#                               fill!, defined here)
#   * iterative_solve! ........ method for IterativeSolverToolkit{<:HIPSparseMatrixCSR} (src/iterative_solvers.jl:31-68):
#                               one ccall per solve
#   * invert!(inversion, b) ... method for InversionToolkit{<:HIPSparseMatrixCSR} (src/inversion.jl:101-110): y = B b + b0
#                               as one copy + one SpMV (no device broadcast machinery needed)
#   * evolve!(model, ...) ..... method for models whose inversion lives on the HIP device (src/model.jl:213-285): the
#                               advection assembly + right-hand-side combination of :269-278 - a serial Gridap loop on the
#                               host in the reference, also in GPU mode - becomes one call of npg_fe_evolution_rhs; the
#                               convection closure (:229-247: kappa_v from b, K_v and its lift, rhs_diff) and the LHS /
#                               Jacobi refresh (:251-261) run on the device too (npg_fe_update_kappa_convection,
#                               npg_fe_assemble_matrix, npg_fe_assemble_rhs_diff, npg_csr_combine, npg_csr_inv_diag)
#   * update_Δt! .............. src/timesteppers.jl:108-119 for the model this extension serves: the per-cell maximum of |u|
#                               over the quadrature points and the minimum over the cells are ONE device reduction over the
#                               inversion solution already in HBM (npg_fe_cfl_ratio)
#   * Model(...) .............. the constructors (src/model.jl:47-62) are where fe_data and the device matrix meet: the two
#                               methods below - typed exactly like the reference's two, but for GPU and a HIP inversion
#                               toolkit, hence strictly more specific - hand A_inversion (uploaded in the reference's own
#                               order, src/dofs.jl:27-41) to npg_csr_block_nodes_dofs with Gridap's (node, component) of every
#                               velocity DoF, so that the solves run on the library's record / windowed-tile layout while
#                               every vector keeps the reference's order (the permutation lives inside the matrix handle),
#                               then `invoke` the reference's own constructor by its exact signature
#   * build_A_inversion! ...... the eddy closure's re-assembly every tenth step (src/model.jl:160-170 -> src/inversion.jl:149-170)
#                               for the model this extension serves: nu_eddy at the quadrature points and the full-stress
#                               matrix on the device (npg_fe_update_nu_eddy, npg_fe_assemble_matrix; the record-form
#                               companion of npg_csr_pack_nodes follows by itself); the upload that run! issues next
#                               (`on_architecture(arch, A_inversion[perm, perm])`, :168) is handed the refreshed device matrix
#   * P_block_setup, mul! ..... src/preconditioners.jl:24-37,101-125 (switched off at src/inversion.jl:60): the GPU P-block's
#                               ILU(0) + CG and the T-block's Jacobi-CG as library calls on device views of the blocks
#
# Dispatch argument for every hooked call site (which method wins and why): INTEGRATION.md, "Julia side".
module nuPGCMHIPExt

using nuPGCM
using SparseArrays, LinearAlgebra
import Krylov
import Gridap

const lib = get(ENV, "NUPGCM_HIP_LIB", "libnupgcm_hip.so")

struct HIPError <: Exception
    code::Cint
    msg::String
end
check(rc::Cint) = rc == 0 ? nothing : throw(HIPError(rc, unsafe_string(@ccall lib.npg_last_error()::Cstring)))

# ---- context ---------------------------------------------------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function ctx()
    if CTX[] == C_NULL
        out = Ref{Ptr{Cvoid}}()
        check(@ccall lib.npg_ctx_create(parse(Cint, get(ENV, "LOCAL_RANK", "0"))::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
        CTX[] = out[]
    end
    return CTX[]
end

# ---- device types ------------------------------------------------------------------------------------------------------
mutable struct HIPVector{T} <: AbstractVector{T}
    h::Ptr{Cvoid}
    n::Int
    parent::Any                 # nothing, or the vector a view handle aliases (kept alive as long as the view)
    function HIPVector{Float64}(::UndefInitializer, n::Integer)          # npg_vec_create zero-fills
        out = Ref{Ptr{Cvoid}}()
        check(@ccall lib.npg_vec_create(ctx()::Ptr{Cvoid}, n::Int64, out::Ptr{Ptr{Cvoid}})::Cint)
        v = new{Float64}(out[], n, nothing)
        finalizer(x -> @ccall(lib.npg_vec_destroy(x.h::Ptr{Cvoid})::Cint), v)
    end
    # device view of p[r] (npg_vec_view: no copy; destroying a view frees the handle only)
    function HIPVector{Float64}(p::HIPVector{Float64}, r::AbstractUnitRange{<:Integer})
        checkbounds(p, r)
        out = Ref{Ptr{Cvoid}}()
        check(@ccall lib.npg_vec_view(p.h::Ptr{Cvoid}, (first(r) - 1)::Int64, length(r)::Int64, out::Ptr{Ptr{Cvoid}})::Cint)
        v = new{Float64}(out[], length(r), p)
        finalizer(x -> @ccall(lib.npg_vec_destroy(x.h::Ptr{Cvoid})::Cint), v)
    end
end
Base.size(v::HIPVector) = (v.n,)
Base.similar(v::HIPVector{Float64}) = HIPVector{Float64}(undef, v.n)
Base.summary(v::HIPVector) = "$(v.n)-element HIPVector{Float64}"
Base.show(io::IO, ::MIME"text/plain", v::HIPVector) = print(io, summary(v))
Base.show(io::IO, v::HIPVector) = print(io, summary(v))                 # (the AbstractArray fallback would index element by element)
Base.getindex(::HIPVector, ::Int) = error("scalar indexing of a HIPVector is not supported; use on_architecture(CPU(), v)")
# `@view x[block.indices]` of src/preconditioners.jl:118-125 is a lazy SubArray (it only needs size / axes); the methods that
# receive one turn it into a device view
const HIPView = SubArray{Float64, 1, HIPVector{Float64}, <:Tuple{AbstractUnitRange{<:Integer}}}
const HIPVecOrView = Union{HIPVector{Float64}, HIPView}
dev(v::HIPVector{Float64}) = v
dev(v::HIPView) = HIPVector{Float64}(parent(v), parentindices(v)[1])

mutable struct HIPSparseMatrixCSR{T} <: AbstractSparseMatrix{T, Int32}
    h::Ptr{Cvoid}
    m::Int
    n::Int
end
Base.size(A::HIPSparseMatrixCSR) = (A.m, A.n)
Base.summary(A::HIPSparseMatrixCSR) = "$(A.m)×$(A.n) HIPSparseMatrixCSR{Float64}"
Base.show(io::IO, ::MIME"text/plain", A::HIPSparseMatrixCSR) = print(io, summary(A))
Base.show(io::IO, A::HIPSparseMatrixCSR) = print(io, summary(A))

# ---- the ten methods of ext/nuPGCMCUDAExt.jl:24-33 ------------------------------------------------------------------------
function nuPGCM.on_architecture(::GPU, a::Array{Float64})
    v = HIPVector{Float64}(undef, length(a))
    check(@ccall lib.npg_vec_upload(v.h::Ptr{Cvoid}, a::Ptr{Float64})::Cint)
    return v
end
function nuPGCM.on_architecture(::CPU, v::HIPVector)
    a = Vector{Float64}(undef, v.n)
    check(@ccall lib.npg_vec_download(v.h::Ptr{Cvoid}, a::Ptr{Float64})::Cint)
    return a
end
nuPGCM.on_architecture(::GPU, v::HIPVector) = v
function upload_csc(A::SparseMatrixCSC{Float64, Int64}, drop_zeros::Integer)
    out = Ref{Ptr{Cvoid}}()
    check(@ccall lib.npg_csr_create_from_csc(ctx()::Ptr{Cvoid}, size(A, 1)::Int64, size(A, 2)::Int64,
                                              (A.colptr .- 1)::Ptr{Int64}, (A.rowval .- 1)::Ptr{Int64},
                                              A.nzval::Ptr{Float64}, drop_zeros::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    M = HIPSparseMatrixCSR{Float64}(out[], size(A)...)
    finalizer(x -> @ccall(lib.npg_csr_destroy(x.h::Ptr{Cvoid})::Cint), M)
end
# the device matrix the extension's build_A_inversion! has just re-assembled in place (see there): the next upload of a matrix
# of its size - `on_architecture(model.arch, A_inversion[perm, perm])`, src/model.jl:168 - returns it instead of uploading
const PENDING_A = Ref{Any}(nothing)
function nuPGCM.on_architecture(::GPU, A::SparseMatrixCSC{Float64, Int64})     # CuSparseMatrixCSR(a): CSC -> CSR on upload
    P = PENDING_A[]
    if P !== nothing && size(P) == size(A)
        PENDING_A[] = nothing
        return P
    end
    # drop_zeros = 0, as CuSparseMatrixCSR(a) keeps them: Gridap stores structural zeros (32 % of a constant-viscosity A_inversion) and
    # the reference RE-ASSEMBLES into the pattern it gets back from on_architecture(CPU(), .) (src/model.jl:112-113,166) - a dropped
    # entry would be a missing slot there.  The constant-viscosity fast path sheds them itself (npg_csr_block_nodes_dofs).
    return upload_csc(A, 0)
end
function nuPGCM.on_architecture(::CPU, A::HIPSparseMatrixCSR)                 # SparseMatrixCSC(a)
    nnz = Ref{Int64}()
    check(@ccall lib.npg_csr_shape(A.h::Ptr{Cvoid}, C_NULL::Ptr{Int64}, C_NULL::Ptr{Int64}, nnz::Ptr{Int64})::Cint)
    colptr, rowval, nzval = Vector{Int64}(undef, A.n + 1), Vector{Int64}(undef, nnz[]), Vector{Float64}(undef, nnz[])
    check(@ccall lib.npg_csr_to_csc(A.h::Ptr{Cvoid}, colptr::Ptr{Int64}, rowval::Ptr{Int64}, nzval::Ptr{Float64})::Cint)
    return SparseMatrixCSC(A.m, A.n, colptr .+ 1, rowval .+ 1, nzval)
end
nuPGCM.on_architecture(::GPU, A::HIPSparseMatrixCSR) = A
nuPGCM.architecture(::HIPVector) = GPU()
nuPGCM.architecture(::HIPSparseMatrixCSR) = GPU()
nuPGCM.vector_type(::GPU, T) = HIPVector{T}
function nuPGCM.print_memory_status(::GPU)
    f, t = Ref{Csize_t}(), Ref{Csize_t}()
    check(@ccall lib.npg_mem_status(ctx()::Ptr{Cvoid}, f::Ptr{Csize_t}, t::Ptr{Csize_t})::Cint)
    println("GPU memory usage: ", round((t[] - f[]) / 2^30, digits = 3), " GiB / ", round(t[] / 2^30, digits = 3), " GiB")
end

# ---- what the reference's GPU path does with the device objects (SURVEY.md 8b) --------------------------------------------
function LinearAlgebra.mul!(y::HIPVector, A::HIPSparseMatrixCSR, x::HIPVector, α::Number = true, β::Number = false)
    check(@ccall lib.npg_spmv(A.h::Ptr{Cvoid}, x.h::Ptr{Cvoid}, y.h::Ptr{Cvoid}, Float64(α)::Float64, Float64(β)::Float64)::Cint)
    return y
end
Base.:*(A::HIPSparseMatrixCSR, x::HIPVector) = mul!(HIPVector{Float64}(undef, A.m), A, x)
LinearAlgebra.dot(x::HIPVector, y::HIPVector) =
    (o = Ref{Float64}(); check(@ccall lib.npg_vec_dot(x.h::Ptr{Cvoid}, y.h::Ptr{Cvoid}, o::Ptr{Float64})::Cint); o[])
LinearAlgebra.norm(x::HIPVector) =
    (o = Ref{Float64}(); check(@ccall lib.npg_vec_nrm2(x.h::Ptr{Cvoid}, o::Ptr{Float64})::Cint); o[])
Base.fill!(x::HIPVector, a) = (check(@ccall lib.npg_vec_fill(x.h::Ptr{Cvoid}, Float64(a)::Float64)::Cint); x)
Base.copyto!(y::HIPVector, x::HIPVector) = (check(@ccall lib.npg_vec_copy(y.h::Ptr{Cvoid}, x.h::Ptr{Cvoid})::Cint); y)
Base.copy(x::HIPVector) = copyto!(similar(x), x)
LinearAlgebra.axpby!(a::Number, x::HIPVector, b::Number, y::HIPVector) =          # y = a x + b y
    (check(@ccall lib.npg_vec_axpby(y.h::Ptr{Cvoid}, Float64(a)::Float64, x.h::Ptr{Cvoid}, Float64(b)::Float64)::Cint); y)
LinearAlgebra.axpy!(a::Number, x::HIPVector, y::HIPVector) = axpby!(a, x, 1.0, y)
LinearAlgebra.rmul!(x::HIPVector, a::Number) = axpby!(0.0, x, a, x)
LinearAlgebra.mul!(y::HIPVector, D::Diagonal{Float64, <:HIPVector}, x::HIPVector) =   # mul!(q, ::Diagonal, w)
    (check(@ccall lib.npg_vec_mul(y.h::Ptr{Cvoid}, D.diag.h::Ptr{Cvoid}, x.h::Ptr{Cvoid})::Cint); y)
function Base.getindex(x::HIPVector, perm::Vector{Int})       # solver.x[inv_perm], src/model.jl:282,312 - result on the HOST
    a = Vector{Float64}(undef, length(perm))
    check(@ccall lib.npg_vec_download_perm(x.h::Ptr{Cvoid}, a::Ptr{Float64}, (perm .- 1)::Ptr{Int64})::Cint)
    return a
end
# (on_architecture(CPU(), x[inv_perm]) then meets the package's own identity method for Arrays, src/architectures.jl:9)

# ---- Krylov workspaces: the constructors the toolkits call, for VT = HIPVector{Float64} -----------------------------------
struct SolveStats
    solved::Int32; niter::Int32; npass::Int32; status::Int32; nreorth::Int32; nflagged::Int32
    rnorm0::Float64; rnorm::Float64; seconds::Float64
end
SolveStats() = SolveStats(0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0)
Base.getproperty(s::SolveStats, f::Symbol) = f === :timer ? getfield(s, :seconds) : getfield(s, f)   # workspace.stats.timer

mutable struct HIPGmresWorkspace
    h::Ptr{Cvoid}
    x::HIPVector{Float64}       # workspace.x: the toolkit's `x` aliases it (src/iterative_solvers.jl:26-29) => warm start
    stats::SolveStats
end
mutable struct HIPCgWorkspace
    h::Ptr{Cvoid}
    x::HIPVector{Float64}
    stats::SolveStats
end
function Krylov.GmresWorkspace(m::Integer, n::Integer, ::Type{HIPVector{Float64}}; memory::Integer = 20)
    out = Ref{Ptr{Cvoid}}()
    check(@ccall lib.npg_gmres_create(ctx()::Ptr{Cvoid}, n::Int64, memory::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    ws = HIPGmresWorkspace(out[], HIPVector{Float64}(undef, n), SolveStats())
    finalizer(w -> @ccall(lib.npg_gmres_destroy(w.h::Ptr{Cvoid})::Cint), ws)
end
function Krylov.CgWorkspace(m::Integer, n::Integer, ::Type{HIPVector{Float64}})
    out = Ref{Ptr{Cvoid}}()
    check(@ccall lib.npg_cg_create(ctx()::Ptr{Cvoid}, n::Int64, out::Ptr{Ptr{Cvoid}})::Cint)
    ws = HIPCgWorkspace(out[], HIPVector{Float64}(undef, n), SolveStats())
    finalizer(w -> @ccall(lib.npg_cg_destroy(w.h::Ptr{Cvoid})::Cint), ws)
end

# (kind, scalar, handle) of npg_*_solve's preconditioner arguments for the reference's `P`: Diagonal(1/h^dim * ones(N)) on the
# inversion (src/inversion.jl:54) -> NPG_PRECOND_SCALAR, Diagonal(1 ./ diag(A)) on the evolution (src/evolution.jl:149,167)
# -> NPG_PRECOND_DIAG.  Decided by a device reduction (min == max), never by indexing the device vector.
function precond_args(P::Diagonal{Float64, <:HIPVector})
    val, flag = Ref{Float64}(), Ref{Cint}()
    check(@ccall lib.npg_vec_is_constant(P.diag.h::Ptr{Cvoid}, val::Ptr{Float64}, flag::Ptr{Cint})::Cint)
    return flag[] != 0 ? (Cint(1), val[], C_NULL) : (Cint(2), 0.0, P.diag.h)
end
precond_args(::Nothing) = (Cint(0), 0.0, C_NULL)

# ---- iterative_solve!: one ccall per solve instead of Krylov.krylov_solve! (src/iterative_solvers.jl:58) -------------------
function nuPGCM.iterative_solve!(tk::nuPGCM.IterativeSolverToolkit{<:HIPSparseMatrixCSR})
    ws, kw = tk.workspace, tk.kwargs
    st = Ref{SolveStats}()
    if tk.P isa nuPGCM.BlockDiagonalPreconditioner
        # src/inversion.jl:60 (commented out in the reference): an INEXACT operator (inner CG per block) needs the flexible method
        fg, pc = fgmres_for(tk)
        check(@ccall lib.npg_fgmres_solve(fg.h::Ptr{Cvoid}, tk.A.h::Ptr{Cvoid}, pc.h::Ptr{Cvoid}, tk.y.h::Ptr{Cvoid}, tk.x.h::Ptr{Cvoid},
                                          1.0::Float64, Float64(kw[:atol])::Float64, Float64(kw[:rtol])::Float64,
                                          Int64(kw[:itmax])::Int64, st::Ptr{SolveStats})::Cint)
    elseif ws isa HIPGmresWorkspace
        kind, scalar, dh = precond_args(tk.P)
        kw[:restart] || error("restart=false is not supported by the HIP GMRES (the reference uses restart=true)")
        check(@ccall lib.npg_gmres_solve(ws.h::Ptr{Cvoid}, tk.A.h::Ptr{Cvoid}, kind::Cint, scalar::Float64, dh::Ptr{Cvoid},
                                         tk.y.h::Ptr{Cvoid}, tk.x.h::Ptr{Cvoid}, Float64(kw[:atol])::Float64,
                                         Float64(kw[:rtol])::Float64, Int64(kw[:itmax])::Int64, 0.1::Float64,
                                         st::Ptr{SolveStats})::Cint)
    else
        kind, scalar, dh = precond_args(tk.P)
        check(@ccall lib.npg_cg_solve(ws.h::Ptr{Cvoid}, tk.A.h::Ptr{Cvoid}, kind::Cint, scalar::Float64, dh::Ptr{Cvoid},
                                      tk.y.h::Ptr{Cvoid}, tk.x.h::Ptr{Cvoid}, Float64(kw[:atol])::Float64,
                                      Float64(kw[:rtol])::Float64, Int64(kw[:itmax])::Int64, st::Ptr{SolveStats})::Cint)
    end
    ws.stats = st[]
    @debug "$(tk.label) iterative solve: solved=$(ws.stats.solved == 1), niter=$(ws.stats.niter), time=$(ws.stats.seconds)"
    return tk
end

# ---- invert!(inversion, b): y = B b + b0 without device broadcasts (src/inversion.jl:101-110) --------------------------------
function nuPGCM.invert!(inversion::nuPGCM.InversionToolkit{<:HIPSparseMatrixCSR}, b)
    s = inversion.solver
    bd = nuPGCM.on_architecture(GPU(), Vector{Float64}(b.free_values))     # B consumes b in native order (src/inversion.jl:38)
    copyto!(s.y, inversion.b)
    mul!(s.y, inversion.B, bd, 1.0, 1.0)
    nuPGCM.iterative_solve!(s)
    return inversion
end

# ---- element kernels: npg_fe_* from the tables Gridap holds ------------------------------------------------------------------
# Filled from Gridap's public accessors; see nupgcm_amd/fe.py + nupgcm_amd/assembly.py for the same tables built without Gridap.
struct FeDesc
    ncell::Int64; nq::Int32; nloc_b::Int32
    grad_lambda::Ptr{Float64}; wdet::Ptr{Float64}; qw::Ptr{Float64}; N2::Ptr{Float64}; dN2::Ptr{Float64}
    Nb::Ptr{Float64}; dNb::Ptr{Float64}; N1::Ptr{Float64}
    cell_u::Ptr{Int32}; cell_p::Ptr{Int32}; cell_b::Ptr{Int32}
    u_diri::Ptr{Float64}; n_u_diri::Int64; b_diri::Ptr{Float64}; n_b_diri::Int64
    n_inv::Int64; n_b::Int64
end
mutable struct HIPFE
    h::Ptr{Cvoid}
    keep::Vector{Any}           # host tables stay alive for the duration of npg_fe_create only; kept for debugging
end
const FE_CACHE = IdDict{Any, HIPFE}()

"0-based device index of a Gridap dof id: free id k > 0 -> inv_perm[k] - 1 (+ offset); Dirichlet id -k < 0 stays -k, which IS
the ABI's `-1 - (k - 1)` (include/nupgcm_hip.h: an entry < 0 is Dirichlet value number -1 - entry)"
devidx(id, inv_perm, off = 0) = id > 0 ? Int32(off + inv_perm[id] - 1) : Int32(id)

function hip_fe(fe_data)
    get!(FE_CACHE, fe_data) do
        mesh, spaces, dofs = fe_data.mesh, fe_data.spaces, fe_data.dofs
        U, P = spaces.X_trial[1], spaces.X_trial[2]
        B = spaces.B_trial
        Ω, dΩ = mesh.Ω, mesh.dΩ
        X = Gridap.Geometry.get_cell_coordinates(Ω)                       # ncell x 4 points (cells sorted by GridapGmsh)
        nc = length(X)
        G, wdet = Matrix{Float64}(undef, 12, nc), Vector{Float64}(undef, nc)   # ABI layout [ncell][4][3] = column-major 12 x nc
        for (c, x) in enumerate(X)
            J = hcat(collect(Tuple(x[2] - x[1])), collect(Tuple(x[3] - x[1])), collect(Tuple(x[4] - x[1])))
            Ji = inv(J)                                                   # rows = gradients of the reference coordinates
            g = vcat(-sum(Ji, dims = 1), Ji)                              # 4 x 3: grad lambda_k
            G[:, c] = vec(permutedims(g))
            wdet[c] = abs(det(J))
        end
        quad = Gridap.CellData.get_data(Gridap.CellData.get_cell_quadrature(dΩ))[1]   # the one reference rule of Measure(Ω, 4)
        xq, qw = Gridap.ReferenceFEs.get_coordinates(quad), Gridap.ReferenceFEs.get_weights(quad)
        lam = [i == 1 ? 1 - sum(Tuple(p)) : Tuple(p)[i - 1] for p in xq, i in 1:4]        # nq x 4 barycentric
        ea, eb = [1, 1, 2, 1, 2, 3], [2, 3, 3, 4, 4, 4]                   # Gridap's local edge order on a TET
        nq = length(qw)
        N2 = hcat(lam .* (2 .* lam .- 1), 4 .* lam[:, ea] .* lam[:, eb])  # nq x 10, nodal P2 = Gridap's (to rounding)
        dN2 = zeros(nq, 10, 4)
        for k in 1:4; dN2[:, k, k] = 4 .* lam[:, k] .- 1; end
        for e in 1:6; dN2[:, 4 + e, ea[e]] = 4 .* lam[:, eb[e]]; dN2[:, 4 + e, eb[e]] = 4 .* lam[:, ea[e]]; end
        dN1 = zeros(nq, 4, 4); for k in 1:4; dN1[:, k, k] .= 1; end
        p1b = Gridap.FESpaces.num_free_dofs(B) + Gridap.FESpaces.num_dirichlet_dofs(B) == Gridap.Geometry.num_nodes(mesh.model)
        Nb, dNb, nlb = p1b ? (lam, dN1, 4) : (N2, dN2, 10)
        rowmajor(a) = vec(permutedims(a, reverse(1:ndims(a))))            # C layout of the ABI
        # DoF tables: velocity dofs are node-major with interleaved components (dof = 3 (node - 1) + comp), SURVEY 8c
        idsu, idsp, idsb = Gridap.FESpaces.get_cell_dof_ids(U), Gridap.FESpaces.get_cell_dof_ids(P), Gridap.FESpaces.get_cell_dof_ids(B)
        ip, ib, nu = dofs.inv_p_inversion, dofs.inv_p_b, dofs.nu
        cu = Int32[devidx(idsu[c][l], ip) for l in 1:30, c in 1:nc]       # 30 x nc column-major = [ncell][10][3]
        cp = Int32[idsp[c][m] > 0 ? Int32(ip[nu + idsp[c][m]] - 1) : Int32(-1) for m in 1:4, c in 1:nc]
        cb = Int32[devidx(idsb[c][i], ib) for i in 1:nlb, c in 1:nc]
        ud = Vector{Float64}(Gridap.FESpaces.get_dirichlet_dof_values(U))
        bd = Vector{Float64}(Gridap.FESpaces.get_dirichlet_dof_values(B))
        tabs = Any[G, wdet, Vector{Float64}(qw), rowmajor(N2), rowmajor(dN2), rowmajor(Nb), rowmajor(dNb), rowmajor(lam), cu, cp, cb, ud, bd]
        d = FeDesc(nc, nq, nlb, pointer(G), pointer(wdet), pointer(tabs[3]), pointer(tabs[4]), pointer(tabs[5]), pointer(tabs[6]),
                   pointer(tabs[7]), pointer(tabs[8]), pointer(cu), pointer(cp), pointer(cb), pointer(ud), length(ud),
                   pointer(bd), length(bd), dofs.nu + dofs.np, dofs.nb)
        out = Ref{Ptr{Cvoid}}()
        GC.@preserve tabs check(@ccall lib.npg_fe_create(ctx()::Ptr{Cvoid}, Ref(d)::Ptr{FeDesc}, out::Ptr{Ptr{Cvoid}})::Cint)
        fe = HIPFE(out[], tabs)
        finalizer(x -> @ccall(lib.npg_fe_destroy(x.h::Ptr{Cvoid})::Cint), fe)
    end
end

# ---- node-block storage from the reference's DoF order ------------------------------------------------------------------------
"(node, component) of every free velocity DoF in Gridap's native numbering: a cell's local DoF l (1..30) is component (l-1)%3 of
its local node (l-1)÷3+1 (node-major local numbering, SURVEY 8c); the node's label is that local node's DoF id in an
unconstrained scalar P2 space on the same triangulation."
function velocity_dof_nodes(fe_data)
    U = fe_data.spaces.X_trial[1]
    S = Gridap.FESpaces.TestFESpace(fe_data.mesh.model, Gridap.ReferenceFEs.ReferenceFE(Gridap.ReferenceFEs.lagrangian, Float64, 2))
    idsu, idsn = Gridap.FESpaces.get_cell_dof_ids(U), Gridap.FESpaces.get_cell_dof_ids(S)
    nu = fe_data.dofs.nu
    node, comp = fill(Int64(-1), nu), zeros(Int32, nu)
    for c in 1:length(idsu), l in 1:30
        id = idsu[c][l]
        if id > 0
            node[id] = idsn[c][(l - 1) ÷ 3 + 1] - 1
            comp[id] = (l - 1) % 3
        end
    end
    return node, comp
end

"npg_csr_block_nodes_dofs on A_inversion as the reference uploaded it (rows / columns in p_inversion order, src/inversion.jl:37)"
function block_nodes!(A::HIPSparseMatrixCSR, fe_data; rtol = 1e-12)
    dofs = fe_data.dofs
    node_native, comp_native = velocity_dof_nodes(fe_data)
    N = dofs.nu + dofs.np
    node, comp = fill(Int64(-1), N), zeros(Int32, N)
    for i in 1:dofs.nu                                                   # p_inversion = [p_u; nu .+ p_p]: row i is native DoF p_u[i]
        node[i] = node_native[dofs.p_inversion[i]]
        comp[i] = comp_native[dofs.p_inversion[i]]
    end
    blocked = Ref{Cint}(0)
    check(@ccall lib.npg_csr_block_nodes_dofs(A.h::Ptr{Cvoid}, node::Ptr{Int64}, comp::Ptr{Int32}, rtol::Float64, blocked::Ptr{Cint})::Cint)
    return blocked[] != 0
end

const ACTIVE = Ref{Any}(nothing)        # the model this extension serves (update_Δt! / build_A_inversion! have no argument that says so)
const HIPInversion = nuPGCM.InversionToolkit{<:HIPSparseMatrixCSR}       # (first type parameter: the type of B, src/inversion.jl:1-5)

"what the two constructors below do before handing over to the reference's: node-block storage (constant viscosity) or the
record-form companion (function-valued viscosity / eddy closure: the full-stress form has nine entries per node pair)"
function prepare_inversion!(forcings, fe_data, inversion)
    A = inversion.solver.A
    size(A, 1) >= 100_000 || return nothing
    if !(forcings.ν isa Function) && !forcings.eddy_param.is_on
        block_nodes!(A, fe_data)
    end
    return nothing
end
# The reference defines exactly two outer constructors (src/model.jl:47-62):
#   Model(::AbstractArchitecture, ::Parameters, ::Forcings, ::FEData, ::InversionToolkit)
#   Model(::AbstractArchitecture, ::Parameters, ::Forcings, ::FEData, ::InversionToolkit, ::EvolutionToolkit, ::AbstractTimestepper)
# The two methods here have the SAME arity and the same types except GPU <: AbstractArchitecture in argument 1 and
# HIPInversion <: InversionToolkit in argument 5: each is strictly more specific than its counterpart and applicable to nothing
# else (in particular not to the 8-argument default constructor of the struct), so there is no ambiguity; `invoke` names the
# reference's method by its exact signature.
function nuPGCM.Model(arch::GPU, params::nuPGCM.Parameters, forcings::nuPGCM.Forcings, fe_data::nuPGCM.FEData, inversion::HIPInversion)
    prepare_inversion!(forcings, fe_data, inversion)
    model = invoke(nuPGCM.Model, Tuple{nuPGCM.AbstractArchitecture, nuPGCM.Parameters, nuPGCM.Forcings, nuPGCM.FEData, nuPGCM.InversionToolkit},
                   arch, params, forcings, fe_data, inversion)
    ACTIVE[] = model
    return model
end
function nuPGCM.Model(arch::GPU, params::nuPGCM.Parameters, forcings::nuPGCM.Forcings, fe_data::nuPGCM.FEData, inversion::HIPInversion,
                      evolution::nuPGCM.EvolutionToolkit, timestepper::nuPGCM.AbstractTimestepper)
    prepare_inversion!(forcings, fe_data, inversion)
    model = invoke(nuPGCM.Model, Tuple{nuPGCM.AbstractArchitecture, nuPGCM.Parameters, nuPGCM.Forcings, nuPGCM.FEData, nuPGCM.InversionToolkit,
                                       nuPGCM.EvolutionToolkit, nuPGCM.AbstractTimestepper},
                   arch, params, forcings, fe_data, inversion, evolution, timestepper)
    ACTIVE[] = model
    return model
end

upload_perm(a::Vector{Float64}, perm::Vector{Int}) = begin               # on_architecture(arch, a[perm]) in one call
    v = HIPVector{Float64}(undef, length(perm))
    check(@ccall lib.npg_vec_upload_perm(v.h::Ptr{Cvoid}, a::Ptr{Float64}, (perm .- 1)::Ptr{Int64})::Cint)
    v
end

# ---- evolve!: src/model.jl:213-285 with :269-278 on the device ---------------------------------------------------------------
const HIPModel = nuPGCM.Model{GPU, <:Any, <:Any, <:Any, <:HIPInversion}
function nuPGCM.evolve!(model::HIPModel, u_prev, b_prev)
    dofs, ev, ts = model.fe_data.dofs, model.evolution, model.timestepper
    solver = ev.solver
    θ = nuPGCM.evolution_parameter(model.params, ts)
    fe = hip_fe(model.fe_data)
    bv(b) = upload_perm(Vector{Float64}(b.free_values), dofs.p_b)
    if model.forcings.conv_param.is_on || ts.adaptive
        dm = device_evolution_matrices(ev, model)                         # M, Kₕ, Kᵥ on the device, one common pattern
        if model.forcings.conv_param.is_on                                # src/model.jl:229-247
            cp = model.forcings.conv_param
            check(@ccall lib.npg_fe_update_kappa_convection(fe.h::Ptr{Cvoid}, C_NULL::Ptr{Float64}, Float64(cp.κᶜ)::Float64,
                                                            Float64(cp.N²min)::Float64, Float64(model.params.α)::Float64,
                                                            Float64(model.params.N²)::Float64, bv(model.state.b).h::Ptr{Cvoid})::Cint)
            check(@ccall lib.npg_fe_assemble_matrix(fe.h::Ptr{Cvoid}, 3::Cint, 1.0::Float64, 0::Cint, dm.Kv.h::Ptr{Cvoid},
                                                    ev.rhsᵥ.h::Ptr{Cvoid})::Cint)                     # NPG_MAT_KV, lift = rhsᵥ
            check(@ccall lib.npg_fe_assemble_rhs_diff(fe.h::Ptr{Cvoid}, Float64(model.params.N²)::Float64, ev.rhs_diff.h::Ptr{Cvoid})::Cint)
        end
        # A = M + θ (Kₕ + Kᵥ), P = Diagonal(1 ./ diag(A)) - src/model.jl:251-261, on the device, in place.  The target is the cached
        # matrix on the common pattern: collect_evolution_LHS! (src/model.jl:134-137, step 2 of a BDF2 run) may have replaced solver.A
        solver.A = dm.A
        check(@ccall lib.npg_csr_combine(solver.A.h::Ptr{Cvoid}, 1.0::Float64, dm.M.h::Ptr{Cvoid}, θ::Float64, dm.Kh.h::Ptr{Cvoid},
                                         dm.Kv.h::Ptr{Cvoid})::Cint)
        check(@ccall lib.npg_csr_inv_diag(solver.A.h::Ptr{Cvoid}, solver.P.diag.h::Ptr{Cvoid})::Cint)
    end
    # state in the solvers' orderings: [u; p] by p_inversion (the pressure part is not read by the advection form), b by p_b
    xi(u) = upload_perm(vcat(Vector{Float64}(u.free_values), zeros(dofs.np)), dofs.p_inversion)
    scheme = ts isa nuPGCM.BDF1 ? Cint(1) : Cint(2)
    check(@ccall lib.npg_fe_evolution_rhs(fe.h::Ptr{Cvoid}, scheme::Cint, ts.Δt[]::Float64, Float64(model.params.N²)::Float64,
                                          θ::Float64, bv(model.state.b).h::Ptr{Cvoid}, bv(b_prev).h::Ptr{Cvoid},
                                          xi(model.state.u).h::Ptr{Cvoid}, xi(u_prev).h::Ptr{Cvoid},
                                          ev.rhs_diff.h::Ptr{Cvoid}, ev.rhs_flux.h::Ptr{Cvoid}, ev.rhsₘ.h::Ptr{Cvoid},
                                          ev.rhsₕ.h::Ptr{Cvoid}, ev.rhsᵥ.h::Ptr{Cvoid}, solver.y.h::Ptr{Cvoid})::Cint)
    nuPGCM.iterative_solve!(solver)
    model.state.b.free_values .= solver.x[dofs.inv_p_b]                  # src/model.jl:282
    return model
end

# device copies of the evolution matrices on ONE pattern (Gridap's structural pattern: explicit zeros kept, drop_zeros = 0), so
# that K_v can be re-assembled in place and the LHS combined on the device; the coefficient tables the closures start from
const EVO_CACHE = IdDict{Any, Any}()
function device_evolution_matrices(ev, model)
    get!(EVO_CACHE, ev) do
        up(A) = upload_csc(A, 0)                                          # explicit zeros kept: one pattern for all four
        pattern = ev.M + ev.Kₕ + ev.Kᵥ                                     # union of the three patterns
        onpat(A) = (B = copy(pattern); fill!(B.nzval, 0.0); B .+ A)       # A's values on the common pattern
        fe = hip_fe(model.fe_data)
        # background kappa_v at the quadrature points, [ncell][nq] (the closure adds the convective part to it)
        xq = Gridap.CellData.get_cell_points(model.fe_data.mesh.dΩ)
        κv = model.forcings.κᵥ
        tab = κv isa Function ? reduce(vcat, [Float64.(κv.(p)) for p in Gridap.CellData.get_array(xq)]) :
              fill(Float64(κv), length(Gridap.CellData.get_array(xq)) * length(first(Gridap.CellData.get_array(xq))))
        check(@ccall lib.npg_fe_set_coeff(fe.h::Ptr{Cvoid}, "kappa_v"::Cstring, tab::Ptr{Float64})::Cint)
        # the solver's own A must live on the same pattern for npg_csr_combine to write into it
        (A = up(onpat(nuPGCM.on_architecture(CPU(), model.evolution.solver.A))), M = up(onpat(ev.M)), Kh = up(onpat(ev.Kₕ)), Kv = up(onpat(ev.Kᵥ)))
    end
end

# ---- update_Δt!: src/timesteppers.jl:108-119 as one device reduction --------------------------------------------------------------
# (The reference's method is update_Δt!(::BDF1, u, dΩ, h_cells) with the last three untyped; this one differs only in typing
#  h_cells - compute_h_cells returns a Vector{Float64}, src/meshes.jl:127-134 - so it is strictly more specific; it serves the model of this
#  extension - recognised by its measure - and defers to the reference's host evaluation otherwise.  `u` is model.state.u, whose
#  values sync_flow! copied from inversion.solver.x after the last invert!: the device vector holds the same numbers.)
function nuPGCM.update_Δt!(ts::nuPGCM.BDF1, u, dΩ, h_cells::AbstractVector{<:Real}; u_min = 0.01)
    model = ACTIVE[]
    if model === nothing || model.fe_data.mesh.dΩ !== dΩ
        return invoke(nuPGCM.update_Δt!, Tuple{nuPGCM.BDF1, Any, Any, Any}, ts, u, dΩ, h_cells; u_min = u_min)
    end
    fe = hip_fe(model.fe_data)
    out = Ref{Float64}()
    hc = convert(Vector{Float64}, h_cells)
    check(@ccall lib.npg_fe_cfl_ratio(fe.h::Ptr{Cvoid}, hc::Ptr{Float64}, Float64(u_min)::Float64,
                                      model.inversion.solver.x.h::Ptr{Cvoid}, out::Ptr{Float64})::Cint)
    ts.Δt[] = ts.CFL_factor * out[]
    return ts
end

# ---- the eddy closure's re-assembly: build_A_inversion! for the served model (src/model.jl:160-170, src/inversion.jl:149-170) ----
# run! caches `A_inversion = on_architecture(CPU(), solver.A)[iperm, iperm]` (host, native order) before its loop and every tenth
# step calls build_A_inversion!(A_inversion, dup, dvq, assembler, fe_data, params, ν) - a Gridap cell loop on the host - followed by
# `solver.A = on_architecture(arch, A_inversion[perm, perm])`: CPU re-assembly + CPU permutation + a full upload, and the new matrix
# has lost any record layout.  The body of run!'s loop cannot be given a method, but build_A_inversion! can: the method below differs
# from the reference's (src/inversion.jl:149-150: A untyped) only in typing A as the host matrix run! passes, so it is strictly more
# specific and every call lands here; for anything but the model this extension serves it `invoke`s the reference's method.
# Served: nu = nu_eddy(alpha (N2 + dz b)) at the quadrature points and the full-stress matrix are formed on the device, in place,
# in the matrix `eddy_state` keeps (p_inversion order, the pattern run! holds); the host matrix A is left as it is, and the upload
# that follows in run! is answered with the device matrix (PENDING_A above).
# (In the reference's DoF order the matrix stays plain CSR: the record-form companion of npg_csr_pack_nodes needs the library's
#  node-block order, which the Python host uses; the Arnoldi kernel then gathers its input from the fp32 copy, 4-byte gathers.)
const EDDY_CACHE = IdDict{Any, Any}()
"coefficient table of a function of x (or a constant) at the quadrature points, [ncell][nq] row-major as npg_fe_set_coeff takes it"
function quad_table(fe_data, g)
    pts = Gridap.CellData.get_array(Gridap.CellData.get_cell_points(fe_data.mesh.dΩ))
    g isa Function ? reduce(vcat, [Float64.(g.(p)) for p in pts]) : fill(Float64(g), length(pts) * length(first(pts)))
end
"once per served model: the device matrix the refreshes write into, and whether the device closure applies at all - it reads ONE
Coriolis table `f`, in the bilinear form and in nu_eddy, so params.f and eddy_param.f must be the same function"
function eddy_state(model, A_native::SparseMatrixCSC{Float64, Int64})
    get!(EDDY_CACHE, model) do
        fe_data, dofs = model.fe_data, model.fe_data.dofs
        tf = quad_table(fe_data, model.params.f)
        if quad_table(fe_data, model.forcings.eddy_param.f) != tf
            (M = nothing, ok = false)
        else
            check(@ccall lib.npg_fe_set_coeff(hip_fe(fe_data).h::Ptr{Cvoid}, "f"::Cstring, tf::Ptr{Float64})::Cint)
            # once: A_inversion in p_inversion order on the pattern run! holds - Gridap's structural one, since uploads keep explicit zeros
            (M = upload_csc(A_native[dofs.p_inversion, dofs.p_inversion], 0), ok = true)
        end
    end
end

function nuPGCM.build_A_inversion!(A::SparseMatrixCSC{Float64, Int64}, dup, dvq, assembler, fe_data::nuPGCM.FEData,
                                   params::nuPGCM.Parameters, ν; friction_only = false, frictionless_only = false)
    model = ACTIVE[]
    served = model !== nothing && model.fe_data === fe_data && model.forcings.eddy_param.is_on && !friction_only && !frictionless_only &&
             model.inversion isa HIPInversion
    if served
        es = eddy_state(model, A)
        if es.ok
            ep, fe = model.forcings.eddy_param, hip_fe(fe_data)
            b = upload_perm(Vector{Float64}(model.state.b.free_values), fe_data.dofs.p_b)
            # ν_eddy(eddy_param, αbz) as run! calls it: smoothing = 10, ν_min = 1 (src/inputs.jl:130-137)
            check(@ccall lib.npg_fe_update_nu_eddy(fe.h::Ptr{Cvoid}, Float64(ep.N²min)::Float64, Float64(params.α)::Float64,
                                                   Float64(params.N²)::Float64, 10.0::Float64, 1.0::Float64, b.h::Ptr{Cvoid})::Cint)
            check(@ccall lib.npg_fe_assemble_matrix(fe.h::Ptr{Cvoid}, 4::Cint, Float64(params.α^2 * params.ε^2)::Float64, 1::Cint,
                                                    es.M.h::Ptr{Cvoid}, C_NULL::Ptr{Cvoid})::Cint)     # NPG_MAT_A, full-stress form
            PENDING_A[] = es.M
            return A
        end
    end
    return invoke(nuPGCM.build_A_inversion!, Tuple{Any, Any, Any, Any, nuPGCM.FEData, nuPGCM.Parameters, Any},
                  A, dup, dvq, assembler, fe_data, params, ν; friction_only = friction_only, frictionless_only = frictionless_only)
end

# ---- BlockDiagonalPreconditioner on the device: src/preconditioners.jl:53-125 ----------------------------------------------------
# P-block, src/preconditioners.jl:101-107: the reference calls P_block_setup(arch, A) with the HOST matrix (:76-81), uploads it
# inside and builds `P_prec = kp_ilu0(P)` + `CgPreconditioner(P, P_prec; ldiv = true, itmax = 100)`.  The method below is typed on
# that host matrix (strictly more specific than the reference's untyped one), uploads inside as the reference does, takes the
# library's ILU(0) (level-scheduled csrilu02 / csrsv2 counterparts, csrc/ilu.hip) and returns a preconditioner whose `mul!` is
# the block's whole CG in one call (npg_cg_ilu0_solve: Krylov.jl's cg with M = the factors, warm-started like `workspace.x`).
mutable struct HIPILU0
    h::Ptr{Cvoid}
    A::HIPSparseMatrixCSR{Float64}          # kept alive: the CG multiplies with it
    function HIPILU0(A::HIPSparseMatrixCSR{Float64})
        out = Ref{Ptr{Cvoid}}()
        check(@ccall lib.npg_ilu0_create(ctx()::Ptr{Cvoid}, A.h::Ptr{Cvoid}, out::Ptr{Ptr{Cvoid}})::Cint)
        f = new(out[], A)
        finalizer(x -> @ccall(lib.npg_ilu0_destroy(x.h::Ptr{Cvoid})::Cint), f)
    end
end
LinearAlgebra.ldiv!(y::HIPVector, F::HIPILU0, x::HIPVector) =
    (check(@ccall lib.npg_ilu0_apply(F.h::Ptr{Cvoid}, x.h::Ptr{Cvoid}, y.h::Ptr{Cvoid})::Cint); y)

struct HIPCgIlu0Preconditioner <: nuPGCM.Preconditioner
    F::HIPILU0
    jac::HIPVector{Float64}                 # 1 ./ diag(P): what npg_precond_blockdiag_set takes beside the factors
    x::HIPVector{Float64}                   # workspace.x of the reference's CgPreconditioner: the warm start, zero at first
    itmax::Int
    label::String
end
function nuPGCM.P_block_setup(::GPU, A::SparseMatrixCSC{Float64, Int64}; tag = "")
    P = upload_csc(A, 0)                    # (the caller has dropped the zeros already, :80; ILU(0) wants the pattern as it is)
    jac = HIPVector{Float64}(undef, size(A, 1))
    check(@ccall lib.npg_csr_inv_diag(P.h::Ptr{Cvoid}, jac.h::Ptr{Cvoid})::Cint)
    return HIPCgIlu0Preconditioner(HIPILU0(P), jac, HIPVector{Float64}(undef, size(A, 1)), 100, "P$tag-block")     # (undef zero-fills)
end
# `mul!(yb, block.P⁻¹, xb)` of src/preconditioners.jl:118-125 arrives with xb, yb = @view x[block.indices]: SubArrays of HIPVectors
function LinearAlgebra.mul!(y::HIPVecOrView, cgp::HIPCgIlu0Preconditioner, x::HIPVecOrView)
    st = Ref{SolveStats}()
    xd, yd = dev(x), dev(y)
    # (the reference passes no tolerances here: Krylov.jl's defaults, atol = rtol = sqrt(eps))
    check(@ccall lib.npg_cg_ilu0_solve(cgp.F.h::Ptr{Cvoid}, cgp.F.A.h::Ptr{Cvoid}, xd.h::Ptr{Cvoid}, cgp.x.h::Ptr{Cvoid},
                                       sqrt(eps(Float64))::Float64, sqrt(eps(Float64))::Float64, cgp.itmax::Int64, st::Ptr{SolveStats})::Cint)
    @debug "$(cgp.label) iterative solve: solved=$(st[].solved != 0), niter=$(st[].niter), time=$(st[].seconds)"
    copyto!(yd, cgp.x)
    return y
end
# T-block, src/preconditioners.jl:84-89: the reference's own CgPreconditioner(T, Diagonal(1 ./ diag(T)); ldiv = false, itmax = 0) - its
# constructor meets Krylov.CgWorkspace(n, n, HIPVector{Float64}) above; its mul! (:24-37) would call Krylov.krylov_solve! on that
# workspace and broadcast into a view, so the method for a HIP matrix is one npg_cg_solve (warm start = workspace.x) + a device copy
function LinearAlgebra.mul!(y::HIPVecOrView, cgp::nuPGCM.CgPreconditioner{<:HIPSparseMatrixCSR}, x::HIPVecOrView)
    cgp.ldiv && error("CgPreconditioner on the HIP device: ldiv = true is served by HIPCgIlu0Preconditioner (P_block_setup) only")
    kind, scalar, dh = precond_args(cgp.preconditioner)
    st = Ref{SolveStats}()
    xd, yd = dev(x), dev(y)
    ws = cgp.workspace
    check(@ccall lib.npg_cg_solve(ws.h::Ptr{Cvoid}, cgp.matrix.h::Ptr{Cvoid}, kind::Cint, scalar::Float64, dh::Ptr{Cvoid},
                                  xd.h::Ptr{Cvoid}, ws.x.h::Ptr{Cvoid}, sqrt(eps(Float64))::Float64, sqrt(eps(Float64))::Float64,
                                  Int64(cgp.itmax)::Int64, st::Ptr{SolveStats})::Cint)
    ws.stats = st[]
    @debug "$(cgp.label) iterative solve: solved=$(st[].solved != 0), niter=$(st[].niter), time=$(st[].seconds)"
    copyto!(yd, ws.x)
    return y
end

# `P = BlockDiagonalPreconditioner(arch, params, fe_data, A)` as the inversion's preconditioner (src/inversion.jl:60): iterative_solve!
# above hands it to the library's flexible GMRES with the same blocks as ONE device-side operator (NPG_PC_BLOCKDIAG: inner CG per
# block, warm-started, no host round trip per application) - the blocks' matrices, Jacobi vector and ILU(0) factors are borrowed
mutable struct HIPHandle
    h::Ptr{Cvoid}
    keep::Any
end
const FGMRES_CACHE = IdDict{Any, Any}()
function fgmres_for(tk)
    get!(FGMRES_CACHE, tk) do
        blocks = tk.P.blocks
        out = Ref{Ptr{Cvoid}}()
        check(@ccall lib.npg_precond_create(ctx()::Ptr{Cvoid}, 1::Cint, length(blocks)::Cint, out::Ptr{Ptr{Cvoid}})::Cint)      # NPG_PC_BLOCKDIAG
        pc = HIPHandle(out[], blocks)
        finalizer(x -> @ccall(lib.npg_precond_destroy(x.h::Ptr{Cvoid})::Cint), pc)
        tol = sqrt(eps(Float64))
        for (k, blk) in enumerate(blocks)
            off = first(blk.indices) - 1
            B = blk.P⁻¹
            if B isa HIPCgIlu0Preconditioner
                check(@ccall lib.npg_precond_blockdiag_set(pc.h::Ptr{Cvoid}, (k - 1)::Cint, off::Int64, B.F.A.h::Ptr{Cvoid}, B.jac.h::Ptr{Cvoid},
                                                           B.itmax::Int64, tol::Float64, tol::Float64)::Cint)
                check(@ccall lib.npg_precond_blockdiag_set_ilu0(pc.h::Ptr{Cvoid}, (k - 1)::Cint, B.F.h::Ptr{Cvoid})::Cint)
            else
                kind, _, dh = precond_args(B.preconditioner)
                kind == 2 || error("BlockDiagonalPreconditioner on the HIP device: block $k needs a Jacobi vector or ILU(0) factors")
                check(@ccall lib.npg_precond_blockdiag_set(pc.h::Ptr{Cvoid}, (k - 1)::Cint, off::Int64, B.matrix.h::Ptr{Cvoid}, dh::Ptr{Cvoid},
                                                           Int64(B.itmax)::Int64, tol::Float64, tol::Float64)::Cint)
            end
        end
        check(@ccall lib.npg_fgmres_create(ctx()::Ptr{Cvoid}, size(tk.A, 1)::Int64, 20::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
        fg = HIPHandle(out[], nothing)
        finalizer(x -> @ccall(lib.npg_fgmres_destroy(x.h::Ptr{Cvoid})::Cint), fg)
        (fg, pc)
    end
end

end # module

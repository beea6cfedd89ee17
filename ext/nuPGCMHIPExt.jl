# nuPGCMHIPExt.jl - package extension a nuPGCM maintainer would add next to ext/nuPGCMCUDAExt.jl.
#
# NOT executed in this repository's CI (the build image has no Julia); it shows the reference-side binding of the C ABI in
# include/nupgcm_hip.h.  It adds methods of the existing `GPU` singleton for two new array types, exactly the way
# ext/nuPGCMCUDAExt.jl:24-33 does for CuArray / CuSparseMatrixCSR (a new Architecture subtype would silently fall into the
# `lu(A)` branches: src/inversion.jl:42, src/evolution.jl:148,166 test `typeof(arch) == GPU`).
module nuPGCMHIPExt

using nuPGCM
using SparseArrays, LinearAlgebra
import Krylov

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
    function HIPVector{Float64}(::UndefInitializer, n::Integer)
        out = Ref{Ptr{Cvoid}}()
        check(@ccall lib.npg_vec_create(ctx()::Ptr{Cvoid}, n::Int64, out::Ptr{Ptr{Cvoid}})::Cint)
        v = new{Float64}(out[], n)
        finalizer(x -> @ccall(lib.npg_vec_destroy(x.h::Ptr{Cvoid})::Cint), v)
    end
end
Base.size(v::HIPVector) = (v.n,)

mutable struct HIPSparseMatrixCSR{T} <: AbstractSparseMatrix{T, Int32}
    h::Ptr{Cvoid}
    m::Int
    n::Int
end
Base.size(A::HIPSparseMatrixCSR) = (A.m, A.n)

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
function nuPGCM.on_architecture(::GPU, A::SparseMatrixCSC{Float64, Int64})     # CuSparseMatrixCSR(a): CSC -> CSR on upload
    out = Ref{Ptr{Cvoid}}()
    check(@ccall lib.npg_csr_create_from_csc(ctx()::Ptr{Cvoid}, size(A, 1)::Int64, size(A, 2)::Int64,
                                              (A.colptr .- 1)::Ptr{Int64}, (A.rowval .- 1)::Ptr{Int64},
                                              A.nzval::Ptr{Float64}, 1::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    M = HIPSparseMatrixCSR{Float64}(out[], size(A)...)
    finalizer(x -> @ccall(lib.npg_csr_destroy(x.h::Ptr{Cvoid})::Cint), M)
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

# ---- what the hot path does with the device objects (SURVEY.md 8b) -----------------------------------------------------
function LinearAlgebra.mul!(y::HIPVector, A::HIPSparseMatrixCSR, x::HIPVector, α::Number = true, β::Number = false)
    check(@ccall lib.npg_spmv(A.h::Ptr{Cvoid}, x.h::Ptr{Cvoid}, y.h::Ptr{Cvoid}, Float64(α)::Float64, Float64(β)::Float64)::Cint)
    return y
end
LinearAlgebra.dot(x::HIPVector, y::HIPVector) =
    (o = Ref{Float64}(); check(@ccall lib.npg_vec_dot(x.h::Ptr{Cvoid}, y.h::Ptr{Cvoid}, o::Ptr{Float64})::Cint); o[])
LinearAlgebra.norm(x::HIPVector) =
    (o = Ref{Float64}(); check(@ccall lib.npg_vec_nrm2(x.h::Ptr{Cvoid}, o::Ptr{Float64})::Cint); o[])
Base.fill!(x::HIPVector, a) = (check(@ccall lib.npg_vec_fill(x.h::Ptr{Cvoid}, Float64(a)::Float64)::Cint); x)
Base.getindex(x::HIPVector, perm::Vector{Int}) = begin      # solver.x[inv_perm], src/model.jl:282,312
    a = Vector{Float64}(undef, x.n)
    check(@ccall lib.npg_vec_download_perm(x.h::Ptr{Cvoid}, a::Ptr{Float64}, (perm .- 1)::Ptr{Int64})::Cint)
    a
end

# ---- iterative_solve!: one ccall per solve instead of Krylov.krylov_solve! (src/iterative_solvers.jl:58) -------------------
struct SolveStats
    solved::Int32; niter::Int32; npass::Int32; status::Int32; nreorth::Int32; nflagged::Int32
    rnorm0::Float64; rnorm::Float64; seconds::Float64
end
mutable struct HIPGmresWorkspace
    h::Ptr{Cvoid}
    x::HIPVector{Float64}       # workspace.x: the toolkit's `x` aliases it (src/iterative_solvers.jl:26-29) => warm start
    stats::SolveStats
end
function nuPGCM.iterative_solve!(tk::nuPGCM.IterativeSolverToolkit{<:HIPSparseMatrixCSR})
    ws, kw = tk.workspace, tk.kwargs
    st = Ref{SolveStats}()
    P = tk.P                                          # Diagonal(1/h^dim * ones(N)) on the inversion, Diagonal(1 ./ diag A) on the evolution
    scalar = allequal(P.diag) ? (1, first(P.diag), C_NULL) : (2, 0.0, P.diag.h)
    if ws isa HIPGmresWorkspace
        check(@ccall lib.npg_gmres_solve(ws.h::Ptr{Cvoid}, tk.A.h::Ptr{Cvoid}, scalar[1]::Cint, scalar[2]::Float64,
                                         scalar[3]::Ptr{Cvoid}, tk.y.h::Ptr{Cvoid}, tk.x.h::Ptr{Cvoid},
                                         kw[:atol]::Float64, kw[:rtol]::Float64, kw[:itmax]::Int64, 0.1::Float64,
                                         st::Ptr{SolveStats})::Cint)
    else
        check(@ccall lib.npg_cg_solve(ws.h::Ptr{Cvoid}, tk.A.h::Ptr{Cvoid}, scalar[1]::Cint, scalar[2]::Float64,
                                      scalar[3]::Ptr{Cvoid}, tk.y.h::Ptr{Cvoid}, tk.x.h::Ptr{Cvoid}, kw[:atol]::Float64,
                                      kw[:rtol]::Float64, kw[:itmax]::Int64, st::Ptr{SolveStats})::Cint)
    end
    ws.stats = st[]
    return tk
end

end # module

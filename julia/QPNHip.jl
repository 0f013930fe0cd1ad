# QPNHip.jl -- `ccall` shim that puts libqpn_hip.so (include/qpn_hip.h) behind the reference's own
# functions for the node-AVI hot path.  SOURCE ONLY: this container has no Julia, so this file has
# never been executed (see INTEGRATION.md); every signature below is checked against the C header
# by hand.  It replaces exactly two bodies in the reference:
#
#   solve_avi(avi::AVI, z0, w)                      src/avi.jl:63-77    (PATHSolver.solve_mcp + check)
#   solve_qp(Q, q, A, l, u; solver=:PATH)           src/qp_processing.jl:12-33
#
# and adds batched entries: solve_avi_batch (many independent AVIs), solve_nodes! (a level of single-node pools), resident
# node records (upload_nodes / solve_nodes!(nodes, ...) / verify_nodes), assemble_pool (combine_gavis) and local_pieces.
#
# Usage from the reference (one line in src/QuadraticProgramNetworks.jl after the includes):
#     include(joinpath(ENV["QPN_HIP_HOME"], "julia", "QPNHip.jl")); using .QPNHip; QPNHip.install!()
module QPNHip

using SparseArrays, LinearAlgebra

const LIB = get(ENV, "QPN_HIP_LIB", joinpath(@__DIR__, "..", "quadraticprogramnetworks.jl_amd", "libqpn_hip.so"))

const QPN_MEM_HOST = Cint(0)
const QPN_SUCCESS = Int32(1)

const QPN_AVI_FLAG_COLD_START = Int32(1)   # include/qpn_hip.h
struct AviOpts                # qpn_avi_opts, include/qpn_hip.h
    check_tol::Cdouble
    piv_tol::Cdouble
    feas_tol::Cdouble
    comp_tol::Cdouble
    max_pivots::Int32
    flags::Int32
end

const CTX = Ref{Ptr{Cvoid}}(C_NULL)

function ctx()
    if CTX[] == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:qpn_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), parse(Cint, get(ENV, "QPN_HIP_DEVICE", "0")), h)
        rc == 0 || error("qpn_ctx_create failed: " * unsafe_string(ccall((:qpn_strerror, LIB), Cstring, (Cint,), rc)))
        CTX[] = h[]
        atexit(() -> ccall((:qpn_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), CTX[]))
    end
    CTX[]
end

function default_opts()
    o = Ref(AviOpts(0, 0, 0, 0, 0, 0))
    ccall((:qpn_avi_default_opts, LIB), Cvoid, (Ref{AviOpts},), o)
    o[]
end

"""
    solve_mcp(M, q, l, u, z0) -> (status::Int32, z, info)

Same argument list as `PATHSolver.solve_mcp(M, q, l, u, z0)` at src/avi.jl:64: `M` is the
reference's own `SparseMatrixCSC{Float64,Int32}`; its `colptr/rowval/nzval` (1-based) go to the
library as they are.  Julia owns every array; the callee reads them only during the call.
"""
function solve_mcp(M::SparseMatrixCSC{Float64,Int32}, q::Vector{Float64}, l::Vector{Float64},
                   u::Vector{Float64}, z0::Vector{Float64})
    N = Int32(size(M, 1))
    z = copy(z0)
    status = Ref{Int32}(0); resid = Ref{Cdouble}(0); pivots = Ref{Int32}(0)
    o = Ref(default_opts())
    rc = ccall((:qpn_solve_mcp_csc, LIB), Cint,
               (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ref{Int32}, Ref{Cdouble}, Ref{Int32}, Ref{AviOpts}),
               ctx(), N, M.colptr, M.rowval, M.nzval, q, l, u, z, status, resid, pivots, o)
    rc == 0 || error("qpn_solve_mcp_csc: " * unsafe_string(ccall((:qpn_ctx_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx())))
    (status[], z, (; resid = resid[], pivots = pivots[]))
end

"""
    solve_avi_batch(M, q, l, u, z0; kind=nothing) -> (z, status, resid, pivots, active)

`M` is N×N×batch (Julia column-major = the ABI layout) or N×N (shared, strideM = 0); `q,l,u,z0`
are N×batch.  `kind` (N or N×batch, UInt8) marks GAVI rows (second condition, src/avi.jl:22-24).
"""
function solve_avi_batch(M::Array{Float64}, q::Matrix{Float64}, l::Matrix{Float64}, u::Matrix{Float64},
                         z0::Matrix{Float64}; kind::Union{Nothing,Array{UInt8}} = nothing)
    N, batch = size(q)
    strideM = ndims(M) == 3 ? Int64(N * N) : Int64(0)
    z = copy(z0)
    status = zeros(Int32, batch); resid = zeros(batch); pivots = zeros(Int32, batch); active = zeros(UInt8, N, batch)
    kp = kind === nothing ? Ptr{UInt8}(C_NULL) : pointer(kind)
    sk = kind === nothing ? Int64(0) : (ndims(kind) == 2 ? Int64(N) : Int64(0))
    o = Ref(default_opts())
    GC.@preserve kind begin
        rc = ccall((:qpn_solve_avi_batch, LIB), Cint,
                   (Ptr{Cvoid}, Int32, Int32, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Int64,
                    Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Int32}, Ptr{UInt8}, Ref{AviOpts}, Cint),
                   ctx(), Int32(batch), Int32(N), M, strideM, q, l, u, kp, sk, z, status, resid, pivots, active, o, QPN_MEM_HOST)
        rc == 0 || error("qpn_solve_avi_batch failed ($rc)")
    end
    (z, status, resid, pivots, active)
end

"""
    solve_nodes!(x, Qd, R, qd, Ad, B, l, u, w) -> (z, status, resid, pivots, active)

One sweep over `batch` single-node pools: `Qd` n×n×batch, `R` n×p×batch, `qd` n×batch, `Ad` m×n×batch,
`B` m×p×batch, `l,u` m×batch, `w` p (shared) or p×batch.  Each node's reduced KKT system is assembled
on the fly and solved (`qpn_solve_nodes_into`); `z` is (n+m)×batch = [x_d; λ] and the primal blocks are
also written into the columns of `x` (n×batch view of the iterate: the write-back of
src/algorithm.jl:97-101).
"""
function solve_nodes!(x::Union{Nothing,StridedMatrix{Float64}}, Qd::Array{Float64,3}, R::Array{Float64,3}, qd::Matrix{Float64},
                      Ad::Array{Float64,3}, B::Array{Float64,3}, l::Matrix{Float64}, u::Matrix{Float64}, w::VecOrMat{Float64})
    n, batch = size(qd); m = size(l, 1); p = size(w, 1)
    N = n + m
    z = zeros(N, batch); status = zeros(Int32, batch); resid = zeros(batch); pivots = zeros(Int32, batch); active = zeros(UInt8, N, batch)
    o = default_opts(); o = AviOpts(o.check_tol, o.piv_tol, o.feas_tol, o.comp_tol, o.max_pivots, o.flags | QPN_AVI_FLAG_COLD_START)
    xp = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    sx = x === nothing ? Int64(0) : Int64(stride(x, 2))
    GC.@preserve x begin
        rc = ccall((:qpn_solve_nodes_into, LIB), Cint,
                   (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                    Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Int32}, Ptr{UInt8},
                    Ref{AviOpts}, Cint, Ptr{Cdouble}, Int64),
                   ctx(), Int32(batch), Int32(n), Int32(m), Int32(p), Qd, R, qd, Ad, B, l, u, w, ndims(w) == 1 ? Int64(0) : Int64(p),
                   z, status, resid, pivots, active, Ref(o), QPN_MEM_HOST, xp, sx)
        rc == 0 || error("qpn_solve_nodes_into failed ($rc)")
    end
    (z, status, resid, pivots, active)
end

# ---- resident node records: upload once, sweep many times (qpn_nodes_*, include/qpn_hip.h) ---------------------------
const QPN_NODE_QD, QPN_NODE_R, QPN_NODE_Q, QPN_NODE_AD, QPN_NODE_B, QPN_NODE_L, QPN_NODE_U = Int32.(0:6)

"""
    nodes = upload_nodes(Qd, R, qd, Ad, B, l, u)      # arrays as in solve_nodes!; the library keeps its own copy in HBM
    (z, status, resid, pivots, active) = solve_nodes!(nodes, x, w)       # one sweep: only w goes up, only outputs come down
    (solution, lambda, path) = verify_nodes(nodes, xd, w)
    update_nodes!(nodes, QPN_NODE_L, l_new); free_nodes!(nodes)

The outer loop (src/algorithm.jl:13-117) sweeps the same nodes with new parameters every iteration: with the records
resident the PCIe traffic per sweep is w in and the requested outputs out (`want_z = false` returns only the statuses and
the primal blocks in `x`), instead of 22 KB per node in.  The handle also remembers whether any of its nodes needs the
general (pivoting) kernel and keeps their longest-first schedule.
"""
mutable struct Nodes
    h::Ptr{Cvoid}
    batch::Int; n::Int; m::Int; p::Int
end

function upload_nodes(Qd::Array{Float64,3}, R::Array{Float64,3}, qd::Matrix{Float64}, Ad::Array{Float64,3}, B::Array{Float64,3},
                      l::Matrix{Float64}, u::Matrix{Float64})
    n, batch = size(qd); m = size(l, 1); p = size(R, 2)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:qpn_nodes_upload, LIB), Cint,
               (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Cint, Ref{Ptr{Cvoid}}),
               ctx(), Int32(batch), Int32(n), Int32(m), Int32(p), Qd, R, qd, Ad, B, l, u, QPN_MEM_HOST, h)
    rc == 0 || error("qpn_nodes_upload failed ($rc)")
    nodes = Nodes(h[], batch, n, m, p)
    finalizer(free_nodes!, nodes)
    nodes
end

function free_nodes!(nodes::Nodes)
    nodes.h == C_NULL && return nothing
    ccall((:qpn_nodes_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), nodes.h)
    nodes.h = C_NULL
    nothing
end

function update_nodes!(nodes::Nodes, field::Int32, data::Array{Float64})
    rc = ccall((:qpn_nodes_update, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cdouble}, Cint), ctx(), nodes.h, field, data, QPN_MEM_HOST)
    rc == 0 || error("qpn_nodes_update failed ($rc)")
    nothing
end

function solve_nodes!(nodes::Nodes, x::Union{Nothing,StridedMatrix{Float64}}, w::VecOrMat{Float64}; want_z::Bool = true)
    N = nodes.n + nodes.m; batch = nodes.batch
    z = want_z ? zeros(N, batch) : nothing
    status = zeros(Int32, batch); resid = zeros(batch); pivots = zeros(Int32, batch)
    active = want_z ? zeros(UInt8, N, batch) : nothing
    xp = x === nothing ? Ptr{Cdouble}(C_NULL) : pointer(x)
    sx = x === nothing ? Int64(0) : Int64(stride(x, 2))
    GC.@preserve x z active begin
        rc = ccall((:qpn_solve_nodes_h, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}, Ptr{Int32}, Ptr{UInt8},
                    Ptr{Cvoid}, Cint, Ptr{Cdouble}, Int64),
                   ctx(), nodes.h, w, ndims(w) == 1 ? Int64(0) : Int64(nodes.p),
                   want_z ? pointer(z) : Ptr{Cdouble}(C_NULL), status, resid, pivots, want_z ? pointer(active) : Ptr{UInt8}(C_NULL),
                   C_NULL, QPN_MEM_HOST, xp, sx)
        rc == 0 || error("qpn_solve_nodes_h failed ($rc)")
    end
    (z, status, resid, pivots, active)
end

function verify_nodes(nodes::Nodes, xd::Matrix{Float64}, w::VecOrMat{Float64}; tol::Float64 = 1e-4)
    solution = zeros(Int32, nodes.batch); path = zeros(Int32, nodes.batch); lambda = zeros(max(nodes.m, 1), nodes.batch)
    rc = ccall((:qpn_verify_nodes_h, LIB), Cint,
               (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Cdouble, Ptr{Int32}, Ptr{Cdouble}, Ptr{Int32}, Cint),
               ctx(), nodes.h, xd, w, ndims(w) == 1 ? Int64(0) : Int64(nodes.p), tol, solution, lambda, path, QPN_MEM_HOST)
    rc == 0 || error("qpn_verify_nodes_h failed ($rc)")
    (solution, lambda, path)
end

# ---- pool assembly (combine_gavis, src/avi.jl:305-377) and local pieces (local_piece, src/avi_solutions.jl:400-496) ----
struct PoolShape          # qpn_pool_shape, include/qpn_hip.h
    players::Int32
    nd::Int32
    p::Int32
    n_i::Ptr{Int32}
    m_i::Ptr{Int32}
    dpos::Ptr{Int32}
end
const QPN_POOL_REDUCED = Cint(0)
const QPN_POOL_REFERENCE = Cint(1)

"""
    (M, q, l, u, kind) = assemble_pool(n_i, m_i, dpos, nd, Qd, Qp, qd, Ad, Bp, lo, hi, w; form = QPN_POOL_REDUCED)

One Nash pool (the players' blocks stacked in pool order, see include/qpn_hip.h; `dpos` 0-based positions in dec_inds) as
the AVI `solve_avi_batch` takes: the device counterpart of `combine_gavis` (+ `convert` for `QPN_POOL_REFERENCE`).
"""
function assemble_pool(n_i::Vector{Int32}, m_i::Vector{Int32}, dpos::Vector{Int32}, nd::Integer, Qd::Matrix{Float64}, Qp::Matrix{Float64},
                       qd::Vector{Float64}, Ad::Matrix{Float64}, Bp::Matrix{Float64}, lo::Vector{Float64}, hi::Vector{Float64},
                       w::Vector{Float64}; form::Cint = QPN_POOL_REDUCED)
    GC.@preserve n_i m_i dpos begin
        shape = Ref(PoolShape(Int32(length(n_i)), Int32(nd), Int32(length(w)), pointer(n_i), pointer(m_i), pointer(dpos)))
        Nr = Ref{Int32}(0)
        ccall((:qpn_pool_size, LIB), Cint, (Ref{PoolShape}, Cint, Ref{Int32}), shape, form, Nr) == 0 || error("qpn_pool_size: bad shape")
        N = Int(Nr[])
        M = zeros(N, N); q = zeros(N); l = zeros(N); u = zeros(N); kind = zeros(UInt8, N)
        rc = ccall((:qpn_assemble_pools, LIB), Cint,
                   (Ptr{Cvoid}, Ref{PoolShape}, Cint, Int32, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Int64,
                    Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Int64, Ptr{Cdouble}, Ptr{Cdouble},
                    Ptr{Cdouble}, Ptr{UInt8}, Cint),
                   ctx(), shape, form, Int32(1), Qd, 0, Qp, 0, qd, 0, Ad, 0, Bp, 0, lo, hi, 0, w, 0, M, Int64(N * N), q, l, u, kind, QPN_MEM_HOST)
        rc == 0 || error("qpn_assemble_pools failed ($rc)")
        return (M, q, l, u, kind)
    end
end

"""
    (Ap, lp, up, keep) = local_pieces(Qd, R, qd, Ad, B, l, u, K)

`local_piece` (src/avi_solutions.jl:400-496, before `simplify`) for the recipes `K` ((n+m)×pieces UInt8 codes 1..8) of ONE
node's GAVI (process_solution_graph, src/avi.jl:447-477): `Ap` is 2(n+m) × (n+m+p) × pieces over [x_d; λ; x_p].
"""
function local_pieces(Qd::Matrix{Float64}, R::Matrix{Float64}, qd::Vector{Float64}, Ad::Matrix{Float64}, B::Matrix{Float64},
                      l::Vector{Float64}, u::Vector{Float64}, K::Matrix{UInt8})
    n = length(qd); m = length(l); p = size(R, 2); N = n + m; pieces = size(K, 2)
    Ap = zeros(2N, N + p, pieces); lp = zeros(2N, pieces); up = zeros(2N, pieces); keep = zeros(UInt8, 2N, pieces)
    node_of = zeros(Int32, pieces)
    rc = ccall((:qpn_local_pieces, LIB), Cint,
               (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Int32}, Ptr{UInt8}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{UInt8}, Cint),
               ctx(), Int32(pieces), Int32(1), Int32(n), Int32(m), Int32(p), Qd, R, qd, Ad, B, l, u, node_of, K, Ap, lp, up, keep, QPN_MEM_HOST)
    rc == 0 || error("qpn_local_pieces failed ($rc)")
    (Ap, lp, up, keep)
end

"""
    recipes_batch(masks, counts) -> (K, node_of)

`all_Ks` (src/avi_solutions.jl:200-215) for MANY solutions in one launch (`qpn_recipes_batch`): `masks` is N x nodes (the `active`
output of a solve, one column per node), `counts[b]` the number of recipes wanted of node b (at most the product of its rows'
code counts).  Returns the recipes K (N x total, one column per recipe) and the 0-based node of each.
"""
function recipes_batch(masks::Matrix{UInt8}, counts::Vector{<:Integer})
    N, nodes = size(masks)
    offsets = Int64[0; cumsum(Int64.(counts))]
    total = Int(offsets[end])
    K = zeros(UInt8, N, total); node_of = zeros(Int32, total)
    rc = ccall((:qpn_recipes_batch, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}, Ptr{Int64}, Ptr{UInt8}, Ptr{Int32}, Cint),
               ctx(), Int32(nodes), Int32(N), masks, offsets, K, node_of, QPN_MEM_HOST)
    rc == 0 || error("qpn_recipes_batch failed ($rc)")
    (K, node_of)
end

"""
    reduced_pieces(Qd, R, qd, Ad, B, l, u, K, node_of; tol = 1e-9) -> (Ar, lr, ur, rows, flags)

`local_piece` (src/avi_solutions.jl:400-496) for the recipes K (one column each) over the node records (third index = node, as
`solve_nodes!` takes them), with the m multiplier columns eliminated through each piece's own equality rows (`qpn_reduced_pieces`):
piece t lives on the records of node `node_of[t]` (0-based).  `Ar[:, :, t]` is the cap x (n + p) row matrix over `[x_d; x_p]`
(cap = n + 2m), of which the first `rows[t]` rows are live; `flags[t] != 0`: a multiplier was pinned by no equality row -- that
piece needs the polyhedral projection (src/avi_solutions.jl:79-91) on the host.
"""
function reduced_pieces(Qd::Array{Float64,3}, R::Array{Float64,3}, qd::Matrix{Float64}, Ad::Array{Float64,3}, B::Array{Float64,3},
                        l::Matrix{Float64}, u::Matrix{Float64}, K::Matrix{UInt8}, node_of::Vector{Int32}; tol::Float64 = 1e-9)
    n, nodes = size(qd); m = size(l, 1); p = size(R, 2); pieces = size(K, 2); cap = n + 2m
    Ar = zeros(cap, n + p, pieces); lr = zeros(cap, pieces); ur = zeros(cap, pieces)
    rows = zeros(Int32, pieces); flags = zeros(Int32, pieces)
    rc = ccall((:qpn_reduced_pieces, LIB), Cint,
               (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                Ptr{Cdouble}, Ptr{Int32}, Ptr{UInt8}, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Int32}, Cint),
               ctx(), Int32(pieces), Int32(nodes), Int32(n), Int32(m), Int32(p), Qd, R, qd, Ad, B, l, u, node_of, K, tol, Ar, lr, ur, rows, flags,
               QPN_MEM_HOST)
    rc == 0 || error("qpn_reduced_pieces failed ($rc)")
    (Ar, lr, ur, rows, flags)
end

"""
    order_nodes_by_pivots!(pivots)

Schedule hint for later `solve_nodes!` calls over the same nodes (longest solves first); `pivots` is the
`pivots` output of an earlier sweep.  `clear_node_order!()` removes it.  Results do not depend on it.
"""
function order_nodes_by_pivots!(pivots::Vector{Int32})
    rc = ccall((:qpn_order_nodes_by_pivots, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int32, Cint), ctx(), pivots, Int32(length(pivots)), QPN_MEM_HOST)
    rc == 0 || error("qpn_order_nodes_by_pivots failed ($rc)")
    nothing
end
# period (in calls) of the context's own refresh of that hint; 0 = off (default 16)
set_auto_schedule!(period::Integer) = (ccall((:qpn_ctx_set_auto_schedule, LIB), Cint, (Ptr{Cvoid}, Int32), ctx(), Int32(period)); nothing)
clear_node_order!() = (ccall((:qpn_set_node_order, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int32, Cint), ctx(), C_NULL, Int32(0), QPN_MEM_HOST); nothing)
# kernel routes with identical contracts (A/B measurements; include/qpn_hip.h, QPN_OPT_*): MID_ROUTE 1 = fused workgroup kernel per
# node of 33 .. 128 variables or constraints (default; one wavefront per node up to 48), 0 = the general route (the cross-check);
# BIG_ROUTE takes 1 only (blocked crash straight from the records for nodes up to 256 x 256)
# SYM_ROUTE 1 = resident records whose Qd blocks are all bitwise symmetric take the kernel variants that use it (default), 0 = never.
const QPN_OPT_MID_ROUTE = Int32(1)
const QPN_OPT_BIG_ROUTE = Int32(2)
const QPN_OPT_SYM_ROUTE = Int32(3)
function set_option!(option::Integer, value::Integer)
    rc = ccall((:qpn_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Int32, Int32), ctx(), Int32(option), Int32(value))
    rc == 0 || error("qpn_ctx_set_option failed ($rc)")
    nothing
end

# ---- multi-GPU (one Julia process per GPU, e.g. under MPI.jl / Distributed): replicas of the iterate ----
# The ABI works on DEVICE pointers here (the host-array routes above stage through the library's workspace):
# `x_dev` below is an address inside a buffer from `shared_alloc`, e.g. wrapped by AMDGPU.jl's unsafe_wrap.
const QPN_IPC_HANDLE_BYTES = 64
const QPN_SHARED_FINE_GRAINED = Cint(1)
const QPN_SWEEP_BOX_BYTES = 512

"""
    shared_alloc(nbytes; fine_grained=false) -> (ptr::Ptr{Cvoid}, handle::Vector{UInt8})

Zeroed device buffer on this process's GPU and its IPC handle; send the handle to the peer processes
(MPI.Allgather, a socket, ...) and `shared_open` it there.
"""
function shared_alloc(nbytes::Integer; fine_grained::Bool = false)
    p = Ref{Ptr{Cvoid}}(C_NULL); h = zeros(UInt8, QPN_IPC_HANDLE_BYTES)
    rc = ccall((:qpn_shared_alloc, LIB), Cint, (Ptr{Cvoid}, Csize_t, Cint, Ref{Ptr{Cvoid}}, Ptr{UInt8}),
               ctx(), Csize_t(nbytes), fine_grained ? QPN_SHARED_FINE_GRAINED : Cint(0), p, h)
    rc == 0 || error("qpn_shared_alloc failed")
    (p[], h)
end
function shared_open(handle::Vector{UInt8})
    p = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:qpn_shared_open, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ref{Ptr{Cvoid}}), ctx(), handle, p) == 0 || error("qpn_shared_open failed")
    p[]
end
shared_close(p::Ptr{Cvoid}) = (ccall((:qpn_shared_close, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), p); nothing)
shared_free(p::Ptr{Cvoid}) = (ccall((:qpn_shared_free, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), ctx(), p); nothing)

"""
    set_primal_mirrors!(own, nbytes, peers)

Later device-memory `qpn_solve_nodes_into` calls whose `x` lies inside `own` also store every primal block at the
same offset of each peer buffer (at most 7).  `set_primal_mirrors!()` clears.
"""
function set_primal_mirrors!(own::Ptr{Cvoid} = C_NULL, nbytes::Integer = 0, peers::Vector{Ptr{Cvoid}} = Ptr{Cvoid}[])
    rc = ccall((:qpn_set_primal_mirrors, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Int32, Ptr{Ptr{Cvoid}}),
               ctx(), own, Csize_t(nbytes), Int32(length(peers)), peers)
    rc == 0 || error("qpn_set_primal_mirrors failed")
    nothing
end

"""
    sweep_status!(out_dev, status_dev, resid_dev, count; rank=0, boxes=Ptr{Cvoid}[], epoch=0, timeout_ms=1000)

`out_dev[1:4]` (device) <- (items not solved, max resid, all ranks arrived, missed barriers so far), combined over
`length(boxes)` ranks through their mailboxes (each `QPN_SWEEP_BOX_BYTES`, `shared_alloc(...; fine_grained=true)`).
Asynchronous; enqueued after the solve it is also the barrier that completes the replicas (src/algorithm.jl:95-109).
"""
function sweep_status!(out_dev::Ptr{Cvoid}, status_dev::Ptr{Cvoid}, resid_dev::Ptr{Cvoid}, count::Integer;
                       rank::Integer = 0, boxes::Vector{Ptr{Cvoid}} = Ptr{Cvoid}[], epoch::Integer = 0, timeout_ms::Integer = 1000)
    world = max(length(boxes), 1)
    rc = ccall((:qpn_sweep_status, LIB), Cint,
               (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Int32, Int32, Ptr{Ptr{Cvoid}}, UInt64, Int32),
               ctx(), status_dev, resid_dev, Int32(count), out_dev, Int32(rank), Int32(world),
               world > 1 ? pointer(boxes) : Ptr{Ptr{Cvoid}}(C_NULL), UInt64(epoch), Int32(timeout_ms))
    rc == 0 || error("qpn_sweep_status failed")
    nothing
end

# A node record with its equality rows (l == u) moved into the free block: z = [x; mu_E; lambda_I].  The multiplier of an equality
# row is a free variable of the node's AVI (src/avi.jl:113-128), so this is the same complementarity system -- but the fused node
# kernels, which decline a record that carries an equality row, take it (DESIGN.md section 5a; the Python mirror's
# level_batch.free_equalities).  Every parent's record has such rows: its children's pieces.  Arrays in MATH layout (row-major
# meaning: Qd n x n, R n x p, Ad m x n, B m x p); returns the rewritten record and nx = n (the first nx entries of z are x).
function free_equalities(Qd::AbstractMatrix, R::AbstractMatrix, qd::AbstractVector, Ad::AbstractMatrix, B::AbstractMatrix,
                         l::AbstractVector, u::AbstractVector)
    E = findall(i -> isfinite(l[i]) && l[i] == u[i], eachindex(l))
    isempty(E) && return (Qd, R, qd, Ad, B, l, u, size(Qd, 1))
    I = setdiff(collect(eachindex(l)), E)
    ne = length(E)
    Qd2 = [Qd -transpose(Ad[E, :]); Ad[E, :] zeros(ne, ne)]
    R2 = [R; B[E, :]]
    qd2 = [qd; -l[E]]
    Ad2 = [Ad[I, :] zeros(length(I), ne)]
    return (Qd2, R2, qd2, Ad2, B[I, :], l[I], u[I], size(Qd, 1))
end

# ---- drop-in bodies -------------------------------------------------------------------------------
# src/avi.jl:63-77 with the PATH call replaced; StatusCode / check_avi_solution stay the reference's.
function solve_avi_hip(QPN, avi, z0, w; convergence_tolerance = 1e-10)
    (st, z, info) = solve_mcp(avi.M, avi.N * w + avi.o, avi.l, avi.u, collect(Float64, z0))
    (; sol_bad, degree, r) = QPN.check_avi_solution(avi, z, w)          # kept: src/avi.jl:71
    sol_bad && return (; z, status = QPN.FAILURE, info = (; path_status = st, info))
    status = st == QPN_SUCCESS ? QPN.SUCCESS : QPN.FAILURE
    (; z, status, info = (; path_status = st, info))
end

# src/qp_processing.jl:12-33 (PATH branch): same MCP, same error convention (:30).
function solve_qp_hip(Q, q, A, l, u)
    n = size(Q, 1); m = size(A, 1)
    M = SparseMatrixCSC{Float64,Int32}([Q -A' spzeros(n, m); A spzeros(m, m) -sparse(1.0I, m, m); spzeros(m, n) sparse(1.0I, m, m) spzeros(m, m)])
    (st, z, _) = solve_mcp(M, [q; zeros(2m)], [fill(-Inf, n + m); l], [fill(Inf, n + m); u], zeros(2m + n))
    st == QPN_SUCCESS || error("Solver failure. Status value is $st")
    z[1:n]
end

"""
Overrides `solve_avi` (src/avi.jl:63-77) of the loaded QuadraticProgramNetworks module.  The PATH
branch of `solve_qp` (src/qp_processing.jl:12-33) shares its function with the OSQP branch, so it is
rerouted by the three-line source edit shown in INTEGRATION.md (calls `QPNHip.solve_qp_hip`).
"""
function install!(QPN = Main.QuadraticProgramNetworks)
    @eval QPN solve_avi(avi::AVI, z0, w; convergence_tolerance = 1e-10) = $(solve_avi_hip)($QPN, avi, z0, w; convergence_tolerance)
    nothing
end

end # module

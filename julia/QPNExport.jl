# QPNExport.jl -- writes a QPNet built by the reference (setup(...)) in the interchange layout that
# quadraticprogramnetworks.jl_amd/interchange.py reads (SURVEY.md section 8(f) F4):
#     <dir>/meta.json + one NPY (v1.0, '<f8', fortran_order = True: Julia's own memory order) per array.
# Plain Julia: an NPY writer and a JSON emitter of a few lines, no package beyond SparseArrays.
#
# THIS CONTAINER HAS NO JULIA: the file is shipped as source and has never been executed; the byte layout it
# writes is the one tests/test_interchange.py builds by hand and loads.  Fields follow src/programs.jl:16-92.
#
#     include("QPNExport.jl"); QPNExport.export_qpnet("net_dir", qpn)
module QPNExport

using SparseArrays

function write_npy(path::AbstractString, A::AbstractArray{Float64})
    shape = ndims(A) == 1 ? "($(length(A)),)" : "(" * join(size(A), ", ") * ")"
    dict = "{'descr': '<f8', 'fortran_order': True, 'shape': $shape, }"
    # magic (6) + version (2) + header length (2) + header, padded with spaces to a multiple of 64, '\n' last
    total = 10 + length(dict) + 1
    pad = (64 - total % 64) % 64
    header = dict * " "^pad * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93, UInt8('N'), UInt8('U'), UInt8('M'), UInt8('P'), UInt8('Y'), 0x01, 0x00])
        write(io, htol(UInt16(length(header))))
        write(io, header)
        write(io, htol.(vec(Array(A))))
    end
end

json(x::AbstractString) = "\"" * replace(x, "\\" => "\\\\", "\"" => "\\\"") * "\""
json(x::Bool) = x ? "true" : "false"
json(x::Integer) = string(x)
json(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : error("non-finite number in meta.json")
json(::Nothing) = "null"
json(x::Symbol) = json(string(x))
json(x::AbstractVector) = "[" * join((json(v) for v in x), ", ") * "]"
json(x::AbstractDict) = "{" * join((json(string(k)) * ": " * json(v) for (k, v) in sort(collect(x); by = p -> string(p[1]))), ", ") * "}"

"""
    export_qpnet(dir, qpn)

`qpn::QuadraticProgramNetworks.QPNet`.  Variable indices and ids stay 1-based (`index_base = 1`).
"""
function export_qpnet(dir::AbstractString, qpn)
    mkpath(dir)
    nv = length(qpn.default_initialization)
    arrays = Dict{String,Any}()
    put(name, A) = (write_npy(joinpath(dir, name * ".npy"), Array{Float64}(A)); arrays[name] = collect(size(A)))
    qps = Any[]
    for pid in sort(collect(keys(qpn.qps)))
        qp = qpn.qps[pid]
        put("qp$(pid)_Q", Matrix(qp.f.Q)); put("qp$(pid)_q", qp.f.q)
        push!(qps, Dict("id" => pid, "k" => qp.f.k, "constraint_indices" => collect(qp.constraint_indices),
                        "var_indices" => collect(qp.var_indices)))
    end
    cons = Any[]
    for cid in sort(collect(keys(qpn.constraints)))
        con = qpn.constraints[cid]
        A = reduce(vcat, (Matrix(s.a') for s in con.poly))          # = vectorize(poly), src/sets.jl:213-221
        l = [s.l for s in con.poly]; u = [s.u for s in con.poly]
        put("con$(cid)_A", A); put("con$(cid)_l", l); put("con$(cid)_u", u)
        push!(cons, Dict("id" => cid, "group_mapping" => Dict(string(k) => v for (k, v) in con.group_mapping)))
    end
    put("default_initialization", qpn.default_initialization)
    o = qpn.options
    lv = o.levels_to_remove_subsets
    opts = Dict{String,Any}(
        "shared_variable_mode" => string(o.shared_variable_mode), "max_iters" => o.max_iters, "tol" => o.tol,
        "high_dimension" => o.high_dimension, "high_dimension_max_iters" => o.high_dimension_max_iters,
        "num_projections" => o.num_projections, "make_requests" => o.make_requests,
        "exploration_vertices" => o.exploration_vertices, "try_hull" => o.try_hull,
        "debug_visualize" => o.debug_visualize, "gen_solution_map" => o.gen_solution_map,
        "levels_to_remove_subsets" => (lv isa AbstractSet{Int} && !(lv isa Set) ? nothing : sort(collect(lv))),   # NaturalNumbers() -> null
        "check_convexity" => o.check_convexity, "check_for_cycling" => o.check_for_cycling,
        "perturb_to_continue" => o.perturb_to_continue)
    meta = Dict{String,Any}(
        "format" => "qpnet-interchange/1", "num_vars" => nv, "index_base" => 1, "qps" => qps, "constraints" => cons,
        "network_edges" => Dict(string(k) => sort(collect(v)) for (k, v) in qpn.network_edges),
        "options" => opts, "arrays" => arrays)
    open(joinpath(dir, "meta.json"), "w") do io
        write(io, json(meta))
    end
    dir
end

end # module

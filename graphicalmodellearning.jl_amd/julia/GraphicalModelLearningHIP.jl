# GraphicalModelLearningHIP.jl -- the reference-side binding of libgml_hip (include/gml.h).
#
# Drop this file next to the reference package (or `include` it after `using
# GraphicalModelLearning`) and point ENV["LIBGML_HIP"] at libgml_hip.so.  It adds ONE new
# GMLMethod subtype, `HIP`, and the `learn(samples, formulation, ::HIP)` methods; every type,
# constructor and default of GraphicalModelLearning.jl:20-65 is reused unchanged, so
#
#     learn(samples, RISE(), HIP())          # instead of learn(samples, RISE(), NLP())
#
# is the whole migration.  The file only marshals arguments: all arithmetic is in the library.
# NOTE: there is no Julia in the build/test image of this repository, so this file is kept 1:1
# with the Python ctypes binding (graphicalmodellearning.jl_amd/_lib.py), which IS exercised by
# the test-suite on MI355X.
module GraphicalModelLearningHIP

using GraphicalModelLearning
import GraphicalModelLearning: learn, GMLMethod, GMLFormulation, RISE, RISEA, logRISE, RPLE, multiRISE, FactorGraph

export HIP, trim_cache

const libgml = get(ENV, "LIBGML_HIP", "libgml_hip.so")

# gml.h constants
const GML_ABI_VERSION = Cint(6)     # the revision of include/gml.h this file mirrors; checked against the library in __init__
const GML_OK, GML_ENOTCONV = Cint(0), Cint(2)
const GML_RISE, GML_LOGRISE, GML_RPLE = Cint(0), Cint(1), Cint(2)
const GML_I64, GML_F64 = Cint(2), Cint(3)
const GML_PREC_F64, GML_PREC_I8X, GML_PREC_AUTO, GML_PREC_I8W = Cint(0), Cint(1), Cint(2), Cint(3)

struct GmlOpts                      # struct gml_opts
    tol::Cdouble; max_iter::Int32; precision::Int32; max_working::Int32; max_add::Int32
    verbose::Int32; hess_samples::Int32; polish::Int32; max_cg::Int32
    limbs_fwd::Int32; hv_limbs_fwd::Int32; hv_limbs_bwd::Int32; debug_row::Int32   # 0 = the library's defaults
    hv_subsample::Int32; coarse::Int32
    cg_viol_frac::Cdouble; cg_eta::Cdouble
end

struct GmlStats                     # struct gml_stats
    iterations::Int32; passes::Int32; forward_passes::Int32; hessian_passes::Int32
    node_evals::Int64; max_kkt::Cdouble; lambda::Cdouble
    t_pack::Cdouble; t_pass::Cdouble; t_hess::Cdouble; t_host::Cdouble; t_total::Cdouble
    not_converged::Int32; polished::Int32
    hv_evals::Int64
    t_assemble::Cdouble
end

# A library built from another revision of gml.h would read GmlOpts / write GmlStats at the wrong offsets without any error:
# refuse it when the module loads (gml.h, "ABI identity").
function __init__()
    v = try
        ccall((:gml_abi_version, libgml), Cint, ())
    catch
        error("$libgml predates gml_abi_version(): rebuild it (this binding is ABI $GML_ABI_VERSION)")
    end
    so, ss = ccall((:gml_sizeof_opts, libgml), Int64, ()), ccall((:gml_sizeof_stats, libgml), Int64, ())
    (v, so, ss) == (GML_ABI_VERSION, sizeof(GmlOpts), sizeof(GmlStats)) ||
        error("$libgml has ABI version / sizeof(gml_opts) / sizeof(gml_stats) = $((v, so, ss)), this binding was written against " *
              "$((GML_ABI_VERSION, sizeof(GmlOpts), sizeof(GmlStats))): library and binding come from different revisions")
end

"""
    HIP(; tol=1e-9, precision=:auto, device=0, devices=nothing, max_iter=100, max_working=512, max_add=64,
        hess_samples=0, polish=true, verbose=0, node_range=nothing)

GMLMethod that solves every node-wise problem on MI355X through libgml_hip (same fields and defaults as the Python
twin, graphicalmodellearning.jl_amd/formulations.py).
`precision = :auto` (default) is `:i8x`, and `:i8w` for tolerances below 2e-10 and for small problems (the reference's fixtures); `:i8x` is the int8-limb fixed-point pass; rows it cannot bring below `tol` are finished on the
FP64 path unless `polish = false`; `:i8w` is the same int8 matrix-core pass at the width of Float64 (theta in 54-bit, the weights in
47-bit limbs, exp in FP64: objective and gradient to the 1e-12 the FP64 path is held to, about 1.5x the time of `:i8x`); `:f64` runs
FP64 MFMA throughout.  `devices = 0:7` shards the nodes over several
GPUs of this node from this one process (gml_multi_*: one handle and one host thread per GPU inside the library, the
row blocks are written straight into the result matrix).
"""
mutable struct HIP <: GMLMethod
    tol::Float64
    precision::Symbol
    device::Int
    devices::Union{Nothing,Vector{Int}}
    max_iter::Int
    max_working::Int
    max_add::Int
    hess_samples::Int
    polish::Bool
    verbose::Int
    node_range::Union{Nothing,Tuple{Int,Int}}   # 1-based inclusive, for one-process-per-GPU sharding
end
HIP(; tol=1e-9, precision=:auto, device=0, devices=nothing, max_iter=100, max_working=512, max_add=64, hess_samples=0,
    polish=true, verbose=0, node_range=nothing) =
    HIP(tol, precision, device, devices === nothing ? nothing : collect(Int, devices), max_iter, max_working, max_add,
        hess_samples, polish, verbose, node_range)

function precision_id(s::Symbol)
    s == :auto && return GML_PREC_AUTO
    s == :i8x && return GML_PREC_I8X
    s == :i8w && return GML_PREC_I8W
    s == :f64 && return GML_PREC_F64
    throw(ArgumentError("HIP: unknown precision :$s (use :auto, :i8x, :i8w or :f64)"))   # as the C ABI and the Python twin do
end

gmlopts(m::HIP) = Ref(GmlOpts(m.tol, m.max_iter, precision_id(m.precision), m.max_working, m.max_add,
                              m.verbose, m.hess_samples, m.polish ? 0 : -1, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0))

lasterr() = unsafe_string(ccall((:gml_last_error, libgml), Cstring, ()))

formulation_id(::Union{RISE,RISEA,multiRISE}) = GML_RISE
formulation_id(::logRISE) = GML_LOGRISE
formulation_id(::RPLE) = GML_RPLE

# rows node0+1 .. node1 of the reconstruction (un-symmetrised), P columns
function solve_rows(samples::Array{T,2}, formulation, method::HIP, order::Int) where T <: Real
    s = T <: AbstractFloat ? convert(Array{Float64,2}, samples) : convert(Array{Int64,2}, samples)
    dtype = eltype(s) == Float64 ? GML_F64 : GML_I64
    K, n = size(s, 1), size(s, 2) - 1
    method.devices === nothing || method.node_range === nothing ||
        throw(ArgumentError("HIP: devices (all nodes over several GPUs) and node_range (one shard) exclude each other"))
    method.devices === nothing || return solve_rows_multi(s, dtype, formulation, method, order)
    n0, n1 = method.node_range === nothing ? (0, n) : (method.node_range[1] - 1, method.node_range[2])
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:gml_problem_create, libgml), Cint,
               (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Cint, Cint, Int64, Int64, Cint, Ref{Ptr{Cvoid}}),
               s, dtype, K, n, K, 1 #= column-major =#, order, n0, n1, method.device, handle)
    rc == GML_OK || error("gml_problem_create: $(lasterr())")
    try
        P = Ref{Int64}(0)
        ccall((:gml_problem_info, libgml), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Int64}, Ptr{Int64}),
              handle[], C_NULL, C_NULL, C_NULL, P, C_NULL, C_NULL)
        R = n1 - n0
        out = Array{Float64}(undef, P[], R)          # C row-major (R x P) == Julia (P x R)
        opts = gmlopts(method)
        stats = Ref{GmlStats}()
        rc = ccall((:gml_learn, libgml), Cint,
                   (Ptr{Cvoid}, Cint, Cdouble, Ref{GmlOpts}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{GmlStats}),
                   handle[], formulation_id(formulation), Float64(formulation.regularizer), opts, out, C_NULL, stats)
        # the reference: @assert JuMP.termination_status(model) == JuMP.MOI.LOCALLY_SOLVED  (:180)
        rc == GML_ENOTCONV && throw(AssertionError(lasterr()))
        rc == GML_OK || error("gml_learn: $(lasterr())")
        return permutedims(out), nothing, n0             # R x P
    finally
        ccall((:gml_problem_destroy, libgml), Cvoid, (Ptr{Cvoid},), handle[])
    end
end

# The library keeps the device blocks of destroyed handles for the next problem of the same shape (gml.h: gml_trim_cache);
# this hands them back to the driver and returns the number of bytes released.
trim_cache() = Int(ccall((:gml_trim_cache, libgml), Int64, ()))

# all nodes over method.devices: gml_multi_* (one handle + one host thread per GPU inside the library)
function solve_rows_multi(s, dtype, formulation, method::HIP, order::Int)
    K, n = size(s, 1), size(s, 2) - 1
    devs = convert(Vector{Cint}, method.devices)
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:gml_multi_create, libgml), Cint,
               (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Cint, Cint, Ptr{Cint}, Cint, Ref{Ptr{Cvoid}}),
               s, dtype, K, n, K, 1, order, devs, length(devs), handle)
    rc == GML_OK || error("gml_multi_create: $(lasterr())")
    try
        P = Ref{Int64}(0)
        ccall((:gml_multi_info, libgml), Cint,
              (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Cdouble}, Ref{Int64}, Ptr{Cint}, Ptr{UInt8}),
              handle[], C_NULL, C_NULL, C_NULL, P, C_NULL, C_NULL)
        out = Array{Float64}(undef, P[], n)               # C row-major (n x P) == Julia (P x n)
        stats = Ref{GmlStats}()
        rc = ccall((:gml_multi_learn, libgml), Cint,
                   (Ptr{Cvoid}, Cint, Cdouble, Ref{GmlOpts}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{GmlStats}, Ptr{Ptr{Cdouble}}),
                   handle[], formulation_id(formulation), Float64(formulation.regularizer), gmlopts(method), out, C_NULL, stats, C_NULL)
        rc == GML_ENOTCONV && throw(AssertionError(lasterr()))   # @assert ... LOCALLY_SOLVED (:180)
        rc == GML_OK || error("gml_multi_learn: $(lasterr())")
        return permutedims(out), nothing, 0
    finally
        ccall((:gml_multi_destroy, libgml), Cvoid, (Ptr{Cvoid},), handle[])
    end
end

# RISE / logRISE / RPLE / RISEA: n x n matrix, diagonal = fields  (:154-189, :263-298, :301-336, :210-260)
# One GPU, all nodes: gml_learn_matrix -- the solve and `0.5 * (reconstruction + transpose(reconstruction))` (:184-186) in one call,
# the rows never leaving the device (on the host that line walks the transposed operand with a stride of n doubles: 0.2 s at n = 4096).
# Node shards and several GPUs: the gathered rows through gml_matrix_symmetrize.
function learn(samples::Array{T,2}, formulation::Union{RISE,RISEA,logRISE,RPLE}, method::HIP) where T <: Real
    n = size(samples, 2) - 1
    if method.devices === nothing && method.node_range === nothing && formulation.symmetrization
        s = T <: AbstractFloat ? convert(Array{Float64,2}, samples) : convert(Array{Int64,2}, samples)
        K = size(s, 1)
        handle = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gml_problem_create, libgml), Cint,
                   (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Cint, Cint, Int64, Int64, Cint, Ref{Ptr{Cvoid}}),
                   s, eltype(s) == Float64 ? GML_F64 : GML_I64, K, n, K, 1, 2, 0, n, method.device, handle)
        rc == GML_OK || error("gml_problem_create: $(lasterr())")
        try
            out = Array{Float64}(undef, n, n)             # symmetric: row-major and column-major coincide
            stats = Ref{GmlStats}()
            rc = ccall((:gml_learn_matrix, libgml), Cint,
                       (Ptr{Cvoid}, Cint, Cdouble, Cint, Ref{GmlOpts}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{GmlStats}),
                       handle[], formulation_id(formulation), Float64(formulation.regularizer), Cint(1), gmlopts(method), out, C_NULL, stats)
            rc == GML_ENOTCONV && throw(AssertionError(lasterr()))              # @assert ... LOCALLY_SOLVED (:180)
            rc == GML_OK || error("gml_learn_matrix: $(lasterr())")
            return out
        finally
            ccall((:gml_problem_destroy, libgml), Cvoid, (Ptr{Cvoid},), handle[])
        end
    end
    reconstruction, _, _ = solve_rows(samples, formulation, method, 2)
    if formulation.symmetrization && size(reconstruction, 1) == size(reconstruction, 2)
        rt = permutedims(reconstruction)                                          # C row-major n x n
        rc = ccall((:gml_matrix_symmetrize, libgml), Cint, (Ptr{Cdouble}, Int64, Int64, Cint, Ptr{Cdouble}),
                   rt, n, n, method.devices === nothing ? method.device : method.devices[1], rt)   # :184-186, in place
        rc == GML_OK || error("gml_matrix_symmetrize: $(lasterr())")
        reconstruction = rt                                                       # symmetric: no transpose back needed
    end
    return reconstruction
end

# multiRISE: FactorGraph keyed by (u, ascending others), optionally symmetrised  (:83-152).
# The per-key storage (:129-132), the grouping by sorted key and the `mean` (:135-149) run on the device (gml_learn_terms /
# gml_terms_assemble: one kernel, keys <-> positions in closed form); what comes back is ONE weight vector in the order the
# reference lists a model's terms in (models.jl:61,72: by (length, key)) and the matching key table from gml_terms_keys, and the
# Dict the FactorGraph constructor wants is built from the two in a single comprehension.
function learn(samples::Array{T,2}, formulation::multiRISE, method::HIP) where T <: Real
    order = formulation.interaction_order
    n = size(samples, 2) - 1
    sym = formulation.symmetrization ? Cint(1) : Cint(0)
    nterms = ccall((:gml_terms_count, libgml), Int64, (Int64, Cint, Cint), n, order, sym)
    nterms >= 0 || error("gml_terms_count: $(lasterr())")
    weights = Vector{Float64}(undef, nterms)
    if method.devices === nothing && method.node_range === nothing
        # one GPU, all nodes: solve + assembly in one call, the n x P rows never leave the device
        s = T <: AbstractFloat ? convert(Array{Float64,2}, samples) : convert(Array{Int64,2}, samples)
        K = size(s, 1)
        handle = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:gml_problem_create, libgml), Cint,
                   (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Cint, Cint, Int64, Int64, Cint, Ref{Ptr{Cvoid}}),
                   s, eltype(s) == Float64 ? GML_F64 : GML_I64, K, n, K, 1, order, 0, n, method.device, handle)
        rc == GML_OK || error("gml_problem_create: $(lasterr())")
        try
            stats = Ref{GmlStats}()
            rc = ccall((:gml_learn_terms, libgml), Cint,
                       (Ptr{Cvoid}, Cint, Cdouble, Cint, Ref{GmlOpts}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{GmlStats}),
                       handle[], GML_RISE, Float64(formulation.regularizer), sym, gmlopts(method), weights, C_NULL, stats)
            rc == GML_ENOTCONV && throw(AssertionError(lasterr()))                # @assert ... LOCALLY_SOLVED (:127)
            rc == GML_OK || error("gml_learn_terms: $(lasterr())")
        finally
            ccall((:gml_problem_destroy, libgml), Cvoid, (Ptr{Cvoid},), handle[])
        end
    else
        method.node_range === nothing ||
            throw(ArgumentError("HIP: multiRISE assembles a FactorGraph from the rows of all nodes; node_range solves a shard"))
        rows, _, _ = solve_rows(samples, formulation, method, order)              # n x P (gml_multi_learn over method.devices)
        rt = permutedims(rows)                                                    # back to C row-major n x P
        rc = ccall((:gml_terms_assemble, libgml), Cint, (Ptr{Cdouble}, Int64, Int64, Cint, Cint, Cint, Ptr{Cdouble}),
                   rt, size(rt, 1), n, order, sym, method.devices[1], weights)
        rc == GML_OK || error("gml_terms_assemble: $(lasterr())")
    end
    keys = Array{Int32}(undef, order, nterms)                                     # C [nterms][order], 0-based, -1 = unused slot
    rc = ccall((:gml_terms_keys, libgml), Cint, (Int64, Cint, Cint, Int64, Int64, Ptr{Int32}), n, order, sym, 0, nterms, keys)
    rc == GML_OK || error("gml_terms_keys: $(lasterr())")
    reconstruction = Dict{Tuple,Real}(tuple((Int(k) + 1 for k in view(keys, :, t) if k >= 0)...) => weights[t] for t in 1:nterms)
    return FactorGraph(order, n, :spin, reconstruction)                           # :151
end

end # module

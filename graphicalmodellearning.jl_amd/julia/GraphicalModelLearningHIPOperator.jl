# GraphicalModelLearningHIPOperator.jl -- the reference's RISEA path (GraphicalModelLearning.jl:191-260) with the
# hand-written operator pair replaced by the device operator of libgml_hip.
#
# The reference registers `obj(x...)` / `grad(g, x...)` with JuMP (`JuMP.register(model, :obj, num_spins, obj, grad)`,
# :233) and sets them as the nonlinear objective (:235-237); Ipopt then drives them with a limited-memory Hessian (a
# multivariate user-defined operator provides no second derivatives).  Here the same two callbacks are thin `ccall`s
# of `gml_objgrad_batch` with ONE row (include/gml.h): the K x n sums run on the GPU, JuMP/Ipopt stay in charge of the
# optimisation, `NLP(solver)` keeps its meaning.  The spins are uploaded once per `learn` call, not once per node.
#
#     using GraphicalModelLearning, Ipopt
#     include("GraphicalModelLearningHIPOperator.jl"); using .GraphicalModelLearningHIPOperator
#     learn(samples, RISEA(), HIPOperator(NLP(Ipopt.Optimizer)))
#
# Written against the JuMP >= 1.15 nonlinear interface (`@operator`); for the legacy interface the reference itself
# uses, replace the `@operator` / `@objective` pair by `JuMP.register` + `JuMP.set_NL_objective` as in :233-237.
# NOTE: there is no Julia in this repository's build/test image; tests/test_gpu_operator_export.py exercises the same
# contract (one node evaluation per call, an external first-order solver on the epigraph form) from Python on MI355X.
module GraphicalModelLearningHIPOperator

using GraphicalModelLearning
import GraphicalModelLearning: learn, GMLMethod, RISEA, RISE, NLP, data_info
using JuMP
import LinearAlgebra

export HIPOperator

const libgml = get(ENV, "LIBGML_HIP", "libgml_hip.so")
const GML_RISE, GML_I64, GML_F64 = Cint(0), Cint(2), Cint(3)
const GML_PREC_AUTO = Cint(2)  # operator calls: the int8 matrix cores at the width of Float64 (objective and gradient to 1e-12), and the
                               # FP64-MFMA path for a trial point whose weights the fixed point cannot hold -- like the reference's
                               # Float64 `risea_obj` (:191-197) it never refuses a finite x (include/gml.h)

"`HIPOperator(NLP(solver); device = 0)`: keep the reference's NLP method, evaluate objective and gradient on the GPU."
struct HIPOperator <: GMLMethod
    nlp::NLP
    device::Int
end
HIPOperator(nlp::NLP; device = 0) = HIPOperator(nlp, device)

lasterr() = unsafe_string(ccall((:gml_last_error, libgml), Cstring, ()))

function learn(samples::Array{T,2}, formulation::Union{RISE,RISEA}, method::HIPOperator) where T <: Real
    num_conf, num_spins, num_samples = data_info(samples)                       # :76-81
    lambda = formulation.regularizer * sqrt(log((num_spins^2) / 0.05) / num_samples)   # :213
    s = T <: AbstractFloat ? convert(Array{Float64,2}, samples) : convert(Array{Int64,2}, samples)
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:gml_problem_create, libgml), Cint,
               (Ptr{Cvoid}, Cint, Int64, Int64, Int64, Cint, Cint, Int64, Int64, Cint, Ref{Ptr{Cvoid}}),
               s, eltype(s) == Float64 ? GML_F64 : GML_I64, num_conf, num_spins, num_conf, 1, 2, 0, num_spins, method.device, handle)
    rc == 0 || error("gml_problem_create: $(lasterr())")
    reconstruction = Array{Float64}(undef, num_spins, num_spins)
    try
        for current_spin in 1:num_spins                                          # :216
            node = Ref{Int64}(current_spin - 1)
            fbuf, gbuf = Ref{Cdouble}(0.0), zeros(Cdouble, num_spins)
            xbuf = zeros(Cdouble, num_spins)
            function evaluate!(x)                                                # one node evaluation on the device
                xbuf .= x
                rc = ccall((:gml_objgrad_batch, libgml), Cint,
                           (Ptr{Cvoid}, Cint, Cint, Int64, Ref{Int64}, Ptr{Cdouble}, Int64, Ref{Cdouble}, Ptr{Cdouble}),
                           handle[], GML_RISE, GML_PREC_AUTO, 1, node, xbuf, num_spins, fbuf, gbuf)
                rc == 0 || error("gml_objgrad_batch: $(lasterr())")
            end
            obj(x...) = (evaluate!(collect(x)); fbuf[])                          # risea_obj       (:191-197)
            function grad(g::AbstractVector, x...)                               # grad_risea_obj  (:199-208)
                evaluate!(collect(x))
                g .= gbuf
                return
            end
            model = Model(method.nlp.solver)
            @variable(model, x[1:num_spins])
            @variable(model, z[1:num_spins])
            @operator(model, op_obj, num_spins, obj, grad)                        # JuMP.register(model, :obj, n, obj, grad)  (:233)
            @objective(model, Min, op_obj(x...) + lambda * sum(z[j] for j in 1:num_spins if current_spin != j))  # :235-237
            for j in 1:num_spins                                                  # :239-242
                @constraint(model, z[j] >= x[j])
                @constraint(model, z[j] >= -x[j])
            end
            JuMP.optimize!(model)
            @assert JuMP.termination_status(model) == JuMP.MOI.LOCALLY_SOLVED    # :251
            reconstruction[current_spin, 1:num_spins] = deepcopy(JuMP.value.(x)) # :253
        end
    finally
        ccall((:gml_problem_destroy, libgml), Cvoid, (Ptr{Cvoid},), handle[])
    end
    if formulation.symmetrization
        reconstruction = 0.5 * (reconstruction + transpose(reconstruction))       # :256-258
    end
    return reconstruction
end

end # module

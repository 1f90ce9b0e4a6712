# DecentralOPFHip.jl — thin Julia host for libdopf_hip (the MI355X-native ADMM consensus-OPF inner loop).
#
# Drop-in for src/opf_admm_decentral.jl of rockstaedt/DecentralOPF.jl: it keeps the reference's own
# element types (Node / Generator / Storage / Line, src/structures/network_elements.jl) and PTDF set-up
# (calculate_ptdf, src/helpers/ptdf.jl) by `include`-ing those two files from the reference checkout —
# nothing is re-declared here — and replaces
#     ADMM(gamma, nodes, generators, storages, lines)      src/structures/admm.jl:23-62
#     run!(admm) / calculate_iteration!(admm)              src/optimization/run.jl:1-16
#     get_nodal_price(iteration)                           src/helpers/network_elements.jl:16-25
# by ccalls into the C ABI of include/dopf.h. JuMP and Gurobi are not needed any more.
#
# NOT RUN in the build container (no Julia there): this file is the reference-side binding a maintainer
# adds; the same calls are exercised from Python (decentralopf.jl_amd/admm.py) in the test-suite.
#
# usage (from the reference's root, with libdopf_hip.so on the loader path or DOPF_LIB set):
#     include("path/to/DecentralOPFHip.jl"); using .DecentralOPFHip
#     include("src/cases/three_node.jl")
#     admm = ADMMHip(0.3, nodes, generators, storages, lines)
#     run!(admm)                       # 476 iterations on the shipped case
#     np = get_nodal_price(admm)       # duals the last solve used, like opf_admm_decentral.jl:9
module DecentralOPFHip

export ADMMHip, run!, calculate_iteration!, get_nodal_price, primal, duals, consensus, residuals

const REF = get(ENV, "DECENTRALOPF_SRC", joinpath(pwd(), "src"))
include(joinpath(REF, "structures", "network_elements.jl"))   # Node, Generator, Storage, Line (verbatim)
include(joinpath(REF, "helpers", "ptdf.jl"))                  # calculate_ptdf (host-side set-up)

const LIB = get(ENV, "DOPF_LIB", "libdopf_hip")

# mirrors struct dopf_problem / dopf_params of include/dopf.h (field order and types)
struct CProblem
    N::Cint; L::Cint; T::Cint; G::Cint; S::Cint
    demand::Ptr{Cdouble}; ptdf::Ptr{Cdouble}; f_max::Ptr{Cdouble}
    gen_mc::Ptr{Cdouble}; gen_pmax::Ptr{Cdouble}; gen_node::Ptr{Cint}
    sto_mc::Ptr{Cdouble}; sto_pmax::Ptr{Cdouble}; sto_emax::Ptr{Cdouble}; sto_node::Ptr{Cint}
end

struct CParams
    gamma::Cdouble; w_flow::Cdouble; w_prox::Cdouble; eps::Cdouble; mask_thr::Cdouble
    max_iters::Cint; n_agents_global::Cint; device::Cint; flags::Cint
    stream::Ptr{Cvoid}
end

mutable struct ADMMHip
    ctx::Ptr{Cvoid}
    iteration::Int
    gamma::Float64
    converged::Bool
    nodes::Vector{Node}; generators::Vector{Generator}; storages::Vector{Storage}; lines::Vector{Line}
    ptdf::Matrix{Float64}
    N::Int; L::Int; T::Int; G::Int; S::Int
end

function check(rc::Cint, ctx::Ptr{Cvoid})
    rc == 0 && return
    msg = unsafe_string(ccall((:dopf_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    error("libdopf_hip error $rc: $msg")
end

"""ADMMHip(gamma, nodes, generators, storages, lines; max_iters=0, device=-1)

Same positional signature as the reference's `ADMM(...)`; struct fields are `Int` there and are promoted
to Float64 when packed. Matrices are handed over column-major, exactly as Julia stores them."""
function ADMMHip(gamma::Float64, nodes::Vector{Node}, generators::Vector{Generator},
                 storages::Vector{Storage}, lines::Vector{Line}; max_iters::Int=0, device::Int=-1,
                 w_flow::Float64=10.0, w_prox::Float64=1.0, eps::Float64=1e-3, mask_thr::Float64=1e-2)
    N, L, T, G, S = length(nodes), length(lines), length(nodes[1].demand), length(generators), length(storages)
    node_id = Dict(n => Cint(i - 1) for (i, n) in enumerate(nodes))
    demand = Float64[nodes[n].demand[t] for n in 1:N, t in 1:T]          # N x T, column-major = [n + N*t]
    ptdf = L > 0 ? Matrix{Float64}(calculate_ptdf(nodes, lines)) : zeros(0, N)
    f_max = Float64[l.max_capacity for l in lines]
    gen_mc = Float64[g.marginal_costs for g in generators]; gen_pmax = Float64[g.max_generation for g in generators]
    gen_node = Cint[node_id[g.node] for g in generators]
    sto_mc = Float64[s.marginal_costs for s in storages]; sto_pmax = Float64[s.max_power for s in storages]
    sto_emax = Float64[s.max_level for s in storages]; sto_node = Cint[node_id[s.node] for s in storages]
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve demand ptdf f_max gen_mc gen_pmax gen_node sto_mc sto_pmax sto_emax sto_node begin
        prob = Ref(CProblem(N, L, T, G, S, pointer(demand), pointer(ptdf), pointer(f_max), pointer(gen_mc),
                            pointer(gen_pmax), pointer(gen_node), pointer(sto_mc), pointer(sto_pmax),
                            pointer(sto_emax), pointer(sto_node)))
        par = Ref(CParams(gamma, w_flow, w_prox, eps, mask_thr, max_iters, 0, device, 0, C_NULL))
        rc = ccall((:dopf_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{CParams}), ctx, prob, par)
        check(rc, Ptr{Cvoid}(C_NULL))            # the library copies every input before returning
    end
    admm = ADMMHip(ctx[], 1, gamma, false, nodes, generators, storages, lines, ptdf, N, L, T, G, S)
    finalizer(a -> ccall((:dopf_destroy, LIB), Cvoid, (Ptr{Cvoid},), a.ctx), admm)
    return admm
end

"""One ADMM iteration on the device (all sub-problems, consensus, dual update, stop test)."""
function calculate_iteration!(admm::ADMMHip; n::Int=1)
    done = Ref{Cint}(0); conv = Ref{Cint}(0)
    check(ccall((:dopf_iterate, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ref{Cint}), admm.ctx, n, done, conv), admm.ctx)
    it = Ref{Cint}(0); r = Ref{Cdouble}(0.0)
    check(ccall((:dopf_get_residuals, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cint}),
                admm.ctx, r, r, r, it), admm.ctx)
    admm.iteration = it[]; admm.converged = conv[] != 0
    return Int(done[])
end

"""run!(admm): iterate until every |dual change| < eps (checked on the device), like src/optimization/run.jl:1-5."""
function run!(admm::ADMMHip; chunk::Int=64)
    while !admm.converged
        calculate_iteration!(admm; n=chunk) == 0 && break      # iteration cap reached
    end
    return admm
end

function duals(admm::ADMMHip; used::Bool=false)
    lam = zeros(admm.T); mu = zeros(admm.L, admm.T); rho = zeros(admm.L, admm.T)
    f = used ? :dopf_get_duals_used : :dopf_get_duals
    check(ccall((f, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), admm.ctx, lam, mu, rho), admm.ctx)
    return lam, mu, rho
end

"""P is T x G, D/C/E are T x S (one column per unit = ResultGenerator.generation etc.)."""
function primal(admm::ADMMHip)
    P = zeros(admm.T, admm.G); D = zeros(admm.T, admm.S); C = zeros(admm.T, admm.S); E = zeros(admm.T, admm.S)
    check(ccall((:dopf_get_primal, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                admm.ctx, P, D, C, E), admm.ctx)
    return P, D, C, E
end

function consensus(admm::ADMMHip)
    inj = zeros(admm.N, admm.T); aU = zeros(admm.L, admm.T); aK = zeros(admm.L, admm.T); fl = zeros(admm.L, admm.T)
    cost = Ref{Cdouble}(0.0)
    check(ccall((:dopf_get_consensus, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
                admm.ctx, inj, aU, aK, fl, cost), admm.ctx)
    return (injection=inj, avg_U=aU, avg_K=aK, line_utilization=fl, total_costs=cost[])
end

function residuals(admm::ADMMHip)
    a = Ref{Cdouble}(0.0); b = Ref{Cdouble}(0.0); c = Ref{Cdouble}(0.0); it = Ref{Cint}(0)
    check(ccall((:dopf_get_residuals, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cint}),
                admm.ctx, a, b, c, it), admm.ctx)
    return (lambda=a[], mue=b[], rho=c[], iteration=Int(it[]))
end

"""Nodal price from the duals the last solve used (`after=false`, what src/opf_admm_decentral.jl:9 evaluates)
or from the duals after the last update."""
function get_nodal_price(admm::ADMMHip; after::Bool=false)
    out = zeros(admm.N, admm.T)
    check(ccall((:dopf_get_nodal_price, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}), admm.ctx, after ? 1 : 0, out), admm.ctx)
    return out
end

end # module

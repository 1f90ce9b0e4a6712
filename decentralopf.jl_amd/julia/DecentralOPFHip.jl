# DecentralOPFHip.jl — Julia host for libdopf_hip (the MI355X-native ADMM consensus-OPF inner loop).
#
# Takes the place of `include("imports.jl")` in src/opf_admm_decentral.jl of rockstaedt/DecentralOPF.jl:
#
#     include("/path/to/DecentralOPFHip.jl")        # instead of include("imports.jl")  (no JuMP, no Gurobi, no licence)
#     include("cases/three_node.jl")
#     admm = ADMM(0.3, nodes, generators, storages, lines)
#     run!(admm)
#     np = get_nodal_price(admm.iteration)
#
# Like the reference's own imports.jl this is a plain script evaluated in `Main` — NOT a module: the element
# types Node / Generator / Storage / Line are the reference's own (its src/structures/network_elements.jl is
# included verbatim, so cases/*.jl construct exactly the types ADMM(...) accepts), calculate_ptdf is the
# reference's own (src/helpers/ptdf.jl), and ADMM / run! / calculate_iteration! / get_nodal_price keep the
# reference's names, argument lists and stopping behaviour:
#     ADMM(gamma, nodes, generators, storages, lines)      src/structures/admm.jl:23-62
#     run!(admm) / calculate_iteration!(admm)              src/optimization/run.jl:1-16
#     get_nodal_price(iteration)  (reads the global `admm`, like the reference)   src/helpers/network_elements.jl:16-25
# Everything between is a ccall into the C ABI of include/dopf.h.
#
# STATUS: written by inspection — the build image has no Julia, so this file has NOT been executed. What is
# checked mechanically (tests/test_julia_shim.py): the C struct mirrors against include/dopf.h field by field,
# every ccall's symbol and arity against the header, and the structural points above. The identical call
# sequence runs from Python ctypes and from C (examples/three_node.c) in the GPU test-suite.
#
# Where the reference's files are found: next to this file if it has been copied into the reference's src/
# directory, else ENV["DECENTRALOPF_SRC"], else ./src. The library: ENV["DOPF_LIB"] or "libdopf_hip" on the
# loader path.

using LinearAlgebra                     # calculate_ptdf uses Diagonal and inv (the reference gets it from imports.jl:3)

const DOPF_SRC = isfile(joinpath(@__DIR__, "structures", "network_elements.jl")) ? @__DIR__ :
                 get(ENV, "DECENTRALOPF_SRC", joinpath(pwd(), "src"))
include(joinpath(DOPF_SRC, "structures", "network_elements.jl"))   # Node, Generator, Storage, Line — verbatim, in Main
include(joinpath(DOPF_SRC, "helpers", "ptdf.jl"))                  # calculate_ptdf(nodes, lines)

const DOPF_LIB = get(ENV, "DOPF_LIB", "libdopf_hip")

# mirrors of struct dopf_problem / struct dopf_params (include/dopf.h): same field order, same types
struct CProblem
    N::Cint
    L::Cint
    T::Cint
    G::Cint
    S::Cint
    demand::Ptr{Cdouble}
    ptdf::Ptr{Cdouble}
    f_max::Ptr{Cdouble}
    gen_mc::Ptr{Cdouble}
    gen_pmax::Ptr{Cdouble}
    gen_node::Ptr{Cint}
    sto_mc::Ptr{Cdouble}
    sto_pmax::Ptr{Cdouble}
    sto_emax::Ptr{Cdouble}
    sto_node::Ptr{Cint}
end

struct CParams
    gamma::Cdouble
    w_flow::Cdouble
    w_prox::Cdouble
    eps::Cdouble
    mask_thr::Cdouble
    max_iters::Cint
    n_agents_global::Cint
    device::Cint
    flags::Cint
    stream::Ptr{Cvoid}
end

# Convergence of the reference (src/structures/convergence.jl): the flags and the residual history
mutable struct Convergence
    lambda::Bool
    lambda_res::Vector{Vector{Float64}}
    mue::Bool
    mue_res::Vector{Matrix{Float64}}
    rho::Bool
    rho_res::Vector{Matrix{Float64}}
    all::Bool
    Convergence() = new(false, [], false, [], false, [], false)
end

# Same field names as the reference's ADMM for everything a caller reads; `results` holds NamedTuples with the
# fields of Result (generation per unit, discharge/charge/level, injection, avg_U, avg_K, total_costs,
# line_utilization). Histories grow only with record = true (the reference always records: O(iterations x agents)
# of host memory and one device read-back per iteration); without it the last TWO dual sets are kept, which is
# what get_nodal_price(admm.iteration) needs.
mutable struct ADMM
    iteration::Int
    gamma::Float64
    lambdas::Vector{Vector{Float64}}
    mues::Vector{Matrix{Float64}}
    rhos::Vector{Matrix{Float64}}
    T::Vector{Int}
    N::Vector{Int}
    L::Vector{Int}
    nodes::Vector{Node}
    generators::Vector{Generator}
    storages::Vector{Storage}
    lines::Vector{Line}
    results::Vector{Any}
    convergence::Convergence
    ptdf::Matrix{Float64}
    total_demand::Vector{Float64}
    node_to_id::Dict{Node, Int}
    f_max::Vector{Float64}
    record::Bool
    n_gpus::Int
    ctx::Ptr{Cvoid}                 # dopf_ctx* (n_gpus == 1) ...
    multi::Ptr{Cvoid}               # ... or dopf_multi* (n_gpus > 1: the library shards the agents and owns RCCL)
end

function dopf_check(rc::Cint, ctx::Ptr{Cvoid})
    rc == 0 && return
    msg = unsafe_string(ccall((:dopf_last_error, DOPF_LIB), Cstring, (Ptr{Cvoid},), ctx))
    error("libdopf_hip error $rc: $msg")
end

function dopf_check_multi(rc::Cint, m::Ptr{Cvoid})
    rc == 0 && return
    msg = unsafe_string(ccall((:dopf_multi_last_error, DOPF_LIB), Cstring, (Ptr{Cvoid},), m))
    error("libdopf_hip error $rc: $msg")
end

const DOPF_F_COMM_P2P = 1024      # include/dopf.h

"""
    ADMM(gamma, nodes, generators, storages, lines; max_iters=0, n_gpus=1, record=false, ...)

The reference's constructor (src/structures/admm.jl:23-27) with the same five positional arguments. The `Int`
struct fields are promoted to Float64 when packed; matrices go over column-major, as Julia stores them.
Keywords are additions: `max_iters` (the reference loops forever on a divergent case), `n_gpus` (> 1: agents are
sharded over that many devices inside the library, one RCCL all-reduce per iteration — or, with
`flags = DOPF_F_COMM_P2P`, the library's peer exchange: direct stores into the other devices' memory, no collective
library, a few microseconds instead of tens for the small vector of a copper plate), `record`, and the
reference's literals `w_flow = 10`, `w_prox = 1`, `eps = 1e-3`, `mask_thr = 1e-2`.
"""
function ADMM(gamma::Float64, nodes::Vector{Node}, generators::Vector{Generator}, storages::Vector{Storage},
              lines::Vector{Line}; max_iters::Int=0, device::Int=-1, n_gpus::Int=1, record::Bool=false,
              w_flow::Float64=10.0, w_prox::Float64=1.0, eps::Float64=1e-3, mask_thr::Float64=1e-2, flags::Int=0)
    N, L, T = length(nodes), length(lines), length(nodes[1].demand)
    G, S = length(generators), length(storages)
    node_to_id = Dict{Node, Int}(n => i for (i, n) in enumerate(nodes))
    demand = Float64[nodes[n].demand[t] for n in 1:N, t in 1:T]          # N x T, column-major = [n + N*t]
    ptdf = L > 0 ? Matrix{Float64}(calculate_ptdf(nodes, lines)) : zeros(Float64, 0, N)
    f_max = Float64[l.max_capacity for l in lines]
    gen_mc = Float64[g.marginal_costs for g in generators]
    gen_pmax = Float64[g.max_generation for g in generators]
    gen_node = Cint[node_to_id[g.node] - 1 for g in generators]          # 0-based
    sto_mc = Float64[s.marginal_costs for s in storages]
    sto_pmax = Float64[s.max_power for s in storages]
    sto_emax = Float64[s.max_level for s in storages]
    sto_node = Cint[node_to_id[s.node] - 1 for s in storages]
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    multi = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve demand ptdf f_max gen_mc gen_pmax gen_node sto_mc sto_pmax sto_emax sto_node begin
        prob = Ref(CProblem(N, L, T, G, S, pointer(demand), pointer(ptdf), pointer(f_max), pointer(gen_mc),
                            pointer(gen_pmax), pointer(gen_node), pointer(sto_mc), pointer(sto_pmax),
                            pointer(sto_emax), pointer(sto_node)))
        par = Ref(CParams(gamma, w_flow, w_prox, eps, mask_thr, max_iters, 0, device, flags, C_NULL))
        if n_gpus == 1
            rc = ccall((:dopf_create, DOPF_LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{CParams}), ctx, prob, par)
            dopf_check(rc, Ptr{Cvoid}(C_NULL))       # the library copies every input before returning
        else
            rc = ccall((:dopf_multi_create, DOPF_LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{CProblem}, Ref{CParams}, Cint, Ptr{Cint}),
                       multi, prob, par, n_gpus, C_NULL)
            dopf_check_multi(rc, Ptr{Cvoid}(C_NULL))
            ctx[] = ccall((:dopf_multi_ctx, DOPF_LIB), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), multi[], 0)   # replicated state: shard 0
        end
    end
    total_demand = zeros(T)
    for node in nodes
        total_demand += node.demand
    end
    admm = ADMM(1, gamma, [zeros(T)], [zeros(L, T)], [zeros(L, T)], collect(1:T), collect(1:N), collect(1:L),
                nodes, generators, storages, lines, [], Convergence(), ptdf, total_demand, node_to_id, f_max,
                record, n_gpus, ctx[], multi[])
    finalizer(admm) do a
        if a.multi != C_NULL
            ccall((:dopf_multi_destroy, DOPF_LIB), Cvoid, (Ptr{Cvoid},), a.multi)
        elseif a.ctx != C_NULL
            ccall((:dopf_destroy, DOPF_LIB), Cvoid, (Ptr{Cvoid},), a.ctx)
        end
    end
    return admm
end

function dopf_duals(admm::ADMM; used::Bool=false)
    lam = zeros(length(admm.T))
    mu = zeros(length(admm.L), length(admm.T))
    rho = zeros(length(admm.L), length(admm.T))
    if used
        dopf_check(ccall((:dopf_get_duals_used, DOPF_LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                         admm.ctx, lam, mu, rho), admm.ctx)
    else
        dopf_check(ccall((:dopf_get_duals, DOPF_LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                         admm.ctx, lam, mu, rho), admm.ctx)
    end
    return lam, mu, rho
end

"""P is T x G, D / C / E are T x S: column u = ResultGenerator.generation resp. ResultStorage.discharge/charge/level of unit u."""
function dopf_primal(admm::ADMM)
    T, G, S = length(admm.T), length(admm.generators), length(admm.storages)
    P = zeros(T, G); D = zeros(T, S); C = zeros(T, S); E = zeros(T, S)
    if admm.multi != C_NULL
        dopf_check_multi(ccall((:dopf_multi_get_primal, DOPF_LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                               admm.multi, P, D, C, E), admm.multi)
    else
        dopf_check(ccall((:dopf_get_primal, DOPF_LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                         admm.ctx, P, D, C, E), admm.ctx)
    end
    return P, D, C, E
end

"""The fields of the reference's Result (src/structures/results.jl:37-48) for the last solved iteration."""
function dopf_result(admm::ADMM)
    N, L, T = length(admm.N), length(admm.L), length(admm.T)
    inj = zeros(N, T); aU = zeros(L, T); aK = zeros(L, T); fl = zeros(L, T)
    cost = Ref{Cdouble}(0.0)
    dopf_check(ccall((:dopf_get_consensus, DOPF_LIB), Cint,
                     (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}),
                     admm.ctx, inj, aU, aK, fl, cost), admm.ctx)
    P, D, C, E = dopf_primal(admm)
    # ResultNode.{generation, discharge, charge} (src/structures/results.jl:19-35): N x T, row n = admm.nodes[n]
    ng = zeros(N, T); nd = zeros(N, T); nc = zeros(N, T)
    dopf_check(ccall((:dopf_get_node_results, DOPF_LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                     admm.ctx, ng, nd, nc), admm.ctx)
    return (generation=P, discharge=D, charge=C, level=E, injection=inj, avg_U=aU, avg_K=aK,
            line_utilization=fl, total_costs=cost[], node_generation=ng, node_discharge=nd, node_charge=nc)
end

"""
Result.penalty_term of the last solved iteration (src/structures/results.jl:66-70: the units' PenaltyTerms summed,
src/helpers/penalty_terms.jl:1-6) as `(energy_balance, upper_flow, lower_flow)`, T values each — one pass on the device.
Needs lines and `flags` containing DOPF_F_KEEP_DELTAS at construction (the device then keeps every unit's injection change).
"""
function dopf_penalty_sums(admm::ADMM)
    T = length(admm.T)
    pen = zeros(3 * T)
    dopf_check(ccall((:dopf_get_penalty_sums, DOPF_LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), admm.ctx, pen), admm.ctx)
    return (energy_balance=pen[1:T], upper_flow=pen[T+1:2T], lower_flow=pen[2T+1:3T])
end

# after a batch of iterations: iteration counter, stop flags, the dual sets get_nodal_price needs
function dopf_refresh!(admm::ADMM, converged::Bool)
    a = Ref{Cdouble}(0.0); b = Ref{Cdouble}(0.0); c = Ref{Cdouble}(0.0); it = Ref{Cint}(0)
    dopf_check(ccall((:dopf_get_residuals, DOPF_LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cdouble}, Ref{Cint}),
                     admm.ctx, a, b, c, it), admm.ctx)
    admm.iteration = Int(it[])
    admm.convergence.all = converged
    if converged
        admm.convergence.lambda = true; admm.convergence.mue = true; admm.convergence.rho = true
    end
    return a[], b[], c[]
end

"""
    calculate_iteration!(admm; n = 1)

`n` ADMM iterations on the device (all sub-problems, consensus sums, dual update, stop test; src/optimization/run.jl:7-16
without the printing). With `admm.record` the duals and a Result-like NamedTuple are pushed per iteration like the
reference does (then n is forced to 1); otherwise only the dual sets of the last step are kept.
"""
function calculate_iteration!(admm::ADMM; n::Int=1)
    n = admm.record ? 1 : n
    done = Ref{Cint}(0); conv = Ref{Cint}(0)
    if admm.multi != C_NULL
        dopf_check_multi(ccall((:dopf_multi_iterate, DOPF_LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ref{Cint}), admm.multi, n, done, conv), admm.multi)
    else
        dopf_check(ccall((:dopf_iterate, DOPF_LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ref{Cint}), admm.ctx, n, done, conv), admm.ctx)
    end
    done[] == 0 && return 0
    dopf_refresh!(admm, conv[] != 0)
    used = dopf_duals(admm; used=true)
    now = dopf_duals(admm)
    if admm.record
        push!(admm.results, dopf_result(admm))
        push!(admm.lambdas, now[1]); push!(admm.mues, now[2]); push!(admm.rhos, now[3])     # update_duals.jl:14,26,38
        push!(admm.convergence.lambda_res, abs.(now[1] - used[1]))
        push!(admm.convergence.mue_res, abs.(now[2] - used[2]))
        push!(admm.convergence.rho_res, abs.(now[3] - used[3]))
    else
        admm.lambdas = [used[1], now[1]]; admm.mues = [used[2], now[2]]; admm.rhos = [used[3], now[3]]
    end
    return Int(done[])
end

"""run!(admm): iterate until every |dual change| < eps — tested on the device exactly like check_convergence!
(no test at iteration 1, the counter is not bumped on the converging step) — or the iteration cap is reached."""
function run!(admm::ADMM; chunk::Int=64)
    while !admm.convergence.all
        calculate_iteration!(admm; n=chunk) == 0 && break      # iteration cap reached
    end
    return admm
end

"""
    get_nodal_price(iteration)

The reference's signature (src/helpers/network_elements.jl:16-25): reads the GLOBAL `admm` and evaluates
`lambdas[iteration] .+ sum((mues[iteration] + rhos[iteration])[l, t] * ptdf[l, :])`. `iteration == admm.iteration`
(what src/opf_admm_decentral.jl:9 asks for) are the duals the last solve used; with `record = true` any past
iteration works, without it only the last two dual sets exist.
"""
function get_nodal_price(iteration::Int)
    a = admm                                   # the global of the driver script, as in the reference
    idx = a.record ? iteration : (iteration == a.iteration ? 1 : (iteration == a.iteration + 1 ? 2 : 0))
    (idx < 1 || idx > length(a.lambdas)) && error("duals of iteration $iteration are not kept (create ADMM(...; record=true))")
    nodal_price = zeros(length(a.N), length(a.T))
    for t in a.T
        nodal_price[:, t] .= a.lambdas[idx][t]
        for l in a.L
            nodal_price[:, t] .+= (a.mues[idx][l, t] + a.rhos[idx][l, t]) .* a.ptdf[l, :]
        end
    end
    return nodal_price
end

"""
    export_results(admm, filename; parent_dir = "results/")

The reference's export (src/helpers/output.jl:1-85): `<filename>_duals.csv` (iteration,dual,timestep,line,value; duals
lambda, rho, mue in that order; `line` empty for lambda; row i = the dual USED in iteration i), `<filename>_generators.csv`
(iteration,generator,timestep,generation) and `<filename>_storages.csv` (iteration,storage,timestep,charge,discharge), one
row per iteration 1..admm.iteration, written as plain text (no CSV.jl / DataFrames needed). Needs the iteration history:
`ADMM(...; record = true)` — which is also what any reference-style script that reads `admm.results[end]` needs.
"""
function export_results(admm::ADMM, filename::String; parent_dir::String="results/")
    admm.record || error("export_results needs the iteration history: create ADMM(...; record = true)")
    mkpath(parent_dir)
    n_it = min(admm.iteration, length(admm.results))
    open(joinpath(parent_dir, filename * "_duals.csv"), "w") do f
        println(f, "iteration,dual,timestep,line,value")
        for (name, hist) in (("lambda", admm.lambdas), ("rho", admm.rhos), ("mue", admm.mues))
            for i in 1:n_it, t in admm.T
                if name == "lambda"
                    println(f, i, ",lambda,", t, ",,", hist[i][t])
                else
                    for l in admm.L
                        println(f, i, ",", name, ",", t, ",", l, ",", hist[i][l, t])
                    end
                end
            end
        end
    end
    open(joinpath(parent_dir, filename * "_generators.csv"), "w") do f
        println(f, "iteration,generator,timestep,generation")
        for (g, gen) in enumerate(admm.generators), i in 1:n_it, t in admm.T
            println(f, i, ",", gen.name, ",", t, ",", admm.results[i].generation[t, g])
        end
    end
    open(joinpath(parent_dir, filename * "_storages.csv"), "w") do f
        println(f, "iteration,storage,timestep,charge,discharge")
        for (s, sto) in enumerate(admm.storages), i in 1:n_it, t in admm.T
            println(f, i, ",", sto.name, ",", t, ",", admm.results[i].charge[t, s], ",", admm.results[i].discharge[t, s])
        end
    end
    return nothing
end

struct CCentralResult          # == struct dopf_central_result (include/dopf.h)
    objective::Cdouble
    dual_objective::Cdouble
    primal_infeasibility::Cdouble
    gap::Cdouble
    iterations::Cint
    converged::Cint
end

"""
    central_reference(nodes, generators, storages, lines; tol = 1e-9, max_iters = 200000)

What src/opf_central_reference.jl computes — objective, P, D, C, line utilisation, system price `dual.(EB)`, nodal price —
from the whole problem as ONE LP, solved on the GPU by libdopf_hip's first-order method (no modelling layer, no licensed solver).
Returns a NamedTuple; matrices are units x timesteps like `value.(P).data`.
"""
function central_reference(nodes::Vector{Node}, generators::Vector{Generator}, storages::Vector{Storage}, lines::Vector{Line};
                           tol::Float64=1e-9, max_iters::Int=200000, device::Int=-1)
    N, L, T = length(nodes), length(lines), length(nodes[1].demand)
    G, S = length(generators), length(storages)
    node_to_id = Dict{Node, Int}(n => i for (i, n) in enumerate(nodes))
    demand = Float64[nodes[n].demand[t] for n in 1:N, t in 1:T]
    ptdf = L > 0 ? Matrix{Float64}(calculate_ptdf(nodes, lines)) : zeros(Float64, 0, N)
    f_max = Float64[l.max_capacity for l in lines]
    gen_mc = Float64[g.marginal_costs for g in generators]
    gen_pmax = Float64[g.max_generation for g in generators]
    gen_node = Cint[node_to_id[g.node] - 1 for g in generators]
    sto_mc = Float64[s.marginal_costs for s in storages]
    sto_pmax = Float64[s.max_power for s in storages]
    sto_emax = Float64[s.max_level for s in storages]
    sto_node = Cint[node_to_id[s.node] - 1 for s in storages]
    P = zeros(T, G); D = zeros(T, S); C = zeros(T, S); E = zeros(T, S)
    lambda = zeros(T); nodal = zeros(N, T); util = zeros(L, T)
    flow_upper = zeros(L, T); flow_lower = zeros(L, T)          # dual.(FlowUpper), dual.(FlowLower), opf_central_reference.jl:71
    res = Ref(CCentralResult(0.0, 0.0, 0.0, 0.0, 0, 0))
    GC.@preserve demand ptdf f_max gen_mc gen_pmax gen_node sto_mc sto_pmax sto_emax sto_node begin
        prob = Ref(CProblem(N, L, T, G, S, pointer(demand), pointer(ptdf), pointer(f_max), pointer(gen_mc),
                            pointer(gen_pmax), pointer(gen_node), pointer(sto_mc), pointer(sto_pmax),
                            pointer(sto_emax), pointer(sto_node)))
        par = Ref(CParams(0.3, 10.0, 1.0, 1e-3, 1e-2, 0, 0, device, 0, C_NULL))
        rc = ccall((:dopf_central_solve, DOPF_LIB), Cint,
                   (Ref{CProblem}, Ref{CParams}, Cdouble, Cint, Ref{CCentralResult}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
                    Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
                   prob, par, tol, max_iters, res, P, D, C, E, lambda, nodal, util, flow_upper, flow_lower)
        dopf_check(rc, Ptr{Cvoid}(C_NULL))
    end
    res[].converged == 0 && error("central LP: gap $(res[].gap) after $(res[].iterations) iterations")
    return (objective=res[].objective, generation=permutedims(P), discharge=permutedims(D), charge=permutedims(C),
            level=permutedims(E), line_utilization=util, system_price=lambda, nodal_price=nodal,
            flow_upper_dual=flow_upper, flow_lower_dual=flow_lower)
end

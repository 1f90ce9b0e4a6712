/*
 * dopf.h — C ABI of libdopf_hip: the MI355X-native ADMM consensus-OPF inner loop.
 *
 * This is the drop-in boundary for the hot path of rockstaedt/DecentralOPF.jl
 * (reference paths are relative to the reference checkout):
 *
 *   dopf_create            replaces  ADMM(gamma, nodes, generators, storages, lines)
 *                                    src/structures/admm.jl:23-62 (state, zero duals, maps)
 *   dopf_iterate           replaces  run!(admm) / calculate_iteration!(admm)
 *                                    src/optimization/run.jl:1-16
 *                                    = optimize_all_subproblems!  src/optimization/subproblems.jl:1-17
 *                                    + update_duals!              src/optimization/update_duals.jl:1-39
 *                                    + check_convergence!         src/optimization/convergence.jl:1-31
 *   dopf_local_update      replaces  optimize_subproblem.(generators|storages)
 *                                    src/optimization/subproblems.jl:19-207 (+ penalty_terms.jl:1-53)
 *                                    and the agent sums of Result(...) src/structures/results.jl:50-106
 *   dopf_apply_consensus   replaces  the rest of Result(...) (avg_U/avg_K, line_utilization,
 *                                    results.jl:108-116), update_duals! and check_convergence!
 *   dopf_get_duals         replaces  admm.lambdas[end], admm.mues[end], admm.rhos[end]
 *   dopf_get_duals_used    replaces  admm.lambdas[admm.iteration] ... (the duals the last solve used;
 *                                    these feed get_nodal_price(admm.iteration),
 *                                    src/opf_admm_decentral.jl:9)
 *   dopf_get_primal        replaces  ResultGenerator.generation / ResultStorage.{discharge,charge,level}
 *                                    src/structures/results.jl:1-17
 *   dopf_get_consensus     replaces  Result.{injection,avg_U,avg_K,line_utilization,total_costs}
 *                                    src/structures/results.jl:37-48
 *   dopf_get_residuals     replaces  Convergence.{lambda_res,mue_res,rho_res} (inf-norms)
 *                                    src/structures/convergence.jl:1-20
 *   dopf_get_nodal_price   replaces  get_nodal_price(iteration) src/helpers/network_elements.jl:16-25
 *   dopf_set_state         (no reference counterpart: resume / trajectory tests; the reference
 *                                    keeps whole histories in admm.results / admm.lambdas instead)
 *
 * Conventions
 *   - all matrices are Julia column-major: demand[n + N*t], ptdf[l + L*n], mu[l + L*t];
 *   - per-agent time series are agent-major: P[t + T*g], D/C/E[t + T*s];
 *   - node ids are 0-based int32; all reals are double (the reference computes in Float64);
 *   - every pointer argument is caller-owned HOST memory, copied in or out before return
 *     (no aliasing of the caller's GC memory after the call returns), except the device
 *     pointer handed to dopf_bind_consensus (see there);
 *   - return 0 = ok, negative = error (DOPF_E_*); message via dopf_last_error;
 *     no exception or longjmp crosses this boundary;
 *   - a context is driven by one host thread at a time.
 *
 * The same signatures, prefixed oracle_ instead of dopf_, are implemented by the CPU oracle
 * (oracle/dopf_oracle.c). The oracle is test infrastructure, never a fallback of this library.
 */
#ifndef DOPF_H
#define DOPF_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DOPF_OK            0
#define DOPF_E_INVALID    -1   /* bad argument / inconsistent sizes */
#define DOPF_E_NOMEM      -2
#define DOPF_E_DEVICE     -3   /* HIP runtime error (message has the hipError string) */
#define DOPF_E_UNSUPPORTED -4
#define DOPF_E_SOLVER     -5   /* an agent sub-problem did not reach its tolerance */

typedef struct dopf_ctx dopf_ctx;

/* One OPF instance (or one rank's shard of it: G and S then count the LOCAL agents,
 * everything else is the global network). Mirrors the fields ADMM(...) derives from the
 * Node/Generator/Storage/Line vectors, src/structures/admm.jl:28-60. */
typedef struct dopf_problem {
    int32_t N, L, T, G, S;
    const double  *demand;    /* N x T, [n + N*t]   (Node.demand, promoted Int -> Float64)      */
    const double  *ptdf;      /* L x N, [l + L*n]   (calculate_ptdf, src/helpers/ptdf.jl:1-41)   */
    const double  *f_max;     /* L                  (Line.max_capacity)                          */
    const double  *gen_mc;    /* G                  (Generator.marginal_costs)                   */
    const double  *gen_pmax;  /* G                  (Generator.max_generation)                   */
    const int32_t *gen_node;  /* G, 0-based                                                      */
    const double  *sto_mc;    /* S                  (Storage.marginal_costs)                     */
    const double  *sto_pmax;  /* S                  (Storage.max_power)                          */
    const double  *sto_emax;  /* S                  (Storage.max_level)                          */
    const int32_t *sto_node;  /* S, 0-based                                                      */
} dopf_problem;

/* The literals of the reference, collected (SURVEY.md section 5 "Config / flags"). */
typedef struct dopf_params {
    double  gamma;      /* 0.3   src/opf_admm_decentral.jl:5  (BASELINE.json calls it rho)      */
    double  w_flow;     /* 10    src/optimization/subproblems.jl:77-78,176-177                  */
    double  w_prox;     /* 1     (the literal 1/2 in subproblems.jl:81,180-181 is w_prox/2)     */
    double  eps;        /* 1e-3  src/optimization/convergence.jl:2                              */
    double  mask_thr;   /* 1e-2  src/optimization/update_duals.jl:24,36                         */
    int32_t max_iters;  /* cap on admm.iteration; 0 = none (the reference has none, run.jl:2-4)  */
    int32_t n_agents_global; /* divisor of avg_U/avg_K (results.jl:108-112); 0 = G+S (one shard) */
    int32_t device;     /* HIP device ordinal; -1 = current device                              */
    int32_t flags;      /* DOPF_F_* below                                                        */
    void   *stream;     /* hipStream_t to enqueue on; NULL = a stream the context creates        */
} dopf_params;

#define DOPF_F_NO_GRAPH   1   /* launch kernels eagerly instead of through a captured hipGraph   */
#define DOPF_F_OVERLAP_AGENTS 2 /* storage kernel forked onto a side stream so it overlaps the generator
                                   kernel (default: one stream, kernels back to back — the per-kernel
                                   durations then mean the same in every tool)                        */

#define DOPF_F_NO_ROW_SKIP   8  /* generators: always sweep every row of P (no skipping of saturated rows)    */
#define DOPF_F_NO_WARM_START 4  /* storages: always the cold price-threshold scan (no warm-start kernel) */
#define DOPF_F_NO_FUSE      16  /* copper plate: generator and storage x-updates as separate launches instead
                                   of the single k_agents launch                                           */
#define DOPF_F_COMM_HOST    64  /* dopf_multi_*: sum the consensus buffers through host memory instead of RCCL — a
                                   debugging transport that lets several shards share one device (tests)      */
#define DOPF_F_KEEP_DELTAS  256  /* networks: keep every agent's injection change of the last iteration on the device, so that
                                   dopf_get_agent_slacks / dopf_get_agent_penalty work without the caller passing it in (by
                                   default it is written only for timesteps whose slack sums need the agents one by one) */
#define DOPF_F_COMM_GRAPH   512  /* contexts joined to a communicator of > 1 ranks: capture the RCCL all-reduce into the iteration
                                   hipGraphs instead of launching the chain eagerly (default: eager — the host enqueues an
                                   iteration faster than the GPU retires it, and plain launches are RCCL's best-trodden path) */
#define DOPF_F_COMM_P2P    1024  /* dopf_multi_*: the consensus sum by the peer exchange (dopf_xchg_* below) instead of RCCL.
                                   (Rehearsals that put several shards on ONE device: every shard's stream then needs a hardware
                                   queue of its own — GPU_MAX_HW_QUEUES — or a waiting exchange kernel sits in front of the peer
                                   it waits for and the wait times out. One shard per device cannot run into that.)        */
#define DOPF_F_NO_TAIL_FUSE 4096  /* one node, no lines, single-GPU chain: keep the consensus sums and the dual step as launches of
                                   their own (k_reduce, k_dual_price_small) instead of finishing the iteration inside the
                                   x-update launch (integer accumulators + the last block's tail) */
#define DOPF_F_TIME_CALLS  8192  /* measurement: dopf_iterate brackets its launches with a HIP event pair on the context's stream;
                                   dopf_last_call_ms returns the device-side span of the last call's iterations (no host launch
                                   latency in front, no status read-back behind) */
#define DOPF_F_STO_GENERAL 16384 /* storages: the general active-set body (kernels_agents.hip: sto_warm_body) also where the lean
                                  * body (sto_lean.h: copper plates; networks with many storage blocks) applies — the two are
                                  * compared by the tests */
#define DOPF_F_NO_QUIET   32768 /* networks, single-GPU chain: always launch k_slack (never the "quiet" chain, in which the dual/price
                                  * kernel forms the node sums while no line is flagged); bitwise comparisons of the two chains */
#define DOPF_F_XCHG_OWNER 65536 /* peer exchange: always the reduce-scatter + all-gather form (every chunk has an owner rank that adds the
                                  * ranks' copies), also for vectors of one chunk per rank — by default the library picks it when the
                                  * vector has more chunks than ranks */
#define DOPF_F_XCHG_ALLGATHER 131072 /* peer exchange: always the all-gather form (every rank stores its vector into every peer's area) */
#define DOPF_F_NO_TAIL_XCHG 262144 /* copper plates on a peer exchange: the exchange runs inside the one-block dual kernel of the
                                  * three-launch chain instead of inside the tail block of the one-launch iteration */
#define DOPF_F_NET_SMALL_ITEMS 524288 /* networks, one launch for all agents: cut the generators into the ~1024 items the separate
                                  * launches use instead of ~512 larger ones (same partial-sum rows on both chains: bitwise comparisons) */
#define DOPF_F_PERSIST 1048576 /* copper plates whose x-update is one launch with every block resident (config1, config2): run the
                                  * iterations of a dopf_iterate call in launches of up to 16 ITERATIONS each — the grid stays, the tail block
                                  * publishes the new prices to the other blocks (csrc/agents_persist.h; round 4 experiment) */
/* Everything else that steers kernel selection is decided from the problem's shape ("which chain runs"). The
 * library reads two environment variables, neither of which changes results: DOPF_GUARD (debug allocator) and DOPF_XCHG_TIMEOUT_MS
 * (how long an exchange kernel waits for a lost peer). Tuning knobs of the experiments (item counts, block counts, launch splits)
 * exist only in builds with -DDOPF_EXPERIMENTS. */
#define DOPF_F_DEBUG_LEAVE  2048  /* tests: the active-set storage body declares every third storage uncertified, so that the
                                   hand-over to the scan body is exercised in every kernel variant                        */
#define DOPF_F_DEBUG_ROOT_CAP 128 /* tests: the scan kernel's root search gives up after 2 iterations instead of
                                   80, so that the DOPF_E_SOLVER path can be exercised                        */

/* Fill q with the reference's defaults (values above). */
void dopf_default_params(dopf_params *q);

int  dopf_create(dopf_ctx **out, const dopf_problem *p, const dopf_params *q);
void dopf_destroy(dopf_ctx *ctx);
/* Message of the last error on ctx; with ctx == NULL the last dopf_create error (thread-local). */
const char *dopf_last_error(const dopf_ctx *ctx);

/* Run up to n_iters ADMM iterations back to back on the device, one host sync at the end.
 * Stops exactly like check_convergence!: no test at iteration 1; on the converging step the
 * iteration counter is NOT bumped and later iterations in the same call are no-ops.
 * iters_done = iterations actually computed in this call; converged = Convergence.all.
 * Returns DOPF_E_SOLVER (outputs are still set, the state is still advanced) when a storage
 * sub-problem's root search hit its iteration cap since the last report: that row is not the
 * minimiser of its QP (the reference would surface a Gurobi failure from value.() at this point,
 * src/optimization/subproblems.jl:189-206). dopf_sync reports the same.
 * On a context joined to a communicator (dopf_comm_init / dopf_multi_create) every iteration
 * includes the all-reduce of the consensus buffer; all ranks must call with the same n_iters. */
int dopf_iterate(dopf_ctx *ctx, int32_t n_iters, int32_t *iters_done, int32_t *converged);

/* Sharded form of one iteration (one process per GPU):
 *   dopf_local_update     x-update of the local agents + local sums -> consensus buffer
 *   <all-reduce(sum) of the consensus buffer across ranks, done by the caller>
 *   dopf_apply_consensus  avg_U/avg_K, flows, dual update, residuals, convergence test
 * Both enqueue on the context's stream and return without synchronising. */
int dopf_local_update(dopf_ctx *ctx);
int dopf_apply_consensus(dopf_ctx *ctx);
/* Number of doubles in the consensus buffer: N*T injections | L*T sum U | L*T sum K | 1 cost. */
int64_t dopf_consensus_size(const dopf_ctx *ctx);
/* Device address of the context's own consensus buffer (for RCCL / a zero-copy tensor view). */
void *dopf_consensus_ptr(dopf_ctx *ctx);
/* Use caller-owned DEVICE memory (>= dopf_consensus_size doubles) as the consensus buffer,
 * e.g. the data_ptr of a torch tensor handed to torch.distributed.all_reduce. */
int dopf_bind_consensus(dopf_ctx *ctx, void *device_ptr);
/* Host sync of the context's stream + status read-back (iteration, converged). */
int dopf_sync(dopf_ctx *ctx, int32_t *iteration, int32_t *converged);

int dopf_get_duals(dopf_ctx *ctx, double *lambda /*T*/, double *mu /*L*T*/, double *rho /*L*T*/);
int dopf_get_duals_used(dopf_ctx *ctx, double *lambda, double *mu, double *rho);
int dopf_get_primal(dopf_ctx *ctx, double *P /*T*G*/, double *D, double *C, double *E /*T*S*/);
int dopf_get_consensus(dopf_ctx *ctx, double *injection /*N*T*/, double *avg_U, double *avg_K,
                       double *line_util /*L*T each*/, double *total_cost /*1*/);
int dopf_get_residuals(dopf_ctx *ctx, double *lam_res, double *mu_res, double *rho_res,
                       int32_t *iteration);
/* Convergence.{lambda_res, mue_res, rho_res}[end] as vectors (src/structures/convergence.jl:5-12):
 * |dual after the last update - dual the last solve used| per entry; any pointer may be NULL. */
int dopf_get_residual_vectors(dopf_ctx *ctx, double *lam_res /*T*/, double *mu_res /*L*T*/, double *rho_res /*L*T*/);
/* ResultGenerator/ResultStorage.{U, K} of the last solve (src/structures/results.jl:1-17), L x T each,
 * [l + L*t]; agent = caller's index, generators 0..G-1 then storages G..G+S-1 (of this context's shard).
 * The device eliminates the slacks in closed form; they are recomputed here from the agent's injection
 * change and the consensus state that solve read. Needs DOPF_F_KEEP_DELTAS. */
int dopf_get_agent_slacks(dopf_ctx *ctx, int32_t agent, double *U, double *K);
/* PenaltyTerm of the agent (src/structures/penalty_terms.jl:1-5, src/optimization/penalty_terms.jl:3-37):
 * penalty[0..T) energy_balance, [T..2T) upper_flow, [2T..3T) lower_flow — the diagnostics print_results shows
 * with print_penalty=true. delta = the agent's injection change of the last iteration (T values), or NULL to
 * use the device's copy, which exists only with lines and DOPF_F_KEEP_DELTAS (DOPF_E_UNSUPPORTED otherwise). */
int dopf_get_agent_penalty(dopf_ctx *ctx, int32_t agent, const double *delta, double *penalty /*3*T*/);
/* Result.penalty_term of the last result (src/structures/results.jl:66-70: `sum_up` of src/helpers/penalty_terms.jl:1-6 over
 * every unit's PenaltyTerm): penalty[0..T) energy_balance, [T..2T) upper_flow, [2T..3T) lower_flow, each the SUM over all
 * agents of this context of what dopf_get_agent_penalty returns for one — one pass on the device instead of one call per
 * agent. Needs lines and DOPF_F_KEEP_DELTAS (the injection changes of the last x-update must be on the device);
 * DOPF_E_UNSUPPORTED otherwise. */
int dopf_get_penalty_sums(dopf_ctx *ctx, double *penalty /*3*T*/);
/* which = 0: duals used by the last solve (what the reference's driver script evaluates),
 * which = 1: duals after the last update. out is N x T, [n + N*t]. */
int dopf_get_nodal_price(dopf_ctx *ctx, int32_t which, double *out);

/* ResultNode.{generation, discharge, charge} of the last result (src/structures/results.jl:19-35, filled by `update`,
 * src/helpers/network_elements.jl:1-14, in the agent loop of Result(...), results.jl:72-106): per node and timestep the
 * sums of its generators' output and of its storages' discharge and charge, layout [n + N*t] like the injection (which
 * is generation + discharge - charge - demand). Any pointer may be NULL. Result.{generation, discharge, charge}
 * (results.jl:37-47) are these summed over the nodes. Computed on request from the primal arrays (not on the hot path). */
int dopf_get_node_results(dopf_ctx *ctx, double *generation /*N*T*/, double *discharge /*N*T*/, double *charge /*N*T*/);
/* Any of P, D, C, avg_U, avg_K, lambda, mu, rho may be NULL (= keep). iteration >= 1 is the
 * value admm.iteration would have before the next solve; iteration == 1 means "no result yet"
 * only if all primal pointers are NULL and the state is untouched. */
int dopf_set_state(dopf_ctx *ctx, const double *P, const double *D, const double *C,
                   const double *avg_U, const double *avg_K,
                   const double *lambda, const double *mu, const double *rho, int32_t iteration);

/* Diagnostics: number of storage sub-problems whose inner root search hit its iteration cap
 * since creation (0 in every healthy run), and a version string. */
/* Measurement: n_iters iterations launched kernel by kernel (no graph) with a hipEvent pair around
 * every kernel on the stream it runs on; one host sync at the end. Average milliseconds per launch. */
typedef struct dopf_timing {
    double tables_ms, gen_ms, sto_ms, slack_ms, reduce_ms, dual_ms;  /* per-kernel averages; with agents_fused
                                                                        gen_ms is the ONE x-update launch
                                                                        (k_agents) and sto_ms an empty pair */
    double iter_ms;                                                  /* whole iteration, event to event */
    double empty_ms;    /* an event pair with nothing between: the fixed cost inside every number above */
    int32_t iters;
    int32_t agents_fused;   /* 1: generators and storages ran as one launch */
    int32_t tail_fused;     /* 1: consensus sums, dual step and stop test ran inside the x-update launch(es): reduce_ms and
                               dual_ms are empty event pairs */
    int32_t slack_in_dual;  /* 1 (networks): node sums left from k_slack, slack sums were formed by the dual/price kernel:
                               reduce_ms is an empty event pair */
    int32_t quiet;          /* 1 (networks): no line was flagged, k_slack was not launched (slack_ms is an empty event pair): the dual/price
                               kernel formed the node sums too */
    int32_t sto_lean;       /* 1: the storages of this problem are solved by the lean active-set body (csrc/sto_lean.h) */
    int32_t persist;        /* 1: dopf_iterate runs several iterations per launch on this context (DOPF_F_PERSIST; this timed call itself
                               launches iteration by iteration) */
} dopf_timing;
int dopf_iterate_timed(dopf_ctx *ctx, int32_t n_iters, dopf_timing *out);
/* DOPF_F_TIME_CALLS: milliseconds between the first launch of the last dopf_iterate call and the end of its last one, on the
 * device (-1 without the flag, or when the call ended early on a stop). */
double dopf_last_call_ms(const dopf_ctx *ctx);

int64_t dopf_solver_failures(dopf_ctx *ctx);

/* ---- the central reference on the device ---------------------------------------------------------------
 * Replaces src/opf_central_reference.jl:16-81 (one JuMP model of the whole multi-period DC-OPF, solved by Gurobi): the same
 * LP — variables P, D, C, E in their boxes, energy balance per timestep, |ptdf * injection| <= f_max, storage balance —
 * solved on the GPU by a first-order primal-dual method (diagonal step sizes, averaging, restarts; csrc/kernels_central.hip).
 * It is the parity target of the decentral run and independent of it. p holds ALL agents; q supplies device and flags.
 * Stops when |primal - dual objective| / (1 + |primal|) + worst violation / (1 + max demand) <= tol, or at max_iters.
 * Outputs (any may be NULL), layouts as in dopf_get_primal / dopf_get_consensus: P, D, C, E; system_price[T] = dual.(EB);
 * nodal_price[N*T] = lambda + sum_l (dual FlowUpper + dual FlowLower)[l,t] ptdf[l,:]; line_utilization[L*T] = ptdf * I;
 * flow_upper_dual[L*T] = dual.(FlowUpper), flow_lower_dual[L*T] = dual.(FlowLower) (opf_central_reference.jl:71-79: both are
 * d objective / d max_capacity <= 0, non-zero only where that limit binds). */
typedef struct dopf_central_result {
    double objective, dual_objective, primal_infeasibility, gap;
    int32_t iterations, converged;
} dopf_central_result;
int dopf_central_solve(const dopf_problem *p, const dopf_params *q, double tol, int32_t max_iters,
                       dopf_central_result *res, double *P, double *D, double *C, double *E,
                       double *system_price, double *nodal_price, double *line_utilization,
                       double *flow_upper_dual, double *flow_lower_dual);

/* ---- consensus sum across GPUs inside the library (RCCL over xGMI, loaded at run time) -------------
 * Replaces nothing in the reference (it has no parallelism); what is distributed is the agent loop of
 * optimize_all_subproblems! (src/optimization/subproblems.jl:1-17) and the agent sums of Result(...)
 * (src/structures/results.jl:72-106), summed over ranks by ONE all-reduce per iteration.
 *
 * One process per GPU: every rank creates its context from its shard (dopf_problem.G/S = local agents,
 * dopf_params.n_agents_global = all agents), rank 0 calls dopf_comm_unique_id and ships the 128 bytes to
 * the other ranks over any host channel (MPI, torch.distributed, a file), every rank calls dopf_comm_init.
 * From then on dopf_iterate runs local_update -> all-reduce -> apply_consensus per iteration with no host round
 * trip: plain launches by default, captured in its hipGraphs with DOPF_F_COMM_GRAPH (back to plain launches if RCCL
 * refuses the capture). */
#define DOPF_COMM_ID_BYTES 128
int dopf_comm_unique_id(void *id128);
int dopf_comm_init(dopf_ctx *ctx, int32_t world, int32_t rank, const void *id128);
/* world / rank of the context's communicator (1 / 0 without one); in_graph = 1 once dopf_iterate has
 * captured the collective into its graphs, 0 while it launches eagerly. Any pointer may be NULL. */
int dopf_comm_info(const dopf_ctx *ctx, int32_t *world, int32_t *rank, int32_t *in_graph);

/* ---- peer exchange: the same sum without a collective library ------------------------------------------------
 * For the sub-kilobyte consensus vector of a copper plate (776 B on BASELINE configs[2]) a library all-reduce is pure
 * latency (tens of microseconds against a 40 us iteration). Here ONE kernel per iteration stores this rank's vector
 * into every peer's receive area over xGMI, publishes a sequence number, waits for the peers' numbers and adds the
 * copies in rank order (bitwise identical sums on every rank; csrc/kernels_consensus.hip k_xchg). The kernel is part of
 * the iteration hipGraphs. One process per GPU: every rank calls dopf_xchg_export (allocates its receive area, returns
 * a 64-byte hipIpc handle), the world handles are gathered in rank order over any host channel, every rank calls
 * dopf_xchg_init (maps the peers' areas, host rendezvous). world <= 16. A peer that does not show up within
 * DOPF_XCHG_TIMEOUT_MS (default 20 000) makes dopf_iterate / dopf_sync return DOPF_E_DEVICE; the kernels always end. */
#define DOPF_XCHG_HANDLE_BYTES 64
int dopf_xchg_export(dopf_ctx *ctx, int32_t world, void *handle64);
int dopf_xchg_init(dopf_ctx *ctx, int32_t world, int32_t rank, const void *handles /* world x 64 bytes, rank order */);

/* One process, n GPUs — what a Julia `ccall` host uses (no launcher): p holds ALL agents; the library
 * cuts the agent lists into n contiguous shards, creates one context per device (devices[i], or 0..n-1
 * when NULL), joins them with ncclCommInitAll and drives each from its own host thread.
 * Replaces ADMM(...) + run!(admm) exactly like dopf_create + dopf_iterate do on one GPU. */
typedef struct dopf_multi dopf_multi;
int  dopf_multi_create(dopf_multi **out, const dopf_problem *p, const dopf_params *q, int32_t n_gpus,
                       const int32_t *devices);
void dopf_multi_destroy(dopf_multi *m);
const char *dopf_multi_last_error(const dopf_multi *m);   /* m == NULL: last dopf_multi_create error */
int  dopf_multi_iterate(dopf_multi *m, int32_t n_iters, int32_t *iters_done, int32_t *converged);
/* Primal rows of all shards in the caller's agent order (layout of dopf_get_primal). */
int  dopf_multi_get_primal(dopf_multi *m, double *P, double *D, double *C, double *E);
int32_t dopf_multi_size(const dopf_multi *m);
/* Shard i's context: duals, consensus state, residuals and prices are replicated, read them from
 * shard 0 with the dopf_get_* calls above. */
dopf_ctx *dopf_multi_ctx(dopf_multi *m, int32_t i);
/* Diagnostics, 15 counters: [0..2] scan-kernel statistics (only in -DDOPF_STATS builds), [3] storages the
 * active-set kernel certified in the LAST iteration, [4] storages it left to the scan kernel, [5..8] contact-set
 * rounds / Newton iterations (sum, max per lane group), [9..14] cycles per section (DOPF_STATS builds). */
int dopf_debug_stats(dopf_ctx *ctx, uint64_t *out15);
/* Diagnostics (-DDOPF_STATS builds; -1 otherwise): 8 wall-clock stamps (100 MHz) per wave of the storage body's last launch. */
int dopf_debug_timeline(dopf_ctx *ctx, uint64_t *out, int32_t n);
/* networks: out3 = { the quiet chain (no k_slack launch while no line is flagged) is allowed for this context, the next launches
 * would use it, times it parked itself because a line got flagged and the host went back to the chain with k_slack } */
int dopf_debug_quiet(dopf_ctx *ctx, int64_t *out3);
/* Diagnostics (L > 0): the breakpoint table of node n, timestep t that the last x-update used:
 * beta, psi: 2L doubles (first *m valid, ascending), slope: 2L+1, psi0 = Psi(0). */
int dopf_debug_table(dopf_ctx *ctx, int32_t n, int32_t t, double *beta, double *psi, double *slope,
                     double *psi0, int32_t *m);
const char *dopf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DOPF_H */

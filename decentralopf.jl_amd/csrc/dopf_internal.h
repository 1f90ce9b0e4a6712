// dopf_internal.h — shared declarations of libdopf_hip (gfx950 only).
//
// Layout of everything that lives in HBM (all fp64 unless noted):
//   P            [t + T*g]      generator output, updated IN PLACE each iteration (the x-update of a
//                               generator needs only its own previous value), generators sorted by node
//   D, C         [t + T*s]      storage discharge / charge, in place, storages sorted by node
//   E            [t + T*s]      storage level = cumsum(C - D): written on request only (dopf_get_primal, the central solver)
//   dltG, dltS   [t + T*a]      change of the agent's net injection in this iteration (only when L > 0:
//                               feeds the slack sums sum_a U_a, sum_a K_a)
//   lam [T], mu/rho [l + L*t]   current duals; *_used = the ones the last solve read
//   inj [n + N*t], s [T], flow [l + L*t], avgU/avgK [l + L*t], price [n + N*t]
//                               consensus state of the previous iteration, replicated per GPU
//   tb_*         per (n,t) breakpoint tables of Psi_{n,t} (only when L > 0)
//   part_*       per work-item partial sums, reduced in a fixed order (bitwise reproducible)
//   cons         [N*T | L*T | L*T | 1]  the consensus vector that is all-reduced across ranks
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dopf.h"

namespace dopf {

// Tuning knobs of the experiments (item counts, block counts, launch splits) are environment variables
// of builds with -DDOPF_EXPERIMENTS only; the shipped library does not change kernel selection on ambient environment.
#ifdef DOPF_EXPERIMENTS
inline const char *exp_env(const char *name) { return getenv(name); }
#else
inline const char *exp_env(const char *) { return nullptr; }
#endif

// one block's share of the agent list: agents [a0, a1) all sit at `node`
struct Item {
    int a0, a1, node, row;          // row (networks): the item's row in DevView::part_T (a storage item's scan partial; its warm-start partial: row + 1)
};

// device-resident status word block (one per context)
struct Status {
    int iteration;          // admm.iteration (1-based)
    int converged;          // Convergence.all
    int halt;               // converged || iteration > max_iters: every kernel returns at once
    int iters_total;        // iterations computed since creation
    unsigned long long solver_fail;
    unsigned long long dbg_scans, dbg_wave_loops, dbg_events;   // storage kernel statistics (DOPF_STATS builds)
    unsigned long long dbg_cyc[6];                              // DOPF_STATS: wave cycles per section of the storage body
    unsigned long long dbg_reason[4];                           // DOPF_STATS: no prices / Newton / level / sign
    unsigned long long resbits[3];   // running max of |dual change| as bit patterns (>= 0 doubles)
    unsigned long long resbits2[2][3];   // k_dual_price_t1024: the same, one set per iteration parity (the stop test does not read them:
                                         // its ticket word carries "some residual >= eps"); the host decodes the reported set
    int walk_last;                  // k_dual_price_t1024: timesteps for which the last dual step flagged a line (k_slack has agents to walk in the next iteration)
    int res_set;                    // -1: res[] holds the residuals of the last checked iteration; 0/1: resbits2[res_set] does
    double res[3];          // lambda / mu / rho residual inf-norms of the last checked iteration
    double total_cost;
    int xchg_timeout;       // peer exchange: a peer's part of the consensus sum did not arrive in time (sticky)
    int tail_timeout;       // tail in the launch: the block sums of an iteration did not all arrive in time (sticky)
    int tail_par;           // tail in the launch: which of the two accumulator sets the next launch adds into
    int pseq;               // persistent iterations (agents_persist.h): dual updates published inside launches so far (wraps; only differences count)
};

struct XchgView;
// accumulators of the one-launch iterations (kernels_agents.hip); lives in device memory
struct TailView {
    long long *acc;                 // [2 sets][kAccRep][accStride]: slots 0..T-1 injection sums, slot T cost
    int accStride;
    int expect;                     // blocks that add to every slot (generator blocks + storage items)
    double scaleInj, invInj, scaleCost, invCost;        // fixed-point scales (powers of two) and their inverses
    const XchgView *xchg;           // non-null (a context joined to a peer exchange): the tail block sums its vector over the ranks
                                    // itself, between its own sums and the dual step (device copy of the exchange's view)
};

struct DevView {
    const DevView *self;            // this view in device memory (what a non-inlined device function is handed)
    int N, L, T, G, S, M2;          // M2 = 2L
    int maxNodeAgents;              // most agents (generators + storages) at one node
    int nGenItems, nStoItems;
    int genTT, genR;                // generator block tiling: TT = min(T, 512) timesteps x R agents
    int genRows;                    // > 0: (one node) the generators' partial sums are this many rows, one per streaming block, not one per item
    int genBlocks;                  // > 0 (needs genChunk): the fused launch has this many generator blocks, each walking items b, b + genBlocks, ...
    int debugLeave;                 // DOPF_F_DEBUG_LEAVE (tests)
    int coldInWarm;                 // networks: k_sto_warm calls the scan body itself for what it leaves over (no k_sto_update launch)
    double *part_T;                 // networks (L > 0): the items' partial injection sums TRANSPOSED, [t][row] with rowsT rows per timestep — node by node
    int rowsT;                      // (a node's generator items, then two rows per storage item: scan partial, warm-start partial). The block of the
                                    // dual/price kernel that owns timestep t reads ALL rows of t as one contiguous vector (the quiet chain: no k_slack
                                    // launch) instead of one 8-byte word out of every row of the [row][t] layout the copper plates keep (part_*inj);
                                    // the sums are formed in the same order from either layout: same bits. Null: part_ginj / part_sinj / part_sinj_w.
    int rowsN;                      // ... of which rowsN are items' rows: the rows are PLACED by the XCD their writer block runs on (block index mod 8,
    const int *row_of_pos, *pos_of_row;   // eight regions on 128-byte boundaries) — written in node order, every 128-byte line of a timestep would collect 8-byte
                                    // pieces in all eight L2s and be written back eight times at the kernel's end. row_of_pos[p]: the node-ordered row
                                    // at position p (-1: padding); pos_of_row: its inverse. Item::row is a POSITION.
    int dualRowsOff, dualSdOff;     // the one-launch dual/price kernel's dynamic LDS, in doubles: where the rows of its timestep (aliased with the tables'
    int dualLdsBytes;               // scratch) and the vectors behind them start, and its size
    int tablesInDual;               // > 0 (networks on the one-launch dual/price kernel): that kernel builds the breakpoint tables of its
                                    // timestep itself, with this many waves; no k_tables launch
    int splitDual;                  // (experiments, DOPF_SPLIT_DUAL=1) networks: dual and price steps as two launches
    int stoChunk;                   // > 0: one node, storage item i = storages [i*stoChunk, (i+1)*stoChunk)
    int genChunk;                   // > 0: one node, generator item i = rows [i*genChunk, (i+1)*genChunk) (no item look-up)
    int genSkip;                    // pair kernel with row skipping (blocks sweep >= 8 passes of agents)
    int genTT2, genR2;              // pair kernel (copper plate, even T <= 1024): T/2 double2 columns x R2 agents; 0 = off
    int sliceDual;                  // per launch: k_reduce stops after the slice sums (level 1) and the one-block dual
                                    // kernel adds the slices itself (single-GPU iterate path, small consensus state)
    int reduceRB;                   // reduce blocks per node (two-level fixed-order sum)
    int slackInDual;                // per launch (networks on the one-launch dual/price kernel, single-GPU chain): k_slack stores the node
                                    // sums and the cost itself, the dual/price block of timestep t forms the slack sums of its lines from the
                                    // PTDF rows it reads anyway — no k_reduce launch
    int slackDualOk;                // the problem and the flags allow that
    int slackGlobal;                // per launch (contexts on a peer exchange, no line flagged): the chain of slackInDual with the exchange of
                                    // the node sums between k_slack and the dual/price kernel — that kernel then takes the nodes' injection
                                    // changes from the SUMMED injections (this iteration's minus the previous one's, both replicated) and counts
                                    // all ranks' agents; a dual step that flags a line parks the chain (as the quiet chain does)
    int quiet;                      // per launch (with slackInDual): the quiet chain — no line is flagged, k_slack is not launched, the dual/price
                                    // kernel forms the node sums too
    int genTT256;                   // networks, fused launch: column tiling of a 256-thread generator block with the same R as genR
    int fuseNet;                    // networks: generators + storages in one launch (k_net_agents), single-GPU chain
    int fuseAgents;                 // copper plate, even T: generators + storages in one launch (k_agents, 256-thread blocks)
    int use_warm;                   // storage warm-start kernel runs first; the scan kernel serves its failures
    int stoLean;                    // copper plates whose horizon fills the lane groups: the lean active-set body (sto_lean.h)
    int persistOk;                  // DOPF_F_PERSIST and a grid whose blocks are all resident at once: several iterations per launch (agents_persist.h)
    int persistIters;               // per launch: iterations this launch runs
    int max_iters;
    int keepDeltas;                 // DOPF_F_KEEP_DELTAS: dltG / dltS are written for every timestep (diagnostic getters)
    int rootCap;                    // iteration cap of the scan kernel's root search (80; 2 with DOPF_F_DEBUG_ROOT_CAP)
    double gamma, w_flow, w_prox, eps, mask_thr, invA;
    double nAgents;                 // the divisor of avg_U / avg_K as a number: all ranks' agents (1 / invA without the rounding)
    double cp_ia, cp_idet, cp_s2;   // copper-plate box2 constants with a = w_prox + gamma, b = gamma: 1/a, 1/(a^2 - b^2), 2/(a + b) (host: no divisions per block)
    // problem (read-only)
    const double *demand, *ptdf, *fmax;
    const double *ptdfT;                            // [n + N*l]: the transpose, for the price kernel's node-major threads
    const double *gen_mc, *gen_pmax;
    const double2 *gen_mp;                          // [g] {mc, pmax} side by side: one 16-byte load per row (streaming blocks)
    const double *sto_mc, *sto_pmax, *sto_emax;
    const Item *gen_items, *sto_items;
    const int *node_gen_beg, *node_sto_beg;         // N+1 each: agent ranges per node
    const double *node_win;                         // N: bound on |change of an agent's net injection| at the node
    const int *node_gitem_beg, *node_sitem_beg;     // N+1 each: item ranges per node
    // primal state
    double *P, *D, *C, *E, *dltG, *dltS;
    int *gen_state;                 // per generator: 0 = P all zero, 1 = all at pmax, 2 = mixed (pair kernel's row skipping)
    // duals and consensus state
    double *lam, *mu, *rho, *lam_used, *mu_used, *rho_used;
    double *inj, *s, *flow, *avgU, *avgK, *price;
    double *s_used, *flow_used, *avgU_used, *avgK_used;   // what the last x-update read (diagnostic getters: per-agent U, K, penalty terms)
    // tables
    double *tb_beta, *tb_psi, *tb_slope, *tb_psi0;
    int *tb_m;
    // partials
    double *part_ginj, *part_gcost;                 // [item*T + t], [item]
    double *part_sinj, *part_scost;                 // storage scan kernel, per item
    double *part_sinj_w, *part_scost_w;             // storage warm-start kernel, per item
    double *nu_prev;                                // [t + T*s] price of stored energy of the last solve (copper plates: + the step's price offset theta)
    int *nu_valid, *sto_fail, *item_fail;           // [s], [s], [item] (= storages of the item the warm start left over)
    double *part_U, *part_K;                        // [(n + N*t)*L + l]: only the entries k_slack had to walk agent by agent
    double *node_dsum;                              // [n + N*t] change of the node's injection in this iteration (L > 0)
    double *prev_node;                              // [n + N*t] the node's (this rank's agents') injection sum of the previous iteration (L > 0)
    const double *line_reach;                       // [l] max over nodes of |kap| W_n: beyond it no agent of any node can flip the line's slack
    int *tab_skip;                                  // [t] the price kernel has written the (empty) tables of timestep t
    int *walk_flag, *walk_any;                      // [l + L*t], [t]: the slack sums of (l,t) need the per-node cases (set by the dual step)
    double *part2, *part2_cost;                     // [(n*RB + rb)*T + t], [rb]
    unsigned long long *dual_ticket;                // [1] one-launch dual/price kernel: blocks whose residuals are in (low word) and how many
                                                    // of them saw a residual >= eps (high word)
    // One-launch iterations (one node, no lines, single-GPU chain; kernels_agents.hip "the tail of the iteration inside the
    // x-update launch"): every block of the x-update adds its per-timestep injection sums and its cost into integer
    // accumulators (fixed point + arrival count in one 64-bit atomic add), and one extra block of the launch waits for the
    // counts and runs the whole tail of the iteration. No k_reduce, no dual kernel, one kernel boundary per iteration.
    const TailView *tail;                           // per launch: non-null = this launch chain works that way
    const TailView *tailDev;                        // the context's TailView in device memory (null: not supported for this problem)
    int *reduce_ticket;                             // [n]
    double *cons;
    Status *st;
};

#ifndef DOPF_ACC_REP
#define DOPF_ACC_REP 16
#endif
constexpr int kAccRep = DOPF_ACC_REP;     // replicas of the accumulators (a block adds into replica blockIdx % kAccRep: 1/16 of the adds per address)

// consensus states up to this many (n,t) / (l,t) entries take the one-block dual step
constexpr size_t kSmallConsensus = 4096;
#ifndef DOPF_STREAM_ROWS
#define DOPF_STREAM_ROWS 6
#endif
constexpr int kGenStreamRows = DOPF_STREAM_ROWS;          // rows per lane and item in the streaming generator blocks (one batch of loads)

// check_convergence!, convergence.jl:1-31 (one thread). The status words a stop test starts from are loaded early by
// callers that can (a load at the very end of a one-block kernel is a round trip on its critical path).
struct StatusPre { int iteration, converged, iters_total; };
__device__ __forceinline__ StatusPre status_load(const DevView &v)
{
    StatusPre s;
    s.iteration = v.st->iteration; s.converged = v.st->converged; s.iters_total = v.st->iters_total;
    return s;
}

__device__ __forceinline__ void status_update(const DevView &v, const StatusPre s, double r0, double r1, double r2)
{
    Status *st = v.st;
    int conv = s.converged, it = s.iteration;
    if (it != 1) {                                                        // convergence.jl:3
        st->res[0] = r0; st->res[1] = r1; st->res[2] = r2;
        st->res_set = -1;
        conv = (r0 < v.eps) && (r1 < v.eps) && (r2 < v.eps);
        st->converged = conv;
    }
    st->iters_total = s.iters_total + 1;
    if (!conv) it += 1;                                                   // convergence.jl:25-30
    st->iteration = it;
    st->halt = conv || (v.max_iters > 0 && it > v.max_iters);
}

__device__ __forceinline__ void status_update(const DevView &v, double r0, double r1, double r2)
{
    status_update(v, status_load(v), r0, r1, r2);
}

struct Launch {
    int stoLPS, stoNCH;
};

// the central reference's view (kernels_central.hip): the context's arrays (P, D, C, E, items, partial sums, cons) plus the
// multipliers, running sums and step sizes of the primal-dual iteration
struct CentralView {
    DevView v;
    double *yb, *yf, *yE;                   // multipliers: balance [T], flows [l + L*t], levels [t + T*s]
    double *aP, *aD, *aC, *aE, *ab, *af;    // running sums of the iterates since the last restart
    double *pi;                             // [n + N*t] yb + ptdf' yf of the candidate being worked on
    const double *tauN;                     // [n] 1 / (1 + sum_l |ptdf[l,n]|): primal step of a generator at node n
    const double *absHn;                    // [n] sum_l |ptdf[l,n]|
    const double *sigF;                     // [l] 1 / sum_n |ptdf[l,n]| (units at n, a storage counting twice)
    double sigB, w;                         // 1 / (G + 2S); primal weight
    double *m_gen, *m_sto, *m_dual;         // metrics: per generator item [1], per storage item [3], per timestep [3]
};

// kernels_central.hip
void central_launch_iteration(const CentralView &c, const DevView &vreduce, hipStream_t s);
void central_launch_metrics(const CentralView &c, const DevView &vreduce, const double *XP, const double *XD, const double *XC,
                            const double *XE, const double *yb, const double *yf, double scale, hipStream_t s);
void central_launch_scale_copy(double *dst, const double *src, double scale, size_t n, hipStream_t s);

// kernels_agents.hip
void launch_gen_update(const DevView &v, hipStream_t s);
void launch_sto_update(const DevView &v, const Launch &lc, hipStream_t s);
void launch_net_agents(const DevView &v, const Launch &lc, hipStream_t s);
void launch_agents_fused(const DevView &v, const Launch &lc, hipStream_t s);
void launch_agents_persist(const DevView &v, const Launch &lc, hipStream_t s);        // v.persistIters iterations in one launch
bool sto_config_supported(int T, Launch *lc);
int debug_timeline(unsigned long long *out, int n);     // DOPF_STATS builds: per-wave stamps of the storage body
// kernels_consensus.hip
void launch_tables(const DevView &v, hipStream_t s);
void launch_slack(const DevView &v, hipStream_t s);
void launch_reduce(const DevView &v, hipStream_t s);
// Peer exchange (dopf_comm.hip sets it up): every rank owns a receive area [2 parities][world source ranks][n doubles] plus
// flags [2][world][chunks]; data[r] / flags[r] are rank r's areas as addressable from THIS device (own allocation, a peer
// device of the same process, or an IPC mapping of another process's allocation).
constexpr int kXchgMaxWorld = 16;
constexpr int kXchgChunk = 2048;                // doubles per block
struct XchgView {
    int world, me, nchunks;
    int rs;                                     // 1: reduce-scatter + all-gather form (k_xchg_rs: vectors of more than one chunk per rank)
    unsigned long long n;                       // doubles in the consensus vector
    unsigned long long timeout_ticks;           // wall_clock64 ticks (100 MHz) a block waits for its peers
    double *data[kXchgMaxWorld];
    unsigned long long *flags[kXchgMaxWorld];
    double *sum[kXchgMaxWorld];                 // [2 parities][n]: the summed chunks, written by each chunk's owner (rs form)
    unsigned long long *sflags[kXchgMaxWorld];  // [2][chunks]
};
void launch_xchg(const DevView &v, const XchgView &x, hipStream_t s, bool inj_only = false);   // cons <- sum over ranks of cons (rank order);
//      // inj_only: just the chunks that hold the node sums and the cost (the slack sums are formed behind the exchange: slackGlobal)
void launch_dual(const DevView &v, hipStream_t s, const XchgView *xd = nullptr);   // xd: peer exchange inside the one-block kernel
//      // consensus -> duals, residuals, prices, status
void launch_derive(const DevView &v, hipStream_t s, bool from_primal);
void launch_derive_level(const DevView &v, hipStream_t s);            // E = cumsum(C - D) into v.E
void launch_penalty_sums(const DevView &v, double *out /* [3][N][T], device */, hipStream_t s);   // Result.penalty_term, per node
void launch_node_results(const DevView &v, double *gen, double *dis, double *chg, hipStream_t s);   // [n + N*t] each, device pointers   // consensus -> inj/s/flow/price (no dual step)

}  // namespace dopf

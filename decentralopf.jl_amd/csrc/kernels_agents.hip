// kernels_agents.hip — the x-update of every agent (gfx950, wave64, fp64).
//
// Replaces optimize_subproblem(::Generator) / optimize_subproblem(::Storage)
// (reference src/optimization/subproblems.jl:19-207 with add_penalty_terms!,
// src/optimization/penalty_terms.jl:1-53) — there: one JuMP model build + one Gurobi barrier solve
// per agent per iteration; here: the slack variables U, K are eliminated in closed form and
//   generator   P*  = clamp(root of a monotone piecewise-linear function, 0, pmax)      per (g,t)
//   storage     the state-of-charge QP is solved exactly by the price-threshold recursion
//               (DESIGN.md "storage kernel"): a group of LPS lanes owns one storage, lanes <-> timesteps,
//               clamp-add maps are composed with wave shuffles (an associative scan), the scalar
//               price of each constant-price segment comes from a safeguarded Newton iteration.
// Both kernels also produce the per-item partial sums of the agents' net injection and cost
// (the agent loop of Result(...), src/structures/results.jl:72-106) in a fixed order.
#include "dopf_internal.h"

#ifdef DOPF_INLINE_CALLS
#define DOPF_CALL_ATTR __forceinline__
#else
#define DOPF_CALL_ATTR __attribute__((noinline))
#endif

namespace dopf {

#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
__device__ unsigned long long g_timeline[8192 * 8 + 8192 * 8];      // per wave of the storage body: wall-clock stamps (100 MHz)
#endif
#ifdef DOPF_BLOCK_STAMPS
#define DOPF_TAIL_STAMP(i) { if (threadIdx.x == 0) g_timeline[(i)] = wall_clock64(); }
#else
#define DOPF_TAIL_STAMP(i)
#endif
#ifdef DOPF_STATS
#define DOPF_STAMP(i) { if (lane == 0 && rep == 0 && round == 0) { const int w_ = blk * 4 + (tid >> 6); if (w_ < 8192) g_timeline[w_ * 8 + (i)] = wall_clock64(); } }
#else
#define DOPF_STAMP(i)
#endif

__device__ __forceinline__ double clampd(double v, double lo, double hi)
{
    return fmin(fmax(v, lo), hi);
}

// 1/x to fp64 rounding without the ~30-instruction division sequence: hardware estimate + two Newton steps
__device__ __forceinline__ double rcp64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Loads / stores of words that ANOTHER block of the SAME launch wrote or will read (the persistent iterations of
// agents_persist.h): agent scope — past the CU's vector L1 and the scalar cache, which are only refreshed between launches.
__device__ __forceinline__ double p_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int p_ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void p_st(double *p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void p_sti(int *p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------------------------------------
// the tail of the iteration inside the x-update launch (DevView::tail; one node, no lines)
// ------------------------------------------------------------------------------------------------
//
// Replaces, for that case, the agent loop of Result(...) (reference src/structures/results.jl:72-106), update_duals!
// (src/optimization/update_duals.jl:1-39) and check_convergence! (src/optimization/convergence.jl:1-31) as two more
// launches (k_reduce, k_dual_price_small) by one extra block of the launch itself:
//   * every block ADDS its per-timestep sums of net injection and its cost into replica blockIdx % kAccRep of a set of
//     64-bit integer accumulators with ONE device-scope atomic add per value, fire and forget (the block does not wait for
//     them and exits). The addend is (round(x * 2^k) << kAccCntBits) + 1: the upper 54 bits carry the value in fixed point
//     (k from the problem's bounds, so that no sum can overflow; integer addition commutes, so the totals do not depend on
//     the order in which blocks finish: bitwise reproducible), the low 10 bits count the contributions that have landed;
//   * the launch's LAST block polls the accumulators (it is the ONLY block that ever waits, and no block waits for it: the
//     others keep being dispatched into the remaining wave slots and finish on their own, whatever the dispatch order and
//     the residency — nothing can deadlock); a lane pair owns a slot and is done when the counts of its slot over the
//     replicas add up to the number of contributing blocks — the data is its own arrival signal, one memory round trip
//     between the last block's adds and the tail, no ticket, no fence: value and count arrive in the same atomic;
//   * it then runs the tail of the iteration — injection, imbalance, lambda step, price, residual, stop test: the
//     arithmetic of k_dual_price_small's copper-plate path. Two accumulator sets take turns: the one a launch used is zeroed
//     by the NEXT launch's tail block while it waits. On a context joined to a peer exchange the tail block also sums its
//     vector over the ranks (TailView::xchg), between its own sums and the dual step.
// The wait is bounded by wall clock (Status::tail_timeout, DOPF_E_DEVICE): the kernel always ends.
// (First version: every block waited for its adds, took a two-level ticket, the last one drained the accumulators with
// atomic exchanges — four dependent device-scope round trips of ~2.5 us each next to the streaming blocks: as slow as the
// two launches it replaced.)
#ifdef DOPF_TAIL_NOINLINE
#define DOPF_TAIL_INLINE __attribute__((noinline))
#else
#define DOPF_TAIL_INLINE __forceinline__
#endif
constexpr int kAccCntBits = 10;
constexpr unsigned long long kAccCntMask = (1ull << kAccCntBits) - 1ull;

// (every block of a launch reads the parity word once: it changes only at the very end of a launch, in the tail block)
// (the accumulators' description lives in device memory, DevView::tail points at it: one pointer in the kernel arguments
// instead of seven more live scalars in kernels that spill scalars already)
template <bool COUNTED = true>
__device__ __forceinline__ void acc_add(const TailView &tv, int par, int slot, double x, double scale)
{
    const long long q = __double2ll_rn(x * scale);
    unsigned long long *p = reinterpret_cast<unsigned long long *>(tv.acc) +
                            ((size_t)par * kAccRep + (size_t)(blockIdx.x % kAccRep)) * tv.accStride + slot;
    __hip_atomic_fetch_add(p, ((unsigned long long)q << kAccCntBits) + (COUNTED ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The tail block (inlined into the kernels that can carry it: it must stay below their register budgets — a lane pair owns
// a slot, each lane polls half of the replicas, 8 loads in flight). Two sets of accumulators take turns (Status::tail_par):
// the set an iteration used is zeroed by the NEXT tail block while it waits, not between the last arrival and the dual step.
// (`self`: the context's view in DEVICE memory — what this block needs is read there, inside its branch, instead of
// widening the set of kernel arguments every block of the launch loads at its start)
__device__ DOPF_TAIL_INLINE void tail_block(const DevView *self)
{
    const DevView &v = *self;
    const TailView tv = *v.tailDev;
    const int expect = tv.expect;
    __shared__ double wmaxT[8];
    __shared__ int badT;
    const int tid = threadIdx.x, T = v.T, nth = (int)blockDim.x;       // 256 or 512 threads
    constexpr int HR = kAccRep / 2;
    DOPF_TAIL_STAMP(0)
    if (v.st->halt) return;
    const int par = v.st->tail_par;
    StatusPre spre{};
    if (tid == 0) { spre = status_load(v); badT = 0; }
    {   // the other set: read by the previous launch's tail, not touched by this launch
        unsigned long long *z = reinterpret_cast<unsigned long long *>(tv.acc) + (size_t)(par ^ 1) * kAccRep * tv.accStride;
        for (int i = tid; i < kAccRep * tv.accStride; i += nth) __hip_atomic_store(z + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long limit = 200000000ull;                      // 2 s of the 100 MHz wall clock
    const unsigned long long tstart = wall_clock64();
    __syncthreads();
    DOPF_TAIL_STAMP(1)
    double rl = 0.0;
    const int half = tid & 1;
    for (int t0 = 0; t0 <= T; t0 += nth / 2) {          // slot T = cost
        const int t = t0 + (tid >> 1);
        const bool act = t <= T;                        // (whole lane pairs)
        const int tc = t < T ? t : 0;
        const double dem = v.demand[tc], lam_old = v.lam[tc], s_old = v.s[tc];
        const unsigned long long *base = reinterpret_cast<const unsigned long long *>(tv.acc) +
                                         ((size_t)par * kAccRep + (size_t)half * HR) * tv.accStride + (act ? t : 0);
        long long isum = 0;
        bool ok = !act;
        for (unsigned round = 1; __any(!ok); ++round) {          // (every lane stays: the pair sums are wave shuffles)
            unsigned long long x[HR];
#pragma unroll
            for (int r = 0; r < HR; ++r)
                x[r] = __hip_atomic_load(base + (size_t)r * tv.accStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned cnt = 0;
            long long sum = 0;
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                cnt += (unsigned)(x[r] & kAccCntMask);
                sum += (long long)(x[r] & ~kAccCntMask) >> kAccCntBits;       // (arithmetic shift: the value part is signed)
            }
            cnt += __shfl_xor(cnt, 1);
            sum += __shfl_xor(sum, 1);
            if (!ok && (int)cnt == expect) { ok = true; isum = sum; }
            // (the clock is a scalar memory read of its own: not in every round; a round is one memory round trip)
            if ((round & 255u) == 0u && wall_clock64() - tstart > limit) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicOr(&badT, 1);             // (no early way out in front of the exchange's barriers)
        DOPF_TAIL_STAMP(2)
        double xsum = t < T ? (double)isum * tv.invInj : (double)isum * tv.invCost;       // this rank's sum of the slot
        if (tv.xchg) {                           // (only offered when all T + 1 slots fit ONE pass of this loop: dopf_comm.hip)
            // A context joined to a peer exchange: the sum over the RANKS happens here, between the rank's own
            // sums and the dual step — the slot's value goes into every peer's receive area, the ranks' copies come back from
            // the own area and are added in rank order (every rank adds the same numbers in the same order: bitwise the same
            // duals everywhere). Protocol as k_xchg's, one chunk; the wait is bounded. (Wave-uniform: every lane is here.)
            const XchgView &x = *tv.xchg;
            const int W = x.world, rk = x.me;
            const unsigned long long seq = (unsigned long long)v.st->iters_total + 1ull;
            const size_t xpar = (size_t)(seq & 1ull), n = x.n;
            // A rank whose OWN sums did not arrive in time (badT & 1) sends nothing and publishes no flag: its peers then fail in
            // this same iteration (their wait for this rank's flag is bounded) instead of running a dual step on a partial vector.
            __syncthreads();
            const bool localBad = (badT & 1) != 0;
            if (act && !half && !localBad)
                for (int q = 0; q < W; ++q) {
                    const int r = (rk + 1 + q) % W;                    // the peers first, the own slot last
                    x.data[r][(xpar * W + rk) * n + t] = xsum;
                }
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < W && !localBad) {
                __hip_atomic_store(x.flags[tid] + (xpar * W + rk) * x.nchunks, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long *f = x.flags[rk] + (xpar * W + tid) * x.nchunks;
                const unsigned long long w0 = wall_clock64();
                bool here = v.st->xchg_timeout == 0;
                while (here && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - w0 > x.timeout_ticks) here = false;
                }
                if (!here) atomicOr(&badT, 2);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            if (badT) continue;
            if (act && !half) {
                const double *mine = x.data[rk] + xpar * W * n;
                xsum = mine[t];
                for (int r = 1; r < W; ++r) xsum += mine[(size_t)r * n + t];
            }
        }
        if (!ok || !act || half) continue;
        if (t < T) {
            const double tot = xsum;
            const double xi = tot - dem;                          // results.jl:58-100 (one node: imbalance = its injection)
            v.cons[t] = tot;
            v.inj[t] = xi;
            v.s_used[t] = s_old;
            v.s[t] = xi;
            const double ln = lam_old + v.gamma * xi;             // update_duals.jl:8-13
            v.lam_used[t] = lam_old;
            v.lam[t] = ln;
            v.price[t] = ln;                                      // no lines: the nodal price is lambda
            rl = fmax(rl, fabs(ln - lam_old));
        } else {
            const double ctot = xsum;
            v.cons[T] = ctot;
            v.st->total_cost = ctot;
        }
    }
    for (int d = 32; d > 0; d >>= 1) rl = fmax(rl, __shfl_xor(rl, d));
    if ((tid & 63) == 0) wmaxT[tid >> 6] = rl;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (only LDS data crosses: the stores above need not be acknowledged first)
    DOPF_TAIL_STAMP(3)
    if (tid == 0) {
        if (badT & 2) { v.st->xchg_timeout = 1; v.st->halt = 1; return; }       // a peer's part did not arrive (sticky; DOPF_E_DEVICE): the chain
                                                                                // stops here — the rest of the graph would add into this accumulator set again
        if (badT) { v.st->tail_timeout = 1; v.st->halt = 1; return; }           // (sticky; the host reports DOPF_E_DEVICE)
        double r0 = 0.0;
        for (int q = 0; q < nth / 64; ++q) r0 = fmax(r0, wmaxT[q]);
        v.st->tail_par = par ^ 1;
        status_update(v, spre, r0, 0.0, 0.0);
    }
}

// box2 coefficients of a step whose Psi is linear with slope kap around the solution: a = w + kap, b = kap,
// 1/a, 1/(a^2 - b^2) = (1/w) / (w + 2 kap), 2/(a + b) = 2 / (w + 2 kap) — rebuilt per evaluation from kap instead of
// keeping three more arrays in registers (the kernels with lines sit at the VGPR limit)
__device__ __forceinline__ void lin_coef(double w, double iw, double kap, double &ia, double &idet, double &s2)
{
    const double r2 = rcp64(w + 2.0 * kap);
    ia = rcp64(w + kap);
    idet = iw * r2;
    s2 = 2.0 * r2;
}

// ------------------------------------------------------------------------------------------------
// generators
// ------------------------------------------------------------------------------------------------
//
// Block = one Item (agents [a0,a1) at one node). Thread (r, tt): timestep tt (+TT, +2TT, ...) of agents
// a0 + r, a0 + r + R, ...; with T <= 512 the block sweeps a contiguous range of P (agent-major), so
// every wave access is a dense 512-byte line set. Algorithmic traffic per update: 8 B read + 8 B write
// of P per (g,t) + 20 B of parameters per agent (L1-broadcast to the T lanes that share an agent).
// One item of generators at a node of a network (tables of the node's Psi): a block of BS threads, thread (r, tt) of an R x TT
// tiling, ceil(T / TT) column passes. k_gen_update<true> runs it with 512 threads, k_net_agents (generators and storages of a
// network in ONE launch) with 256 and the same R — the same rows meet in the same order, the sums carry the same bits.
template <int BS, int FL = 4>
__device__ __forceinline__ void gen_lines_body(const DevView &v, const int item, const int TT, const int R)
{
    __shared__ double red[BS];
    const Item it = v.gen_items[item];
    const int T = v.T, N = v.N;
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const double w = v.w_prox;
    double cost = 0.0;

    for (int tc = 0; tc < T; tc += TT) {
        const int t = tc + tt;
        double acc = 0.0;
        if (r < R && t < T) {
            const size_t at = (size_t)it.node + (size_t)N * t;
            const int m = v.tb_m[at];
            const double *beta = v.tb_beta + at * v.M2, *psi = v.tb_psi + at * v.M2;
            const double *slope = v.tb_slope + at * (v.M2 + 1);
            const double psi0 = v.tb_psi0[at];
            const double slope0 = slope[0];
            const double inv0 = rcp64(slope0 + w);             // (empty table: one piece for every agent of the node)
            const bool keepd = v.keepDeltas || v.walk_any[t];      // the change is needed agent by agent only for walked slack sums
            for (int g0 = it.a0 + r; g0 < it.a1; g0 += FL * R) {         // FL agents' rows in flight per lane (added in row order)
                double mc[FL], pm[FL], p0[FL];
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int g = g0 + u * R < it.a1 ? g0 + u * R : g0;
                    mc[u] = v.gen_mc[g]; pm[u] = v.gen_pmax[g]; p0[u] = v.P[(size_t)g * T + t];
                }
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int g = g0 + u * R;
                    if (g >= it.a1) break;
                    const size_t e = (size_t)g * T + t;
                    double dl;
                    if (m == 0) {
                        dl = -(mc[u] + psi0) * inv0;
                    } else {
                        int lo = 0, hi = m;           // first kink with psi + w beta >= -mc
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (psi[mid] + w * beta[mid] >= -mc[u]) hi = mid; else lo = mid + 1;
                        }
                        const int a = lo < m ? lo : m - 1;
                        dl = beta[a] - (mc[u] + psi[a] + w * beta[a]) * rcp64(slope[lo] + w);
                    }
                    const double pn = clampd(p0[u] + dl, 0.0, pm[u]);
                    v.P[e] = pn;
                    if (keepd) v.dltG[e] = pn - p0[u];
                    acc += pn;
                    cost += mc[u] * pn;
                }
            }
        }
        // fixed-order reduction over the R agent lanes that share a timestep (only LDS data crosses these barriers:
        // __syncthreads() would also wait for the acknowledgement of the rows just stored)
        red[tid] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (r == 0 && t < T) {
            double sum = 0.0;
            for (int q = 0; q < R; ++q) sum += red[q * TT + tt];
            v.part_T[(size_t)t * v.rowsT + it.row] = sum;          // (networks: [t][row], read along the rows by the consensus kernels)
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // cost: butterfly inside each wave, then the waves in order — one barrier instead of one per tree level
    __shared__ double wcost[BS / 64];
    for (int d = 32; d > 0; d >>= 1) cost += __shfl_xor(cost, d);
    if ((tid & 63) == 0) wcost[tid >> 6] = cost;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0) {
        double c = 0.0;
        for (int q = 0; q < BS / 64; ++q) c += wcost[q];
        v.part_gcost[item] = c;
    }
}

// The same item with the block's two column passes merged (needs 2 TT >= T: the 256-thread tiling of k_net_agents): thread
// (r, tt) owns timesteps tt and tt + TT, both columns' tables and rows are on their way before anything is computed, and the
// rows' costs are loaded once. A block of that launch is one of a few hundred that pass through the wave slots the storage
// blocks leave free, each a chain of dependent round trips: two chains in parallel instead of one behind the other. Per
// timestep the same rows meet in the same order as in gen_lines_body: the injection sums carry the same bits.
template <int BS, int FL>
__device__ __forceinline__ void gen_lines_body2(const DevView &v, const int item, const int TT, const int R)
{
    __shared__ double red[2][BS];
    const Item it = v.gen_items[item];
    const int T = v.T, N = v.N;
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const double w = v.w_prox;
    double cost = 0.0, acc[2] = {0.0, 0.0};
    const bool on[2] = {r < R && tt < T, r < R && tt + TT < T};
    if (on[0]) {
        int m[2];
        const double *beta[2], *psi[2], *slope[2];
        double psi0[2], inv0[2];
        bool keepd[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = on[c] ? tt + c * TT : tt;
            const size_t at = (size_t)it.node + (size_t)N * t;
            m[c] = v.tb_m[at];
            beta[c] = v.tb_beta + at * v.M2; psi[c] = v.tb_psi + at * v.M2;
            slope[c] = v.tb_slope + at * (v.M2 + 1);
            psi0[c] = v.tb_psi0[at];
            inv0[c] = slope[c][0];
            keepd[c] = v.keepDeltas || v.walk_any[t];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) inv0[c] = rcp64(inv0[c] + w);          // (empty table: one piece for every agent of the node)
        for (int g0 = it.a0 + r; g0 < it.a1; g0 += FL * R) {              // FL agents' rows in flight per lane and column
            double mc[FL], pm[FL], p0[2][FL];
#pragma unroll
            for (int u = 0; u < FL; ++u) {
                const int g = g0 + u * R < it.a1 ? g0 + u * R : g0;
                mc[u] = v.gen_mc[g]; pm[u] = v.gen_pmax[g];
                p0[0][u] = v.P[(size_t)g * T + tt];
                p0[1][u] = v.P[(size_t)g * T + (on[1] ? tt + TT : tt)];
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!on[c]) continue;
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int g = g0 + u * R;
                    if (g >= it.a1) break;
                    const size_t e = (size_t)g * T + tt + c * TT;
                    double dl;
                    if (m[c] == 0) {
                        dl = -(mc[u] + psi0[c]) * inv0[c];
                    } else {
                        int lo = 0, hi = m[c];           // first kink with psi + w beta >= -mc
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (psi[c][mid] + w * beta[c][mid] >= -mc[u]) hi = mid; else lo = mid + 1;
                        }
                        const int a = lo < m[c] ? lo : m[c] - 1;
                        dl = beta[c][a] - (mc[u] + psi[c][a] + w * beta[c][a]) * rcp64(slope[c][lo] + w);
                    }
                    const double pn = clampd(p0[c][u] + dl, 0.0, pm[u]);
                    v.P[e] = pn;
                    if (keepd[c]) v.dltG[e] = pn - p0[c][u];
                    acc[c] += pn;
                    cost += mc[u] * pn;
                }
            }
        }
    }
    // fixed-order reduction over the R agent lanes that share a timestep (only LDS data crosses the barrier)
    red[0][tid] = acc[0]; red[1][tid] = acc[1];
    __shared__ double wcost[BS / 64];
    for (int d = 32; d > 0; d >>= 1) cost += __shfl_xor(cost, d);
    if ((tid & 63) == 0) wcost[tid >> 6] = cost;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (r == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = tt + c * TT;
            if (tt < TT && t < T) {
                double sum = 0.0;
                for (int q = 0; q < R; ++q) sum += red[c][q * TT + tt];
                v.part_T[(size_t)t * v.rowsT + it.row] = sum;
            }
        }
    }
    if (tid == 0) {
        double c = 0.0;
        for (int q = 0; q < BS / 64; ++q) c += wcost[q];
        v.part_gcost[item] = c;
    }
}

template <bool LINES>
__global__ __launch_bounds__(512) void k_gen_update(DevView v)
{
    if (v.st->halt) return;
    if (LINES) {
        gen_lines_body<512>(v, blockIdx.x, v.genTT, v.genR);
        return;
    }
    __shared__ double red[512];
    const Item it = v.gen_items[blockIdx.x];
    const int T = v.T, N = v.N, TT = v.genTT, R = v.genR;
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const double w = v.w_prox, gam = v.gamma;
    const double inv = 1.0 / (w + gam);
    double cost = 0.0;

    for (int tc = 0; tc < T; tc += TT) {
        const int t = tc + tt;
        double acc = 0.0;
        if (r < R && t < T) {
            // copper plate / no line touches this problem: Psi(d) = price + gamma (s + d)
            const double shift = (v.price[it.node + N * t] + gam * v.s[t]) * inv;
#pragma unroll 4
            for (int g = it.a0 + r; g < it.a1; g += R) {
                const size_t e = (size_t)g * T + t;
                const double mc = v.gen_mc[g], pm = v.gen_pmax[g];
                const double p0 = v.P[e];
                const double pn = clampd(p0 - (mc * inv + shift), 0.0, pm);
                v.P[e] = pn;
                acc += pn;
                cost += mc * pn;
            }
        }
        // fixed-order reduction over the R agent lanes that share a timestep (only LDS data crosses these barriers:
        // __syncthreads() would also wait for the acknowledgement of the rows just stored)
        red[tid] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (r == 0 && t < T) {
            double sum = 0.0;
            for (int q = 0; q < R; ++q) sum += red[q * TT + tt];
            v.part_ginj[(size_t)blockIdx.x * T + t] = sum;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // cost: butterfly inside each wave, then the eight waves in order — one barrier instead of one per tree level
    __shared__ double wcost[8];
    for (int d = 32; d > 0; d >>= 1) cost += __shfl_xor(cost, d);
    if ((tid & 63) == 0) wcost[tid >> 6] = cost;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0) {
        double c = 0.0;
        for (int q = 0; q < 8; ++q) c += wcost[q];
        v.part_gcost[blockIdx.x] = c;
    }
}

// End of a generator block of the pair kernels: per-column sums of the R agent lanes and the block's cost, every sum in
// a fixed order (cost: butterfly inside each wave, then the waves in order), ONE barrier — a block lives for a few
// microseconds and a barrier per tree level was a quarter of that.
// LDS_ONLY: the barrier waits for this wave's LDS traffic only. __syncthreads() also drains every global load in
// flight (s_waitcnt vmcnt(0)) — in the streaming blocks those are the NEXT item's rows, i.e. exactly the overlap the
// streaming is for. Only LDS data crosses this barrier.
template <int BS, bool TAIL, bool LDS_ONLY = false, bool PERSIST = false>
__device__ __forceinline__ void gen_pair_sums(const DevView &v, const int blk, const int tid, const int r, const int tt,
                                              double acc0, double acc1, double cost, double (*red)[BS], double *wc, const int ppar = 0)
{
    const int T = v.T, TT = v.genTT2, R = v.genR2;
    TailView tv{};                                           // TAIL: the launch chain carries the iteration's tail
    int par = 0;
    if (TAIL) { tv = *v.tail; par = PERSIST ? ppar : v.st->tail_par; }        // (uniform scalar loads, in flight with the LDS traffic below)
    for (int d = 32; d > 0; d >>= 1) cost += __shfl_xor(cost, d);
    red[0][tid] = acc0; red[1][tid] = acc1;
    if ((tid & 63) == 0) wc[tid >> 6] = cost;
    if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
    if (r == 0 && tt < TT) {
        double s0 = 0.0, s1 = 0.0;
        for (int q = 0; q < R; ++q) { s0 += red[0][q * TT + tt]; s1 += red[1][q * TT + tt]; }
        if (TAIL) {
            acc_add(tv, par, 2 * tt, s0, tv.scaleInj);
            acc_add(tv, par, 2 * tt + 1, s1, tv.scaleInj);
        } else {
            v.part_ginj[(size_t)blk * T + 2 * tt] = s0;
            v.part_ginj[(size_t)blk * T + 2 * tt + 1] = s1;
        }
    }
    if (tid == 0) {
        double c = 0.0;
        for (int q = 0; q < BS / 64; ++q) c += wc[q];
        if (TAIL) acc_add(tv, par, T, c, tv.scaleCost);
        else v.part_gcost[blk] = c;
    }
}

// Copper plate, even T: each thread owns TWO consecutive timesteps of an agent, so every P access is a
// 16-byte-per-lane double2 (the widest coalesced form), half as many load/store instructions per byte.
// returns false when the block found the halted state (nothing stored, nothing added)
template <int BS, bool TAIL, bool CHECK_HALT = false>
__device__ __forceinline__ bool gen_pair_body(const DevView &v, const int blk)
{
    const int halt = CHECK_HALT ? v.st->halt : 0;        // (the load is in flight with the ones below)
    __shared__ double red[2][BS];
    __shared__ double wc[BS / 64];
    // one node (every copper plate of BASELINE.json): the items are equal cuts of the generator list — no table look-up
    // between the block's start and its first row loads
    Item it;
    if (v.genChunk > 0) { it.a0 = blk * v.genChunk; it.a1 = min(v.G, it.a0 + v.genChunk); it.node = 0; }
    else it = v.gen_items[blk];
    const int T = v.T, N = v.N, TT = v.genTT2, R = v.genR2;     // TT = T/2 pair columns
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const double w = v.w_prox, gam = v.gamma;
    const double inv = 1.0 / (w + gam);
    double cost = 0.0, acc0 = 0.0, acc1 = 0.0;
    if (r < R) {
        const int t = 2 * tt;
        const double sh0 = fma(gam, v.s[t], v.price[it.node + N * t]) * inv;
        const double sh1 = fma(gam, v.s[t + 1], v.price[it.node + N * (t + 1)]) * inv;
        double2 *P2 = reinterpret_cast<double2 *>(v.P);
        const size_t half = (size_t)(T >> 1);
        // rows in batches of GU: every load of the batch is issued before the first store (the compiler may not move a
        // load across a store to the same array on its own), so a block keeps GU x 16 B per lane in flight — what lets the
        // generator blocks of the fused launch stream while the storage blocks hold most of the wave slots
        // (the fused launch cuts config2's generators into items of 6 rows per lane: one batch)
        constexpr int GU = BS == 256 ? 6 : 4;
        for (int g0 = it.a0 + r; g0 < it.a1; g0 += GU * R) {
            double2 p0[GU];
            double mc[GU], pm[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int g = g0 + u * R < it.a1 ? g0 + u * R : g0;          // (a valid row; result dropped below)
                mc[u] = v.gen_mc[g]; pm[u] = v.gen_pmax[g];
                p0[u] = P2[(size_t)g * half + tt];
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int g = g0 + u * R;
                if (g >= it.a1) break;
                double2 pn;
                // explicit fma: the row-skipping variant below must round exactly like this sweep
                pn.x = clampd(p0[u].x - fma(mc[u], inv, sh0), 0.0, pm[u]);
                pn.y = clampd(p0[u].y - fma(mc[u], inv, sh1), 0.0, pm[u]);
                P2[(size_t)g * half + tt] = pn;
                acc0 += pn.x; acc1 += pn.y;
                cost = fma(mc[u], pn.x + pn.y, cost);
            }
        }
    }
    if (CHECK_HALT && halt) return false;
    gen_pair_sums<BS, TAIL>(v, blk, tid, r, tt, acc0, acc1, cost, red, wc);
    return true;
}

// Generator blocks of the fused launch (one node, items = equal cuts of <= GU rows per lane): a block STAYS and walks the
// items first, first + stride, ... with the next item's rows already on their way while the current one is computed,
// stored and summed. While the storage blocks hold most of the wave slots only ~200 generator blocks are resident:
// launched one per item each of them spent 3.7 us mostly waiting on its one batch of loads (390 of 1516 items done
// when the storage blocks retired, 2.5 TB/s); streaming, the same 200 blocks keep two batches in flight per lane
// (3.3 TB/s next to the storage blocks, 5.3 TB/s alone) and finish 3 us after the last storage block instead of 7.
// Static assignment: every block ends after its last item; a block always walks the same items in the same order, so
// its sums run over ALL its items in registers and meet once, at the block's end (one partial row per BLOCK: fixed
// order, bitwise reproducible; no LDS traffic or barrier inside the loop). (Tried: items drawn from a counter so that blocks starting late can help
// — a same-address device-scope atomic per item costs more than it balances on eight L2s: 26 vs 21 us.)
// returns false when the block found the halted state (nothing stored, nothing added)
template <int BS, bool TAIL>
__device__ __forceinline__ bool gen_pair_stream(const DevView &v, const int first, const int stride)
{
    constexpr int GU = kGenStreamRows;
    __shared__ double red[2][BS];
    __shared__ double wc[BS / 64];
    const int halt = v.st->halt;                             // (in flight with the loads below)
    const int T = v.T, N = v.N, TT = v.genTT2, R = v.genR2, nI = v.nGenItems, chunk = v.genChunk, G = v.G;
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const bool rowlane = r < R;
    const double gam = v.gamma, inv = 1.0 / (v.w_prox + gam);
    const int t2c = 2 * tt;                                  // one node: its price is entry 0 of every timestep
    const double sh0 = fma(gam, v.s[t2c], v.price[(size_t)N * t2c]) * inv;
    const double sh1 = fma(gam, v.s[t2c + 1], v.price[(size_t)N * (t2c + 1)]) * inv;
    if (first >= nI) return halt == 0;
    double2 *P2 = reinterpret_cast<double2 *>(v.P);
    const size_t half = (size_t)(T >> 1);
    double2 pa[GU], pb[GU];
    double mca[GU], pma[GU], mcb[GU], pmb[GU];
    // Straight-line on purpose. The wait before a use is a count of NEWER operations the in-order counter may leave
    // outstanding, fixed per program point: any path on which the next batch's loads are skipped (no next item, a lane
    // without a row, an early exit from the row loop) makes the compiler assume that path everywhere — the use then
    // waits for the next batch as well and the overlap is gone. So: every lane loads every time (a valid row, dropped
    // if it is not its own), the last item is loaded once more instead of nothing, rows are masked, not skipped.
#define DOPF_GEN_LOAD(item, p, mc, pm)                                                      \
    {                                                                                       \
        const int a0_ = (item) * chunk, a1_ = min(G, a0_ + chunk);                          \
        _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                    \
            const int g_ = (rowlane && a0_ + r + u * R < a1_) ? a0_ + r + u * R : a0_;      \
            const double2 mp_ = v.gen_mp[g_];                                               \
            mc[u] = mp_.x; pm[u] = mp_.y;                                                   \
            p[u] = P2[(size_t)g_ * half + tt];                                              \
        }                                                                                   \
    }
#define DOPF_GEN_WORK(item, p, mc, pm)                                                      \
    {                                                                                       \
        const int a0_ = (item) * chunk, a1_ = min(G, a0_ + chunk);                          \
        _Pragma("unroll") for (int u = 0; u < GU; ++u) {                                    \
            const int g_ = a0_ + r + u * R;                                                 \
            const bool mine = rowlane && g_ < a1_;                                          \
            double2 pn;            /* explicit fma: rounds exactly like gen_pair_body */     \
            pn.x = clampd(p[u].x - fma(mc[u], inv, sh0), 0.0, pm[u]);                       \
            pn.y = clampd(p[u].y - fma(mc[u], inv, sh1), 0.0, pm[u]);                       \
            if (mine) {                                                                     \
                P2[(size_t)g_ * half + tt] = pn;                                            \
                acc0 += pn.x; acc1 += pn.y;                                                 \
                cost = fma(mc[u], pn.x + pn.y, cost);                                       \
            }                                                                               \
        }                                                                                   \
    }
    double acc0 = 0.0, acc1 = 0.0, cost = 0.0;
    int i = first;
    DOPF_GEN_LOAD(i, pa, mca, pma)
    for (;;) {
        const int j = i + stride;
        DOPF_GEN_LOAD(min(j, nI - 1), pb, mcb, pmb)
        if (halt) return false;                              // (uniform) nothing is stored in a halted state
        DOPF_GEN_WORK(i, pa, mca, pma)
        if (j >= nI) break;
        const int k = j + stride;
        DOPF_GEN_LOAD(min(k, nI - 1), pa, mca, pma)
        DOPF_GEN_WORK(j, pb, mcb, pmb)
        if (k >= nI) break;
        i = k;
    }
    gen_pair_sums<BS, TAIL>(v, first, tid, r, tt, acc0, acc1, cost, red, wc);       // row `first` of the partials = this block
#undef DOPF_GEN_LOAD
#undef DOPF_GEN_WORK
    return true;
}

// MODE 0: partial rows for k_reduce; 1: sums into the accumulators, a later launch of the chain (k_sto) carries the tail block;
// 2: no storage launch follows, this launch carries the tail block itself. (Instantiations of their own: the tail's registers
// would cost the common kernel a wave per SIMD, and the chains without a tail keep the code they had.)
template <int MODE>
__global__ __launch_bounds__(512) void k_gen_update_pair(DevView v)
{
    if (MODE == 2 && (int)blockIdx.x == v.nGenItems) { tail_block(v.self); return; }
    if (v.st->halt) return;
    gen_pair_body<512, MODE != 0>(v, blockIdx.x);
}

// Row skipping variant (used when a block sweeps many agents, so that its fixed cost is amortised): in a
// settled dispatch two thirds of the generators sit at 0 or at pmax for ALL timesteps and stay there. A word
// per generator remembers "all zero" / "all at pmax" / "mixed"; an all-zero row stays all zero iff
// mc/(w+gamma) + min_t shift_t >= 0, an all-pmax row stays iff mc/(w+gamma) + max_t shift_t <= 0 (the update
// then clamps every element back onto the same bound), so such a row is neither read nor written — its
// contribution to the sums is 0 or pmax. Results are identical to the full sweep, bit for bit.
template <int BS, bool TAIL>
__device__ __forceinline__ void gen_pair_skip_body(const DevView &v, const int blk)
{
    __shared__ double red[2][BS];
    __shared__ double wc[BS / 64], wlo[BS / 64], whi[BS / 64];
    __shared__ int flg[2][BS];
    const Item it = v.gen_items[blk];
    const int T = v.T, N = v.N, TT = v.genTT2, R = v.genR2;     // TT = T/2 pair columns
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const double w = v.w_prox, gam = v.gamma;
    const double inv = 1.0 / (w + gam);
    double cost = 0.0, acc0 = 0.0, acc1 = 0.0;
    double sh0 = 0.0, sh1 = 0.0;
    if (r < R) {
        const int t = 2 * tt;
        sh0 = fma(gam, v.s[t], v.price[it.node + N * t]) * inv;
        sh1 = fma(gam, v.s[t + 1], v.price[it.node + N * (t + 1)]) * inv;
    }
    // min and max of the shift over the horizon (same for every agent of the item)
    // (min / max are exact: any order gives the same bits)
    double lo = (r == 0 && tt < TT) ? fmin(sh0, sh1) : INFINITY, hi = (r == 0 && tt < TT) ? fmax(sh0, sh1) : -INFINITY;
    for (int d = 32; d > 0; d >>= 1) { lo = fmin(lo, __shfl_xor(lo, d)); hi = fmax(hi, __shfl_xor(hi, d)); }
    if ((tid & 63) == 0) { wlo[tid >> 6] = lo; whi[tid >> 6] = hi; }
    flg[0][tid] = 3; flg[1][tid] = 3;
    __syncthreads();
    double smin = wlo[0], smax = whi[0];
    for (int q = 1; q < BS / 64; ++q) { smin = fmin(smin, wlo[q]); smax = fmax(smax, whi[q]); }

    double2 *P2 = reinterpret_cast<double2 *>(v.P);
    const size_t half = (size_t)(T >> 1);
    const int nPass = (it.a1 - it.a0 + R - 1) / R;
    // the row's state word and parameters are fetched one pass ahead: the decision "sweep or skip" then costs no round
    // trip of its own in front of the row load (a valid row is read when the lane has none: straight-line, dropped)
    double mcN, pmN;
    int sttN;
    {
        const int g0_ = it.a0 + r < it.a1 && r < R ? it.a0 + r : it.a0;
        mcN = v.gen_mc[g0_]; pmN = v.gen_pmax[g0_]; sttN = v.gen_state[g0_];
    }
    for (int p = 0; p < nPass; ++p) {
        const int g = it.a0 + p * R + r;
        const bool on = r < R && g < it.a1;
        bool full = false;
        const double mc = mcN, pm = pmN;
        const int stt = sttN;                               // 0 all zero, 1 all at pmax, 2 mixed
        {
            const int gn_ = g + R < it.a1 && r < R ? g + R : it.a0;
            mcN = v.gen_mc[gn_]; pmN = v.gen_pmax[gn_]; sttN = v.gen_state[gn_];
        }
        if (on) {
            // fma(mc, inv, .) is monotone in its addend, so its extremes over t are at smin / smax
            if (stt == 0 && fma(mc, inv, smin) >= 0.0) {
                // stays all zero: nothing to read, write or add
            } else if (stt == 1 && fma(mc, inv, smax) <= 0.0) {
                acc0 += pm; acc1 += pm; cost = fma(mc, pm + pm, cost);      // stays all at pmax
            } else {
                full = true;
                const size_t e = (size_t)g * half + tt;
                const double2 p0 = P2[e];
                double2 pn;
                pn.x = clampd(p0.x - fma(mc, inv, sh0), 0.0, pm);
                pn.y = clampd(p0.y - fma(mc, inv, sh1), 0.0, pm);
                P2[e] = pn;
                acc0 += pn.x; acc1 += pn.y;
                cost = fma(mc, pn.x + pn.y, cost);
                const int bits = ((pn.x == 0.0 && pn.y == 0.0) ? 1 : 0) | ((pn.x == pm && pn.y == pm) ? 2 : 0);
                if (bits != 3) atomicAnd(&flg[p & 1][r], bits);
            }
        }
        // only LDS data (the flags) crosses this barrier: __syncthreads() would also wait for this pass's row stores to be
        // acknowledged (s_waitcnt vmcnt(0)) before the next pass may issue its loads
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (full && tt == 0) {
            const int bits = flg[p & 1][r];
            v.gen_state[g] = (bits & 1) ? 0 : ((bits & 2) ? 1 : 2);
        }
        if (tt == 0 && r < R) flg[p & 1][r] = 3;              // free again two passes later
    }
    gen_pair_sums<BS, TAIL>(v, blk, tid, r, tt, acc0, acc1, cost, red, wc);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_gen_update_pair_skip(DevView v)
{
    if (MODE == 2 && (int)blockIdx.x == v.nGenItems) { tail_block(v.self); return; }
    if (v.st->halt) return;
    gen_pair_skip_body<512, MODE != 0>(v, blockIdx.x);
}

void launch_gen_update(const DevView &v, hipStream_t s)
{
    if (v.nGenItems == 0) return;
    if (v.L > 0) hipLaunchKernelGGL(k_gen_update<true>, dim3(v.nGenItems), dim3(512), 0, s, v);
    else if (v.genTT2 > 0 && v.tail && v.nStoItems == 0) {
        if (v.genSkip) hipLaunchKernelGGL(k_gen_update_pair_skip<2>, dim3(v.nGenItems + 1), dim3(512), 0, s, v);
        else hipLaunchKernelGGL(k_gen_update_pair<2>, dim3(v.nGenItems + 1), dim3(512), 0, s, v);
    } else if (v.genTT2 > 0 && v.tail) {
        if (v.genSkip) hipLaunchKernelGGL(k_gen_update_pair_skip<1>, dim3(v.nGenItems), dim3(512), 0, s, v);
        else hipLaunchKernelGGL(k_gen_update_pair<1>, dim3(v.nGenItems), dim3(512), 0, s, v);
    }
    else if (v.genTT2 > 0 && v.genSkip) hipLaunchKernelGGL(k_gen_update_pair_skip<0>, dim3(v.nGenItems), dim3(512), 0, s, v);
    else if (v.genTT2 > 0) hipLaunchKernelGGL(k_gen_update_pair<0>, dim3(v.nGenItems), dim3(512), 0, s, v);
    else hipLaunchKernelGGL(k_gen_update<false>, dim3(v.nGenItems), dim3(512), 0, s, v);
}

// ------------------------------------------------------------------------------------------------
// storages
// ------------------------------------------------------------------------------------------------
//
// A group of LPS lanes owns one storage; lane li owns the NCH CONSECUTIVE timesteps li*NCH .. li*NCH+NCH-1.
// One "scan" evaluates, for a trial price nu, the whole forward recursion
//     F_t = clamp(F_{t-1} + x_t(nu), 0, emax),  F_0 = 0
// as an associative scan of clamp-add maps e -> clamp(e + A, LO, HI): NCH maps are composed inside the
// lane, the lane composites are scanned across the group with DPP row shifts / row broadcasts (no LDS
// traffic), and the prefix is applied back inside the lane.

struct Map3 {
    double A, LO, HI;
};

__device__ __forceinline__ Map3 compose(const Map3 &p, const Map3 &c)   // p first, then c
{
    Map3 r;
    r.A = p.A + c.A;
    r.LO = clampd(p.LO + c.A, c.LO, c.HI);
    r.HI = clampd(p.HI + c.A, c.LO, c.HI);
    return r;
}

// lanes whose DPP source is out of range (or whose row is masked off) keep `oldv`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dppd(double oldv, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(oldv), __double2loint(src), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(oldv), __double2hiint(src), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// DPP read whose lanes without a source get 0 — one move per dword (dppd with old == src is a copy + a move per dword). For
// callers that discard what such lanes read. (A DPP read must be made by ALL lanes: a switched-off source lane reads as 0 here.)
template <int CTRL>
__device__ __forceinline__ double dppz(double src)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// v_min_f64 / v_max_f64 as they are: the compiler puts a canonicalising `v_max_f64 x, x` in front of every fmin / fmax whose
// operand was assembled from the integer halves a DPP move (or a 64-bit select) delivers. No NaN enters the chains these are used
// in (finite values and +-inf, combined by min / max only).
__device__ __forceinline__ double lz_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double lz_minabs(double a, double b)          // min(|a|, |b|)
{
    double r;
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double lz_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double lz_clamp(double v, double lo, double hi) { return lz_min(lz_max(v, lo), hi); }      // = clampd

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Map3 dppm(const Map3 &m)
{
    Map3 p;
    p.A = dppd<CTRL, ROW_MASK>(m.A, m.A);
    p.LO = dppd<CTRL, ROW_MASK>(m.LO, m.LO);
    p.HI = dppd<CTRL, ROW_MASK>(m.HI, m.HI);
    return p;
}

// inclusive scan of the lane composites over the lanes of each group
template <int LPS>
__device__ __forceinline__ void scan_maps(Map3 &m, int lane)
{
    const int r = lane & ((LPS < 16 ? LPS : 16) - 1);
    { const Map3 p = dppm<0x111, 0xF>(m); if (r >= 1) m = compose(p, m); }                     // row_shr:1
    if (LPS >= 4) { const Map3 p = dppm<0x112, 0xF>(m); if (r >= 2) m = compose(p, m); }       // row_shr:2
    if (LPS >= 8) { const Map3 p = dppm<0x114, 0xF>(m); if (r >= 4) m = compose(p, m); }       // row_shr:4
    if (LPS >= 16) { const Map3 p = dppm<0x118, 0xF>(m); if (r >= 8) m = compose(p, m); }      // row_shr:8
    if (LPS >= 32) { const Map3 p = dppm<0x142, 0xA>(m); if (lane & 16) m = compose(p, m); }   // row_bcast:15
    if (LPS >= 64) { const Map3 p = dppm<0x143, 0xC>(m); if (lane & 32) m = compose(p, m); }   // row_bcast:31
}

// value of the previous lane of the group (garbage for the group's first lane: caller masks it)
template <int LPS>
__device__ __forceinline__ double prev_lane(double x)
{
    if (LPS <= 16) return dppz<0x111>(x);              // row_shr:1
    return dppz<0x138>(x);                             // wave_shr:1
}

template <int LPS>
__device__ __forceinline__ double group_sum(double x)
{
    if (LPS >= 2) x += dppd<0xB1, 0xF>(x, x);          // quad_perm [1,0,3,2]
    if (LPS >= 4) x += dppd<0x4E, 0xF>(x, x);          // quad_perm [2,3,0,1]
    if (LPS >= 8) x += dppd<0x141, 0xF>(x, x);         // row_half_mirror
    if (LPS >= 16) x += dppd<0x140, 0xF>(x, x);        // row_mirror
    if (LPS >= 32) x += __shfl_xor(x, 16);
    if (LPS >= 64) x += __shfl_xor(x, 32);
    return x;
}

template <int LPS>
__device__ __forceinline__ double group_min(double x)
{
    if (LPS >= 2) x = fmin(x, dppd<0xB1, 0xF>(x, x));
    if (LPS >= 4) x = fmin(x, dppd<0x4E, 0xF>(x, x));
    if (LPS >= 8) x = fmin(x, dppd<0x141, 0xF>(x, x));
    if (LPS >= 16) x = fmin(x, dppd<0x140, 0xF>(x, x));
    if (LPS >= 32) x = fmin(x, __shfl_xor(x, 16));
    if (LPS >= 64) x = fmin(x, __shfl_xor(x, 32));
    return x;
}

// inclusive prefix sum over the lanes of each group
template <int LPS>
__device__ __forceinline__ double scan_sum(double x, int lane)
{
    const int r = lane & ((LPS < 16 ? LPS : 16) - 1);
    { const double p = dppd<0x111, 0xF>(0.0, x); if (r >= 1) x += p; }
    if (LPS >= 4) { const double p = dppd<0x112, 0xF>(0.0, x); if (r >= 2) x += p; }
    if (LPS >= 8) { const double p = dppd<0x114, 0xF>(0.0, x); if (r >= 4) x += p; }
    if (LPS >= 16) { const double p = dppd<0x118, 0xF>(0.0, x); if (r >= 8) x += p; }
    if (LPS >= 32) { const double p = dppd<0x142, 0xA>(0.0, x); if (lane & 16) x += p; }
    if (LPS >= 64) { const double p = dppd<0x143, 0xC>(0.0, x); if (lane & 32) x += p; }
    return x;
}

template <int LPS>
__device__ __forceinline__ unsigned long long group_bits(bool pred, int gbase)
{
    const unsigned long long b = __ballot(pred);
    if (LPS == 64) return b;
    return (b >> gbase) & ((1ull << LPS) - 1ull);
}

// argmin over [0,pm]^2 of the strictly convex quadratic with gradient (a D - b C - rD, a C - b D - rC),
// a > b >= 0, given ia = 1/a, idet = 1/(a^2 - b^2); sg = d(C - D)/d(nu) on the active piece
// (rD falls, rC rises with nu at unit rate): 2/(a+b) with both free, 1/a with one, 0 with none.
// With C "free" D solves a 1-D convex problem (clamp of its stationary point); if the implied C leaves
// the box, C sits on that bound (monotone contraction argument).
__device__ __forceinline__ void box2(double a, double b, double ia, double idet, double s2, double rD,
                                     double rC, double pm, double &D, double &C, double &sg)
{
    const double Df = clampd((a * rD + b * rC) * idet, 0.0, pm);
    const double Cf = (rC + b * Df) * ia;
    C = clampd(Cf, 0.0, pm);
    // D given that C: with C on a bound the 1-D problem's clamp (rD/a resp. (rD + b pm)/a); with Cf inside the box it is Df
    // again — interior: a^2 D = a rD + b rC + b^2 Df = a^2 Df; Df on a bound: the unclamped value lies beyond the same bound
    D = clampd((rD + b * C) * ia, 0.0, pm);
    const bool fD = D > 0.0 && D < pm, fC = C > 0.0 && C < pm;
    sg = (fD && fC) ? s2 : ((fD || fC) ? ia : 0.0);
}

// breakpoint table of Psi_{n,t} as one lane-timestep sees it
struct TabRef {
    const double *beta, *psi, *slope;
    int m;
    double psi0;
};

__device__ __forceinline__ TabRef tab_ref(const DevView &v, int node, int t)
{
    const size_t at = (size_t)node + (size_t)v.N * t;
    TabRef r;
    r.beta = v.tb_beta + at * v.M2;
    r.psi = v.tb_psi + at * v.M2;
    r.slope = v.tb_slope + at * (v.M2 + 1);
    r.m = v.tb_m[at];
    r.psi0 = v.tb_psi0[at];
    return r;
}

// Psi_{n,t}(dl) from the table (first kink >= dl, then the piece left of it)
__device__ __forceinline__ double tab_psi_at(const TabRef &tb, double dl)
{
    if (tb.m == 0) return tb.psi0 + tb.slope[0] * dl;
    int lo = 0, hi = tb.m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tb.beta[mid] >= dl) hi = mid; else lo = mid + 1;
    }
    const int a = lo < tb.m ? lo : tb.m - 1;
    return tb.psi[a] + tb.slope[lo] * (dl - tb.beta[a]);
}

// (D, C)(nu) of one storage timestep with lines: find the piece of Psi the solution lies on — first kink
// whose residual r(beta) = beta - (D(z) - C(z) - q0), z = Psi(beta) + nu, is >= 0 (r is increasing) — then
// the 2x2 box QP with that piece's slope. `hint` remembers the piece between calls: successive prices are
// close, so two probes around the hint usually replace the 9 dependent table reads of a full bisection.
__device__ __forceinline__ void eval_lines(const TabRef &tb, int &hint, double w, double iw, double mc, double pm,
                                           double D0, double C0, double nu, double &dd, double &cc, double &s1,
                                           double &psi_cur)
{
    const double q0 = D0 - C0;
    double ab = 0.0, ap = tb.psi0, kap = tb.slope[0];
    const int m = tb.m;
    if (m > 0) {
        auto rneg = [&](int idx) -> bool {
            const double z = tb.psi[idx] + nu;
            const double Dz = clampd(D0 - (mc + z) * iw, 0.0, pm), Cz = clampd(C0 - (mc - z) * iw, 0.0, pm);
            return tb.beta[idx] - (Dz - Cz - q0) < 0.0;
        };
        int l2 = hint < 0 ? 0 : (hint > m ? m : hint);
        const bool okLo = l2 == 0 || rneg(l2 - 1);
        const bool okHi = l2 == m || !rneg(l2);
        if (!(okLo && okHi)) {
            int lo = okLo ? l2 + 1 : 0, hi = okLo ? m : l2 - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (!rneg(mid)) hi = mid; else lo = mid + 1;
            }
            l2 = lo;
        }
        hint = l2;
        const int a = l2 < m ? l2 : m - 1;
        ab = tb.beta[a]; ap = tb.psi[a]; kap = tb.slope[l2];
    }
    const double theta = ap - kap * (ab + q0);
    const double a = w + kap;
    double ia, idet, s2;
    lin_coef(w, iw, kap, ia, idet, s2);
    box2(a, kap, ia, idet, s2, w * D0 - mc - theta - nu, w * C0 - mc + theta + nu, pm, dd, cc, s1);
    psi_cur = theta + kap * (dd - cc);       // Psi at the step's current net injection
}

// The same two as functions of their own, for the active-set body: there a (node, timestep) with a non-empty table is
// the rare case (none in the settled state), and inlined their registers cost the common path a wave per SIMD.
struct EvalOut { double dd, cc, s1, pc; int hint; };
__device__ DOPF_CALL_ATTR EvalOut eval_lines_call(const DevView *self, int node, int t, int hint, double mc, double pm,
                                                             double D0, double C0, double nu)
{
    const DevView &v = *self;
    const TabRef tb = tab_ref(v, node, t);
    EvalOut o;
    o.hint = hint;
    eval_lines(tb, o.hint, v.w_prox, 1.0 / v.w_prox, mc, pm, D0, C0, nu, o.dd, o.cc, o.s1, o.pc);
    return o;
}
__device__ DOPF_CALL_ATTR double tab_psi_call(const DevView *self, int node, int t, double dl)
{
    return tab_psi_at(tab_ref(*self, node, t), dl);
}

struct StoAgent {
    double mc, pm, em;
};

// `item_fail`: number of storages of this item the warm start left over (block-uniform); < 0 = read it
// FULLT: the horizon fills the lane group exactly (T == LPS * NCH: 24 = 8 x 3, 48 = 16 x 3, 96 = 32 x 3): T is then a
// compile-time constant and every "is this step inside the horizon" test folds away
template <int LPS, int NCH, bool LINES, bool TAIL = false, bool FULLT = false, bool PERSIST = false>
__device__ __forceinline__ void sto_cold_body(const DevView &v, const int blk, int item_fail, const int ppar = 0)
{
    constexpr int NG = 256 / LPS;
    __shared__ double red[NG * LPS * NCH];
    __shared__ double redc[256];
    const int tid = threadIdx.x, lane = tid & 63, li = tid & (LPS - 1), grp = tid / LPS;
    const int gbase = lane & ~(LPS - 1);
    const Item it = v.sto_items[blk];
    const int T = FULLT ? LPS * NCH : v.T, N = v.N;
    if (item_fail < 0) item_fail = v.item_fail[blk];
    if (v.use_warm && item_fail == 0) {      // the warm start solved this whole item
        if (TAIL) return;                    // (nothing to add)
        if (LINES) for (int t = tid; t < T; t += 256) v.part_T[(size_t)t * v.rowsT + it.row] = 0.0;
        else for (int t = tid; t < T; t += 256) v.part_sinj[(size_t)blk * T + t] = 0.0;
        if (tid == 0) v.part_scost[blk] = 0.0;
        return;
    }
    const double w = v.w_prox, gam = v.gamma;
    const double a0 = w + gam, ia0 = 1.0 / a0, idet0 = 1.0 / (a0 * a0 - gam * gam), s20 = 2.0 / (a0 + gam);
    const int tbase = li * NCH;

    double th0[NCH], accQ[NCH];
    double accCost = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = tbase + c;
        accQ[c] = 0.0;
        th0[c] = (!LINES && t < T) ? (PERSIST ? p_ld(v.price + it.node + N * t) + gam * p_ld(v.s + t) : v.price[it.node + N * t] + gam * v.s[t]) : 0.0;
    }
    // with lines: a (node, timestep) whose table is empty — no kink of Psi inside the node's window, the usual case —
    // is the copper-plate closed form with (Psi(0), slope) in place of (theta, gamma): cached here, no table reads
    double lp0[NCH], lkap[NCH];
    bool lin[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = tbase + c;
        lin[c] = false; lp0[c] = 0.0; lkap[c] = 0.0;
        if (LINES && t < T) {
            const size_t at = (size_t)it.node + (size_t)N * t;
            // (all three loaded at once: the size word in front of the other two would be a round trip of its own)
            const int m_ = v.tb_m[at];
            const double p0_ = v.tb_psi0[at], k_ = v.tb_slope[at * (v.M2 + 1)];
            if (m_ == 0) {
                lin[c] = true;
                lp0[c] = p0_;
                lkap[c] = k_;
            }
        }
    }
    unsigned long long fails = 0;
#ifdef DOPF_STATS
    unsigned long long st_scans = 0, st_loops = 0, st_events = 0;
#endif
    const int nRep = (it.a1 - it.a0 + NG - 1) / NG;

    for (int rep = 0; rep < nRep; ++rep) {
        const int s = it.a0 + rep * NG + grp;
        const bool live = s < it.a1 && (!v.use_warm || v.sto_fail[s] != 0);
        StoAgent ag;
        ag.mc = live ? v.sto_mc[s] : 0.0;
        ag.pm = live ? v.sto_pmax[s] : 0.0;
        ag.em = live ? v.sto_emax[s] : 0.0;
        // rD0/rC0: the nu-independent part of the two gradient offsets (copper plate); D0/C0 otherwise
        double D0[NCH], C0[NCH], nuf[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            const bool ok = live && t < T;
            const size_t e = (size_t)s * T + t;
            D0[c] = ok ? v.D[e] : 0.0;
            C0[c] = ok ? v.C[e] : 0.0;
            nuf[c] = 0.0;
        }
        const double tol = 1e-11 * (1.0 + ag.em);

        // x_t(nu): net charge of timestep (li, c) at price nu
        int hint[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) hint[c] = 0;
        const double iw = 1.0 / w;
        auto eval = [&](int c, double nu, double &dd, double &cc, double &s1, double &pc) {
            if (!LINES) {
                const double q0 = D0[c] - C0[c], theta = th0[c] - gam * q0;
                box2(a0, gam, ia0, idet0, s20, w * D0[c] - ag.mc - theta - nu, w * C0[c] - ag.mc + theta + nu,
                     ag.pm, dd, cc, s1);
                pc = theta + gam * (dd - cc);
            } else if (lin[c]) {
                const double q0 = D0[c] - C0[c], theta = lp0[c] - lkap[c] * q0;
                double lia, lidet, ls2;
                lin_coef(w, iw, lkap[c], lia, lidet, ls2);
                box2(w + lkap[c], lkap[c], lia, lidet, ls2, w * D0[c] - ag.mc - theta - nu, w * C0[c] - ag.mc + theta + nu,
                     ag.pm, dd, cc, s1);
                pc = theta + lkap[c] * (dd - cc);
            } else {
                const TabRef tb = tab_ref(v, it.node, tbase + c);
                eval_lines(tb, hint[c], w, iw, ag.mc, ag.pm, D0[c], C0[c], nu, dd, cc, s1, pc);
            }
        };

        // ---- price-threshold recursion, backwards over constant-price segments -------------------
        double nu = 0.0;
        int k = live ? T - 1 : -1;      // timesteps 0..k are still open
        int mode = 0;                   // 0: classify at nu, 1: root search for timestep vv
        int vv = -1, rit = 0;
        double target = 0.0, lo = -INFINITY, hi = INFINITY, step = 1.0;

        while (__any(k >= 0)) {
            const bool active = k >= 0;
#ifdef DOPF_STATS
            if (li == 0 && active) ++st_scans;
            if (lane == 0) ++st_loops;
#endif
            // -- one scan at price nu
            double x[NCH], sg[NCH], Sv[NCH], psc[NCH];
            Map3 loc;
            loc.A = 0.0; loc.LO = -INFINITY; loc.HI = INFINITY;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int t = tbase + c;
                double dd = 0.0, cc = 0.0, s1 = 0.0, pc = 0.0;
                if (t <= k && t < T) eval(c, nu, dd, cc, s1, pc);
                x[c] = cc - dd;
                sg[c] = s1;
                psc[c] = pc;
                loc.A += x[c];
                loc.LO = clampd(loc.LO + x[c], 0.0, ag.em);
                loc.HI = clampd(loc.HI + x[c], 0.0, ag.em);
            }
            Map3 inc = loc;
            scan_maps<LPS>(inc, lane);
            Map3 ex;
            ex.A = prev_lane<LPS>(inc.A); ex.LO = prev_lane<LPS>(inc.LO); ex.HI = prev_lane<LPS>(inc.HI);
            double e = li == 0 ? 0.0 : clampd(ex.A, ex.LO, ex.HI);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                Sv[c] = e + x[c];
                e = clampd(Sv[c], 0.0, ag.em);
            }

            // -- helpers on the scan just made: unclamped level at a timestep, its slope in nu, and
            //    the way out of a flat piece
            auto level_at = [&](int idx) -> double {
                const int lown = idx / NCH, cown = idx - lown * NCH;
                double sel = 0.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (c == cown) sel = Sv[c];
                return __shfl(sel, gbase + lown);
            };
            // dS_idx/dnu = sum of dx_t/dnu over the run of unclamped steps that ends at idx
            int jlast = -1;      // last clamped step before idx (set by slope_at)
            auto slope_at = [&](int idx) -> double {
                jlast = -1;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    const unsigned long long bts = group_bits<LPS>(t < idx && (Sv[c] <= 0.0 || Sv[c] >= ag.em), gbase);
                    if (bts) { const int j = (63 - __clzll(bts)) * NCH + c; jlast = j > jlast ? j : jlast; }
                }
                double part = 0.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    if (t > jlast && t <= idx) part += sg[c];
                }
                return group_sum<LPS>(part);
            };
            // S_idx is flat at nu: jump just past the nearest kink of the x_t in its run (jlast, idx] in direction dir.
            // Any trial point is legitimate (the bracket keeps the search safe); this one has the right scale.
            auto flat_jump = [&](int idx, double dir) -> double {
                // every step of a flat run has D and C on bounds, so its net injection (hence Psi) does not
                // move with nu and the four prices at which D or C would leave a bound are closed form
                double best = INFINITY;                   // distance to the nearest such price ahead
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (tbase + c > idx || tbase + c <= jlast) continue;
                    const double bD = w * D0[c] - ag.mc - psc[c], bC = ag.mc - w * C0[c] - psc[c], wp = w * ag.pm;
                    const double cand[4] = {bD, bD - wp, bC, bC + wp};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double d = (cand[q] - nu) * dir;
                        if (d > 0.0) best = fmin(best, d);
                    }
                }
                best = group_min<LPS>(best);
                if (best < INFINITY) return nu + dir * (best + 1e-9 * (1.0 + fabs(nu) + best));
                const double tr = nu + dir * step;
                step *= 4.0;
                return tr;
            };

            bool classify = active && mode == 0;
            if (active && mode == 1) {
                const double res = level_at(vv) - target;
                if (res < 0.0) lo = nu; else hi = nu;
                bool conv = fabs(res) <= 1e-12 * (1.0 + ag.em) || rit >= v.rootCap;
                double trial = nu;
                if (!conv) {
                    const double sl = slope_at(vv);
                    const bool both = lo > -INFINITY && hi < INFINITY;
                    if (sl > 0.0) {
                        double r = __builtin_amdgcn_rcp(sl);
                        r = r * (2.0 - sl * r);
                        trial = nu - res * r;
                    } else {
                        trial = both ? 0.5 * (lo + hi) : flat_jump(vv, res < 0.0 ? 1.0 : -1.0);
                    }
                    const bool forceBis = both && rit >= 6 && (rit & 1);
                    if (!(trial > lo && trial < hi) || forceBis) {
                        if (both) trial = 0.5 * (lo + hi);
                        else { trial = (res < 0.0) ? nu + step : nu - step; step *= 4.0; }
                    }
                    if (!(trial > lo && trial < hi)) conv = true;   // bracket is two adjacent doubles
                }
                if (conv) {
                    if (rit >= v.rootCap && fabs(res) > 1e-7 * (1.0 + ag.em) && li == 0) ++fails;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        if (tbase + c == vv) nuf[c] = nu;
                    k = vv - 1;
                    mode = 0;
                    classify = k >= 0;
                } else {
                    nu = trial;
                    ++rit;
                }
            }
            if (classify) {
                // largest open timestep whose unclamped level leaves [0, em] at this price
                int vnew = -1;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    const unsigned long long bts = group_bits<LPS>(t <= k && t < T && (Sv[c] < -tol || Sv[c] > ag.em + tol), gbase);
                    if (bts) { const int j = (63 - __clzll(bts)) * NCH + c; vnew = j > vnew ? j : vnew; }
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    if (t > vnew && t <= k) nuf[c] = nu;
                }
                if (vnew < 0) {
                    k = -1;
                } else {
                    const double sv = level_at(vnew), sl = slope_at(vnew);
                    vv = vnew;
                    target = sv < 0.0 ? 0.0 : ag.em;
                    const double res = sv - target;
                    lo = -INFINITY; hi = INFINITY;
                    if (res < 0.0) lo = nu; else hi = nu;
                    step = 1.0 + fabs(nu);
                    if (sl > 0.0) {
                        double r = __builtin_amdgcn_rcp(sl);
                        r = r * (2.0 - sl * r);
                        nu -= res * r;
                    } else {
                        nu = flat_jump(vnew, res < 0.0 ? 1.0 : -1.0);
                    }
                    mode = 1;
                    rit = 0;
#ifdef DOPF_STATS
                    if (li == 0) ++st_events;
#endif
                }
            }
        }

        // ---- final (D, C) at each timestep's price, level E = cumsum(C - D), outputs, partial sums ------
        double Dn[NCH], Cn[NCH], run = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double s1;
            Dn[c] = 0.0; Cn[c] = 0.0;
            double pcx;
            if (live && tbase + c < T) eval(c, nuf[c], Dn[c], Cn[c], s1, pcx);
            run += Cn[c] - Dn[c];
        }
        if (live && li == 0) v.nu_valid[s] = 1;
        const double incl = scan_sum<LPS>(run, lane);
        const double incl_prev = prev_lane<LPS>(incl);     // DPP: every lane must execute it (no ?: around it)
        double ev = li == 0 ? 0.0 : incl_prev;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            ev += Cn[c] - Dn[c];
            if (live && tbase + c < T) {
                const size_t e = (size_t)s * T + (tbase + c);
                v.D[e] = Dn[c];
                v.C[e] = Cn[c];
                // (the level is not stored: nothing on the path reads it back — a solve takes its contacts from cumsum(C - D) of
                // the rows it loads anyway — and dopf_get_primal rebuilds it on request; 8T bytes per storage and iteration less)
                if (LINES && (v.keepDeltas || v.walk_any[tbase + c])) v.dltS[e] = (Dn[c] - Cn[c]) - (D0[c] - C0[c]);
                v.nu_prev[e] = LINES ? nuf[c] : nuf[c] + (th0[c] - gam * (D0[c] - C0[c]));      // (nu + theta: see the active-set body)
                accQ[c] += Dn[c] - Cn[c];
                accCost += ag.mc * (Dn[c] + Cn[c]);
            }
        }
    }

    // fixed-order block reduction of the per-timestep sums over the NG groups
#pragma unroll
    for (int c = 0; c < NCH; ++c) red[(grp * LPS + li) * NCH + c] = accQ[c];
    redc[tid] = accCost;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            if (t < T) {
                double sum = 0.0;
                for (int g2 = 0; g2 < NG; ++g2) sum += red[(g2 * LPS + li) * NCH + c];
                // (tail in the launch: this thread wrote the active-set body's sum of slot t itself, a moment ago)
                if (TAIL) { const TailView tv = *v.tail; acc_add(tv, PERSIST ? ppar : v.st->tail_par, t, sum + v.part_sinj_w[(size_t)blk * T + t], tv.scaleInj); }
                else if (LINES) v.part_T[(size_t)t * v.rowsT + it.row] = sum;
                else v.part_sinj[(size_t)blk * T + t] = sum;
            }
        }
    }
    for (int sft = 128; sft > 0; sft >>= 1) {
        if (tid < sft) redc[tid] += redc[tid + sft];
        __syncthreads();
    }
    if (tid == 0) {
        if (TAIL) { const TailView tv = *v.tail; acc_add(tv, PERSIST ? ppar : v.st->tail_par, T, redc[0] + v.part_scost_w[blk], tv.scaleCost); }
        else v.part_scost[blk] = redc[0];
    }
    if (fails) atomicAdd(&v.st->solver_fail, fails);
#ifdef DOPF_STATS
    if (st_scans) atomicAdd(&v.st->dbg_scans, st_scans);
    if (st_loops) atomicAdd(&v.st->dbg_wave_loops, st_loops);
    if (st_events) atomicAdd(&v.st->dbg_events, st_events);
#endif
}

template <int LPS, int NCH, bool LINES>
__global__ __launch_bounds__(256, 2) void k_sto_update(DevView v)
{
    if (v.st->halt) return;
    sto_cold_body<LPS, NCH, LINES>(v, blockIdx.x, -1);
}

// ------------------------------------------------------------------------------------------------
// storages, active-set solve: all constant-price segments at once, contact set repaired between rounds
// ------------------------------------------------------------------------------------------------
//
// The storage QP in dual form is max_nu sum_t phi_t(nu_t) - emax sum_t max(0, nu_{t+1} - nu_t): the price of
// stored energy nu is piecewise constant in time and may only fall going forward where the level sits at 0
// ("empty contact") and only rise where it sits at emax ("full contact"). For a GUESS of the contact set
// every segment between two contacts has one price, fixed by sum_{t in seg} x_t(nu) = level change of the
// segment; all segments are solved at once (segmented, bracketed Newton on the piecewise-linear sums: a
// segmented DPP scan supplies sum x and sum dx/dnu per segment, LDS hands the new price back to the segment's
// timesteps; a segment whose steps all sit on corners of their (D, C) box jumps just past the nearest kink).
// Then the KKT conditions of the QP are CHECKED (levels within [0, emax]; price jumps at the contacts have
// the right sign, with whole intervals of multipliers for segments that do not move with their price;
// nu = 0 on the open last segment). The QP is strictly convex in (D, C): a point that passes is THE
// minimiser, whatever produced the guess. A guess that fails is repaired the primal-dual-active-set way —
// every step whose level left the band becomes a contact at the bound it crossed, every contact whose price
// jump has the wrong sign is released — and the solve is repeated. Round 0 starts from the previous
// iteration's contact set (read off the previous level trajectory: no extra state) and prices (8T bytes of
// solver state per storage): near convergence it certifies at once (the "warm start"); from the zero state,
// or after the contact structure moved, a handful of rounds do. Cost per round does not depend on the number
// of contacts. Storages that are not certified within the round budget are left untouched and flagged for
// the scan kernel (the safeguarded, exact fallback).

// inclusive segmented scan of (a, b) over the lanes of each group; f = "a segment starts in this lane"
template <int LPS>
__device__ __forceinline__ void seg_scan2(int &f, double &a, double &b, int lane)
{
    const int r = lane & ((LPS < 16 ? LPS : 16) - 1);
#define DOPF_SEG_STEP(CTRL, RM, COND)                                                        \
    {                                                                                        \
        const int pf = __builtin_amdgcn_update_dpp(f, f, CTRL, RM, 0xF, false);              \
        const double pa = (RM) == 0xF ? dppz<CTRL>(a) : dppd<CTRL, RM>(a, a), pb = (RM) == 0xF ? dppz<CTRL>(b) : dppd<CTRL, RM>(b, b); \
        if (COND) { if (!f) { a += pa; b += pb; } f |= pf; }                                 \
    }
    DOPF_SEG_STEP(0x111, 0xF, r >= 1)
    if (LPS >= 4) DOPF_SEG_STEP(0x112, 0xF, r >= 2)
    if (LPS >= 8) DOPF_SEG_STEP(0x114, 0xF, r >= 4)
    if (LPS >= 16) DOPF_SEG_STEP(0x118, 0xF, r >= 8)
    if (LPS >= 32) DOPF_SEG_STEP(0x142, 0xA, lane & 16)
    if (LPS >= 64) DOPF_SEG_STEP(0x143, 0xC, lane & 32)
#undef DOPF_SEG_STEP
}

// same, with (max, min) instead of (+, +): a <- max over the segment so far, b <- min
template <int LPS>
__device__ __forceinline__ void seg_scan_maxmin(int &f, double &a, double &b, int lane)
{
    const int r = lane & ((LPS < 16 ? LPS : 16) - 1);
#define DOPF_SEG_STEP(CTRL, RM, COND)                                                        \
    {                                                                                        \
        const int pf = __builtin_amdgcn_update_dpp(f, f, CTRL, RM, 0xF, false);              \
        const double pa = (RM) == 0xF ? dppz<CTRL>(a) : dppd<CTRL, RM>(a, a), pb = (RM) == 0xF ? dppz<CTRL>(b) : dppd<CTRL, RM>(b, b); \
        if (COND) { if (!f) { a = lz_max(a, pa); b = lz_min(b, pb); } f |= pf; }             \
    }
    DOPF_SEG_STEP(0x111, 0xF, r >= 1)
    if (LPS >= 4) DOPF_SEG_STEP(0x112, 0xF, r >= 2)
    if (LPS >= 8) DOPF_SEG_STEP(0x114, 0xF, r >= 4)
    if (LPS >= 16) DOPF_SEG_STEP(0x118, 0xF, r >= 8)
    if (LPS >= 32) DOPF_SEG_STEP(0x142, 0xA, lane & 16)
    if (LPS >= 64) DOPF_SEG_STEP(0x143, 0xC, lane & 32)
#undef DOPF_SEG_STEP
}

template <int LPS>
__device__ __forceinline__ double group_max(double x)
{
    if (LPS >= 2) x = lz_max(x, dppz<0xB1>(x));
    if (LPS >= 4) x = lz_max(x, dppz<0x4E>(x));
    if (LPS >= 8) x = lz_max(x, dppz<0x141>(x));
    if (LPS >= 16) x = lz_max(x, dppz<0x140>(x));
    if (LPS >= 32) x = fmax(x, __shfl_xor(x, 16));
    if (LPS >= 64) x = fmax(x, __shfl_xor(x, 32));
    return x;
}

// suffix (right-to-left) inclusive scan of clamp maps x -> clamp(x, lo, hi) over the lanes of each group:
// lane l ends up with M_l o M_{l+1} o ... o M_last (the right-most map is applied first)
template <int LPS>
__device__ __forceinline__ void scan_clamps_rev(double &lo, double &hi, int lane)
{
    constexpr int RL = LPS < 16 ? LPS : 16;
    const int r = lane & (RL - 1);
#define DOPF_REV_STEP(CTRL, D)                                                       \
    {                                                                                \
        const double glo = dppz<CTRL>(lo), ghi = dppz<CTRL>(hi);                     \
        if (r + D < RL) { const double nlo = lz_clamp(glo, lo, hi), nhi = lz_clamp(ghi, lo, hi); lo = nlo; hi = nhi; } \
    }
    DOPF_REV_STEP(0x101, 1)                     // row_shl:1
    if (LPS >= 4) DOPF_REV_STEP(0x102, 2)
    if (LPS >= 8) DOPF_REV_STEP(0x104, 4)
    if (LPS >= 16) DOPF_REV_STEP(0x108, 8)
#undef DOPF_REV_STEP
    if (LPS >= 32) {                            // rows 0, 2 take the whole of the next row (its lane 0)
        const double glo = __shfl(lo, (lane | 15) + 1), ghi = __shfl(hi, (lane | 15) + 1);
        if ((lane & 16) == 0) { const double nlo = lz_clamp(glo, lo, hi), nhi = lz_clamp(ghi, lo, hi); lo = nlo; hi = nhi; }
    }
    if (LPS >= 64) {                            // rows 0, 1 take rows 2-3 (lane 32)
        const double glo = __shfl(lo, 32), ghi = __shfl(hi, 32);
        if ((lane & 32) == 0) { const double nlo = lz_clamp(glo, lo, hi), nhi = lz_clamp(ghi, lo, hi); lo = nlo; hi = nhi; }
    }
}

// prefix (left-to-right) inclusive scan of clamp maps over the lanes of each group: lane l ends up with
// M_l o ... o M_1 o M_0 (the left-most map is applied first)
template <int LPS>
__device__ __forceinline__ void scan_clamps_fwd(double &lo, double &hi, int lane)
{
    const int r = lane & ((LPS < 16 ? LPS : 16) - 1);
#define DOPF_FWD_STEP(CTRL, RM, COND)                                                \
    {                                                                                \
        const double glo = (RM) == 0xF ? dppz<CTRL>(lo) : dppd<CTRL, RM>(lo, lo), ghi = (RM) == 0xF ? dppz<CTRL>(hi) : dppd<CTRL, RM>(hi, hi); \
        if (COND) { const double nlo = lz_clamp(glo, lo, hi), nhi = lz_clamp(ghi, lo, hi); lo = nlo; hi = nhi; } \
    }
    DOPF_FWD_STEP(0x111, 0xF, r >= 1)                           // row_shr:1
    if (LPS >= 4) DOPF_FWD_STEP(0x112, 0xF, r >= 2)
    if (LPS >= 8) DOPF_FWD_STEP(0x114, 0xF, r >= 4)
    if (LPS >= 16) DOPF_FWD_STEP(0x118, 0xF, r >= 8)
    if (LPS >= 32) DOPF_FWD_STEP(0x142, 0xA, lane & 16)         // row_bcast:15
    if (LPS >= 64) DOPF_FWD_STEP(0x143, 0xC, lane & 32)         // row_bcast:31
#undef DOPF_FWD_STEP
}

// suffix (right-to-left) inclusive min over the lanes of each group
template <int LPS>
__device__ __forceinline__ int scan_min_rev_i(int x, int lane)
{
    constexpr int RL = LPS < 16 ? LPS : 16;
    const int r = lane & (RL - 1);
#define DOPF_REVI_STEP(CTRL, D)                                                      \
    {                                                                                \
        const int g = __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false);      \
        if (r + D < RL) x = g < x ? g : x;                                           \
    }
    DOPF_REVI_STEP(0x101, 1)                    // row_shl:1
    if (LPS >= 4) DOPF_REVI_STEP(0x102, 2)
    if (LPS >= 8) DOPF_REVI_STEP(0x104, 4)
    if (LPS >= 16) DOPF_REVI_STEP(0x108, 8)
#undef DOPF_REVI_STEP
    if (LPS >= 32) { const int g = __shfl(x, (lane | 15) + 1); if ((lane & 16) == 0) x = g < x ? g : x; }
    if (LPS >= 64) { const int g = __shfl(x, 32); if ((lane & 32) == 0) x = g < x ? g : x; }
    return x;
}

// value of the next lane of the group (garbage for the group's last lane: caller masks it)
template <int LPS>
__device__ __forceinline__ double next_lane(double x)
{
    if (LPS <= 16) return dppz<0x101>(x);              // row_shl:1
    return dppz<0x130>(x);                             // wave_shl:1
}

template <int LPS>
__device__ __forceinline__ int next_lane_i(int x)
{
    if (LPS <= 16) return __builtin_amdgcn_update_dpp(x, x, 0x101, 0xF, 0xF, false);
    return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xF, 0xF, false);
}

// returns the number of storages of the item left to the scan (block-uniform)
template <int LPS, int NCH, bool LINES, bool TAIL = false, bool FULLT = false>
__device__ __forceinline__ int sto_warm_body(const DevView &v, const int blk, const int halt = 0)
{
    constexpr int NG = 256 / LPS, TP = LPS * NCH;
    constexpr int MAXR = 16;                 // contact-set rounds per storage
    constexpr int MAXN = 40;                 // Newton iterations per round
    constexpr int BIG = 0x3fffffff;
    // per lane group, indexed by timestep; the last three only at segment ends
    __shared__ double red[NG * TP];          // nuL: price of the segment that ends here (also the final reduction's buffer)
    __shared__ double baseL[NG * TP];        // level at which the segment that ends here starts
    __shared__ double loL[NG * TP], hiL[NG * TP];    // Newton bracket of the segment's price
    __shared__ double fdL[NG * TP];          // flat segment: signed distance to the nearest kink ahead
    __shared__ double redc[256];
    const int tid = threadIdx.x, lane = tid & 63, li = tid & (LPS - 1), grp = tid / LPS;
    const int gbase = lane & ~(LPS - 1);
    // one node: the items are equal cuts of the storage list — no table look-up between the block's start and its loads
    Item it;
    if (v.stoChunk > 0) { it.a0 = blk * v.stoChunk; it.a1 = min(v.S, it.a0 + v.stoChunk); it.node = 0; }
    else it = v.sto_items[blk];
    // `halt`: the caller's halt word, loaded but not yet looked at: it travels with the first storage's rows (looked at
    // below, once those loads have been issued; nothing is stored before that. Uniform; -1 = halted)
    const int T = FULLT ? LPS * NCH : v.T, N = v.N;
    const double w = v.w_prox, gam = v.gamma;
    const double a0 = w + gam, ia0 = 1.0 / a0, idet0 = 1.0 / (a0 * a0 - gam * gam), s20 = 2.0 / (a0 + gam);
    const int tbase = li * NCH;
    double *nuL = red + grp * TP, *base = baseL + grp * TP, *lo_ = loL + grp * TP, *hi_ = hiL + grp * TP, *fd_ = fdL + grp * TP;

    double accQ[NCH];
    double accCost = 0.0;
    int anyFail = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) accQ[c] = 0.0;
    // (price + gamma * imbalance of a step is read per storage pass, with its rows, instead of held across the solve: the
    // kernel sits at its register limit and a pass's second read of two cached doubles costs less than six registers)
    // with lines: a (node, timestep) whose table is empty — no kink of Psi inside the node's window, the usual case —
    // is the copper-plate closed form with (Psi(0), slope) in place of (theta, gamma): cached here, no table reads
    double lp0[NCH], lkap[NCH];
    bool lin[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = tbase + c;
        lin[c] = false; lp0[c] = 0.0; lkap[c] = 0.0;
        if (LINES && t < T) {
            const size_t at = (size_t)it.node + (size_t)N * t;
            // (all three loaded at once: the size word in front of the other two would be a round trip of its own)
            const int m_ = v.tb_m[at];
            const double p0_ = v.tb_psi0[at], k_ = v.tb_slope[at * (v.M2 + 1)];
            if (m_ == 0) {
                lin[c] = true;
                lp0[c] = p0_;
                lkap[c] = k_;
            }
        }
    }
#ifdef DOPF_STATS
    unsigned long long st_rounds = 0, st_newton = 0, cyc[6] = {0, 0, 0, 0, 0, 0}, tq = 0, st_lvl = 0, st_sgn = 0, st_nnc = 0;
#define DOPF_TIC() tq = clock64()
#define DOPF_TOC(i) { const unsigned long long t2_ = clock64(); cyc[i] += t2_ - tq; tq = t2_; }
#else
#define DOPF_TIC()
#define DOPF_TOC(i)
#endif
    const int nRep = (it.a1 - it.a0 + NG - 1) / NG;
    { const int rep = 0, round = 0; DOPF_STAMP(0) }
    for (int rep = 0; rep < nRep; ++rep) {
        const int s = it.a0 + rep * NG + grp;
        const bool live = s < it.a1;
        const double mc = live ? v.sto_mc[s] : 0.0, pm = live ? v.sto_pmax[s] : 0.0, em = live ? v.sto_emax[s] : 0.0;
        const bool havenu = live && v.nu_valid[s] != 0;
        // copper plate: the nu-independent parts of the two gradient offsets, rD = rD0 - nu, rC = rC0 + nu
        // (with lines Psi depends on the step itself: D0/C0 are kept and the offsets are built per evaluation)
        double A0[NCH], B0[NCH], nuv[NCH];       // (rD0, rC0) without lines, (D0, C0) with
        int hint[NCH];
        double dq[NCH];                          // C0 - D0 per step: the level trajectory below needs only these
        const double iw = 1.0 / w;
        double run = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            const bool ok = live && t < T;
            const size_t e = (size_t)s * T + (ok ? t : 0);
            const double d0 = ok ? v.D[e] : 0.0, c0 = ok ? v.C[e] : 0.0;
            const double nu_st = v.nu_prev[ok ? e : 0];        // (always loaded — from a valid address: a lane group without a storage
                                                               // has s past the end — a load behind the nu_valid word would be a second round trip)
            nuv[c] = (ok && havenu) ? nu_st : 0.0;             // no stored prices (zero state, set_state): start from 0
            hint[c] = 0;
            dq[c] = c0 - d0;
            run += c0 - d0;
            if (LINES) { A0[c] = d0; B0[c] = c0; }
            else {
                const double th0 = t < T ? v.price[it.node + N * t] + gam * v.s[t] : 0.0;
                const double theta = th0 - gam * (d0 - c0);
                A0[c] = w * d0 - mc - theta; B0[c] = w * c0 - mc + theta;
                // Newton starts where the price move has put the segment: what a step's (D, C) problem sees of the two prices
                // is nu + theta, and with the same contact structure it is that sum, not nu, that stays when lambda moves — so
                // nu + theta is what is stored per step. config2 from the zero state: 7.6 -> 2.8 Newton iterations per storage
                // in iteration 3, 2.35 -> 2.0 in the settled state.
                nuv[c] = (ok && havenu) ? nu_st - theta : 0.0;
            }
        }
        if (rep == 0 && halt) return -1;         // (uniform; the loads above are on their way, nothing has been stored)
        // (D, C)(nu) of step c, d(C - D)/dnu; shift = Psi at the step's net injection minus its nu-independent anchor
        auto eval = [&](int c, double nu, double &dd, double &cc, double &s1) {
            if (!LINES) {
                box2(a0, gam, ia0, idet0, s20, A0[c] - nu, B0[c] + nu, pm, dd, cc, s1);
            } else if (lin[c]) {
                const double q0 = A0[c] - B0[c], theta = lp0[c] - lkap[c] * q0;
                double lia, lidet, ls2;
                lin_coef(w, iw, lkap[c], lia, lidet, ls2);
                box2(w + lkap[c], lkap[c], lia, lidet, ls2, w * A0[c] - mc - theta - nu, w * B0[c] - mc + theta + nu,
                     pm, dd, cc, s1);
            } else {
                const EvalOut o = eval_lines_call(v.self, it.node, tbase + c, hint[c], mc, pm, A0[c], B0[c], nu);
                dd = o.dd; cc = o.cc; s1 = o.s1; hint[c] = o.hint;
            }
        };
        // the four prices at which D or C of step c would leave a bound with the step's net injection (hence Psi)
        // frozen at (dd, cc): bD, bD - w pm, bC, bC + w pm
        auto kinks = [&](int c, double dd, double cc, double &bD, double &bC) {
            if (!LINES) {
                bD = A0[c] - gam * (dd - cc); bC = -B0[c] - gam * (dd - cc);
            } else {
                double pc;
                if (lin[c]) pc = lp0[c] + lkap[c] * ((dd - cc) - (A0[c] - B0[c]));
                else pc = tab_psi_call(v.self, it.node, tbase + c, (dd - cc) - (A0[c] - B0[c]));
                bD = w * A0[c] - mc - pc; bC = mc - w * B0[c] - pc;
            }
        };
        // previous level trajectory -> contacts
        const double inclE = scan_sum<LPS>(run, lane);
        const double prevE = prev_lane<LPS>(inclE);
        const double tolc = 1e-9 * (1.0 + em), tolE = 1e-11 * (1.0 + em), tolr = 1e-12 * (1.0 + em);
        int kind[NCH];                       // 0 free, 1 empty, 2 full
        {
            double eo = li == 0 ? 0.0 : prevE;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int t = tbase + c;
                eo += dq[c];
                kind[c] = t < T ? (eo <= tolc ? 1 : (eo >= em - tolc ? 2 : 0)) : 0;
            }
        }

        bool gdone = !live, good = false;
        { const int round = 0; DOPF_STAMP(1) }
        for (int round = 0; round < MAXR; ++round) {
#ifdef DOPF_STATS
            if (li == 0 && !gdone) ++st_rounds;
#endif
            // ---- A. segments of the current contact set --------------------------------------------------
            DOPF_TIC();
            // end of a segment: a contact, or the last step; start: the step after an end (or step 0)
#define ISEND(c) (tbase + (c) < T && (kind[c] != 0 || tbase + (c) == T - 1))
#define TGT(c) (kind[c] == 2 ? em : 0.0)
            double bs[NCH];
            int send[NCH];
            bool st0;                                            // this lane's first step starts a segment
            {
                int mfirst = BIG;
#pragma unroll
                for (int c = NCH - 1; c >= 0; --c)
                    if (ISEND(c)) mfirst = tbase + c;
                const int incl = scan_min_rev_i<LPS>(mfirst, lane);
                const int nxt = next_lane_i<LPS>(incl);
                int carry = li == LPS - 1 ? BIG : nxt;       // first segment end in the lanes to the right
#pragma unroll
                for (int c = NCH - 1; c >= 0; --c) {
                    if (ISEND(c)) carry = tbase + c;
                    send[c] = carry < TP ? carry : TP - 1;   // (steps past the horizon: any valid slot)
                }
                // info of the step before this lane's first step: target level if it is a segment end, else -1
                const double lastInfo = ISEND(NCH - 1) ? TGT(NCH - 1) : -1.0;
                const double plInfo = prev_lane<LPS>(lastInfo);
                st0 = tbase < T && (li == 0 || plInfo >= 0.0);
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    const bool stc = c == 0 ? st0 : (t < T && ISEND(c - 1));
                    const double pinfo = c == 0 ? (li == 0 ? 0.0 : plInfo) : TGT(c - 1);
                    if (stc) base[send[c]] = pinfo;                  // level at which this segment starts
                    if (ISEND(c)) {
                        nuL[t] = kind[c] != 0 ? nuv[c] : 0.0;        // one price per segment; open last segment: 0
                        lo_[t] = -INFINITY; hi_[t] = INFINITY;
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bool ok = tbase + c < T;
                    nuv[c] = ok ? nuL[send[c]] : 0.0;
                    bs[c] = ok ? base[send[c]] : 0.0;
                }
                __builtin_amdgcn_wave_barrier();
            }
#define STARTS(c) ((c) == 0 ? st0 : (tbase + (c) < T && ISEND((c) - 1)))

            // ---- B. segmented Newton, bracketed --------------------------------------------------------------
            DOPF_TOC(0)
            DOPF_STAMP(2)
            double px[NCH], ps[NCH], Dv[NCH], Cv[NCH];
            bool nconv = false, nfail = false;
            for (int itn = 0; itn < MAXN; ++itn) {
#ifdef DOPF_STATS
                if (li == 0 && !gdone) ++st_newton;
#endif
                {
                    int f = 0;
                    double rx = 0.0, rs = 0.0;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        double dd = 0.0, cc = 0.0, s1 = 0.0;
                        if (tbase + c < T) eval(c, nuv[c], dd, cc, s1);
                        Dv[c] = dd; Cv[c] = cc;            // (of the last evaluation: the certified values when the round passes)
                        if (STARTS(c)) { rx = 0.0; rs = 0.0; f = 1; }
                        rx += cc - dd; rs += s1;
                        px[c] = rx; ps[c] = rs;
                    }
                    int fl = f;
                    double ax = rx, as = rs;
                    seg_scan2<LPS>(fl, ax, as, lane);
                    double cx = prev_lane<LPS>(ax), cs = prev_lane<LPS>(as);
                    if (li == 0) { cx = 0.0; cs = 0.0; }
                    bool seen = false;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        seen = seen || STARTS(c);
                        if (!seen) { px[c] += cx; ps[c] += cs; }
                    }
                }
                // does a segment that has to move sit on a flat piece? Then every step of it has D and C on bounds, its
                // net injection (hence Psi) does not move with nu and the four prices at which D or C would leave a bound
                // are closed form: signed distance to the nearest one in the direction the residual asks for
                bool flatNeed = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (ISEND(c) && kind[c] != 0 && fabs(bs[c] + px[c] - TGT(c)) > tolr && !(ps[c] > 0.0)) flatNeed = true;
                if (__any(flatNeed && !gdone)) {
                    int f2 = 0;
                    double ru = -INFINITY, rd = INFINITY;        // (-min distance above, min distance below) so far
                    double fu[NCH], fn[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        double du = INFINITY, dn = INFINITY;
                        if (tbase + c < T) {
                            double bD, bC;
                            kinks(c, Dv[c], Cv[c], bD, bC);          // (D, C) of this Newton iteration's evaluation at nuv[c]
                            const double wp = w * pm;
                            const double cand[4] = {bD, bD - wp, bC, bC + wp};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const double d = cand[q] - nuv[c];
                                if (d > 0.0) du = fmin(du, d);
                                if (d < 0.0) dn = fmin(dn, -d);
                            }
                        }
                        if (STARTS(c)) { ru = -INFINITY; rd = INFINITY; f2 = 1; }
                        ru = fmax(ru, -du); rd = fmin(rd, dn);
                        fu[c] = ru; fn[c] = rd;
                    }
                    int fl = f2;
                    double au = ru, ad = rd;
                    seg_scan_maxmin<LPS>(fl, au, ad, lane);
                    double cu = prev_lane<LPS>(au), cd = prev_lane<LPS>(ad);
                    if (li == 0) { cu = -INFINITY; cd = INFINITY; }
                    bool seen = false;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        seen = seen || STARTS(c);
                        if (!seen) { fu[c] = fmax(fu[c], cu); fn[c] = fmin(fn[c], cd); }
                        if (ISEND(c) && kind[c] != 0) fd_[tbase + c] = (bs[c] + px[c] - TGT(c)) < 0.0 ? -fu[c] : -fn[c];
                    }
                }
                double worst = 0.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    if (ISEND(c) && kind[c] != 0) {
                        const double r = bs[c] + px[c] - TGT(c);
                        double ar = fabs(r);
                        if (ar > tolr) {
                            const double nu = nuv[c];
                            double blo = lo_[t], bhi = hi_[t];
                            if (r < 0.0) { blo = nu; lo_[t] = nu; } else { bhi = nu; hi_[t] = nu; }
                            const bool both = blo > -INFINITY && bhi < INFINITY;
                            double trial;
                            if (ps[c] > 0.0) {
                                trial = nu - r * rcp64(ps[c]);
                            } else {
                                const double sd = fd_[t];                  // signed; +-inf when no kink lies ahead
                                trial = nu + sd + copysign(1e-9 * (1.0 + fabs(nu) + fabs(sd)), sd);
                            }
                            if (!(trial > blo && trial < bhi) || (both && itn >= 8 && (itn & 1))) {
                                if (both) trial = 0.5 * (blo + bhi);
                            }
                            if (trial > blo && trial < bhi) nuL[t] = trial;
                            else if (both) ar = 0.0;           // bracket is two adjacent doubles: this is the root
                            else nfail = true;                 // nothing ahead moves this segment: not a valid contact set
                        }
                        worst = fmax(worst, ar);
                    }
                }
                worst = group_max<LPS>(worst);
                nfail = group_bits<LPS>(nfail, gbase) != 0ull;
                nconv = worst <= tolr && !nfail;
                // every group in the wave runs the same number of rounds (DPP scans need all lanes)
                if (__all(gdone || nconv || nfail)) break;
                __builtin_amdgcn_wave_barrier();
                if (!nconv) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        if (tbase + c < T) nuv[c] = nuL[send[c]];
                }
                __builtin_amdgcn_wave_barrier();
            }

            // ---- C. certificate: levels inside the band, price jumps have the right sign ----------------------
            // A contact step that is a segment of its own with zero net charge (the storage idles on a bound)
            // accepts every price of its dead band [rD0, -rC0]; all other segment prices are points. Prices
            // are then chosen right to left, nu_e = clamp(nu_next, band_e) starting from nu_{T+1} = 0 — the
            // choice that satisfies the sign condition at e whenever any does — with one suffix scan of
            // clamp maps, and the sign conditions are checked on that choice.
            DOPF_TOC(1)
            DOPF_STAMP(3)
            bool okk = true;
            int nkind[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) nkind[c] = kind[c];
            // Copper plate: a segment whose steps ALL sit on a corner of their (D, C) box (charging or discharging at
            // full rate, or idle) does not move with its price: every price in the intersection of the steps' corner
            // intervals is a multiplier of that segment, not just the one Newton happened to stop at. With
            // grad_D = a D - gam C - (rD0 - nu), grad_C = a C - gam D - (rC0 + nu):
            //   D = 0: nu >= rD0 + gam C      D = pm: nu <= rD0 - a pm + gam C
            //   C = 0: nu <= -rC0 - gam D     C = pm: nu >= a pm - gam D - rC0
            double slo[NCH], shi[NCH];
            if (!LINES) {
                int f = 0;
                double rl = -INFINITY, rh = INFINITY;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    double lo = -INFINITY, hi = INFINITY;
                    if (t < T) {
                        const double rD0 = A0[c], rC0 = B0[c];
                        const double dd = Dv[c], cc = Cv[c];
                        if (dd <= 0.0) lo = rD0 + gam * cc; else if (dd >= pm) hi = rD0 - a0 * pm + gam * cc;
                        if (cc <= 0.0) hi = lz_min(hi, -rC0 - gam * dd); else if (cc >= pm) lo = lz_max(lo, a0 * pm - gam * dd - rC0);
                    }
                    if (STARTS(c)) { rl = -INFINITY; rh = INFINITY; f = 1; }
                    rl = lz_max(rl, lo); rh = lz_min(rh, hi);
                    slo[c] = rl; shi[c] = rh;
                }
                int fl = f;
                double al = rl, ah = rh;
                seg_scan_maxmin<LPS>(fl, al, ah, lane);
                double cl = prev_lane<LPS>(al), ch = prev_lane<LPS>(ah);
                if (li == 0) { cl = -INFINITY; ch = INFINITY; }
                bool seen = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    seen = seen || STARTS(c);
                    if (!seen) { slo[c] = lz_max(slo[c], cl); shi[c] = lz_min(shi[c], ch); }
                }
            }
            double mlo[NCH], mhi[NCH];
            // Prices: segment i (ending at contact e_i) may take any nu_i in [mlo, mhi] (a point unless the segment is
            // flat); an empty contact needs nu_i >= nu_{i+1}, a full one nu_i <= nu_{i+1}, and nu past the horizon is 0.
            // Right to left, the set of nu_i that can be completed to the right is the interval
            //   empty: [max(mlo, flo_{i+1}), mhi]      full: [mlo, min(mhi, fhi_{i+1})]      open last segment: {0}
            // — two chains of clamp maps (one for the lower ends, one for the upper ends), one suffix scan each;
            // the certificate holds iff no interval is empty.
            double alo = -INFINITY, ahi = INFINITY, blo2 = -INFINITY, bhi2 = INFINITY;      // this lane's composed maps
#pragma unroll
            for (int c = NCH - 1; c >= 0; --c) {
                const int t = tbase + c;
                mlo[c] = -INFINITY; mhi[c] = INFINITY;
                if (t < T) {
                    const double Ev = bs[c] + px[c];
                    if (!(ISEND(c) && kind[c] != 0)) {           // (a contact's level is its Newton target)
                        if (Ev < -tolE) { okk = false; nkind[c] = 1; }
                        else if (Ev > em + tolE) { okk = false; nkind[c] = 2; }
                    }
                    if (ISEND(c)) {
                        mlo[c] = mhi[c] = kind[c] != 0 ? nuv[c] : 0.0;
                        if (!LINES) {
                            // flat segment (zero slope at its end = every step on a corner): the whole interval
                            if (kind[c] != 0 && ps[c] == 0.0 && slo[c] <= shi[c] && nuv[c] >= slo[c] - 1e-9 && nuv[c] <= shi[c] + 1e-9) {
                                mlo[c] = slo[c]; mhi[c] = shi[c];
                            }
                        } else if (kind[c] != 0 && STARTS(c) && bs[c] == TGT(c)) {
                            // idle on a bound: with D = C = 0 the step's net injection is unchanged at q = 0: dlt = -q0
                            const double q0 = A0[c] - B0[c];
                            const double theta = lin[c] ? lp0[c] - lkap[c] * q0 : tab_psi_call(v.self, it.node, t, -q0);
                            const double rD0 = w * A0[c] - mc - theta, rC0 = w * B0[c] - mc + theta;
                            if (rD0 <= -rC0 && nuv[c] >= rD0 - 1e-9 && nuv[c] <= -rC0 + 1e-9) { mlo[c] = rD0; mhi[c] = -rC0; }
                        }
                        // lower-end chain: empty x -> max(mlo, x), full/open x -> mlo; upper-end chain: empty/open x -> mhi, full x -> min(mhi, x)
                        const double a_lo = mlo[c], a_hi = kind[c] == 1 ? INFINITY : mlo[c];
                        const double b_lo = kind[c] == 2 ? -INFINITY : mhi[c], b_hi = mhi[c];
                        const double nal = lz_clamp(alo, a_lo, a_hi), nah = lz_clamp(ahi, a_lo, a_hi);
                        const double nbl = lz_clamp(blo2, b_lo, b_hi), nbh = lz_clamp(bhi2, b_lo, b_hi);
                        alo = nal; ahi = nah; blo2 = nbl; bhi2 = nbh;
                    }
                }
            }
            scan_clamps_rev<LPS>(alo, ahi, lane);
            scan_clamps_rev<LPS>(blo2, bhi2, lane);
            // ends of the feasible interval arriving from the right of this lane: (lanes to the right)(0)
            const double rightA = next_lane<LPS>(lz_clamp(0.0, alo, ahi)), rightB = next_lane<LPS>(lz_clamp(0.0, blo2, bhi2));
            double flo = li == LPS - 1 ? 0.0 : rightA, fhi = li == LPS - 1 ? 0.0 : rightB;
            double nuc[NCH];                                     // the certified prices (move only inside a flat segment's interval)
#pragma unroll
            for (int c = NCH - 1; c >= 0; --c) {
                const int t = tbase + c;
                nuc[c] = nuv[c];
                if (t < T && ISEND(c)) {
                    flo = kind[c] == 1 ? lz_max(mlo[c], flo) : mlo[c];
                    fhi = kind[c] == 2 ? lz_min(mhi[c], fhi) : mhi[c];
                    if (kind[c] != 0) {
                        const double tn = 1e-10 * (1.0 + lz_minabs(flo, fhi));
                        if (flo > fhi + tn) { okk = false; nkind[c] = 0; }      // wrong sign: release the contact
                        nuc[c] = lz_clamp(nuv[c], flo, lz_max(flo, fhi));
                    }
                }
            }
            // The same question from the LEFT, asked only when a price jump has the wrong sign somewhere in the wave: the interval a
            // segment's price may take given everything to its left (an empty contact in front of it caps it at the left price, a
            // full one floors it). An empty interval releases the contact IN FRONT of the segment, and the inverted bound travels
            // on. Without it a run of idle contacts that a cheaper segment on its left wants to charge through (a storage that sat
            // empty for sixteen steps and now starts earlier) lost ONE contact per round from the left — the right-to-left pass
            // above only sees the first of them: 7 rounds in iteration 8 of config2, 3 with it (prototype: scripts/proto_repair.py,
            // FWD=4).
            // Only when the right-to-left pass released a FEW contacts (1..4 of the storage): where it releases many — the zero
            // state, where every step is a contact — both passes carrying their inverted bounds on release nearly everything,
            // the next round adds it back, and the rounds cycle (prototype: all storages at the round cap in iteration 1).
            {
                int nrel = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) nrel += __popcll(group_bits<LPS>(nkind[c] == 0 && kind[c] != 0, gbase));
                const bool fwd = nrel >= 1 && nrel <= 4 && !gdone;          // (uniform over the lane group)
                if (__any(fwd)) {
                    double Llo = -INFINITY, Lhi = INFINITY, Ulo = -INFINITY, Uhi = INFINITY;     // this lane's composed maps
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (tbase + c < T && ISEND(c)) {
                            // state for the NEXT segment: lower bound max(mlo, .) behind a full contact, none otherwise;
                            // upper bound min(mhi, .) behind an empty contact, none otherwise
                            const double l_lo = kind[c] == 2 ? mlo[c] : -INFINITY, l_hi = kind[c] == 2 ? INFINITY : -INFINITY;
                            const double u_lo = kind[c] == 1 ? -INFINITY : INFINITY, u_hi = kind[c] == 1 ? mhi[c] : INFINITY;
                            const double a1 = lz_clamp(Llo, l_lo, l_hi), a2 = lz_clamp(Lhi, l_lo, l_hi);
                            const double b1 = lz_clamp(Ulo, u_lo, u_hi), b2 = lz_clamp(Uhi, u_lo, u_hi);
                            Llo = a1; Lhi = a2; Ulo = b1; Uhi = b2;
                        }
                    }
                    scan_clamps_fwd<LPS>(Llo, Lhi, lane);
                    scan_clamps_fwd<LPS>(Ulo, Uhi, lane);
                    const double leftL = prev_lane<LPS>(Llo), leftU = prev_lane<LPS>(Uhi);       // (maps of the lanes to the left)(-inf), (+inf)
                    double pin = li == 0 ? -INFINITY : leftL, phin = li == 0 ? INFINITY : leftU;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int t = tbase + c;
                        if (t < T && ISEND(c)) {
                            const double glo = lz_max(mlo[c], pin), ghi = lz_min(mhi[c], phin);
                            const double tn = 1e-10 * (1.0 + lz_minabs(glo, ghi));
                            fd_[t] = glo > ghi + tn ? 1.0 : 0.0;           // (fd_: the Newton loop's scratch, free here; indexed by segment end)
                            pin = kind[c] == 2 ? glo : -INFINITY;
                            phin = kind[c] == 1 ? ghi : INFINITY;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int sendNext = next_lane_i<LPS>(send[0]);        // the segment the next lane's first step belongs to
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int t = tbase + c;
                        if (t + 1 < T && ISEND(c) && kind[c] != 0) {
                            const int sn = c + 1 < NCH ? send[c + 1 < NCH ? c + 1 : c] : sendNext;
                            if (fwd && fd_[sn] != 0.0) { okk = false; nkind[c] = 0; }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // (debugLeave, tests: every third storage is declared uncertified, so that the hand-over to the scan body runs)
            const bool cert = nconv && group_bits<LPS>(!okk, gbase) == 0ull && !(v.debugLeave && s % 3 == 0);
            bool chg = false;
#pragma unroll
            for (int c = 0; c < NCH; ++c) chg = chg || nkind[c] != kind[c];
            const bool changed = group_bits<LPS>(chg, gbase) != 0ull;
#ifdef DOPF_STATS
            {   // why a round did not certify: levels outside the band (contacts to add) / price jumps of the wrong sign (to release) /
                // Newton did not converge — per lane group and round
                bool lv = false, sg = false;
#pragma unroll
                for (int c = 0; c < NCH; ++c) { lv = lv || (nkind[c] != 0 && kind[c] == 0); sg = sg || (nkind[c] == 0 && kind[c] != 0); }
                const bool anyLv = group_bits<LPS>(lv, gbase) != 0ull, anySg = group_bits<LPS>(sg, gbase) != 0ull;
                if (li == 0 && !gdone && !cert) {
                    if (anyLv) ++st_lvl;
                    if (anySg) ++st_sgn;
                    if (!nconv) ++st_nnc;
                }
            }
#endif

            // ---- D. accept, repair the contact set, or give up --------------------------------------------------
            DOPF_TOC(2)
            DOPF_STAMP(4)
            if (!gdone && cert) {
                // (the rows' addresses are formed again from the storage's index: held since the loads at the top they are six
                // registers the solve in between spills for)
                int s_ = s;
                asm volatile("" : "+v"(s_));
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = tbase + c;
                    if (t < T) {
                        const size_t e = (size_t)s_ * T + t;
                        v.D[e] = Dv[c];
                        v.C[e] = Cv[c];
                        // (nu + theta, see the loads; theta back from the step's offsets: B0 - A0 = w (c0 - d0) + 2 theta. Kept in
                        // registers: a second array of solver state in memory cost 1.2 us per iteration, the registers nothing)
                        v.nu_prev[e] = LINES ? nuc[c] : nuc[c] + 0.5 * (B0[c] - A0[c] - w * dq[c]);
                        if (LINES && (v.keepDeltas || v.walk_any[t])) v.dltS[e] = (Dv[c] - Cv[c]) - (A0[c] - B0[c]);
                        accQ[c] += Dv[c] - Cv[c];
                        accCost += mc * (Dv[c] + Cv[c]);
                    }
                }
                good = true;
                gdone = true;
            }
            if (!gdone && (!nconv || !changed)) gdone = true;       // Newton stalled / nothing to repair: scan kernel
            DOPF_TOC(3)
            DOPF_STAMP(5)
            if (__all(gdone)) break;
            if (!gdone) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) kind[c] = nkind[c];
            }
            __builtin_amdgcn_wave_barrier();
#undef ISEND
#undef TGT
#undef STARTS
        }

        if (live && li == 0) { v.sto_fail[s] = good ? 0 : 1; if (good) v.nu_valid[s] = 1; }
        if (live && !good && li == 0) anyFail += 1;
        __builtin_amdgcn_wave_barrier();
    }
#ifdef DOPF_STATS
    if (st_lvl) atomicAdd(&v.st->dbg_scans, st_lvl);               // (the scan body's counters, unused while it has nothing to do)
    if (st_sgn) atomicAdd(&v.st->dbg_wave_loops, st_sgn);
    if (st_nnc) atomicAdd(&v.st->dbg_events, st_nnc);
    if (st_rounds) atomicAdd(&v.st->dbg_reason[0], st_rounds);
    if (st_newton) atomicAdd(&v.st->dbg_reason[1], st_newton);
    if (lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(&v.st->dbg_cyc[i], cyc[i]);
    if (st_rounds) atomicMax(&v.st->dbg_reason[2], st_rounds);        // most rounds / Newton iterations of one lane group
    if (st_newton) atomicMax(&v.st->dbg_reason[3], st_newton);
#endif
#undef DOPF_TIC
#undef DOPF_TOC

    // fixed-order block reduction of the per-timestep sums over the NG groups
    // the block's sums with ONE barrier: cost and the count of storages left over are added inside each wave first
    // (butterfly: a fixed order), the four waves then in order. (A lane group's slice of red[] is its own nuL region:
    // nothing of another wave's is overwritten here. The barrier below waits for LDS traffic only — __syncthreads()
    // would add the acknowledgement of the rows just stored to the chain.)
#pragma unroll
    for (int c = 0; c < NCH; ++c) red[(grp * LPS + li) * NCH + c] = accQ[c];
    __shared__ double wcostS[4];
    __shared__ int wfailS[4];
    {
        double cw = accCost;
        int fw = anyFail;
        for (int d = 32; d > 0; d >>= 1) { cw += __shfl_xor(cw, d); fw += __shfl_xor(fw, d); }
        if (lane == 0) { wcostS[tid >> 6] = cw; wfailS[tid >> 6] = fw; }
    }
    TailView tv{};                                              // TAIL: the launch carries the iteration's tail
    int tpar = 0;
    if (TAIL) { tv = *v.tail; tpar = v.st->tail_par; }          // (uniform scalar loads, in flight across the barrier)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int blockFail = wfailS[0] + wfailS[1] + wfailS[2] + wfailS[3];     // storages of this item left to the scan kernel
    if (grp == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = tbase + c;
            if (t < T) {
                double sum = 0.0;
                for (int g2 = 0; g2 < NG; ++g2) sum += red[(g2 * LPS + li) * NCH + c];
                // tail in the launch: ONE counted add per block and slot. A block with storages left over for the scan body
                // parks its sums in its partial row instead; the scan body (same block, same thread per slot) adds both.
                if (TAIL && blockFail == 0) acc_add(tv, tpar, t, sum, tv.scaleInj);
                else if (LINES) v.part_T[(size_t)t * v.rowsT + it.row + 1] = sum;
                else v.part_sinj_w[(size_t)blk * T + t] = sum;
            }
        }
    }
    if (tid == 0) {
        const double cw = ((wcostS[0] + wcostS[1]) + wcostS[2]) + wcostS[3];
        if (TAIL && blockFail == 0) acc_add(tv, tpar, T, cw, tv.scaleCost);
        else v.part_scost_w[blk] = cw;
        v.item_fail[blk] = blockFail;
    }
    // The scan body reads the sto_fail words other waves have just stored: that needs the full barrier (stores
    // acknowledged). With nothing left over — the usual case — nobody reads anything of this block's again: its waves
    // end without waiting for their rows to be acknowledged.
    if (blockFail != 0) __syncthreads();
    { const int rep = 0, round = 0; DOPF_STAMP(6) }
    return blockFail;
}

}  // namespace dopf
#include "sto_lean.h"
namespace dopf {

// The scan body as a function of its own (networks): called by k_sto_warm for the rare item the active-set body leaves
// something of. Inlined, its registers crowd the active-set body (255 VGPRs and spills, 40 % slower, measured); as a
// separate launch it cost 4 us + a launch gap per iteration for finding nothing to do.
template <int LPS, int NCH>
__device__ DOPF_CALL_ATTR void sto_cold_lines_call(const DevView *self, const int blk, const int left)
{
    sto_cold_body<LPS, NCH, true>(*self, blk, left);
}

#ifndef DOPF_WARM_WAVES
#define DOPF_WARM_WAVES 2
#endif
template <int LPS, int NCH, bool LINES, bool LEAN = false>
__global__ __launch_bounds__(256, DOPF_WARM_WAVES) void k_sto_warm(DevView v)
{
    // (networks, LEAN: the lean body where every table of the item's node is empty — the settled state — else the general one.
    // A template argument, not a branch on v.stoLean: with both bodies in one function the general one ran 6 % slower, measured)
    static_assert(LINES || !LEAN, "copper plates: k_sto_l");
    const int left = LEAN ? sto_lean_body<LPS, NCH, false, false, true, false>(v, blockIdx.x, v.st->halt)
                          : sto_warm_body<LPS, NCH, LINES>(v, blockIdx.x, v.st->halt);         // ends on a __syncthreads
    if (left < 0) return;                                                               // halted
    if (LINES && v.coldInWarm) {
        if (left == 0) {                         // (what the scan body writes when there is nothing for it)
            const int row = v.sto_items[blockIdx.x].row;
            for (int t = threadIdx.x; t < v.T; t += 256) v.part_T[(size_t)t * v.rowsT + row] = 0.0;
            if (threadIdx.x == 0) v.part_scost[blockIdx.x] = 0.0;
        } else {
            sto_cold_lines_call<LPS, NCH>(v.self, blockIdx.x, left);
        }
    }
}

#ifndef DOPF_NET_GEN_FLIGHT
#define DOPF_NET_GEN_FLIGHT 4
#endif
// Networks, single-GPU chain: generators and storages in ONE launch — the storage items first (a block lives for the whole
// launch: a chain of dependent round trips), the generator items behind them in 256-thread blocks that pass through the
// wave slots the storages leave free. Alone, either launch is a few hundred short blocks bound by its own latency chain
// (configs[3]'s share: 12 + 13 us and a kernel boundary); together they overlap.
template <int LPS, int NCH, bool LEAN>
__global__ __launch_bounds__(256, DOPF_WARM_WAVES) void k_net_agents(DevView v)
{
    const int nS = v.nStoItems;
#ifdef DOPF_NET_GEN_FIRST            // (experiment: generator blocks in front)
    const int nG_ = (int)gridDim.x - nS;
    const bool isGen = (int)blockIdx.x < nG_;
    const int gi = blockIdx.x, si = (int)blockIdx.x - nG_;
#else
    const bool isGen = (int)blockIdx.x >= nS;
    const int gi = (int)blockIdx.x - nS, si = blockIdx.x;
#endif
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_timeline[32768 + 2 * blockIdx.x] = wall_clock64();
#endif
    if (isGen) {
        if (v.st->halt) return;
        if (2 * v.genTT256 >= v.T) gen_lines_body2<256, DOPF_NET_GEN_FLIGHT>(v, gi, v.genTT256, v.genR);
        else gen_lines_body<256, DOPF_NET_GEN_FLIGHT>(v, gi, v.genTT256, v.genR);
    } else {
        const int left = LEAN ? sto_lean_body<LPS, NCH, false, false, true, false>(v, si, v.st->halt)
                              : sto_warm_body<LPS, NCH, true>(v, si, v.st->halt);                  // ends on a __syncthreads
        if (left < 0) return;                                                           // halted
        if (left == 0) {                     // (what the scan body writes when there is nothing for it)
            const int row = v.sto_items[si].row;
            for (int t = threadIdx.x; t < v.T; t += 256) v.part_T[(size_t)t * v.rowsT + row] = 0.0;
            if (threadIdx.x == 0) v.part_scost[si] = 0.0;
        } else {
            sto_cold_lines_call<LPS, NCH>(v.self, si, left);
        }
    }
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_timeline[32768 + 2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// Warm start and, in the same block, the cold scan for what it left over: one launch for the storages of the big
// copper-plate grids (the separate k_sto_update launch mostly found nothing to do).
template <int LPS, int NCH, bool LINES, bool TAIL, bool FULLT>
__global__ __launch_bounds__(256, 3) void k_sto(DevView v)
{
    // (TAIL: the generator launch in front of this one has added its sums; the grid's last block is the tail block)
    if (TAIL && (int)blockIdx.x == v.nStoItems) { tail_block(v.self); return; }
    const int left = sto_warm_body<LPS, NCH, LINES, TAIL, FULLT>(v, blockIdx.x, v.st->halt);   // (with something left over it ends on a
    if (left < 0) return;                                                                       // __syncthreads: the sto_fail flags are visible)
    sto_cold_body<LPS, NCH, LINES, TAIL, FULLT>(v, blockIdx.x, left);
}

// All x-updates of one copper-plate iteration in ONE launch: blocks [0, nStoItems) solve storages (warm start,
// then the scan for what it left over, in the same block), the rest sweep generators. The storage blocks are
// latency/VALU work, the generator blocks are pure streaming, so sharing the CUs hides one behind the other
// and two kernel boundaries (~4 us each of fixed cost) disappear. Storage blocks come first: they are the long
// ones (interleaving the two kinds in dispatch order starts the last storage blocks late and costs 50 %).
// The launch runs at the storage code's 3 waves/SIMD, which starves the streaming generator blocks once the
// grid is large, so dopf_create only fuses grids whose storage blocks are all resident from the start.
template <int LPS, int NCH, bool SKIP, bool TAIL, bool FULLT>
__global__ __launch_bounds__(256, 3) void k_agents(DevView v)
{
    const int nS = v.nStoItems;
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_timeline[32768 + 2 * blockIdx.x] = wall_clock64();
#endif
    if (TAIL && blockIdx.x == gridDim.x - 1) {
        tail_block(v.self);
    } else if (!SKIP && (int)blockIdx.x >= nS) {
        // generator block: its loads do not wait for the halt word
        if (v.genBlocks > 0) gen_pair_stream<256, TAIL>(v, blockIdx.x - nS, v.genBlocks);
        else gen_pair_body<256, TAIL, true>(v, blockIdx.x - nS);
    } else {
        if ((int)blockIdx.x < nS) {
            const int left = sto_warm_body<LPS, NCH, false, TAIL, FULLT>(v, blockIdx.x, v.st->halt);   // (with something left over it ends on a
            if (left >= 0) sto_cold_body<LPS, NCH, false, TAIL, FULLT>(v, blockIdx.x, left);            // __syncthreads: the sto_fail flags are visible)
        } else {
            if (v.st->halt) return;
            gen_pair_skip_body<256, TAIL>(v, blockIdx.x - nS);
        }
    }
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_timeline[32768 + 2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// The same two launches with the lean copper-plate storage body (sto_lean.h): horizon == LPS * NCH, at most 32 lanes per storage.
#ifndef DOPF_LEAN_STO_WAVES
#define DOPF_LEAN_STO_WAVES 3
#endif
template <int LPS, int NCH, bool TAIL, bool FULLT = true>
__global__ __launch_bounds__(256, DOPF_LEAN_STO_WAVES) void k_sto_l(DevView v)
{
    if (TAIL && (int)blockIdx.x == v.nStoItems) { tail_block(v.self); return; }
    const int left = sto_lean_body<LPS, NCH, TAIL, false, false, FULLT>(v, blockIdx.x, v.st->halt);
    if (left < 0) return;
    sto_cold_body<LPS, NCH, false, TAIL, FULLT>(v, blockIdx.x, left);
}

template <int LPS, int NCH, bool SKIP, bool TAIL, bool FULLT = true>
__global__ __launch_bounds__(256, 3) void k_agents_l(DevView v)
{
    const int nS = v.nStoItems;
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_timeline[32768 + 2 * blockIdx.x] = wall_clock64();
#endif
    if (TAIL && blockIdx.x == gridDim.x - 1) {
        tail_block(v.self);
    } else if (!SKIP && (int)blockIdx.x >= nS) {
        if (v.genBlocks > 0) gen_pair_stream<256, TAIL>(v, blockIdx.x - nS, v.genBlocks);
        else gen_pair_body<256, TAIL, true>(v, blockIdx.x - nS);
    } else {
        if ((int)blockIdx.x < nS) {
            const int left = sto_lean_body<LPS, NCH, TAIL, false, false, FULLT>(v, blockIdx.x, v.st->halt);
            if (left >= 0) sto_cold_body<LPS, NCH, false, TAIL, FULLT>(v, blockIdx.x, left);
        } else {
            if (v.st->halt) return;
            gen_pair_skip_body<256, TAIL>(v, blockIdx.x - nS);
        }
    }
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 8192 * 4) g_timeline[32768 + 2 * blockIdx.x + 1] = wall_clock64();
#endif
}

}  // namespace dopf
#include "agents_persist.h"
namespace dopf {

int debug_timeline(unsigned long long *out, int n)
{
#if defined(DOPF_STATS) || defined(DOPF_BLOCK_STAMPS)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
#else
    (void)out; (void)n;
    return -1;
#endif
}

bool sto_config_supported(int T, Launch *lc)
{
    // lane group x consecutive timesteps per lane; 3 timesteps per lane keeps the kernel at 2 waves/SIMD
    if (T <= 24) { lc->stoLPS = 8; lc->stoNCH = (T + 7) / 8; return true; }
    if (T <= 48) { lc->stoLPS = 16; lc->stoNCH = 3; return true; }
    if (T <= 96) { lc->stoLPS = 32; lc->stoNCH = 3; return true; }
    if (T <= 192) { lc->stoLPS = 64; lc->stoNCH = 3; return true; }
    if (T <= 384) { lc->stoLPS = 64; lc->stoNCH = 6; return true; }
    if (T <= 512) { lc->stoLPS = 64; lc->stoNCH = 8; return true; }
    return false;
}

template <int LPS, int NCH>
static void launch_sto_t(const DevView &v, hipStream_t s)
{
    if (v.use_warm && v.L == 0) {            // (NCH <= 3 whenever the warm start is on)
        constexpr int NC = NCH <= 3 ? NCH : 3;
#ifdef DOPF_NO_FULLT
        const bool full = false;
#else
        const bool full = v.T == LPS * NC;
#endif
        if (v.stoLean) {
            if (v.tail) { if (full) hipLaunchKernelGGL((k_sto_l<LPS, NC, true, true>), dim3(v.nStoItems + 1), dim3(256), 0, s, v);
                          else hipLaunchKernelGGL((k_sto_l<LPS, NC, true, false>), dim3(v.nStoItems + 1), dim3(256), 0, s, v); }
            else { if (full) hipLaunchKernelGGL((k_sto_l<LPS, NC, false, true>), dim3(v.nStoItems), dim3(256), 0, s, v);
                   else hipLaunchKernelGGL((k_sto_l<LPS, NC, false, false>), dim3(v.nStoItems), dim3(256), 0, s, v); }
            return;
        }
        if (v.tail) {
            if (full) hipLaunchKernelGGL((k_sto<LPS, NC, false, true, true>), dim3(v.nStoItems + 1), dim3(256), 0, s, v);
            else hipLaunchKernelGGL((k_sto<LPS, NC, false, true, false>), dim3(v.nStoItems + 1), dim3(256), 0, s, v);
        } else {
            if (full) hipLaunchKernelGGL((k_sto<LPS, NC, false, false, true>), dim3(v.nStoItems), dim3(256), 0, s, v);
            else hipLaunchKernelGGL((k_sto<LPS, NC, false, false, false>), dim3(v.nStoItems), dim3(256), 0, s, v);
        }
        return;
    }
    // with lines the two kernels stay apart: fused, the warm part runs 40 % slower (255 VGPRs, measured)
    if (v.use_warm) {
        if (v.stoLean && v.L > 0) hipLaunchKernelGGL((k_sto_warm<LPS, (NCH <= 3 ? NCH : 3), true, true>), dim3(v.nStoItems), dim3(256), 0, s, v);
        else hipLaunchKernelGGL((k_sto_warm<LPS, (NCH <= 3 ? NCH : 3), true, false>), dim3(v.nStoItems), dim3(256), 0, s, v);
    }
    if (v.use_warm && v.L > 0 && v.coldInWarm) return;            // the warm kernel has called the scan body where needed
    if (v.L > 0) hipLaunchKernelGGL((k_sto_update<LPS, NCH, true>), dim3(v.nStoItems), dim3(256), 0, s, v);
    else hipLaunchKernelGGL((k_sto_update<LPS, NCH, false>), dim3(v.nStoItems), dim3(256), 0, s, v);
}

template <int LPS, int NCH>
static void launch_agents_t(const DevView &v, hipStream_t s)
{
    const dim3 grid(v.nStoItems + (v.genBlocks > 0 && !v.genSkip ? v.genBlocks : v.nGenItems) + (v.tail ? 1 : 0));
#ifdef DOPF_NO_FULLT
    const bool fullA = false;
#else
    const bool fullA = v.T == LPS * NCH;
#endif
    if (v.stoLean) {
#define DOPF_AGL(SKIP_, TAIL_) { if (fullA) hipLaunchKernelGGL((k_agents_l<LPS, NCH, SKIP_, TAIL_, true>), grid, dim3(256), 0, s, v); \
                               else hipLaunchKernelGGL((k_agents_l<LPS, NCH, SKIP_, TAIL_, false>), grid, dim3(256), 0, s, v); }
        if (v.tail) { if (v.genSkip) DOPF_AGL(true, true) else DOPF_AGL(false, true) }
        else { if (v.genSkip) DOPF_AGL(true, false) else DOPF_AGL(false, false) }
#undef DOPF_AGL
        return;
    }
#define DOPF_AG(SKIP_, TAIL_) { if (fullA) hipLaunchKernelGGL((k_agents<LPS, NCH, SKIP_, TAIL_, true>), grid, dim3(256), 0, s, v); \
                              else hipLaunchKernelGGL((k_agents<LPS, NCH, SKIP_, TAIL_, false>), grid, dim3(256), 0, s, v); }
    if (v.tail) {
        if (v.genSkip) DOPF_AG(true, true) else DOPF_AG(false, true)
    } else {
        if (v.genSkip) DOPF_AG(true, false) else DOPF_AG(false, false)
    }
#undef DOPF_AG
}

void launch_agents_fused(const DevView &v, const Launch &lc, hipStream_t s)
{
#define DOPF_CASE(LPS_, NCH_) if (lc.stoLPS == LPS_ && lc.stoNCH == NCH_) { launch_agents_t<LPS_, NCH_>(v, s); return; }
    DOPF_CASE(8, 1) DOPF_CASE(8, 2) DOPF_CASE(8, 3)
    DOPF_CASE(16, 3)
    DOPF_CASE(32, 3)
    DOPF_CASE(64, 3)
#undef DOPF_CASE
}

void launch_agents_persist(const DevView &v, const Launch &lc, hipStream_t s)
{
    const dim3 grid(v.nStoItems + v.genBlocks + 1);
#define DOPF_CASE(LPS_, NCH_) if (lc.stoLPS == LPS_ && lc.stoNCH == NCH_) { hipLaunchKernelGGL((k_agents_p<LPS_, NCH_>), grid, dim3(256), 0, s, v); return; }
    DOPF_CASE(8, 1) DOPF_CASE(8, 2) DOPF_CASE(8, 3)
    DOPF_CASE(16, 3)
    DOPF_CASE(32, 3)
#undef DOPF_CASE
}

void launch_net_agents(const DevView &v, const Launch &lc, hipStream_t s)
{
    const dim3 grid(v.nStoItems + v.nGenItems);
#define DOPF_CASE(LPS_, NCH_) if (lc.stoLPS == LPS_ && lc.stoNCH == NCH_) { \
        if (v.stoLean) hipLaunchKernelGGL((k_net_agents<LPS_, NCH_, true>), grid, dim3(256), 0, s, v); \
        else hipLaunchKernelGGL((k_net_agents<LPS_, NCH_, false>), grid, dim3(256), 0, s, v); return; }
    DOPF_CASE(8, 1) DOPF_CASE(8, 2) DOPF_CASE(8, 3)
    DOPF_CASE(16, 3)
    DOPF_CASE(32, 3)
    DOPF_CASE(64, 3)
#undef DOPF_CASE
}

void launch_sto_update(const DevView &v, const Launch &lc, hipStream_t s)
{
    if (v.nStoItems == 0) return;
#define DOPF_CASE(LPS_, NCH_) if (lc.stoLPS == LPS_ && lc.stoNCH == NCH_) { launch_sto_t<LPS_, NCH_>(v, s); return; }
    DOPF_CASE(8, 1) DOPF_CASE(8, 2) DOPF_CASE(8, 3)
    DOPF_CASE(16, 3)
    DOPF_CASE(32, 3)
    DOPF_CASE(64, 3) DOPF_CASE(64, 6) DOPF_CASE(64, 8)
#undef DOPF_CASE
}

}  // namespace dopf

// agents_persist.h — several ADMM iterations of a copper plate in ONE launch (round 4 experiment: DOPF_F_PERSIST).
//
// The one-launch iteration (kernels_agents.hip: k_agents / k_agents_l, the tail block) still pays, per iteration, a kernel
// boundary and a launch ramp (~2.9 us of the 25 us of config2 in a 16-kernel graph replay), and every block's first loads
// wait for that boundary although only the PRICES depend on the previous iteration. Here the grid stays: every block runs
// its share of iteration k, k+1, ... (replacing run!'s loop over calculate_iteration!, reference src/optimization/run.jl:1-16,
// for K iterations per launch):
//   * the tail block (the grid's last) finishes iteration k as before — waits for the blocks' sums in the fixed-point
//     accumulators, dual step (src/optimization/update_duals.jl:8-13), stop test (src/optimization/convergence.jl:1-31) —
//     and then PUBLISHES: lambda / imbalance / price of every timestep as agent-scope stores, every storing wave drained
//     (s_waitcnt vmcnt(0)), a block barrier, then the sequence word Status::pseq = count of published updates | the halt word
//     (one agent-scope store);
//   * every other block issues the price-independent loads of its next share (generator blocks: the rows of their first
//     item; storage blocks: the rows of their first pass), then one lane polls Status::pseq (agent-scope loads, s_sleep
//     between polls, bounded by wall clock), and the block reads the prices with agent-scope loads — past the vector L1 and
//     the scalar cache, which are only refreshed between launches;
//   * the accumulator sets alternate with the sequence number (the set of iteration k is zeroed by the tail block while it
//     waits for iteration k + 1, as before).
// Needs every block resident at once (a block that is not cannot add its sums, the tail cannot publish, nobody advances):
// dopf_create offers it only for grids sized that way (3 blocks of 256 threads per CU: config1, config2), and every wait is
// bounded (Status::tail_timeout, DOPF_E_DEVICE) — the kernel always ends. A block's own rows (P / D / C / the stored prices of
// its storages) are read and written by that block only: same CU, same L1.
#pragma once

namespace dopf {

constexpr unsigned long long kPersistWaitTicks = 200000000ull;      // 2 s of the 100 MHz wall clock

// The tail of iteration `seq` (tail_block's arithmetic; one node, no lines, no exchange), then the hand-over to the other blocks.
// Returns the halt word it has published.
__device__ __forceinline__ int persist_tail(const DevView &v, const TailView &tv, const int par, const unsigned seq_now)
{
    __shared__ double wmaxP[8];
    __shared__ int badP, haltP;
    const int tid = threadIdx.x, T = v.T, nth = (int)blockDim.x;
    constexpr int HR = kAccRep / 2;
    const int expect = tv.expect;
    StatusPre spre{};
    if (tid == 0) { spre = status_load(v); badP = 0; }
    {   // the other set: used by the previous iteration, read by its tail, not touched by this one
        unsigned long long *z = reinterpret_cast<unsigned long long *>(tv.acc) + (size_t)(par ^ 1) * kAccRep * tv.accStride;
        for (int i = tid; i < kAccRep * tv.accStride; i += nth) __hip_atomic_store(z + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long tstart = wall_clock64();
    __syncthreads();
    double rl = 0.0;
    const int half = tid & 1;
    for (int t0 = 0; t0 <= T; t0 += nth / 2) {          // slot T = cost
        const int t = t0 + (tid >> 1);
        const bool act = t <= T;
        const int tc = t < T ? t : 0;
        const double dem = v.demand[tc], lam_old = p_ld(v.lam + tc), s_old = p_ld(v.s + tc);
        const unsigned long long *base = reinterpret_cast<const unsigned long long *>(tv.acc) +
                                         ((size_t)par * kAccRep + (size_t)half * HR) * tv.accStride + (act ? t : 0);
        long long isum = 0;
        bool ok = !act;
        for (unsigned round = 1; __any(!ok); ++round) {
            unsigned long long x[HR];
#pragma unroll
            for (int r = 0; r < HR; ++r)
                x[r] = __hip_atomic_load(base + (size_t)r * tv.accStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned cnt = 0;
            long long sum = 0;
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                cnt += (unsigned)(x[r] & kAccCntMask);
                sum += (long long)(x[r] & ~kAccCntMask) >> kAccCntBits;
            }
            cnt += __shfl_xor(cnt, 1);
            sum += __shfl_xor(sum, 1);
            if (!ok && (int)cnt == expect) { ok = true; isum = sum; }
            if ((round & 255u) == 0u && wall_clock64() - tstart > kPersistWaitTicks) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) atomicOr(&badP, 1);
        if (!ok || !act || half) continue;
        const double xsum = t < T ? (double)isum * tv.invInj : (double)isum * tv.invCost;
        if (t < T) {
            const double xi = xsum - dem;                         // results.jl:58-100 (one node: imbalance = its injection)
            v.cons[t] = xsum;
            v.inj[t] = xi;
            v.s_used[t] = s_old;
            const double ln = lam_old + v.gamma * xi;             // update_duals.jl:8-13
            v.lam_used[t] = lam_old;
            p_st(v.s + t, xi);                                    // what the other blocks read next: agent scope
            p_st(v.lam + t, ln);
            p_st(v.price + t, ln);                                // no lines: the nodal price is lambda
            rl = fmax(rl, fabs(ln - lam_old));
        } else {
            v.cons[T] = xsum;
            v.st->total_cost = xsum;
        }
    }
    for (int d = 32; d > 0; d >>= 1) rl = fmax(rl, __shfl_xor(rl, d));
    if ((tid & 63) == 0) wmaxP[tid >> 6] = rl;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's published words have left
    __syncthreads();
    if (tid == 0) {
        int h = 1;
        if (badP) { v.st->tail_timeout = 1; v.st->halt = 1; }       // (sticky; the host reports DOPF_E_DEVICE)
        else {
            double r0 = 0.0;
            for (int q = 0; q < nth / 64; ++q) r0 = fmax(r0, wmaxP[q]);
            v.st->tail_par = par ^ 1;
            status_update(v, spre, r0, 0.0, 0.0);
            h = v.st->halt;                                       // (this lane wrote it)
        }
        // the sequence word carries the halt word: one round trip for the blocks that wait (count in the low 30 bits)
        p_sti(&v.st->pseq, (int)(((seq_now + 1u) & kPersistSeqMask) | (h ? kPersistHaltBit : 0u)));
        haltP = h;
    }
    __syncthreads();
    return haltP;
}

// a streaming generator block over K iterations: gen_pair_stream's loop, the first item's rows loaded BEFORE the wait
template <int BS>
__device__ __forceinline__ void persist_gen(const DevView &v, const int first, const int stride, const int K, const unsigned s0, const int par0)
{
    constexpr int GU = kGenStreamRows;
    __shared__ double red[2][BS];
    __shared__ double wc[BS / 64];
    __shared__ int goP;
    const int T = v.T, N = v.N, TT = v.genTT2, R = v.genR2, nI = v.nGenItems, chunk = v.genChunk, G = v.G;
    const int tid = threadIdx.x;
    const int r = tid / TT, tt = tid - r * TT;
    const bool rowlane = r < R;
    const double gam = v.gamma, inv = 1.0 / (v.w_prox + gam);
    const int t2c = 2 * tt;
    double2 *P2 = reinterpret_cast<double2 *>(v.P);
    const size_t half = (size_t)(T >> 1);
    const bool any = first < nI;
    Status *st = v.st;
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
    for (int k = 0; k < K; ++k) {
        double2 pa[GU], pb[GU];
        double mca[GU], pma[GU], mcb[GU], pmb[GU];
        int i = any ? first : 0;
        // the block's own stores of the previous iteration to these rows have been issued by these same lanes: in order
        DOPF_GEN_LOAD(i, pa, mca, pma)
        if (tid == 0) {
            int go = p_ldi(&st->halt) ? 0 : 1;                       // (k = 0: the launch's own look at the halt word)
            if (k > 0) {
                const unsigned w = persist_wait_word(st, s0 + (unsigned)k);
                go = w == 0xffffffffu ? -1 : ((w & kPersistHaltBit) ? 0 : 1);
            }
            goP = go;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // (only the LDS word crosses: the rows stay in flight)
        const int go = goP;
        if (go <= 0) {
            if (go < 0 && tid == 0) st->tail_timeout = 1;
            return;                                                              // halted (or lost): nothing stored, nothing added
        }
        const double sh0 = fma(gam, p_ld(v.s + t2c), p_ld(v.price + (size_t)N * t2c)) * inv;
        const double sh1 = fma(gam, p_ld(v.s + t2c + 1), p_ld(v.price + (size_t)N * (t2c + 1))) * inv;
        double acc0 = 0.0, acc1 = 0.0, cost = 0.0;
        if (any)
            for (;;) {
                const int j = i + stride;
                DOPF_GEN_LOAD(min(j, nI - 1), pb, mcb, pmb)
                DOPF_GEN_WORK(i, pa, mca, pma)
                if (j >= nI) break;
                const int kk = j + stride;
                DOPF_GEN_LOAD(min(kk, nI - 1), pa, mca, pma)
                DOPF_GEN_WORK(j, pb, mcb, pmb)
                if (kk >= nI) break;
                i = kk;
            }
        gen_pair_sums<BS, true, false, true>(v, first, tid, r, tt, acc0, acc1, cost, red, wc, (par0 + k) & 1);
        __syncthreads();                                                         // (red / wc / goP are reused by the next iteration)
    }
#undef DOPF_GEN_LOAD
#undef DOPF_GEN_WORK
}

// K iterations of the fused copper-plate launch (k_agents_l's grid: storage items, streaming generator blocks, the tail block)
template <int LPS, int NCH>
__global__ __launch_bounds__(256, 3) void k_agents_p(DevView v)
{
    const int nS = v.nStoItems, K = v.persistIters;
    Status *st = v.st;
    // The sequence word and the accumulator set as this launch finds them: the tail block cannot advance either before EVERY
    // block has added its sums of the launch's first iteration, i.e. before every block has read them here.
    const unsigned s0 = (unsigned)p_ldi(reinterpret_cast<const int *>(&st->pseq)) & kPersistSeqMask;
    const int par0 = p_ldi(&st->tail_par);
    if (blockIdx.x == gridDim.x - 1) {
        if (p_ldi(&st->halt)) return;
        const TailView tv = *v.tailDev;
        for (int k = 0; k < K; ++k)
            if (persist_tail(v, tv, (par0 + k) & 1, s0 + (unsigned)k)) return;
        return;
    }
    if ((int)blockIdx.x >= nS) {
        persist_gen<256>(v, blockIdx.x - nS, v.genBlocks, K, s0, par0);
        return;
    }
    if (p_ldi(&st->halt)) return;                // (the launch's own look at the halt word; later iterations learn it from the sequence word)
    for (int k = 0; k < K; ++k) {
        const int par = (par0 + k) & 1;
        // (the wait for iteration k's prices happens inside, behind the first pass's row loads)
        const int left = sto_lean_body<LPS, NCH, true, true>(v, blockIdx.x, 0, par, k == 0 ? 0u : (s0 + (unsigned)k) | 0x40000000u);
        if (left < 0) return;
        if (left > 0) sto_cold_body<LPS, NCH, false, true, true, true>(v, blockIdx.x, left, par);
        __syncthreads();                         // (the bodies' LDS is reused by the next iteration)
    }
}

}  // namespace dopf

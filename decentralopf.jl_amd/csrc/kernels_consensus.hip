// kernels_consensus.hip — everything between two x-updates (gfx950, fp64).
//
//   k_tables   per (n,t): breakpoint table of Psi_{n,t} (only L > 0)                 [DESIGN.md]
//   k_slack    per (n,t): sum over the node's agents of the closed-form slacks U, K  (only L > 0)
//              = `result.avg_U += result_unit.U` of Result(...), src/structures/results.jl:83-84
//   k_reduce   per-item partials -> consensus vector [inj sums | sum U | sum K | cost]
//              = the agent loop of Result(...), results.jl:72-106, in a fixed summation order
//   k_dual     injection, avg_U/avg_K, line_utilization (results.jl:108-116); lambda/mu/rho steps with
//              the slack mask (src/optimization/update_duals.jl:1-39); |dual change| inf-norms
//   k_price    price[n,t] = lambda_t + sum_l ptdf[l,n] (mu - rho)[l,t] for the next x-update
//              (src/optimization/subproblems.jl:67-74) and the stop test of check_convergence!
//              (src/optimization/convergence.jl:1-31)
//   k_derive_* rebuild the consensus state from a primal state handed in by dopf_set_state
#include <algorithm>

#include <atomic>

#include "dopf_internal.h"

namespace dopf {

// rows of the PTDF matrix a thread keeps in flight in the dot products of the network kernels (every sum stays in index
// order: the batch size does not change a bit of the result)
#ifndef DOPF_PTDF_FLIGHT
#define DOPF_PTDF_FLIGHT 16
#endif
constexpr int kFlight = DOPF_PTDF_FLIGHT;

__device__ __forceinline__ double dmax0(double a) { return a > 0.0 ? a : 0.0; }

// deterministic block sum (256 threads), result broadcast to all threads
__device__ __forceinline__ double block_sum256(double x, double *sh)
{
    const int tid = threadIdx.x;
    __syncthreads();
    sh[tid] = x;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] += sh[tid + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// network part of Psi at offset dl for line l of node n (closed-form slacks inside)
__device__ __forceinline__ double line_term(const DevView &v, int l, int t, double h, double dl)
{
    const double w2 = 2.0 * v.w_flow, g = v.gamma;
    const double f = v.flow[l + v.L * t] + h * dl, F = v.fmax[l];
    const double U = dmax0((g * v.avgU[l + v.L * t] - w2 * (f - F)) / (w2 + g));
    const double K = dmax0((g * v.avgK[l + v.L * t] + w2 * (f + F)) / (w2 + g));
    return w2 * h * ((f + U - F) - (K - f - F));
}

// ------------------------------------------------------------------------------------------------
// breakpoint tables (L > 0): one wave per (n,t)
// ------------------------------------------------------------------------------------------------
//
// Psi_{n,t}(dlt) is piecewise linear with up to 2L kinks, but an agent at node n can only move its injection by
// |dlt| <= W_n (a generator by pmax, a storage by 2 pmax): kinks left of -W_n only add their slope jumps to the
// slope at the window's left end, kinks right of +W_n are never reached. Typically a handful of the 2L kinks lie
// inside the window, so the table that the agent kernels search has a handful of entries, and building it is a
// classification pass + a rank sort of the few survivors instead of a 512-key bitonic sort per (n,t).
// Psi is anchored at dlt = 0 (direct evaluation) and walked outwards piece by piece, which keeps full precision
// where the agents' steps live. Every sum has a fixed order (wave butterflies, list order).
__device__ __forceinline__ double wave_sum64(double x)
{
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

// one wave builds the table of (n, t); shm: 4 * 2L + 1 doubles of LDS of the wave's own
__device__ __forceinline__ void build_table(const DevView &v, const int n, const int t, double *shm)
{
    const int N = v.N, L = v.L, M2 = v.M2;
    double *key = shm, *jmp = shm + M2, *skey = shm + 2 * M2, *sslope = shm + 3 * M2;   // sslope: M2 + 1
    const int lane = threadIdx.x & 63;
    const size_t at = (size_t)n + (size_t)N * t;
    const double w2 = 2.0 * v.w_flow, g = v.gamma, act = g / (w2 + g);
    const double W = v.node_win[n];

    // ---- classify the 2L candidate kinks; keep the ones inside [-W, W] in list order
    double s0part = 0.0, left = 0.0, pz = 0.0;
    int c = 0;
    for (int base = 0; base < M2; base += 64 * 4) {          // four passes of 64 candidates, their loads issued together
        double hh[4], ff[4], FF[4], au[4], ak[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + 64 * u + lane;
            const int l = (i < M2 ? i : 0) >> 1;
            hh[u] = v.ptdf[l + L * n]; ff[u] = v.flow[l + L * t]; FF[u] = v.fmax[l];
            au[u] = v.avgU[l + L * t]; ak[u] = v.avgK[l + L * t];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + 64 * u + lane;
            if (base + 64 * u >= M2) break;                    // (wave-uniform)
            double kv = INFINITY, jv = 0.0;
            if (i < M2) {
                const double h = hh[u];
                if (h != 0.0) {
                    const double f = ff[u], F = FF[u];
                    const double dj = w2 * h * h * (1.0 - act);
                    if ((i & 1) == 0) {          // U switches: active where h*dlt < ...
                        kv = (g * au[u] / w2 - f + F) / h;
                        jv = h > 0.0 ? dj : -dj;
                        s0part += w2 * h * h * (1.0 + act);     // at -inf exactly one of U, K is active
                        // Psi(0), network part (line_term at dl = 0 with the values already loaded)
                        const double U = dmax0((g * au[u] - w2 * (f - F)) / (w2 + g));
                        const double K = dmax0((g * ak[u] + w2 * (f + F)) / (w2 + g));
                        pz += w2 * h * ((f + U - F) - (K - f - F));
                    } else {                     // K switches
                        kv = (-g * ak[u] / w2 - f - F) / h;
                        jv = h > 0.0 ? -dj : dj;
                    }
                }
            }
            const bool fin = kv < INFINITY;                      // (also false for NaN)
            if (fin && kv < -W) left += jv;
            const bool in = fin && kv >= -W && kv <= W;
            const unsigned long long mask = __ballot(in);
            if (in) {
                const int pos = c + __popcll(mask & ((1ull << lane) - 1ull));
                key[pos] = kv;
                jmp[pos] = jv;
            }
            c += __popcll(mask);
        }
    }
    const double slope_left = g + wave_sum64(s0part) + wave_sum64(left);    // slope just right of -W
    const double psiZ = v.price[n + N * t] + g * v.s[t] + wave_sum64(pz);
    __builtin_amdgcn_wave_barrier();

    // ---- rank sort (ties by list position), sorted keys and jumps into skey / sslope[1..]
    for (int e = lane; e < c; e += 64) {
        const double ke = key[e];
        int r = 0;
        for (int k = 0; k < c; ++k) {
            const double kk = key[k];
            r += (kk < ke || (kk == ke && k < e)) ? 1 : 0;
        }
        skey[r] = ke;
        sslope[r + 1] = jmp[e];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- slopes: piece 0 = left of the first kept kink, piece j+1 = (kink j, kink j+1)
    double *ob = v.tb_beta + at * M2, *op = v.tb_psi + at * M2, *os = v.tb_slope + at * (M2 + 1);
    int j0 = 0;                                              // number of kept kinks < 0: 0 lies on piece j0
    for (int k = 0; k < c; ++k) j0 += skey[k] < 0.0 ? 1 : 0;
    for (int j = lane; j < c; j += 64) {
        double sl = slope_left;
        for (int k = 0; k <= j; ++k) sl += sslope[k + 1];
        key[j] = sl;                                         // (key[] is free now) slope on piece j+1
    }
    __builtin_amdgcn_wave_barrier();
    // ---- Psi at the kinks, outwards from 0
    const double slope_j0 = j0 == 0 ? slope_left : key[j0 - 1];
    for (int j = lane; j < c; j += 64) {
        double ps;
        if (j >= j0) {
            ps = psiZ + slope_j0 * skey[j0];
            for (int i = j0 + 1; i <= j; ++i) ps += key[i - 1] * (skey[i] - skey[i - 1]);       // slope on piece i = key[i-1]
        } else {
            ps = psiZ + slope_j0 * skey[j0 - 1];
            for (int i = j0 - 2; i >= j; --i) ps -= key[i] * (skey[i + 1] - skey[i]);            // slope on piece i+1 = key[i]
        }
        ob[j] = skey[j];
        op[j] = ps;
        os[j + 1] = key[j];
    }
    if (lane == 0) {
        os[0] = slope_left;
        v.tb_m[at] = c;
        v.tb_psi0[at] = psiZ;
    }
    __builtin_amdgcn_wave_barrier();         // (the wave's LDS scratch is free for its next table)
}

__global__ __launch_bounds__(64) void k_tables(DevView v)
{
    if (v.st->halt) return;
    extern __shared__ double shm[];
    const size_t at = blockIdx.x;
    const int n = (int)(at % v.N), t = (int)(at / v.N);
    if (v.tab_skip[t]) return;               // linear inside every window: the price kernel wrote Psi(0) and the slope
    build_table(v, n, t, shm);
}

// A function attribute belongs to the CURRENT device: "raised once per process" leaves every other device of a dopf_multi_* run
// (or of a process with contexts on several GPUs) at the default limit, and the launch fails there. One bit per device, under a
// lock (for_each_shard launches from one host thread per shard).
static bool first_time_on_this_device(std::atomic<unsigned long long> &mask)
{
    int dev = 0;
    hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    return (mask.fetch_or(bit) & bit) == 0ull;
}

void launch_tables(const DevView &v, hipStream_t s)
{
    if (v.L == 0 || v.tablesInDual) return;  // (tablesInDual: the dual/price kernel builds the tables of its timestep itself)
    const size_t shm = (size_t)(4 * v.M2 + 1) * sizeof(double);
    static std::atomic<unsigned long long> big_lds{0ull};
    if (shm > 64 * 1024 && first_time_on_this_device(big_lds))     // worst case (every kink inside the window) needs 4 * 2L doubles
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_tables), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_tables, dim3(v.N * v.T), dim3(64), shm, s, v);
}

// ------------------------------------------------------------------------------------------------
// slack sums (L > 0)
// ------------------------------------------------------------------------------------------------
// sum over the agents a of node n of U_a[l,t] = max(0, aU - kap d_a) (and K_a = max(0, aK + kap d_a)), d_a = the agent's
// injection change, |d_a| <= W_n. With reach = |kap| W_n three cases per (n, l, t):
//     aU - reach >= 0   every agent's slack is active:   sum = n_a aU - kap sum_a d_a      (closed form from the node's sum)
//     aU + reach <= 0   none is:                         sum = 0
//     otherwise         the agents are walked one by one (k_slack, agents staged in LDS) and the sum is stored
// Only the third kind — a line whose switch point lies inside the window of what the node's agents can do — travels
// through memory (part_U / part_K); k_reduce re-derives the case with the same arithmetic and adds the closed forms
// itself from the N x T node sums. (Round 1 wrote all N x L x T entries and read them back: 130 MB per iteration on
// the 118-node share against 40 MB of algorithmic traffic.)
struct SlackCase {
    double aU, aK, kap, reach;
};

__device__ __forceinline__ SlackCase slack_case(double g, double w2, double inv, double h, double f, double F, double cu, double ck, double W)
{
#pragma clang fp contract(off)          // the two kernels must classify identically: no fused multiply-adds here
    SlackCase c;
    c.aU = (g * cu - w2 * (f - F)) * inv;
    c.aK = (g * ck + w2 * (f + F)) * inv;
    c.kap = w2 * h * inv;
    c.reach = fabs(c.kap) * W;
    return c;
}

// sum_a max(0, base + k d_a) over the node's agents at one timestep: the staged ones from LDS, eight reads in
// flight, the rest (nodes with more agents than the tile holds) from memory; list order
__device__ __forceinline__ double walk_sum(const DevView &v, const double *dtile, int tt, int t, int T, int gb, int ng,
                                           int sb, int na, int cap, double base, double k)
{
    double acc = 0.0;
    const int nl = na < cap ? na : cap;
    for (int a0 = 0; a0 < nl; a0 += 8) {
        double d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = dtile[(a0 + u < nl ? a0 + u : a0) * 33 + tt];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += a0 + u < nl ? dmax0(base + k * d[u]) : 0.0;
    }
    for (int a = nl; a < na; ++a)
        acc += dmax0(base + k * (a < ng ? v.dltG[(size_t)(gb + a) * T + t] : v.dltS[(size_t)(sb + a - ng) * T + t]));
    return acc;
}

// lane r of eight: rows i0 + r, i0 + r + 8, ... (below i1) of a node's partial rows at timestep t — generator items, then the
// storage items' scan partials, then their warm-start partials. Four loads in flight, no branch between them (one select
// on the address); the grouping is fixed (k_reduce level 1, and k_slack when it stores the node sums itself: same bits)
// Networks keep the rows transposed (DevView::part_T, [t][row]: the node's rows side by side): where row jj of the node sits there
__device__ __forceinline__ int row_pos(int jj, int ngi, int nsi)
{
    return jj < ngi ? jj : (jj < ngi + nsi ? ngi + 2 * (jj - ngi) : ngi + 2 * (jj - ngi - nsi) + 1);
}
__device__ __forceinline__ double rows_sum(const DevView &v, int g0, int ngi, int s0, int nsi, int i0, int i1, int r, int t)
{
    constexpr int R = 8;
    const int T = v.T;
    double acc = 0.0;
    const double *rowsT = v.part_T ? v.part_T + (size_t)t * v.rowsT : nullptr;      // (uniform choice)
    const int base = g0 + 2 * s0;
    for (int i = i0 + r; i < i1; i += 4 * R) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = i + u * R;
            const int jj = j < i1 ? j : i0;                               // in range: always a valid row
            double val;
            if (rowsT) val = rowsT[v.pos_of_row[base + row_pos(jj, ngi, nsi)]];
            else {
                const double *row = jj < ngi ? v.part_ginj + (size_t)(g0 + jj) * T
                                  : (jj < ngi + nsi ? v.part_sinj + (size_t)(s0 + jj - ngi) * T
                                                    : v.part_sinj_w + (size_t)(s0 + jj - ngi - nsi) * T);
                val = row[t];
            }
            x[u] = j < i1 ? val : 0.0;
        }
        acc += (x[0] + x[1]) + (x[2] + x[3]);
    }
    return acc;
}

// total cost of the iteration from the items' partials, by one block of 256 (fixed order)
__device__ __forceinline__ double cost_rows_sum(const DevView &v, int c0, int c1, int ngr, double *red)
{
    double c = 0.0;
    for (int i = c0 + (int)threadIdx.x; i < c1; i += 256)
        c += i < ngr ? v.part_gcost[i] : v.part_scost[i - ngr] + v.part_scost_w[i - ngr];
    return block_sum256(c, red);
}

__global__ __launch_bounds__(256) void k_slack(DevView v, const int cap)
{
    if (v.st->halt) return;
    constexpr int TS = 32;                   // timesteps per block: the agents' changes are read along t (coalesced)
    extern __shared__ double dtile[];        // [a * 33 + tt] for the node's first `cap` agents (padded rows)
    const int tid = threadIdx.x;
    const int N = v.N, L = v.L, T = v.T;
    const int TC = (T + TS - 1) / TS;
    const int n = blockIdx.x / TC, t0 = (blockIdx.x - n * TC) * TS;
    const int gb = v.node_gen_beg[n], ng = v.node_gen_beg[n + 1] - gb;
    const int sb = v.node_sto_beg[n], ns = v.node_sto_beg[n + 1] - sb;
    const int na = ng + ns;
    const double w2 = 2.0 * v.w_flow, g = v.gamma, inv = 1.0 / (w2 + g);
    const double W = v.node_win[n];
    const int nt = min(TS, T - t0), np = L * nt;
    {   // the node's injection sum of this iteration (its items' partial rows, read along t: coalesced, k_reduce's order and
        // bits) and its change against the previous iteration's — what the closed-form slack sums need
        const int r = tid >> 5, tt = tid & 31, t = t0 + tt;
        __shared__ double redn[256];
        const size_t at = (size_t)n + (size_t)N * (t < T ? t : 0);
        const double was = (r == 0 && t < T) ? v.prev_node[at] : 0.0;          // (on its way while the rows are read)
        const int g0 = v.node_gitem_beg[n], ngi = v.node_gitem_beg[n + 1] - g0, s0 = v.node_sitem_beg[n], nsi = v.node_sitem_beg[n + 1] - s0;
        {   // (transposed rows: the eight row lanes of a timestep side by side in a wave — they read neighbouring words)
            const int rr = v.part_T ? (tid & 7) : r, tr = v.part_T ? (tid >> 3) : tt;
            redn[rr * 32 + tr] = t0 + tr < T ? rows_sum(v, g0, ngi, s0, nsi, 0, ngi + 2 * nsi, rr, t0 + tr) : 0.0;
        }
        __syncthreads();
        if (r == 0 && t < T) {
            double sum = 0.0;
            for (int q = 0; q < 8; ++q) sum += redn[q * 32 + tt];
            v.node_dsum[at] = sum - was;
            v.prev_node[at] = sum;
            if (v.slackInDual) v.cons[at] = sum;        // no k_reduce launch behind this one: the node's sum leaves from here
        }
        if (v.slackInDual && blockIdx.x == 0) {          // ... and the cost
            const int ngr = v.genRows > 0 ? v.genRows : v.nGenItems;
            const double c = cost_rows_sum(v, 0, ngr + v.nStoItems, ngr, redn);
            if (tid == 0) v.cons[(size_t)N * T + 2 * (size_t)L * T] = c;
        }
    }
    {   // usually no (line, timestep) of this chunk has a switch point within anybody's reach: k_reduce takes the closed forms
        int any = 0;
        if (tid < nt) any = v.walk_any[t0 + tid];
        if (!__syncthreads_or(any)) return;
    }
    // the node's injection changes of the chunk's timesteps, staged in LDS for the walks (read along t: coalesced)
    {
        const int r = tid >> 5, tt = tid & 31, t = t0 + tt;
        if (t < T)
            for (int a0 = r; a0 < na && a0 < cap; a0 += 32) {         // four rows in flight per lane
                double d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int a = a0 + 8 * u;
                    const int aa = a < na ? a : r;            // (a valid row; value dropped below)
                    d[u] = aa < ng ? v.dltG[(size_t)(gb + aa) * T + t] : v.dltS[(size_t)(sb + aa - ng) * T + t];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int a = a0 + 8 * u;
                    if (a < na && a < cap) dtile[a * 33 + tt] = d[u];
                }
            }
        __syncthreads();
    }
    for (int p0 = tid; p0 < np; p0 += 4 * 256) {          // pairs (l fastest: coalesced), four per lane in flight
        double h[4], f[4], F[4], cu[4], ck[4];
        int ls[4], tts[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + 256 * u < np ? p0 + 256 * u : tid;      // (a valid pair; result dropped below)
            const int tt = p / L, l = p - tt * L, t = t0 + tt;
            ls[u] = l; tts[u] = tt;
            h[u] = v.ptdf[l + L * n]; f[u] = v.flow[l + L * t]; F[u] = v.fmax[l];
            cu[u] = v.avgU[l + L * t]; ck[u] = v.avgK[l + L * t];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + 256 * u >= np) continue;
            const int l = ls[u], tt = tts[u], t = t0 + tt;
            if (!v.walk_flag[l + L * t]) continue;            // k_reduce takes the dot-product form
            const SlackCase c = slack_case(g, w2, inv, h[u], f[u], F[u], cu[u], ck[u], W);
            const size_t at = (size_t)n + (size_t)N * t;
            // (closed-form and all-zero cases: k_reduce adds them from the node changes)
            if (!(c.aU - c.reach >= 0.0) && c.aU + c.reach > 0.0)
                v.part_U[at * L + l] = walk_sum(v, dtile, tt, t, T, gb, ng, sb, na, cap, c.aU, -c.kap);
            if (!(c.aK - c.reach >= 0.0) && c.aK + c.reach > 0.0)
                v.part_K[at * L + l] = walk_sum(v, dtile, tt, t, T, gb, ng, sb, na, cap, c.aK, c.kap);
        }
    }
}

void launch_slack(const DevView &v, hipStream_t s)
{
    if (v.L == 0) return;
    const int cap = std::min(v.maxNodeAgents, 224);         // 224 agents x 33 doubles = 58 KB (+ 2.3 KB static: under 64 KB)
    hipLaunchKernelGGL(k_slack, dim3(v.N * ((v.T + 31) / 32)), dim3(256), (size_t)cap * 33 * sizeof(double), s, v, cap);
}

// does the slack sum of (line, timestep) need the per-node cases? Not if the offsets clear the line's largest reach on
// either side for U and for K: then every agent of every node has the slack active (sum = dot product) or inactive (0)
__device__ __forceinline__ bool slack_needs_cases(double g, double w2, double inv, double f, double F, double cu, double ck, double R)
{
    const SlackCase c = slack_case(g, w2, inv, 0.0, f, F, cu, ck, 0.0);
    const bool uOk = c.aU - R >= 0.0 || c.aU + R <= 0.0, kOk = c.aK - R >= 0.0 || c.aK + R <= 0.0;
    return !(uOk && kOk);
}

// closed form of a part's contribution when every agent of every node has the slack active (offset a > 0): the two kernels
// that form slack sums must round identically — no contraction left to the compiler
__device__ __forceinline__ double slack_sum_plain(int which, double cnt, double a, double w2inv, double dot)
{
#pragma clang fp contract(off)
    const double x = cnt * a, y = w2inv * dot;
    return which ? x + y : x - y;
}

// The part [nbeg, nend) of the nodes' contributions to the slack sum (which: 0 = U, 1 = K) of a line whose switch point lies
// inside some node's window. Pass 1, the part's nodes in order: closed forms are added at once, the nodes k_slack had to walk
// are remembered in a bit mask (a load whose address hangs on this arithmetic would put two dependent memory round trips
// into every batch); pass 2 fetches the walked sums, eight in flight. Both orders are fixed. (k_reduce, and the one-launch
// dual/price kernel when it forms the slack sums itself: same bits.)
__device__ __forceinline__ double slack_sum_cases(const DevView &v, int which, int l, int t, int nbeg, int nend, const double *sdL,
                                                  const double *winL, const double *naL, double f, double F, double cu, double ck)
{
#pragma clang fp contract(off)
    const int N = v.N, L = v.L;
    const double *src = which ? v.part_K : v.part_U;
    const double w2 = 2.0 * v.w_flow, g = v.gamma, inv = 1.0 / (w2 + g);
    constexpr int MW = 4;                              // mask words: a part of up to 128 nodes (more: fetched inline)
    unsigned wm[MW];
#pragma unroll
    for (int q = 0; q < MW; ++q) wm[q] = 0u;
    double sum = 0.0;
    for (int n0 = nbeg; n0 < nend; n0 += 8) {
        double h[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) h[u] = v.ptdf[l + (size_t)L * (n0 + u < nend ? n0 + u : nend - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = n0 + u < nend ? n0 + u : nend - 1;
            const SlackCase c = slack_case(g, w2, inv, h[u], f, F, cu, ck, winL[n]);
            const double a = which ? c.aK : c.aU;
            const bool live = n0 + u < nend, all_on = a - c.reach >= 0.0, walked = live && !all_on && a + c.reach > 0.0;
            x[u] = (live && all_on) ? (which ? naL[n] * c.aK + c.kap * sdL[n] : naL[n] * c.aU - c.kap * sdL[n]) : 0.0;
            if (walked) {
                const int k = n - nbeg;
                if (k < 32 * MW) {
#pragma unroll
                    for (int q = 0; q < MW; ++q)
                        if (q == (k >> 5)) wm[q] |= 1u << (k & 31);
                } else {
                    x[u] = src[((size_t)n + (size_t)N * t) * L + l];
                }
            }
        }
        sum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
#pragma unroll
    for (int q = 0; q < MW; ++q) {
        unsigned m = wm[q];
        while (m) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x[u] = 0.0;
                if (m) {
                    const int n = nbeg + 32 * q + __builtin_ctz(m);
                    m &= m - 1u;
                    x[u] = src[((size_t)n + (size_t)N * t) * L + l];
                }
            }
            sum += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        }
    }
    return sum;
}

// (the one-launch dual/price kernel's rare case, kept out of line: its registers are not that kernel's problem)
__device__ __attribute__((noinline)) void slack_sum_cases_both(const DevView *self, int l, int t, int nbeg, int nend, const double *sdL,
                                                               const double *winL, const double *naL, double f, double F, double cu,
                                                               double ck, double *pu, double *pk)
{
    const DevView &v = *self;
    *pu = slack_sum_cases(v, 0, l, t, nbeg, nend, sdL, winL, naL, f, F, cu, ck);
    *pk = slack_sum_cases(v, 1, l, t, nbeg, nend, sdL, winL, naL, f, F, cu, ck);
}

// ------------------------------------------------------------------------------------------------
// reduce: RB blocks per node sum slices of the node's item partials (level 1); the block that
// finishes last for a node adds the RB slice sums in slice order (level 2) — the order of every
// addition is fixed, so the result is bitwise reproducible whichever block happens to be last.
// Remaining blocks sum part_U / part_K over nodes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce(DevView v)
{
    // (the halt word is looked at after the loads below have been issued: one round trip, not two; nothing is stored
    // before that)
    const int halt = v.st->halt;
    __shared__ double red[256];
    __shared__ int last_sh;
    const int tid = threadIdx.x;
    const int N = v.N, L = v.L, T = v.T, RB = v.reduceRB;
    constexpr int TT = 32, R = 8;                       // block = 8 item lanes x 32 timesteps
    const int TC = (T + TT - 1) / TT;
    if ((int)blockIdx.x < N * RB * TC) {
        const int tcx = blockIdx.x % TC, nrb = blockIdx.x / TC;
        const int n = nrb / RB, rb = nrb - n * RB;
        const int r = tid >> 5, tt = tid & 31, t = tcx * TT + tt;
        // (one node: the row ranges are known without the item tables — a round trip less in front of the row loads)
        const bool one = N == 1;
        const int g0 = one ? 0 : v.node_gitem_beg[n];
        const int ngi = v.genRows > 0 ? v.genRows : (one ? v.nGenItems : v.node_gitem_beg[n + 1] - g0);    // (genRows: one node)
        const int s0 = one ? 0 : v.node_sitem_beg[n], nsi = one ? v.nStoItems : v.node_sitem_beg[n + 1] - s0;
        // rows to add: generator items, then the storage items' scan partials, then their warm-start partials
        const int ni = ngi + 2 * nsi;
        const int per = (ni + RB - 1) / RB;
        const int i0 = rb * per, i1 = min(ni, i0 + per);
        const int rr = v.part_T ? (tid & 7) : r, tr = v.part_T ? (tid >> 3) : tt;      // (transposed rows: see k_slack)
        const double acc = tcx * TT + tr < T ? rows_sum(v, g0, ngi, s0, nsi, i0, i1, rr, tcx * TT + tr) : 0.0;
        if (halt) return;                                   // (uniform)
        red[rr * TT + tr] = acc;
        __syncthreads();
        const bool direct = RB == 1 && !v.sliceDual;       // one slice per node: its sum IS the node's sum
        if (r == 0 && t < T) {
            double sum = 0.0;
            for (int q = 0; q < R; ++q) sum += red[q * TT + tt];
            if (direct) v.cons[n + (size_t)N * t] = sum;
            else v.part2[((size_t)n * RB + rb) * T + t] = sum;
        }
        // cost partials ride with the first node's slices of the first timestep chunk
        if (n == 0 && tcx == 0) {
            const int ngr = v.genRows > 0 ? v.genRows : v.nGenItems;
            const int nc = ngr + v.nStoItems, cper = (nc + RB - 1) / RB;
            const int c0 = rb * cper, c1 = min(nc, c0 + cper);
            const double c = cost_rows_sum(v, c0, c1, ngr, red);
            if (tid == 0) {
                if (direct) v.cons[(size_t)N * T + 2 * (size_t)L * T] = c;
                else v.part2_cost[rb] = c;
            }
        }
        if (v.sliceDual || direct) return;      // slices added by the dual kernel / nothing left to add
        // publish, take a ticket; the last block of this (node, timestep chunk) finishes the sum.
        // Hand-off per cdna_hip_programming.md G16: every storing wave drains its stores, the block meets,
        // ONE lane releases at agent scope and takes the ticket; the last block's lane acquires, the block
        // meets again, then everybody reads the other blocks' slices with plain loads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int last = atomicAdd(&v.reduce_ticket[n * TC + tcx], 1) == RB - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            last_sh = last;
        }
        __syncthreads();
        if (last_sh) {
            // 8 lanes per timestep take interleaved slices; combined in a fixed order
            double sum = 0.0;
            if (t < T) {
                double x[8];                              // RB <= 64: all of this lane's slices in flight at once
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = r + R * k;
                    x[k] = q < RB ? v.part2[((size_t)n * RB + q) * T + t] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += x[k];
            }
            __syncthreads();
            red[tid] = sum;
            __syncthreads();
            if (r == 0 && t < T) {
                double tot = 0.0;
                for (int q = 0; q < R; ++q) tot += red[q * TT + tt];
                v.cons[n + (size_t)N * t] = tot;
            }
            if (n == 0 && tcx == 0 && tid < 64) {          // cost slices: side by side in wave 0, butterfly sum
                double c = tid < RB ? v.part2_cost[tid] : 0.0;
                for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
                if (tid == 0) v.cons[(size_t)N * T + 2 * (size_t)L * T] = c;
            }
            if (tid == 0) v.reduce_ticket[n * TC + tcx] = 0;       // ready for the next iteration
        }
    } else {
        // slack sums: block (timestep, U | K, group of 64 lines); thread (node part, line) covers a quarter of the nodes —
        // 4x the loads in flight of one thread per line walking all N nodes (these are chains of L2-latency-bound ptdf
        // reads) — and the four partial sums meet in LDS, added in part order. The change of every node's injection in
        // this iteration (new minus previous sum of its items' partials) and the nodes' constants are staged in LDS.
        if (halt) return;
        extern __shared__ double nsh[];               // [N] node changes of timestep t | [N] window | [N] agents at the node
        __shared__ double pred[256];
        double *sdL = nsh, *winL = nsh + N, *naL = nsh + 2 * N;
        const int LB = (L + 63) / 64;
        const int b2 = blockIdx.x - N * RB * TC, lb = b2 % LB, tw = b2 / LB, t = tw >> 1, which = tw & 1;
        for (int n = tid; n < N; n += 256) {
            sdL[n] = v.node_dsum[n + (size_t)N * t];
            winL[n] = v.node_win[n];
            naL[n] = (double)((v.node_gen_beg[n + 1] - v.node_gen_beg[n]) + (v.node_sto_beg[n + 1] - v.node_sto_beg[n]));
        }
        __syncthreads();
        const int pr = tid >> 6, ll = tid & 63, l = lb * 64 + ll;
        const int Nc = (((N + 3) / 4) + 7) & ~7, nbeg = pr * Nc, nend = min(N, nbeg + Nc);
        const double w2 = 2.0 * v.w_flow, g = v.gamma, inv = 1.0 / (w2 + g);
        double partial = 0.0;
        if (l < L && nbeg < nend) {
            const size_t rem = l + (size_t)L * t;
            const double f = v.flow[rem], F = v.fmax[l], cu = v.avgU[rem], ck = v.avgK[rem];
            if (!v.walk_flag[rem]) {
                // every agent of every node has this slack active, or none has: sum_a (aX -+ kap_n d_a) = A aX -+ (w2 inv) sum_n h_n D_n
                const SlackCase c0 = slack_case(g, w2, inv, 0.0, f, F, cu, ck, 0.0);
                const double a = which ? c0.aK : c0.aU;
                if (a > 0.0) {
                    double dot = 0.0, cnt = 0.0;
                    for (int n0 = nbeg; n0 < nend; n0 += kFlight) {
                        double h[kFlight];
#pragma unroll
                        for (int u = 0; u < kFlight; ++u) h[u] = n0 + u < nend ? v.ptdf[l + (size_t)L * (n0 + u)] : 0.0;
#pragma unroll
                        for (int u = 0; u < kFlight; ++u)
                            if (n0 + u < nend) { dot = fma(h[u], sdL[n0 + u], dot); cnt += naL[n0 + u]; }
                    }
                    partial = slack_sum_plain(which, cnt, a, w2 * inv, dot);
                }
            } else {
                partial = slack_sum_cases(v, which, l, t, nbeg, nend, sdL, winL, naL, f, F, cu, ck);
            }
        }
        pred[tid] = partial;
        __syncthreads();
        if (pr == 0 && l < L)
            v.cons[(size_t)N * T + (size_t)which * L * T + l + (size_t)L * t] = ((pred[ll] + pred[64 + ll]) + pred[128 + ll]) + pred[192 + ll];
    }
}


void launch_reduce(const DevView &v, hipStream_t s)
{
    const int TC = (v.T + 31) / 32;
    const int blocks = v.N * v.reduceRB * TC + (v.L > 0 ? 2 * v.T * ((v.L + 63) / 64) : 0);
    hipLaunchKernelGGL(k_reduce, dim3(blocks), dim3(256), v.L > 0 ? 3 * (size_t)v.N * sizeof(double) : 0, s, v);
}

// ------------------------------------------------------------------------------------------------
// peer exchange: the sum of the consensus vector over the ranks WITHOUT a collective library. Every block owns a chunk of
// the vector: it stores its rank's chunk into slot [parity][me] of EVERY rank's receive area (plain stores over xGMI for
// the peers), publishes a sequence number per (source rank, chunk) behind a system-scope release, waits until the same
// chunk of every rank has arrived in its own area, and adds the world copies in RANK ORDER — every rank adds the same
// numbers in the same order, so the sums (and the duals computed from them) are bitwise identical everywhere.
//   * no block waits before it has sent, and a block waits only for remote blocks of its own chunk: no deadlock, whatever
//     the order blocks run in on any device;
//   * two parities: a rank that is writing iteration k+2 into a slot has received iteration k+1 from everybody, and a
//     rank sends k+1 only after it has consumed k (same stream) — two slots suffice; sequence numbers only grow;
//   * a wait is bounded (wall clock): a lost peer sets Status.xchg_timeout, later exchanges do not wait again, the host
//     reports DOPF_E_DEVICE. The kernel always ends.
// Latency: one kernel, one xGMI store + one flag round trip (a few microseconds) against the 20-30 us of a library
// all-reduce for the sub-kilobyte vector of a copper plate (776 B on config2).
// ------------------------------------------------------------------------------------------------
// (sub > 0: only the first sub - 1 chunks and the vector's last chunk are exchanged — the node sums and the cost; a chunk's flags
// and slots are its own, so chunks that sit an iteration out just keep their older sequence numbers)
__global__ __launch_bounds__(256) void k_xchg(DevView v, XchgView x, int sub)
{
    if (v.st->halt) return;
    __shared__ int bad;
    const int tid = threadIdx.x, W = x.world, me = x.me;
    const int chunk = (sub > 0 && (int)blockIdx.x == sub - 1) ? x.nchunks - 1 : (int)blockIdx.x;
    const unsigned long long seq = (unsigned long long)v.st->iters_total + 1ull;
    const size_t par = (size_t)(seq & 1ull), n = x.n, j0 = (size_t)chunk * kXchgChunk;
    constexpr int U = kXchgChunk / 256;
    if (tid == 0) bad = 0;
    double own[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = j0 + tid + 256 * (size_t)u;
        own[u] = j < n ? v.cons[j] : 0.0;
    }
    for (int q = 0; q < W; ++q) {
        const int r = (me + 1 + q) % W;                    // the peers first, the own slot last
        double *dst = x.data[r] + (par * W + me) * n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = j0 + tid + 256 * (size_t)u;
            if (j < n) dst[j] = own[u];
        }
    }
    __threadfence_system();                                // this thread's stores have left for their destinations
    __syncthreads();
    if (tid < W)
        __hip_atomic_store(x.flags[tid] + (par * W + me) * x.nchunks + chunk, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < W) {
        const unsigned long long *f = x.flags[me] + (par * W + tid) * x.nchunks + chunk;
        const unsigned long long t0 = wall_clock64();
        bool ok = v.st->xchg_timeout == 0;
        while (ok && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > x.timeout_ticks) ok = false;
        }
        if (!ok) atomicOr(&bad, 1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // every wave: nothing older than the flags is read below
    if (bad) {
        if (tid == 0) v.st->xchg_timeout = 1;
        return;
    }
    const double *mine = x.data[me] + par * W * n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = j0 + tid + 256 * (size_t)u;
        if (j < n) {
            double sum = mine[j];
            for (int r = 1; r < W; ++r) sum += mine[(size_t)r * n + j];
            v.cons[j] = sum;
        }
    }
}

// The same sum for vectors of more than one chunk per rank (the 118-node / 186-line network: 659 kB, 41 chunks): as an
// all-gather every rank would send and receive world x n doubles per iteration (5.3 MB at 8 ranks). Here chunk c has an OWNER,
// rank c % world:
//   1. every rank stores its copy of chunk c into the owner's receive area only and publishes its flag there;
//   2. the owner's block waits for the world copies, adds them in RANK ORDER, stores the sum into every rank's sum
//      region and publishes a sum flag there;
//   3. every rank's block waits for the sum flag of its chunk and copies the sum into its consensus vector.
// 2 n doubles out and 2 n in per rank and iteration, whatever the world size; every rank receives the owner's sum, so
// the replicated state stays bitwise identical. A block sends before it waits, an owner waits only for sends, the
// others only for the owner: no cycle, whatever the order blocks run in. Two parities suffice as before (a rank sends
// k+2 only after it has consumed the sum of k+1, which exists only after everybody's k+1 has been consumed).
__global__ __launch_bounds__(256) void k_xchg_rs(DevView v, XchgView x, int sub)
{
    if (v.st->halt) return;
    __shared__ int bad;
    const int tid = threadIdx.x, W = x.world, me = x.me;
    const int chunk = (sub > 0 && (int)blockIdx.x == sub - 1) ? x.nchunks - 1 : (int)blockIdx.x, owner = chunk % W;
    const unsigned long long seq = (unsigned long long)v.st->iters_total + 1ull;
    const size_t par = (size_t)(seq & 1ull), n = x.n, j0 = (size_t)chunk * kXchgChunk;
    constexpr int U = kXchgChunk / 256;
    if (tid == 0) bad = 0;
    {   // 1. this rank's copy -> the owner
        double *dst = x.data[owner] + (par * W + me) * n;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = j0 + tid + 256 * (size_t)u;
            if (j < n) dst[j] = v.cons[j];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
        __hip_atomic_store(x.flags[owner] + (par * W + me) * x.nchunks + chunk, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    if (me == owner) {
        // 2. wait for the world copies, add them in rank order, hand the sum to everybody
        if (tid < W) {
            const unsigned long long *f = x.flags[me] + (par * W + tid) * x.nchunks + chunk;
            bool ok = v.st->xchg_timeout == 0;
            while (ok && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > x.timeout_ticks) ok = false;
            }
            if (!ok) atomicOr(&bad, 1);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        if (!bad) {
            const double *mine = x.data[me] + par * W * n;
            double sum[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t j = j0 + tid + 256 * (size_t)u;
                sum[u] = 0.0;
                if (j < n) {
                    sum[u] = mine[j];
                    for (int r = 1; r < W; ++r) sum[u] += mine[(size_t)r * n + j];
                }
            }
            for (int q = 0; q < W; ++q) {
                const int r = (me + 1 + q) % W;                // the peers first, the own region last
                double *dst = x.sum[r] + par * n;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const size_t j = j0 + tid + 256 * (size_t)u;
                    if (j < n) dst[j] = sum[u];
                }
            }
            __threadfence_system();
            __syncthreads();
            if (tid < W)
                __hip_atomic_store(x.sflags[tid] + par * x.nchunks + chunk, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // 3. the summed chunk (a lost peer: the owner publishes nothing, everybody's wait ends at the deadline)
    if (tid == 0) {
        const unsigned long long *f = x.sflags[me] + par * x.nchunks + chunk;
        bool ok = v.st->xchg_timeout == 0 && !bad;
        while (ok && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > x.timeout_ticks) ok = false;
        }
        if (!ok) bad = 1;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (bad) {
        if (tid == 0) v.st->xchg_timeout = 1;
        return;
    }
    const double *res = x.sum[me] + par * n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t j = j0 + tid + 256 * (size_t)u;
        if (j < n) v.cons[j] = res[j];
    }
}

void launch_xchg(const DevView &v, const XchgView &x, hipStream_t s, bool inj_only)
{
    int sub = 0, grid = x.nchunks;
    if (inj_only) {                      // the chunks of the node sums [0, N T) and the chunk of the cost (the vector's last entry)
        const int ninj = (int)(((size_t)v.N * v.T + kXchgChunk - 1) / kXchgChunk);
        if (ninj < x.nchunks) { sub = ninj + 1; grid = sub; }
    }
    if (x.rs) hipLaunchKernelGGL(k_xchg_rs, dim3(grid), dim3(256), 0, s, v, x, sub);
    else hipLaunchKernelGGL(k_xchg, dim3(grid), dim3(256), 0, s, v, x, sub);
}

// ------------------------------------------------------------------------------------------------
// dual update
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_pos(unsigned long long *addr, double vpos)
{
    atomicMax(addr, (unsigned long long)__double_as_longlong(vpos));   // vpos >= 0: bit order = value order
}

// element i of the dual step: injection / imbalance / flows from the consensus vector, then the
// lambda (i < T) and mu, rho (i < L*T) ascent steps; returns this element's |dual change|
template <bool UPDATE>
__device__ __forceinline__ void dual_body(const DevView &v, size_t i, double &rl, double &rm, double &rr)
{
    const int N = v.N, L = v.L, T = v.T;
    const size_t NT = (size_t)N * T, LT = (size_t)L * T;
    const double *cinj = v.cons, *cU = v.cons + NT, *cK = cU + LT;
    if (i < NT) v.inj[i] = cinj[i] - v.demand[i];                         // results.jl:58-100
    if (i < (size_t)T) {
        const int t = (int)i;
        double sum = 0.0;
        for (int n0 = 0; n0 < N; n0 += 8) {               // eight nodes in flight (one lane walks all N of them)
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + u;
                x[u] = n < N ? cinj[n + (size_t)N * t] - v.demand[n + (size_t)N * t] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += x[u];
        }
        if (UPDATE) v.s_used[t] = v.s[t];
        v.s[t] = sum;
        if (UPDATE) {
            const double lo = v.lam[t], ln = lo + v.gamma * sum;          // update_duals.jl:8-13
            v.lam_used[t] = lo;
            v.lam[t] = ln;
            rl = fmax(rl, fabs(ln - lo));
        }
    }
    if (i < LT) {
        const int l = (int)(i % L), t = (int)(i / L);
        double f = 0.0;
        for (int n0 = 0; n0 < N; n0 += 8) {               // eight rows of ptdf in flight
            double h[8], q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + u < N ? n0 + u : N - 1;
                h[u] = n0 + u < N ? v.ptdf[l + (size_t)L * n] : 0.0;
                q[u] = cinj[n + (size_t)N * t] - v.demand[n + (size_t)N * t];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) f += h[u] * q[u];
        }
        if (UPDATE) { v.flow_used[i] = v.flow[i]; v.avgU_used[i] = v.avgU[i]; v.avgK_used[i] = v.avgK[i]; }
        v.flow[i] = f;                                                    // results.jl:114
        if (UPDATE) {
            const double aU = v.invA * cU[i], aK = v.invA * cK[i];       // results.jl:108-112
            v.avgU[i] = aU;
            v.avgK[i] = aK;
            const double mo = v.mu[i], ro = v.rho[i], F = v.fmax[l];
            const double mn = (mo + v.gamma * (f + aU - F)) * (aU <= v.mask_thr ? 1.0 : 0.0);   // :18-25
            const double rn = (ro + v.gamma * (aK - f - F)) * (aK <= v.mask_thr ? 1.0 : 0.0);   // :30-37
            v.mu_used[i] = mo; v.rho_used[i] = ro;
            v.mu[i] = mn; v.rho[i] = rn;
            rm = fmax(rm, fabs(mn - mo));
            rr = fmax(rr, fabs(rn - ro));
        }
        {   // what the next slack sums of (l,t) will need (k_slack / k_reduce)
            const double w2 = 2.0 * v.w_flow, inv = 1.0 / (w2 + v.gamma);
            const int need = slack_needs_cases(v.gamma, w2, inv, f, v.fmax[l], v.avgU[i], v.avgK[i], v.line_reach[l]) ? 1 : 0;
            v.walk_flag[i] = need;
            if (need) atomicOr(&v.walk_any[t], 1);          // (zeroed by the caller before the sweep; an OR has no order)
        }
    }
}

// price[n,t] = lambda_t + sum_l ptdf[l,n] (mu - rho)[l,t]   (subproblems.jl:67-74), element i = n + N*t
__device__ __forceinline__ void price_body(const DevView &v, size_t i)
{
    const int N = v.N, L = v.L;
    const int n = (int)(i % N), t = (int)(i / N);
    double p = v.lam[t];
    for (int l0 = 0; l0 < L; l0 += 8) {                   // ptdfT[n + N l]: coalesced over the nodes, eight lines in flight
        double h[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int l = l0 + u < L ? l0 + u : L - 1;
            h[u] = l0 + u < L ? v.ptdfT[n + (size_t)N * l] : 0.0;
            q[u] = v.mu[l + (size_t)L * t] - v.rho[l + (size_t)L * t];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) p += h[u] * q[u];
    }
    v.price[i] = p;
}

template <bool UPDATE>
__global__ __launch_bounds__(256) void k_dual(DevView v)
{
    if (UPDATE && v.st->halt) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double rl = 0.0, rm = 0.0, rr = 0.0;
    dual_body<UPDATE>(v, i, rl, rm, rr);
    if (UPDATE) {
        // wave max, then one atomic per wave (max is order independent: deterministic)
        for (int d = 32; d > 0; d >>= 1) {
            rl = fmax(rl, __shfl_xor(rl, d));
            rm = fmax(rm, __shfl_xor(rm, d));
            rr = fmax(rr, __shfl_xor(rr, d));
        }
        if ((threadIdx.x & 63) == 0) {
            if (rl > 0.0) atomic_max_pos(&v.st->resbits[0], rl);
            if (rm > 0.0) atomic_max_pos(&v.st->resbits[1], rm);
            if (rr > 0.0) atomic_max_pos(&v.st->resbits[2], rr);
        }
        if (i == 0) v.st->total_cost = v.cons[(size_t)v.N * v.T + 2 * (size_t)v.L * v.T];
    }
}

template <bool UPDATE>
__global__ __launch_bounds__(256) void k_price(DevView v)
{
    const size_t NT = (size_t)v.N * v.T;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    // recomputed even after a halt: the duals are frozen then, so the values are identical
    if (i < NT) price_body(v, i);
    if (UPDATE && i == 0) {
        Status *st = v.st;
        if (st->halt) return;
        const double r0 = __longlong_as_double((long long)st->resbits[0]);
        const double r1 = __longlong_as_double((long long)st->resbits[1]);
        const double r2 = __longlong_as_double((long long)st->resbits[2]);
        st->resbits[0] = st->resbits[1] = st->resbits[2] = 0ull;
        status_update(v, r0, r1, r2);
    }
}

// Big networks, one block per timestep: the nodal injections of the timestep are staged in LDS once and every line
// of the block reads them from there (flows = ptdf . inj is the only O(N L T) piece of the dual step); the
// imbalance is a fixed-order block sum. The price kernel stages (mu - rho)[., t] the same way and skips the
// ptdf^T product altogether for a timestep on which no line carries a multiplier.
template <bool UPDATE>
__global__ __launch_bounds__(256) void k_dual_t(DevView v)
{
    if (UPDATE && v.st->halt) return;
    extern __shared__ double q[];            // N
    __shared__ double red[256];
    // block (timestep t, group of 64 lines): thread (part, line) adds a quarter of the nodes, the four partial flows meet
    // in LDS and are added in part order — 4x the loads in flight of one thread per line walking all N nodes
    const int LB = v.L > 0 ? (v.L + 63) / 64 : 1;
    const int tid = threadIdx.x, t = blockIdx.x / LB, lb = blockIdx.x - t * LB;
    const int N = v.N, L = v.L, T = v.T;
    const size_t NT = (size_t)N * T, LT = (size_t)L * T;
    const double *cinj = v.cons, *cU = v.cons + NT, *cK = cU + LT;
    double part = 0.0;
    for (int n = tid; n < N; n += 256) {
        const double x = cinj[n + (size_t)N * t] - v.demand[n + (size_t)N * t];
        q[n] = x;
        if (lb == 0) v.inj[n + (size_t)N * t] = x;                        // results.jl:58-100
        part += x;
    }
    const double sum = block_sum256(part, red);                              // (barriers inside: q[] is complete)
    double rl = 0.0, rm = 0.0, rr = 0.0;
    if (tid == 0 && lb == 0) {
        if (UPDATE) v.s_used[t] = v.s[t];
        v.s[t] = sum;
        if (UPDATE) {
            const double lo = v.lam[t], ln = lo + v.gamma * sum;             // update_duals.jl:8-13
            v.lam_used[t] = lo;
            v.lam[t] = ln;
            rl = fabs(ln - lo);
        }
    }
    {
        const int pr = tid >> 6, ll = tid & 63, l = lb * 64 + ll;
        const int Nc = (((N + 3) / 4) + 7) & ~7, nbeg = pr * Nc, nend = min(N, nbeg + Nc);
        double f = 0.0;
        if (l < L)
            for (int n0 = nbeg; n0 < nend; n0 += 8) {                        // eight rows of ptdf in flight
                double h[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) h[u] = n0 + u < nend ? v.ptdf[l + (size_t)L * (n0 + u)] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) f += h[u] * (n0 + u < nend ? q[n0 + u] : 0.0);
            }
        __syncthreads();
        red[tid] = f;
        __syncthreads();
        if (pr == 0 && l < L) {
            f = ((red[ll] + red[64 + ll]) + red[128 + ll]) + red[192 + ll];
            const size_t i = l + (size_t)L * t;
            if (UPDATE) { v.flow_used[i] = v.flow[i]; v.avgU_used[i] = v.avgU[i]; v.avgK_used[i] = v.avgK[i]; }
            v.flow[i] = f;                                                       // results.jl:114
            if (UPDATE) {
                const double aU = v.invA * cU[i], aK = v.invA * cK[i];          // results.jl:108-112
                v.avgU[i] = aU;
                v.avgK[i] = aK;
                const double mo = v.mu[i], ro = v.rho[i], F = v.fmax[l];
                const double mn = (mo + v.gamma * (f + aU - F)) * (aU <= v.mask_thr ? 1.0 : 0.0);   // update_duals.jl:18-25
                const double rn = (ro + v.gamma * (aK - f - F)) * (aK <= v.mask_thr ? 1.0 : 0.0);   // :30-37
                v.mu_used[i] = mo; v.rho_used[i] = ro;
                v.mu[i] = mn; v.rho[i] = rn;
                rm = fabs(mn - mo);
                rr = fabs(rn - ro);
            }
            // what the next slack sums of (l,t) will need (k_slack / k_reduce; the price kernel ORs them per timestep)
            const double w2 = 2.0 * v.w_flow, inv = 1.0 / (w2 + v.gamma);
            v.walk_flag[i] = slack_needs_cases(v.gamma, w2, inv, f, v.fmax[l], v.avgU[i], v.avgK[i], v.line_reach[l]) ? 1 : 0;
        }
    }
    if (UPDATE) {
        // block max (order independent), then one atomic per block and residual
        __syncthreads();
        red[tid] = rm;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
        const double bm = red[0];
        __syncthreads();
        red[tid] = rr;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
        if (tid == 0) {
            if (rl > 0.0) atomic_max_pos(&v.st->resbits[0], rl);
            if (bm > 0.0) atomic_max_pos(&v.st->resbits[1], bm);
            if (red[0] > 0.0) atomic_max_pos(&v.st->resbits[2], red[0]);
            if (t == 0 && lb == 0) v.st->total_cost = v.cons[NT + 2 * LT];
        }
    }
}

template <bool UPDATE>
__global__ __launch_bounds__(256) void k_price_t(DevView v)
{
    extern __shared__ double d[];            // L: (mu - rho)[., t] | L: G[., t] | L: S[., t]
    const int tid = threadIdx.x, t = blockIdx.x;
    const int N = v.N, L = v.L;
    double *Gl = d + L, *Sl = d + 2 * L;
    // A timestep on which no line has a switch point within any node's reach (walk_any[t] == 0: the settled state) has
    // EMPTY breakpoint tables: every line's slacks are in one regime for every move an agent can make, Psi_{n,t} is linear
    // inside the node's window, Psi(0) = price + gamma s + sum_l h G_l and its slope = gamma + sum_l h^2 S_l with
    //   G_l = w2 ((f + U0 - F) - (K0 - f - F)),  S_l = w2 (2 - (U on + K on) w2 / (w2 + gamma))
    // — two more dot products along the loop that forms the price; k_tables then skips the timestep altogether.
    int anyNeed = 0;
    for (int l = tid; l < L; l += 256) anyNeed |= v.walk_flag[l + (size_t)L * t];
    anyNeed = __syncthreads_or(anyNeed);
    if (tid == 0 && L > 0) v.walk_any[t] = anyNeed ? 1 : 0;
    const bool lin = L > 0 && !anyNeed;
    const double w2 = 2.0 * v.w_flow, g = v.gamma, inv = 1.0 / (w2 + g);
    int nz = 0;
    for (int l = tid; l < L; l += 256) {
        const size_t i = l + (size_t)L * t;
        const double x = v.mu[i] - v.rho[i];
        d[l] = x;
        nz |= x != 0.0;
        if (lin) {
            const double f = v.flow[i], F = v.fmax[l];
            const double U0 = dmax0((g * v.avgU[i] - w2 * (f - F)) * inv), K0 = dmax0((g * v.avgK[i] + w2 * (f + F)) * inv);
            Gl[l] = w2 * ((f + U0 - F) - (K0 - f - F));
            Sl[l] = w2 * (2.0 - ((U0 > 0.0 ? 1.0 : 0.0) + (K0 > 0.0 ? 1.0 : 0.0)) * w2 * inv);
        }
    }
    nz = __syncthreads_or(nz);
    // recomputed even after a halt: the duals are frozen then, so the values are identical
    const double lam = v.lam[t], gs = g * v.s[t];
    // thread (part, n): the lines are cut into P contiguous parts so that a block of N < 256 nodes still fills its
    // 256 threads; partial dot products meet in LDS and are added in part order (fixed: deterministic)
    __shared__ double redp[3][256];
    const int NP = N <= 256 ? ((N + 31) & ~31) : 256, P = 256 / NP;
    const int part = tid / NP, nn = tid - part * NP;
    const int Lc = (((L + P - 1) / P) + 7) & ~7, lbeg = part * Lc, lend = min(L, lbeg + Lc);
    for (int nb = 0; nb < N; nb += NP) {
        const int n = nb + nn;
        double p = 0.0, ps = 0.0, sl = 0.0;
        if (n < N && part < P && (nz || lin))
            for (int l0 = lbeg; l0 < lend; l0 += 8) {                        // ptdfT[n + N l]: coalesced over the nodes
                double h[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) h[u] = l0 + u < lend ? v.ptdfT[n + (size_t)N * (l0 + u)] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) p += h[u] * (l0 + u < lend ? d[l0 + u] : 0.0);
                if (lin) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (l0 + u < lend) { ps += h[u] * Gl[l0 + u]; sl += h[u] * h[u] * Sl[l0 + u]; }
                }
            }
        if (P > 1) {
            __syncthreads();
            redp[0][tid] = p; redp[1][tid] = ps; redp[2][tid] = sl;
            __syncthreads();
            if (part == 0) {
                for (int q = 1; q < P; ++q) { p += redp[0][q * NP + nn]; ps += redp[1][q * NP + nn]; sl += redp[2][q * NP + nn]; }
            }
        }
        if (n < N && part == 0) {
            const size_t at = n + (size_t)N * t;
            p += lam;
            v.price[at] = p;
            if (lin) {
                v.tb_m[at] = 0;
                v.tb_psi0[at] = (p + gs) + ps;
                v.tb_slope[at * (v.M2 + 1)] = g + sl;
            }
        }
    }
    if (L > 0 && tid == 0) v.tab_skip[t] = lin ? 1 : 0;
    if (UPDATE && t == 0 && tid == 0) {
        Status *st = v.st;
        if (st->halt) return;
        const double r0 = __longlong_as_double((long long)st->resbits[0]);
        const double r1 = __longlong_as_double((long long)st->resbits[1]);
        const double r2 = __longlong_as_double((long long)st->resbits[2]);
        st->resbits[0] = st->resbits[1] = st->resbits[2] = 0ull;
        status_update(v, r0, r1, r2);
    }
}

// Networks with L <= 256 lines and N <= 256 nodes (BASELINE configs[3]: 118 / 186): dual step, prices, linear Psi
// pieces and the stop test of timestep t in ONE block of 1024 threads — what k_dual_t + k_price_t do in two launches
// with the new duals travelling through global memory in between. Both halves are chains of L2-latency-bound dot
// products over the PTDF matrix; here a thread's share of a dot product is two batches of 16 rows, the old
// line state is on its way while the flows are formed, mu - rho and the G / S terms of the linear pieces go from the
// dual half to the price half through LDS, and the residual maxima meet in a ticket: the block that finishes last runs
// the stop test (one atomic round trip per block instead of a launch). Every sum has a fixed order.
// QUIET (with SID; the host's "quiet" chain, dopf_api.hip): no line of any timestep is flagged, so k_slack has nothing to walk and
// is not launched — the block forms the nodes' sums of its timestep itself, from the items' partial rows (one 8-byte read per
// row: 76 KB of sectors per block on the 118-node share, one round trip), with k_slack's arithmetic and bits. If the dual step
// flags a line for the NEXT iteration the last block parks the chain (Status::halt = 2) and the host goes back to the chain
// with k_slack.
template <bool UPDATE, bool SID = false, bool QUIET = false>
__global__ __launch_bounds__(1024) void k_dual_price_t1024(DevView v)
{
    const int halt = UPDATE ? v.st->halt : 0;           // (looked at once the first loads are on their way; nothing stored before)
    const StatusPre spre = UPDATE ? status_load(v) : StatusPre{0, 0, 0};       // (what the stop test starts from: loaded now, used at the end)
#ifdef DOPF_DUAL_STAMPS          // (measurement build: wall-clock stamps of block T/2's phases in the status block's counters, scripts/dual_stamps.py)
    if (UPDATE && threadIdx.x == 0 && (int)blockIdx.x == v.T / 2) v.st->dbg_reason[0] = wall_clock64();
#endif
    extern __shared__ double sh[];               // q[N] injections | d[L] mu - rho | G[L] | S[L] | rows of t (later: table scratch) | slack vectors
    __shared__ double red[3][1024];
    __shared__ double wsum[4], wmx[2][4];
    __shared__ int wany[4], wnz[4], lastBlock;
    const int tid = threadIdx.x, lane = tid & 63, t = blockIdx.x;
    const int N = v.N, L = v.L, T = v.T;
    const size_t NT = (size_t)N * T, LT = (size_t)L * T;
    double *q = sh, *dd = sh + N, *Gl = dd + L, *Sl = Gl + L;
    const double *cinj = v.cons, *cU = v.cons + NT, *cK = cU + LT;
    const double g = v.gamma, w2 = 2.0 * v.w_flow, inv = 1.0 / (w2 + g);

    // ---- dual half: thread (node part pl, line l) --------------------------------------------------------------
    // SID: the slack sums of this timestep's lines are formed HERE, from the PTDF rows the flows need anyway and the nodes'
    // injection changes (k_slack) — with k_reduce's arithmetic, part by part — instead of by a launch in between
    constexpr bool sid = UPDATE && SID;
    double *sdL = sh + v.dualSdOff, *winL = sdL + N, *naL = winL + N;
    const int pl = tid >> 8, l = tid & 255;
    const bool lt = pl == 0 && l < L;
    const size_t i = (size_t)(l < L ? l : 0) + (size_t)L * t;
    const int Nc = (((N + 3) / 4) + 7) & ~7, nbeg = pl * Nc, nend = min(N, nbeg + Nc);
    const bool mine = sid && l < L && nbeg < nend;           // this thread forms a part of the slack sums of line l
    // QUIET: the thread's first batch of PTDF rows depends on nothing the kernel computes — issued here, it arrives while the nodes'
    // sums are formed (two dependent round trips). (Not in the other variants: there the registers cost more than the wait.)
    double h0[kFlight];
    if (QUIET) {
#pragma unroll
        for (int u = 0; u < kFlight; ++u) h0[u] = (l < L && nbeg + u < nend) ? v.ptdf[l + (size_t)L * (nbeg + u)] : 0.0;
    }
    double f_old = 0.0, aU_old = 0.0, aK_old = 0.0, F = 0.0, cntp = 0.0;
    int wf = 0;
    if (lt || mine) {                            // (in flight while the flows are formed)
        f_old = v.flow[i]; aU_old = v.avgU[i]; aK_old = v.avgK[i]; F = v.fmax[l];
    }
    if (mine) {
        wf = (QUIET || v.slackGlobal) ? 0 : v.walk_flag[i];       // (slackGlobal: the host runs this chain only while no line is flagged)
        cntp = (double)((v.node_gen_beg[nend] - v.node_gen_beg[nbeg]) + (v.node_sto_beg[nend] - v.node_sto_beg[nbeg]));   // agents at the part's nodes
        // behind an exchange the nodes' changes are all ranks': so is the count (the parts' counts only ever meet as their sum)
        if (v.slackGlobal) cntp = pl == 0 ? v.nAgents : 0.0;
    }
    const double lam_old = v.lam[t], s_old = v.s[t];
    double x = 0.0, cost_q = 0.0;
    if (QUIET) {
        // Node sums of timestep t. The items' partial sums are kept transposed (DevView::part_T): ALL rows of t are one contiguous
        // vector — brought into LDS by coalesced loads that depend on nothing (the node tables load beside them), instead of one
        // 8-byte word out of every row behind the tables' round trip (76 KB of sectors per block on the 118-node share). Then
        // thread (node, lane r of eight) adds rows r, r + 8, ... of its node from LDS, the eight lanes are added in lane order:
        // rows_sum's grouping, k_slack's bits.
        double *rowsL = sh + v.dualRowsOff;
        const double *src = v.part_T + (size_t)t * v.rowsT;
        double xr[4];                                    // four loads in flight per lane: the first 4 096 rows
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = tid + 1024 * u; xr[u] = j < v.rowsT ? src[j] : 0.0; }
        int gr[4];                                       // (where the position's row sits in node order: the rows are placed by writer XCD)
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = tid + 1024 * u; gr[u] = j < v.rowsT ? v.row_of_pos[j] : -1; }
        for (int nb = 0; nb < N; nb += 128) {
            const int n = nb + (tid >> 3), r = tid & 7;
            double acc = 0.0, was = 0.0, dem = 0.0;
            int base = 0, ngi = 0, nsi = 0;
            if (n < N) {                                 // (the node tables: beside the rows, not in front of them)
                const int g0 = v.node_gitem_beg[n], s0 = v.node_sitem_beg[n];
                ngi = v.node_gitem_beg[n + 1] - g0; nsi = v.node_sitem_beg[n + 1] - s0; base = g0 + 2 * s0;
                if (r == 0) { was = v.prev_node[n + (size_t)N * t]; dem = v.demand[n + (size_t)N * t]; }
            }
            if (nb == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (gr[u] >= 0) rowsL[gr[u]] = xr[u];
                for (int j = 4096 + tid; j < v.rowsT; j += 1024) { const int g_ = v.row_of_pos[j]; if (g_ >= 0) rowsL[g_] = src[j]; }
            }
            if (halt) return;                            // (uniform; nothing has been stored)
            // (everything in flight is a load; the stores come behind the kernel's barriers, which wait for LDS only — see below)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (n < N) {
                const int i1 = ngi + 2 * nsi;
                for (int i = r; i < i1; i += 32) {
                    double x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = i + 8 * u;
                        x[u] = j < i1 ? rowsL[base + row_pos(j, ngi, nsi)] : 0.0;
                    }
                    acc += (x[0] + x[1]) + (x[2] + x[3]);
                }
            }
            red[0][tid] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (n < N && r == 0) {
                double sn = 0.0;
                for (int k = 0; k < 8; ++k) sn += red[0][tid + k];
                const size_t at = n + (size_t)N * t;
                sdL[n] = sn - was;
                q[n] = sn - dem;                                             // results.jl:58-100
                v.prev_node[at] = sn; v.node_dsum[at] = sn - was; v.cons[at] = sn;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (t == 0) {                                    // the cost, by the block of the first timestep (k_slack's order)
            const int ngr = v.genRows > 0 ? v.genRows : v.nGenItems, nc = ngr + v.nStoItems;
            double c = 0.0;
            if (tid < 256)
                for (int k0 = tid; k0 < nc; k0 += 8 * 256) {          // eight items' costs in flight per lane, added in item order
                    double xc[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = k0 + 256 * u < nc ? k0 + 256 * u : k0;
                        xc[u] = k < ngr ? v.part_gcost[k] : v.part_scost[k - ngr] + v.part_scost_w[k - ngr];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) c += k0 + 256 * u < nc ? xc[u] : 0.0;
                }
            // (block_sum256's tree — lanes i, i + 128, then + 64, ... + 1 — with barriers that wait for LDS only: the two levels that
            // cross waves through LDS, the six inside wave 0 by shuffles; same operands at every level, same bits)
            if (tid < 256) red[0][tid] = c;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < 64) {
                double cs = (red[0][tid] + red[0][tid + 128]) + (red[0][tid + 64] + red[0][tid + 192]);
                for (int d = 32; d > 0; d >>= 1) cs += __shfl_xor(cs, d);
                if (tid == 0) { wmx[0][0] = cs; v.cons[NT + 2 * LT] = cs; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cost_q = wmx[0][0];
        }
        if (tid < N) x = q[tid];
    } else if (tid < N) {
        x = cinj[tid + (size_t)N * t] - v.demand[tid + (size_t)N * t];       // results.jl:58-100
        q[tid] = x;
        // the node's injection change of this iteration: k_slack's (this rank's agents), or — behind an exchange — the summed
        // injection against the previous iteration's (replicated; v.inj is overwritten further down)
        if (sid) sdL[tid] = v.slackGlobal ? x - v.inj[tid + (size_t)N * t] : v.node_dsum[tid + (size_t)N * t];
    }
    {   // imbalance: butterfly inside each of the (at most four) waves that hold nodes, waves in order
        double ps = x;
        for (int d = 32; d > 0; d >>= 1) ps += __shfl_xor(ps, d);
        if (lane == 0 && tid < 256) wsum[tid >> 6] = ps;
    }
    // (only LDS data crosses this kernel's barriers, and what a later phase loads it waits for itself: the barriers wait for LDS
    // alone — __syncthreads() also waits for the acknowledgement of every store issued so far)
#define DOPF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    int anyWalk = 0;
    if (sid && !QUIET) anyWalk = __syncthreads_or(wf);
    else DOPF_LDS_BARRIER();

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_reason[1] = wall_clock64();
#endif
    if (halt) return;                                    // (uniform)
    if (sid && !QUIET && anyWalk) {
        // (rare, uniform) a line of t has its switch point inside some node's window: the parts of those lines' sums, node
        // by node, before anything else is held in registers
        if (tid < N) {
            winL[tid] = v.node_win[tid];
            naL[tid] = (double)((v.node_gen_beg[tid + 1] - v.node_gen_beg[tid]) + (v.node_sto_beg[tid + 1] - v.node_sto_beg[tid]));
        }
        __syncthreads();
        if (mine && wf) {
            double pu, pk;
            slack_sum_cases_both(v.self, l, t, nbeg, nend, sdL, winL, naL, f_old, F, aU_old, aK_old, &pu, &pk);
            red[1][tid] = pu; red[2][tid] = pk;
        }
    }
    if (tid < N) v.inj[tid + (size_t)N * t] = x;
    const double sum = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    // (the imbalance and the new lambda are needed again at the kernel's end: parked in LDS, not in registers the dot
    // products need — the compiler spilled them to scratch)
    __shared__ double parked[2];
    if (tid == 0) {
        const double ln0 = UPDATE ? lam_old + g * sum : lam_old;                 // update_duals.jl:8-13
        if (UPDATE) { v.s_used[t] = s_old; v.lam_used[t] = lam_old; v.lam[t] = ln0; }
        v.s[t] = sum;
        parked[0] = sum; parked[1] = ln0;
    }
    double mo = 0.0, ro = 0.0, sU = 0.0, sK = 0.0, reach = 0.0;
    if (lt) {
        mo = v.mu[i]; ro = v.rho[i]; reach = v.line_reach[l];
        if (!sid) { sU = cU[i]; sK = cK[i]; }
    }
    double f = 0.0, ds = 0.0;
    if (l < L) {
        auto use = [&](const double (&h)[kFlight], int n0) {
#pragma unroll
            for (int u = 0; u < kFlight; ++u) f += h[u] * (n0 + u < nend ? q[n0 + u] : 0.0);
            if (sid) {
#pragma unroll
                for (int u = 0; u < kFlight; ++u)
                    if (n0 + u < nend) ds = fma(h[u], sdL[n0 + u], ds);
            }
        };
        if (QUIET) use(h0, nbeg);                                              // (on its way since the kernel's entry)
        for (int n0 = QUIET ? nbeg + kFlight : nbeg; n0 < nend; n0 += kFlight) {      // kFlight rows of ptdf in flight
            double h[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; ++u) h[u] = n0 + u < nend ? v.ptdf[l + (size_t)L * (n0 + u)] : 0.0;
            use(h, n0);
        }
    }

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_reason[2] = wall_clock64();
#endif
    red[0][tid] = f;
    if (sid && !wf) {
        double pu = 0.0, pk = 0.0;
        if (mine) {
            const SlackCase c0 = slack_case(g, w2, inv, 0.0, f_old, F, aU_old, aK_old, 0.0);
            if (c0.aU > 0.0) pu = slack_sum_plain(0, cntp, c0.aU, w2 * inv, ds);
            if (c0.aK > 0.0) pk = slack_sum_plain(1, cntp, c0.aK, w2 * inv, ds);
        }
        red[1][tid] = pu; red[2][tid] = pk;
    }
    DOPF_LDS_BARRIER();

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_reason[3] = wall_clock64();
#endif
    double rm = 0.0, rr = 0.0;
    int flag = 0, nzl = 0;
    if (lt) {
        f = ((red[0][l] + red[0][256 + l]) + red[0][512 + l]) + red[0][768 + l];
        if (sid) {                               // (stored: dopf_get_consensus, and what a later k_reduce-chain call would have left)
            sU = ((red[1][l] + red[1][256 + l]) + red[1][512 + l]) + red[1][768 + l];
            sK = ((red[2][l] + red[2][256 + l]) + red[2][512 + l]) + red[2][768 + l];
            v.cons[NT + i] = sU; v.cons[NT + LT + i] = sK;
        }
        double aU = aU_old, aK = aK_old, mn = mo, rn = ro;
        if (UPDATE) {
            v.flow_used[i] = f_old; v.avgU_used[i] = aU_old; v.avgK_used[i] = aK_old;
            aU = v.invA * sU; aK = v.invA * sK;                                  // results.jl:108-112
            v.avgU[i] = aU; v.avgK[i] = aK;
            mn = (mo + g * (f + aU - F)) * (aU <= v.mask_thr ? 1.0 : 0.0);       // update_duals.jl:18-25
            rn = (ro + g * (aK - f - F)) * (aK <= v.mask_thr ? 1.0 : 0.0);       // :30-37
            v.mu_used[i] = mo; v.rho_used[i] = ro;
            v.mu[i] = mn; v.rho[i] = rn;
            rm = fabs(mn - mo); rr = fabs(rn - ro);
        }
        v.flow[i] = f;                                                            // results.jl:114
        // what the next slack sums of (l,t) will need (k_slack / k_reduce)
        flag = slack_needs_cases(g, w2, inv, f, F, aU, aK, reach) ? 1 : 0;
        v.walk_flag[i] = flag;
        // price half: mu - rho, and the terms of the linear Psi pieces (used when no line of t is flagged)
        const double dx = mn - rn;
        dd[l] = dx;
        nzl = dx != 0.0;
        const double U0 = dmax0((g * aU - w2 * (f - F)) * inv), K0 = dmax0((g * aK + w2 * (f + F)) * inv);
        Gl[l] = w2 * ((f + U0 - F) - (K0 - f - F));
        Sl[l] = w2 * (2.0 - ((U0 > 0.0 ? 1.0 : 0.0) + (K0 > 0.0 ? 1.0 : 0.0)) * w2 * inv);
    }
    if (tid < 256) {                             // the four waves that hold lines: OR / max inside the wave (order independent)
        const int a = __any(flag), z = __any(nzl);
        for (int d = 32; d > 0; d >>= 1) { rm = fmax(rm, __shfl_xor(rm, d)); rr = fmax(rr, __shfl_xor(rr, d)); }
        if (lane == 0) { wany[tid >> 6] = a; wnz[tid >> 6] = z; wmx[0][tid >> 6] = rm; wmx[1][tid >> 6] = rr; }
    }
    DOPF_LDS_BARRIER();
    const int anyNeed = wany[0] | wany[1] | wany[2] | wany[3];
    const int nz = wnz[0] | wnz[1] | wnz[2] | wnz[3];
    const bool lin = !anyNeed;
    // Stop test, part 1 (thread 0). The residual maxima of this timestep go into this iteration's set of maxima words as
    // device-scope atomics nobody waits for (the host decodes them), and ONE 64-bit add draws the ticket: low word = blocks
    // that are this far, high word = how many of them saw a residual >= eps. The block that draws the last ticket knows
    // from the returned word alone whether the iteration converged — no second round trip for the maxima, and the one
    // round trip there is runs under the end of the price half (see below).

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_cyc[0] = wall_clock64();
#endif
    // ---- price half: thread (line part pp, node n) -------------------------------------------------------------
    const int NP = N <= 128 ? 128 : 256, P = 1024 / NP;
    const int pp = tid / NP, n = tid - pp * NP;
    const int Lc = (((L + P - 1) / P) + 7) & ~7, lbeg = pp * Lc, lend = min(L, lbeg + Lc);
    double pr = 0.0, psx = 0.0, sl = 0.0;
    const bool priced = n < N && (nz || lin);
    auto batch_use = [&](const double (&h)[kFlight], int l0) {
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            if (l0 + u < lend) {
                pr += h[u] * dd[l0 + u];
                if (lin) { psx += h[u] * Gl[l0 + u]; sl += h[u] * h[u] * Sl[l0 + u]; }
            }
        }
    };
    // batches of kFlight rows (ptdfT[n + N l]: coalesced over the nodes): all but the last ...
    int l0 = lbeg;
    if (priced)
        for (; l0 + kFlight < lend; l0 += kFlight) {
            double h[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; ++u) h[u] = v.ptdfT[n + (size_t)N * (l0 + u)];
            batch_use(h, l0);
        }
    if (priced) {                                   // ... and the last one
        double h[kFlight];
#pragma unroll
        for (int u = 0; u < kFlight; ++u) h[u] = l0 + u < lend ? v.ptdfT[n + (size_t)N * (l0 + u)] : 0.0;
        batch_use(h, l0);
    }

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_cyc[1] = wall_clock64();
#endif
    // The ticket is drawn HERE, behind the last use of a load: a wave's vector memory operations retire through one counter,
    // and with an atomic among them the compiler can only wait for all of them (vmcnt(0)) — drawn in front of the price
    // half, it held that wave's PTDF rows back for its 2.5 us. What is left of the round trip runs under the barrier, the
    // sums and the stores below. (The address goes through a register the compiler knows nothing about: for an address it
    // can prove uniform it rewrites the add into a wave-wide one whose result it reads back at once.)
    unsigned long long tk_ = 0, viol_ = 0;
    if (UPDATE && tid == 0) {
        const double rl = fabs(parked[1] - lam_old);
        const double bm = fmax(fmax(wmx[0][0], wmx[0][1]), fmax(wmx[0][2], wmx[0][3]));
        const double br = fmax(fmax(wmx[1][0], wmx[1][1]), fmax(wmx[1][2], wmx[1][3]));
        unsigned long long *rb = v.st->resbits2[spre.iters_total & 1];
        if (rl > 0.0) (void)__hip_atomic_fetch_max(&rb[0], (unsigned long long)__double_as_longlong(rl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bm > 0.0) (void)__hip_atomic_fetch_max(&rb[1], (unsigned long long)__double_as_longlong(bm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (br > 0.0) (void)__hip_atomic_fetch_max(&rb[2], (unsigned long long)__double_as_longlong(br), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        viol_ = (rl < v.eps && bm < v.eps && br < v.eps) ? 0ull : 1ull;               // convergence.jl:15-23
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        tk_ = atomicAdd(v.dual_ticket + zero, 1ull | (viol_ << 32) | ((unsigned long long)(anyNeed ? 1 : 0) << 48));
    }
    red[0][tid] = pr; red[1][tid] = psx; red[2][tid] = sl;
    DOPF_LDS_BARRIER();

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_cyc[2] = wall_clock64();
#endif
    if (pp == 0 && n < N) {
        for (int k = 1; k < P; ++k) { pr += red[0][k * NP + n]; psx += red[1][k * NP + n]; sl += red[2][k * NP + n]; }
        const size_t at = n + (size_t)N * t;
        pr += parked[1];
        v.price[at] = pr;
        if (lin) {
            v.tb_m[at] = 0;
            v.tb_psi0[at] = (pr + g * parked[0]) + psx;
            v.tb_slope[at * (v.M2 + 1)] = g + sl;
        }
    }
    if (tid == 0) { v.walk_any[t] = anyNeed ? 1 : 0; v.tab_skip[t] = lin ? 1 : 0; }
    if (!lin && v.tablesInDual) {
        // A line's switch point lies inside some node's window: this timestep's tables are built HERE, by this block's
        // waves (node n by wave n % tablesInDual, each wave with LDS scratch of its own behind the kernel's other dynamic
        // LDS), from the prices, flows and mean slacks the block has just stored — instead of by a launch of their own
        // (k_tables) that in the settled state finds nothing to do: 5 us + a kernel boundary per iteration.
        __syncthreads();                             // this block's stores above are visible to all its waves
        const int TW = v.tablesInDual, wv = tid >> 6;
        double *tsh = sh + v.dualRowsOff + (size_t)wv * (4 * v.M2 + 1);
        if (wv < TW)
            for (int nn = wv; nn < N; nn += TW) build_table(v, nn, t, tsh);
    }

#ifdef DOPF_DUAL_STAMPS
    if (UPDATE && tid == 0 && t == T / 2) v.st->dbg_cyc[3] = wall_clock64();
#endif
    if (UPDATE && tid == 0) {
        // Stop test, part 2: whoever drew the last ticket has every block's verdict in the word it got back
        if (t == 0) v.st->total_cost = QUIET ? cost_q : v.cons[NT + 2 * LT];
        const bool last_ = (unsigned)(tk_ & 0xffffffffull) == (unsigned)(T - 1);
#ifdef DOPF_DUAL_STAMPS
        if (t == T / 2) v.st->dbg_cyc[4] = wall_clock64();        // (the ticket has returned)
#endif
        if (last_) {
            Status *st = v.st;
            const int par = spre.iters_total & 1;
            st->resbits2[1 - par][0] = st->resbits2[1 - par][1] = st->resbits2[1 - par][2] = 0ull;      // the next iteration's set
            *v.dual_ticket = 0ull;
            int conv = spre.converged, it = spre.iteration;
            if (it != 1) {                                                        // convergence.jl:3
                conv = ((tk_ >> 32) & 0xffffull) + viol_ == 0ull;
                st->converged = conv;
                st->res_set = par;
            }
            st->iters_total = spre.iters_total + 1;
            if (!conv) it += 1;                                                   // convergence.jl:25-30
            st->iteration = it;
            // timesteps with a flagged line for the next iteration: the quiet chain cannot run it (halt = 2: parked, not finished)
            const int walks = (int)(tk_ >> 48) + (anyNeed ? 1 : 0);
            st->walk_last = walks;
            const int stop = conv || (v.max_iters > 0 && it > v.max_iters);
            st->halt = stop ? 1 : (((QUIET || v.slackGlobal) && walks > 0) ? 2 : 0);
        }
    }
}

#undef DOPF_LDS_BARRIER

// dual step + prices + stop test in ONE block when the consensus state is small (every copper-plate case):
// saves a launch per iteration, which is what the small configurations are bound by
// XCHG (copper plate, peer exchange): the sum over the ranks happens HERE, between the slice sums and the dual step —
// the rank's vector goes from registers into every peer's receive area, the copies come back from the own area; no
// launch is added to the single-GPU chain (protocol: see k_xchg).
template <bool UPDATE, bool XCHG>
__global__ __launch_bounds__(256) void k_dual_price_small(DevView v, XchgView x)
{
    const int halt = UPDATE ? v.st->halt : 0;              // (looked at once the loads below are on their way)
    __shared__ double red[8][256];
    const int tid = threadIdx.x;
    const size_t NT = (size_t)v.N * v.T, LT = (size_t)v.L * v.T;
    const size_t n1 = NT > LT ? NT : LT;
    if (UPDATE && v.sliceDual && v.L == 0) {
        // Copper plate, single-GPU chain (config1, config2, config4): the whole tail of the iteration in registers and
        // LDS — slice sums, injections, imbalance, lambda step, prices, residual, stop test — with every global load
        // issued up front and nothing read back from memory that this block has just written (the general path below
        // pays an L2 round trip for each of cons -> inj -> lam -> price). Same arithmetic, same order of additions.
        const int RB = v.reduceRB, N = v.N, T = v.T, R = 8;
        const int r = tid >> 5, tt = tid & 31;
        __shared__ double injL[256], lamL[256], wmax[4];
        const size_t me = tid;                                // entry n + N*t of this thread in the later stages
        const StatusPre spre = status_load(v);                // (with the other loads, not behind the last barrier)
        const double dem = me < NT ? v.demand[me] : 0.0;
        const double lam_old = tid < T ? v.lam[tid] : 0.0, s_old = tid < T ? v.s[tid] : 0.0;
        const double cslice = tid < 64 && tid < RB ? v.part2_cost[tid] : 0.0;
        double sc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const size_t i = (size_t)c * 32 + tt;
            sc[c] = 0.0;
            if (i < NT) {
                const int n = (int)(i % N), t = (int)(i / N);
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = r + R * k;
                    x[k] = q < RB ? v.part2[((size_t)n * RB + q) * T + t] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sc[c] += x[k];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) red[c][tid] = sc[c];
        __syncthreads();
        if (halt) return;                                     // (uniform; nothing has been stored or sent yet)
        double tot = 0.0;
        if (me < NT) {
            const int c = (int)(me >> 5), t5 = (int)(me & 31);
            for (int q = 0; q < R; ++q) tot += red[c][q * 32 + t5];
        }
        double ctot = cslice;                                 // cost slices: butterfly in wave 0 (a fixed order too)
        if (tid < 64)
            for (int d = 32; d > 0; d >>= 1) ctot += __shfl_xor(ctot, d);
        if (XCHG) {
            __shared__ int bad;
            const int W = x.world, rk = x.me;
            const unsigned long long seq = (unsigned long long)v.st->iters_total + 1ull;
            const size_t par = (size_t)(seq & 1ull), n = x.n;          // n = NT + 1: injections | cost
            if (tid == 0) bad = 0;
            for (int q = 0; q < W; ++q) {
                const int r = (rk + 1 + q) % W;
                double *dst = x.data[r] + (par * W + rk) * n;
                if (me < NT) dst[me] = tot;
                if (tid == 0) dst[NT] = ctot;
            }
            __threadfence_system();
            __syncthreads();
            if (tid < W) {
                __hip_atomic_store(x.flags[tid] + (par * W + rk) * x.nchunks, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long *f = x.flags[rk] + (par * W + tid) * x.nchunks;
                const unsigned long long t0 = wall_clock64();
                bool ok = v.st->xchg_timeout == 0;
                while (ok && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > x.timeout_ticks) ok = false;
                }
                if (!ok) atomicOr(&bad, 1);
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            if (bad) {
                if (tid == 0) v.st->xchg_timeout = 1;
                return;
            }
            const double *mine = x.data[rk] + par * W * n;
            if (me < NT) {
                tot = mine[me];
                for (int r = 1; r < W; ++r) tot += mine[(size_t)r * n + me];
            }
            if (tid == 0) {
                ctot = mine[NT];
                for (int r = 1; r < W; ++r) ctot += mine[(size_t)r * n + NT];
            }
        }
        if (me < NT) {
            v.cons[me] = tot;
            const double xi = tot - dem;                      // results.jl:58-100
            v.inj[me] = xi;
            injL[me] = xi;
        }
        if (tid == 0) { v.cons[NT] = ctot; v.st->total_cost = ctot; }
        __syncthreads();
        double rl = 0.0;
        if (tid < T) {
            double sum = 0.0;
            for (int n0 = 0; n0 < N; n0 += 8) {               // same grouping as dual_body
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = n0 + u < N ? injL[n0 + u + (size_t)N * tid] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) sum += x[u];
            }
            v.s_used[tid] = s_old;
            v.s[tid] = sum;
            const double ln = lam_old + v.gamma * sum;        // update_duals.jl:8-13
            v.lam_used[tid] = lam_old;
            v.lam[tid] = ln;
            lamL[tid] = ln;
            rl = fabs(ln - lam_old);
        }
        for (int d = 32; d > 0; d >>= 1) rl = fmax(rl, __shfl_xor(rl, d));
        if ((tid & 63) == 0) wmax[tid >> 6] = rl;
        __syncthreads();
        if (me < NT) v.price[me] = lamL[me / N];              // no lines: the nodal price is lambda
        if (tid == 0) status_update(v, spre, fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3])), 0.0, 0.0);
        return;
    }
    if (halt) return;
    if (UPDATE && v.sliceDual) {
        // level 2 of the consensus sum, here instead of behind a ticket in k_reduce: slice order, so the bits are
        // the ones the two-level kernel produces
        const int RB = v.reduceRB, N = v.N, T = v.T, R = 8;
        const int r = tid >> 5, tt = tid & 31;            // 8 slice lanes x 32 entries, as in k_reduce
        // N*T <= 256 here (slice_dual() in dopf_api.hip): up to 8 chunks of 32 entries, every load of every chunk
        // issued before the first use, ONE barrier
        double sc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const size_t i = (size_t)c * 32 + tt;
            sc[c] = 0.0;
            if (i < NT) {
                const int n = (int)(i % N), t = (int)(i / N);
                double x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = r + R * k;
                    x[k] = q < RB ? v.part2[((size_t)n * RB + q) * T + t] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) sc[c] += x[k];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) red[c][tid] = sc[c];
        __syncthreads();
        if (r == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const size_t i = (size_t)c * 32 + tt;
                if (i < NT) {
                    double tot = 0.0;
                    for (int q = 0; q < R; ++q) tot += red[c][q * 32 + tt];
                    v.cons[i] = tot;
                }
            }
        }
        __syncthreads();
        // cost slices: loaded side by side (one lane adding them from memory is a chain of RB dependent loads —
        // 8 us for 64 slices)
        if (tid < 64) {                                   // wave 0: a butterfly is a fixed order too
            double c = tid < RB ? v.part2_cost[tid] : 0.0;
            for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
            if (tid == 0) v.cons[NT + 2 * LT] = c;
        }
        __syncthreads();
    }
    double rl = 0.0, rm = 0.0, rr = 0.0;
    if (v.L > 0) {
        for (int t = tid; t < v.T; t += 256) v.walk_any[t] = 0;
        __syncthreads();
    }
    for (size_t i = tid; i < n1; i += 256) dual_body<UPDATE>(v, i, rl, rm, rr);
    if (UPDATE) {
        red[0][tid] = rl; red[1][tid] = rm; red[2][tid] = rr;
        if (tid == 0) v.st->total_cost = v.cons[NT + 2 * LT];
    }
    __syncthreads();                       // the block's own global writes (new duals) are visible after this
    for (size_t i = tid; i < NT; i += 256) price_body(v, i);
    if (UPDATE) {
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                red[0][tid] = fmax(red[0][tid], red[0][tid + s]);
                red[1][tid] = fmax(red[1][tid], red[1][tid + s]);
                red[2][tid] = fmax(red[2][tid], red[2][tid + s]);
            }
            __syncthreads();
        }
        if (tid == 0) status_update(v, red[0][0], red[1][0], red[2][0]);
    }
}

// dynamic LDS of k_dual_price_t1024: its own vectors, plus table scratch for tablesInDual waves / the rows of its timestep (dopf_create)
static size_t t1024_lds(const DevView &v)
{
    const size_t bytes = (size_t)v.dualLdsBytes;
    static std::atomic<unsigned long long> raised{0ull};
    if (bytes > 48 * 1024 && first_time_on_this_device(raised)) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_dual_price_t1024<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_dual_price_t1024<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_dual_price_t1024<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_dual_price_t1024<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }
    return bytes;
}

void launch_dual(const DevView &v, hipStream_t s, const XchgView *xd)
{
    const size_t NT = (size_t)v.N * v.T, LT = (size_t)v.L * v.T;
    const size_t n1 = NT > LT ? NT : LT;
    if (n1 <= kSmallConsensus) {
        if (xd) hipLaunchKernelGGL((k_dual_price_small<true, true>), dim3(1), dim3(256), 0, s, v, *xd);
        else hipLaunchKernelGGL((k_dual_price_small<true, false>), dim3(1), dim3(256), 0, s, v, XchgView{});
        return;
    }
    if (v.L > 0 && v.L <= 256 && v.N <= 256 && !v.splitDual) {
        if (v.slackInDual && v.quiet) hipLaunchKernelGGL((k_dual_price_t1024<true, true, true>), dim3(v.T), dim3(1024), t1024_lds(v), s, v);
        else if (v.slackInDual) hipLaunchKernelGGL((k_dual_price_t1024<true, true>), dim3(v.T), dim3(1024), t1024_lds(v), s, v);
        else hipLaunchKernelGGL(k_dual_price_t1024<true>, dim3(v.T), dim3(1024), t1024_lds(v), s, v);
        return;
    }
    if ((size_t)std::max(v.N, 3 * v.L) * sizeof(double) <= 48 * 1024) {
        hipLaunchKernelGGL(k_dual_t<true>, dim3(v.T * (v.L > 0 ? (v.L + 63) / 64 : 1)), dim3(256), (size_t)v.N * sizeof(double), s, v);
        hipLaunchKernelGGL(k_price_t<true>, dim3(v.T), dim3(256), 3 * (size_t)v.L * sizeof(double), s, v);
        return;
    }
    if (v.L > 0) hipMemsetAsync(v.walk_any, 0, sizeof(int) * v.T, s);
    hipLaunchKernelGGL(k_dual<true>, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, v);
    hipLaunchKernelGGL(k_price<true>, dim3((unsigned)((NT + 255) / 256)), dim3(256), 0, s, v);
}

// ------------------------------------------------------------------------------------------------
// set_state support: consensus sums straight from the primal arrays (slow path, not on the hot loop)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_derive_cons(DevView v)
{
    const int N = v.N, T = v.T;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * T) return;
    const int n = (int)(i % N), t = (int)(i / N);
    double sum = 0.0;
    for (int g = v.node_gen_beg[n]; g < v.node_gen_beg[n + 1]; ++g) sum += v.P[(size_t)g * T + t];
    for (int s = v.node_sto_beg[n]; s < v.node_sto_beg[n + 1]; ++s) sum += v.D[(size_t)s * T + t] - v.C[(size_t)s * T + t];
    v.cons[i] = sum;
    if (v.L > 0) v.prev_node[i] = sum;        // what the next iteration's node changes refer to
}

__global__ __launch_bounds__(256) void k_derive_level(DevView v)
{
    const int T = v.T;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= v.S) return;
    double e = 0.0;
    for (int t = 0; t < T; ++t) {
        e += v.C[(size_t)s * T + t] - v.D[(size_t)s * T + t];
        v.E[(size_t)s * T + t] = e;
    }
}

// ResultNode.{generation, discharge, charge} (results.jl:19-35): per node and timestep the sums over the node's units,
// on request. Block = (node, chunk of 32 timesteps), 8 agent lanes per timestep, fixed-order sums.
__global__ __launch_bounds__(256) void k_node_results(DevView v, double *gen, double *dis, double *chg)
{
    __shared__ double red[3][256];
    const int N = v.N, T = v.T, TC = (T + 31) / 32;
    const int n = blockIdx.x / TC, t = (blockIdx.x % TC) * 32 + (threadIdx.x & 31), r = threadIdx.x >> 5;
    double g = 0.0, d = 0.0, c = 0.0;
    if (t < T) {
        for (int a = v.node_gen_beg[n] + r; a < v.node_gen_beg[n + 1]; a += 8) g += v.P[(size_t)a * T + t];
        for (int a = v.node_sto_beg[n] + r; a < v.node_sto_beg[n + 1]; a += 8) { d += v.D[(size_t)a * T + t]; c += v.C[(size_t)a * T + t]; }
    }
    red[0][threadIdx.x] = g; red[1][threadIdx.x] = d; red[2][threadIdx.x] = c;
    __syncthreads();
    if (r == 0 && t < T) {
        double sg = 0.0, sd = 0.0, sc = 0.0;
        for (int q = 0; q < 8; ++q) { sg += red[0][q * 32 + threadIdx.x]; sd += red[1][q * 32 + threadIdx.x]; sc += red[2][q * 32 + threadIdx.x]; }
        const size_t i = (size_t)n + (size_t)N * t;
        gen[i] = sg; dis[i] = sd; chg[i] = sc;
    }
}

// ResultStorage.level (results.jl:4): E = cumsum(C - D), on request (the solve kernels do not store it)
void launch_derive_level(const DevView &v, hipStream_t s)
{
    if (v.S > 0) hipLaunchKernelGGL(k_derive_level, dim3((unsigned)((v.S + 255) / 256)), dim3(256), 0, s, v);
}

// Result.penalty_term (reference src/structures/results.jl:66-70 with sum_up, src/helpers/penalty_terms.jl:1-6): the three
// penalty vectors of src/optimization/penalty_terms.jl:3-37 summed over ALL agents, from the injection changes the last
// x-update left on the device (dltG / dltS: networks with DOPF_F_KEEP_DELTAS) and the consensus state that solve read. One
// block per (node, 64 timesteps), thread = timestep: the node's agents are walked in list order and, per agent, the lines in
// order — dopf_get_agent_penalty's arithmetic, agent by agent — then the host adds the N rows in node order.
__global__ __launch_bounds__(64) void k_penalty_sums(DevView v, double *out /* [3][N][T] */)
{
    const int TB = (v.T + 63) / 64;
    const int n = blockIdx.x / TB, t = (blockIdx.x % TB) * 64 + threadIdx.x;
    if (t >= v.T) return;
    const int L = v.L, T = v.T;
    const double w2 = 2.0 * v.w_flow, g = v.gamma, s = v.s_used[t];
    double eb = 0.0, up = 0.0, lo = 0.0;
    for (int kind = 0; kind < 2; ++kind) {
        const int *beg = kind == 0 ? v.node_gen_beg : v.node_sto_beg;
        const double *dlt = kind == 0 ? v.dltG : v.dltS;
        for (int a = beg[n]; a < beg[n + 1]; ++a) {
            const double dl = dlt[(size_t)a * T + t];
            eb += (s + dl) * (s + dl);                                               // penalty_terms.jl:3-7
            double au = 0.0, al = 0.0;
            for (int l = 0; l < L; ++l) {
                const size_t i = l + (size_t)L * t;
                const double F = v.fmax[l];
                const double fl = v.flow_used[i] + v.ptdf[l + (size_t)L * n] * dl;
                const double u = fmax(0.0, (g * v.avgU_used[i] - w2 * (fl - F)) / (w2 + g));
                const double k = fmax(0.0, (g * v.avgK_used[i] + w2 * (fl + F)) / (w2 + g));
                au += (fl + u - F) * (fl + u - F);                                    // penalty_terms.jl:10-20
                al += (k - fl - F) * (k - fl - F);                                    // :23-37
            }
            up += au; lo += al;
        }
    }
    const size_t NT = (size_t)v.N * T, at = (size_t)n * T + t;
    out[at] = eb; out[NT + at] = up; out[2 * NT + at] = lo;
}

void launch_penalty_sums(const DevView &v, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_penalty_sums, dim3(v.N * ((v.T + 63) / 64)), dim3(64), 0, s, v, out);
}

void launch_node_results(const DevView &v, double *gen, double *dis, double *chg, hipStream_t s)
{
    hipLaunchKernelGGL(k_node_results, dim3(v.N * ((v.T + 31) / 32)), dim3(256), 0, s, v, gen, dis, chg);
}

void launch_derive(const DevView &v, hipStream_t s, bool from_primal)
{
    const size_t NT = (size_t)v.N * v.T, LT = (size_t)v.L * v.T;
    const size_t n1 = NT > LT ? NT : LT;
    if (from_primal) {        // serial over a node's agents: fine for tests / resume, not a hot path
        hipLaunchKernelGGL(k_derive_cons, dim3((unsigned)((NT + 255) / 256)), dim3(256), 0, s, v);
        if (v.S > 0) hipLaunchKernelGGL(k_derive_level, dim3((unsigned)((v.S + 255) / 256)), dim3(256), 0, s, v);
    }
    if (n1 <= kSmallConsensus) {
        hipLaunchKernelGGL((k_dual_price_small<false, false>), dim3(1), dim3(256), 0, s, v, XchgView{});
        return;
    }
    if (v.L > 0 && v.L <= 256 && v.N <= 256 && !v.splitDual) {
        hipLaunchKernelGGL(k_dual_price_t1024<false>, dim3(v.T), dim3(1024), t1024_lds(v), s, v);
        return;
    }
    if ((size_t)std::max(v.N, 3 * v.L) * sizeof(double) <= 48 * 1024) {
        hipLaunchKernelGGL(k_dual_t<false>, dim3(v.T * (v.L > 0 ? (v.L + 63) / 64 : 1)), dim3(256), (size_t)v.N * sizeof(double), s, v);
        hipLaunchKernelGGL(k_price_t<false>, dim3(v.T), dim3(256), 3 * (size_t)v.L * sizeof(double), s, v);
        return;
    }
    if (v.L > 0) hipMemsetAsync(v.walk_any, 0, sizeof(int) * v.T, s);
    hipLaunchKernelGGL(k_dual<false>, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, v);
    hipLaunchKernelGGL(k_price<false>, dim3((unsigned)((NT + 255) / 256)), dim3(256), 0, s, v);
}

}  // namespace dopf

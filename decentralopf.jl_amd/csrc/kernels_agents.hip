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

namespace dopf {

__device__ __forceinline__ double clampd(double v, double lo, double hi)
{
    return fmin(fmax(v, lo), hi);
}

// ------------------------------------------------------------------------------------------------
// generators
// ------------------------------------------------------------------------------------------------
//
// Block = one Item (agents [a0,a1) at one node). Thread (r, tt): timestep tt (+TT, +2TT, ...) of agents
// a0 + r, a0 + r + R, ...; with T <= 512 the block sweeps a contiguous range of P (agent-major), so
// every wave access is a dense 512-byte line set. Algorithmic traffic per update: 8 B read + 8 B write
// of P per (g,t) + 20 B of parameters per agent (L1-broadcast to the T lanes that share an agent).
template <bool LINES>
__global__ __launch_bounds__(512) void k_gen_update(DevView v)
{
    if (v.st->halt) return;
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
            if (!LINES) {
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
            } else {
                const size_t at = (size_t)it.node + (size_t)N * t;
                const int m = v.tb_m[at];
                const double *beta = v.tb_beta + at * v.M2, *psi = v.tb_psi + at * v.M2;
                const double *slope = v.tb_slope + at * (v.M2 + 1);
                const double psi0 = v.tb_psi0[at];
                for (int g = it.a0 + r; g < it.a1; g += R) {
                    const size_t e = (size_t)g * T + t;
                    const double mc = v.gen_mc[g], pm = v.gen_pmax[g];
                    const double p0 = v.P[e];
                    double dl;
                    if (m == 0) {
                        dl = -(mc + psi0) / (slope[0] + w);
                    } else {
                        int lo = 0, hi = m;           // first kink with psi + w beta >= -mc
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (psi[mid] + w * beta[mid] >= -mc) hi = mid; else lo = mid + 1;
                        }
                        const int a = lo < m ? lo : m - 1;
                        dl = beta[a] - (mc + psi[a] + w * beta[a]) / (slope[lo] + w);
                    }
                    const double pn = clampd(p0 + dl, 0.0, pm);
                    v.P[e] = pn;
                    v.dltG[e] = pn - p0;
                    acc += pn;
                    cost += mc * pn;
                }
            }
        }
        // fixed-order reduction over the R agent lanes that share a timestep
        red[tid] = acc;
        __syncthreads();
        if (r == 0 && t < T) {
            double sum = 0.0;
            for (int q = 0; q < R; ++q) sum += red[q * TT + tt];
            v.part_ginj[(size_t)blockIdx.x * T + t] = sum;
        }
        __syncthreads();
    }
    red[tid] = cost;
    __syncthreads();
    for (int sft = 256; sft > 0; sft >>= 1) {
        if (tid < sft) red[tid] += red[tid + sft];
        __syncthreads();
    }
    if (tid == 0) v.part_gcost[blockIdx.x] = red[0];
}

void launch_gen_update(const DevView &v, hipStream_t s)
{
    if (v.nGenItems == 0) return;
    if (v.L > 0) hipLaunchKernelGGL(k_gen_update<true>, dim3(v.nGenItems), dim3(512), 0, s, v);
    else hipLaunchKernelGGL(k_gen_update<false>, dim3(v.nGenItems), dim3(512), 0, s, v);
}

// ------------------------------------------------------------------------------------------------
// storages
// ------------------------------------------------------------------------------------------------

// argmin over [0,pm]^2 of the strictly convex quadratic with gradient (a D - b C - rD, a C - b D - rC),
// a > b >= 0; sg = d(C - D)/d(nu) on the active piece (rD falls, rC rises with nu at unit rate).
__device__ __forceinline__ void box2(double a, double b, double rD, double rC, double pm, double &D,
                                     double &C, double &sg)
{
    const double ia = 1.0 / a;
    const double Df = clampd((a * rD + b * rC) / (a * a - b * b), 0.0, pm);
    const double Cf = (rC + b * Df) * ia;
    if (Cf < 0.0) { C = 0.0; D = clampd(rD * ia, 0.0, pm); }
    else if (Cf > pm) { C = pm; D = clampd((rD + b * pm) * ia, 0.0, pm); }
    else { C = Cf; D = Df; }
    const bool fD = D > 0.0 && D < pm, fC = C > 0.0 && C < pm;
    sg = (fD && fC) ? 2.0 / (a + b) : ((fD || fC) ? ia : 0.0);
}

template <int LPS>
__device__ __forceinline__ unsigned long long group_bits(bool pred, int gbase)
{
    const unsigned long long b = __ballot(pred);
    if (LPS == 64) return b;
    return (b >> gbase) & ((1ull << LPS) - 1ull);
}

struct StoAgent {
    double mc, pm, em;
};

// Block = one Item of storages at one node; a group of LPS lanes owns one storage at a time,
// lane li of chunk c owns timestep c*LPS + li.
template <int LPS, int NCH, bool LINES>
__global__ __launch_bounds__(256) void k_sto_update(DevView v)
{
    if (v.st->halt) return;
    constexpr int NG = 256 / LPS;
    __shared__ double red[NG * NCH * LPS];
    __shared__ double redc[256];
    const int tid = threadIdx.x, lane = tid & 63, li = tid & (LPS - 1), grp = tid / LPS;
    const int gbase = lane & ~(LPS - 1);
    const Item it = v.sto_items[blockIdx.x];
    const int T = v.T, N = v.N;
    const double w = v.w_prox, gam = v.gamma;

    bool val[NCH];
    double th0[NCH], accQ[NCH];
    double accCost = 0.0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = c * LPS + li;
        val[c] = t < T;
        accQ[c] = 0.0;
        th0[c] = (!LINES && val[c]) ? v.price[it.node + N * t] + gam * v.s[t] : 0.0;
    }
    unsigned long long fails = 0;
#ifdef DOPF_STATS
    unsigned long long st_scans = 0, st_loops = 0, st_events = 0;
#endif
    const int nRep = (it.a1 - it.a0 + NG - 1) / NG;

    for (int rep = 0; rep < nRep; ++rep) {
        const int s = it.a0 + rep * NG + grp;
        const bool live = s < it.a1;
        StoAgent ag;
        ag.mc = live ? v.sto_mc[s] : 0.0;
        ag.pm = live ? v.sto_pmax[s] : 0.0;
        ag.em = live ? v.sto_emax[s] : 0.0;
        double D0[NCH], C0[NCH], Dn[NCH], Cn[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool ok = live && val[c];
            const size_t e = (size_t)s * T + (c * LPS + li);
            D0[c] = ok ? v.D[e] : 0.0;
            C0[c] = ok ? v.C[e] : 0.0;
            Dn[c] = 0.0;
            Cn[c] = 0.0;
        }
        const double tol = 1e-11 * (1.0 + ag.em);

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
            // -- forward scan of the clamp-add maps e -> clamp(e + x_t(nu), 0, em) at price nu
            double Dv[NCH], Cv[NCH], Sv[NCH], sg[NCH];
            double e_in = 0.0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int t = c * LPS + li;
                double dd = 0.0, cc = 0.0, s1 = 0.0;
                if (val[c] && t <= k) {
                    const double q0 = D0[c] - C0[c];
                    double theta, kap;
                    if (!LINES) {
                        theta = th0[c] - gam * q0;
                        kap = gam;
                    } else {
                        const size_t at = (size_t)it.node + (size_t)N * t;
                        const int m = v.tb_m[at];
                        const double *beta = v.tb_beta + at * v.M2, *psi = v.tb_psi + at * v.M2;
                        const double *slope = v.tb_slope + at * (v.M2 + 1);
                        double ab = 0.0, ap = v.tb_psi0[at];
                        kap = slope[0];
                        if (m > 0) {
                            int l2 = 0, h2 = m;   // first kink where dlt - (D(z) - C(z) - q0) >= 0
                            while (l2 < h2) {
                                const int mid = (l2 + h2) >> 1;
                                const double z = psi[mid] + nu;
                                const double Dz = clampd(D0[c] - (ag.mc + z) / w, 0.0, ag.pm);
                                const double Cz = clampd(C0[c] - (ag.mc - z) / w, 0.0, ag.pm);
                                if (beta[mid] - (Dz - Cz - q0) >= 0.0) h2 = mid; else l2 = mid + 1;
                            }
                            const int a = l2 < m ? l2 : m - 1;
                            ab = beta[a]; ap = psi[a]; kap = slope[l2];
                        }
                        theta = ap - kap * (ab + q0);
                    }
                    box2(w + kap, kap, w * D0[c] - ag.mc - theta - nu, w * C0[c] - ag.mc + theta + nu,
                         ag.pm, dd, cc, s1);
                }
                Dv[c] = dd; Cv[c] = cc; sg[c] = s1;
                const double x = cc - dd;
                double A = x, LO = 0.0, HI = ag.em;
#pragma unroll
                for (int d = 1; d < LPS; d <<= 1) {
                    const double pA = __shfl_up(A, d, LPS), pLO = __shfl_up(LO, d, LPS), pHI = __shfl_up(HI, d, LPS);
                    if (li >= d) {
                        const double nLO = clampd(pLO + A, LO, HI), nHI = clampd(pHI + A, LO, HI);
                        A += pA; LO = nLO; HI = nHI;
                    }
                }
                const double F = clampd(e_in + A, LO, HI);
                double Fp = __shfl_up(F, 1, LPS);
                if (li == 0) Fp = e_in;
                Sv[c] = Fp + x;
                e_in = __shfl(F, LPS - 1, LPS);
            }

            bool classify = active && mode == 0;
            if (active && mode == 1) {
                // value of S_vv(nu) and its slope: sum of sg over the run of unclamped steps ending at vv
                const int cv = vv / LPS, lv = vv & (LPS - 1);
                double sel = 0.0;
                int jlast = -1;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c == cv) sel = Sv[c];
                    const int t = c * LPS + li;
                    const unsigned long long b = group_bits<LPS>(t < vv && (Sv[c] <= 0.0 || Sv[c] >= ag.em), gbase);
                    if (b) jlast = c * LPS + (63 - __clzll(b));
                }
                const double sv = __shfl(sel, lv, LPS);
                double sl = 0.0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = c * LPS + li;
                    if (t > jlast && t <= vv) sl += sg[c];
                }
#pragma unroll
                for (int d = LPS >> 1; d > 0; d >>= 1) sl += __shfl_xor(sl, d, LPS);
                const double res = sv - target;
                if (res < 0.0) lo = nu; else hi = nu;
                bool conv = fabs(res) <= 1e-12 * (1.0 + ag.em) || rit >= 80;
                double trial = nu;
                if (!conv) {
                    const bool both = lo > -INFINITY && hi < INFINITY;
                    trial = sl > 0.0 ? nu - res / sl : NAN;
                    const bool forceBis = both && rit >= 6 && (rit & 1);
                    if (!(trial > lo && trial < hi) || forceBis) {
                        if (both) trial = 0.5 * (lo + hi);
                        else { trial = (res < 0.0) ? nu + step : nu - step; step *= 4.0; }
                    }
                    if (!(trial > lo && trial < hi)) conv = true;   // bracket is two adjacent doubles
                }
                if (conv) {
                    if (rit >= 80 && fabs(res) > 1e-7 * (1.0 + ag.em) && li == 0) ++fails;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        if (c * LPS + li == vv) { Dn[c] = Dv[c]; Cn[c] = Cv[c]; }
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
                    const int t = c * LPS + li;
                    const unsigned long long b = group_bits<LPS>(val[c] && t <= k && (Sv[c] < -tol || Sv[c] > ag.em + tol), gbase);
                    if (b) vnew = c * LPS + (63 - __clzll(b));
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int t = c * LPS + li;
                    if (t > vnew && t <= k) { Dn[c] = Dv[c]; Cn[c] = Cv[c]; }
                }
                if (vnew < 0) {
                    k = -1;
                } else {
                    const int cv = vnew / LPS, lv = vnew & (LPS - 1);
                    double sel = 0.0;
                    int jlast = -1;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (c == cv) sel = Sv[c];
                        const int t = c * LPS + li;
                        const unsigned long long b = group_bits<LPS>(t < vnew && (Sv[c] <= 0.0 || Sv[c] >= ag.em), gbase);
                        if (b) jlast = c * LPS + (63 - __clzll(b));
                    }
                    const double sv = __shfl(sel, lv, LPS);
                    double sl = 0.0;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int t = c * LPS + li;
                        if (t > jlast && t <= vnew) sl += sg[c];
                    }
#pragma unroll
                    for (int d = LPS >> 1; d > 0; d >>= 1) sl += __shfl_xor(sl, d, LPS);
                    vv = vnew;
                    target = sv < 0.0 ? 0.0 : ag.em;
                    const double res = sv - target;
                    lo = -INFINITY; hi = INFINITY;
                    if (res < 0.0) lo = nu; else hi = nu;
                    step = 1.0 + fabs(nu);
                    if (sl > 0.0) nu -= res / sl;
                    else { nu = (res < 0.0) ? nu + step : nu - step; step *= 4.0; }
                    mode = 1;
                    rit = 0;
#ifdef DOPF_STATS
                    if (li == 0) { ++st_events; if (!(sl > 0.0)) st_loops += (1ull << 32); }
#endif
                }
            }
        }

        // ---- level E = cumsum(C - D), outputs, partial sums ----------------------------------------
        double carry = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            double x = Cn[c] - Dn[c];
#pragma unroll
            for (int d = 1; d < LPS; d <<= 1) {
                const double px = __shfl_up(x, d, LPS);
                if (li >= d) x += px;
            }
            const double ev = carry + x;
            carry = __shfl(ev, LPS - 1, LPS);
            if (live && val[c]) {
                const size_t e = (size_t)s * T + (c * LPS + li);
                v.D[e] = Dn[c];
                v.C[e] = Cn[c];
                v.E[e] = ev;
                if (LINES) v.dltS[e] = (Dn[c] - Cn[c]) - (D0[c] - C0[c]);
                accQ[c] += Dn[c] - Cn[c];
                accCost += ag.mc * (Dn[c] + Cn[c]);
            }
        }
    }

    // fixed-order block reduction of the per-timestep sums over the NG groups
#pragma unroll
    for (int c = 0; c < NCH; ++c) red[(grp * NCH + c) * LPS + li] = accQ[c];
    redc[tid] = accCost;
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int t = c * LPS + li;
            if (t < T) {
                double sum = 0.0;
                for (int g2 = 0; g2 < NG; ++g2) sum += red[(g2 * NCH + c) * LPS + li];
                v.part_sinj[(size_t)blockIdx.x * T + t] = sum;
            }
        }
    }
    for (int sft = 128; sft > 0; sft >>= 1) {
        if (tid < sft) redc[tid] += redc[tid + sft];
        __syncthreads();
    }
    if (tid == 0) v.part_scost[blockIdx.x] = redc[0];
    if (fails) atomicAdd(&v.st->solver_fail, fails);
#ifdef DOPF_STATS
    if (st_scans) atomicAdd(&v.st->dbg_scans, st_scans);
    if (st_loops) atomicAdd(&v.st->dbg_wave_loops, st_loops);
    if (st_events) atomicAdd(&v.st->dbg_events, st_events);
#endif
}

bool sto_config_supported(int T, Launch *lc)
{
    // smallest lane group that covers T in <= 3 chunks; groups wider than 8 lanes are always
    // instantiated with 3 chunks (a chunk past T is masked out) to keep the kernel count small
    if (T <= 24) { lc->stoLPS = 8; lc->stoNCH = (T + 7) / 8; return true; }
    if (T <= 48) { lc->stoLPS = 16; lc->stoNCH = 3; return true; }
    if (T <= 96) { lc->stoLPS = 32; lc->stoNCH = 3; return true; }
    if (T <= 192) { lc->stoLPS = 64; lc->stoNCH = 3; return true; }
    if (T <= 256) { lc->stoLPS = 64; lc->stoNCH = 4; return true; }
    if (T <= 512) { lc->stoLPS = 64; lc->stoNCH = 8; return true; }
    return false;
}

template <int LPS, int NCH>
static void launch_sto_t(const DevView &v, hipStream_t s)
{
    if (v.L > 0) hipLaunchKernelGGL((k_sto_update<LPS, NCH, true>), dim3(v.nStoItems), dim3(256), 0, s, v);
    else hipLaunchKernelGGL((k_sto_update<LPS, NCH, false>), dim3(v.nStoItems), dim3(256), 0, s, v);
}

void launch_sto_update(const DevView &v, const Launch &lc, hipStream_t s)
{
    if (v.nStoItems == 0) return;
#define DOPF_CASE(LPS_, NCH_) if (lc.stoLPS == LPS_ && lc.stoNCH == NCH_) { launch_sto_t<LPS_, NCH_>(v, s); return; }
    DOPF_CASE(8, 1) DOPF_CASE(8, 2) DOPF_CASE(8, 3)
    DOPF_CASE(16, 3)
    DOPF_CASE(32, 3)
    DOPF_CASE(64, 3) DOPF_CASE(64, 4) DOPF_CASE(64, 8)
#undef DOPF_CASE
}

}  // namespace dopf

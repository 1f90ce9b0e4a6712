// kernels_central.hip — the central reference on the device (gfx950, fp64).
//
// Replaces src/opf_central_reference.jl:16-57 (one JuMP model of the whole multi-period DC-OPF handed to Gurobi): the same
// LP solved by a first-order primal-dual method whose every step is the kind of work the decentral kernels do — a
// sweep over the agents' time series, the nodal sums, one PTDF product — so it runs on the same data layout and reuses
// k_reduce. It is the parity target of the decentral run ("converged objective within 1e-3 of the central optimum") for
// cases no LP solver on the host can take, and an algorithm independent of the ADMM iteration.
//
//   variables   x = (P[g,t], D[s,t], C[s,t]) in their boxes           (E is eliminated: E[s,t] = sum_{tau <= t} (C - D))
//   rows        balance_t:  sum_n I[n,t] = 0                           multiplier yb[t]   (free)
//               flows:      -f_max <= (ptdf I)[l,t] <= f_max           multiplier yf[l,t] (sign = which limit binds)
//               levels:     0 <= E[s,t] <= e_max                       multiplier yE[s,t]
//   PDHG        x+ = clamp(x - tau (c + K'y)),  y+ = prox(y + sigma K(2x+ - x))   with the diagonal step sizes of Pock &
//               Chambolle (tau_j = 1 / sum_i |K_ij|, sigma_i = 1 / sum_j |K_ij|), running averages and restarts to the better
//               of (last iterate, average) whenever its normalised gap has halved (the scheme of PDLP, without its line
//               search). Prototype and convergence record: scripts/proto_pdlp.py (machine precision in 4k-12k iterations on
//               config1, config2/10 and a 118-node case).
//   outputs     objective, P, D, C, E; system price = -yb (dual(EB) of the reference), nodal price = -(yb + ptdf' yf)
//               (src/opf_central_reference.jl:66-79).
#include "dopf_internal.h"

namespace dopf {

__device__ __forceinline__ double cclamp(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// inclusive prefix sum over the 64 lanes of a wave
__device__ __forceinline__ double wave_prefix(double x, int lane)
{
    for (int d = 1; d < 64; d <<= 1) {
        const double y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    return x;
}

// pi[n,t] = yb[t] + sum_l ptdf[l,n] yf[l,t]   (the column of K' every unit at node n sees), candidate = scale * (yb, yf)
__global__ __launch_bounds__(256) void kc_price(CentralView c, const double *yb, const double *yf, double scale)
{
    extern __shared__ double d[];            // L
    const DevView &v = c.v;
    const int tid = threadIdx.x, t = blockIdx.x, N = v.N, L = v.L;
    for (int l = tid; l < L; l += 256) d[l] = scale * yf[l + (size_t)L * t];
    __syncthreads();
    const double b = scale * yb[t];
    for (int n = tid; n < N; n += 256) {
        double p = b;
        for (int l0 = 0; l0 < L; l0 += 8) {
            double h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = l0 + u < L ? v.ptdfT[n + (size_t)N * (l0 + u)] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) p += h[u] * (l0 + u < L ? d[l0 + u] : 0.0);
        }
        c.pi[n + (size_t)N * t] = p;
    }
}

// generators. ITER: one PDHG step of every (g,t) of the item; per-item sums of the extrapolated point 2x+ - x.
// !ITER: metrics of the candidate scale * X: per-item sums of x, cost, and sum min(reduced cost, 0) * pmax.
template <bool ITER>
__global__ __launch_bounds__(512) void kc_gen(CentralView c, const double *X, double scale)
{
    __shared__ double red[512];
    const DevView &v = c.v;
    const Item it = v.gen_items[blockIdx.x];
    const int T = v.T, N = v.N, TT = v.genTT, R = v.genR;
    const int tid = threadIdx.x, r = tid / TT, tt = tid - r * TT;
    const double tau = c.tauN[it.node] / c.w;
    double cost = 0.0, dpart = 0.0;
    for (int tc = 0; tc < T; tc += TT) {
        const int t = tc + tt;
        double acc = 0.0;
        if (r < R && t < T) {
            const double pi = c.pi[it.node + (size_t)N * t];
            for (int g = it.a0 + r; g < it.a1; g += R) {
                const size_t e = (size_t)g * T + t;
                const double mc = v.gen_mc[g], pm = v.gen_pmax[g];
                if (ITER) {
                    const double x0 = v.P[e];
                    const double xn = cclamp(x0 - tau * (mc + pi), 0.0, pm);
                    v.P[e] = xn;
                    c.aP[e] += xn;
                    acc += 2.0 * xn - x0;
                } else {
                    const double x = scale * X[e];
                    acc += x;
                    cost += mc * x;
                    dpart += fmin(mc + pi, 0.0) * pm;
                }
            }
        }
        red[tid] = acc;
        __syncthreads();
        if (r == 0 && t < T) {
            double sum = 0.0;
            for (int q = 0; q < R; ++q) sum += red[q * TT + tt];
            v.part_ginj[(size_t)blockIdx.x * T + t] = sum;
        }
        __syncthreads();
    }
    if (!ITER) {
        red[tid] = cost;
        __syncthreads();
        for (int sft = 256; sft > 0; sft >>= 1) { if (tid < sft) red[tid] += red[tid + sft]; __syncthreads(); }
        if (tid == 0) v.part_gcost[blockIdx.x] = red[0];
        __syncthreads();
        red[tid] = dpart;
        __syncthreads();
        for (int sft = 256; sft > 0; sft >>= 1) { if (tid < sft) red[tid] += red[tid + sft]; __syncthreads(); }
        if (tid == 0) c.m_gen[blockIdx.x] = red[0];
    }
}

// storages: one wave per storage, lane li owns the K = ceil(T/64) consecutive timesteps li*K .. li*K+K-1 (K <= 8).
// ITER: D+, C+ from the gradient c -+ (pi - sum_{tau >= t} yE), levels of the extrapolated point by a prefix sum, the level
// multipliers' proximal step, running sums. !ITER: metrics of the candidate (scale * XD, XC, XE): cost, reduced-cost term,
// worst level violation, -sum max(yE, 0) e_max; the levels themselves are written to v.E.
template <bool ITER>
__global__ __launch_bounds__(256) void kc_sto(CentralView c, const double *XD, const double *XC, const double *XE, double scale)
{
    constexpr int KMAX = 8;
    __shared__ double red[4][512];
    __shared__ double redm[4][4];
    const DevView &v = c.v;
    const Item it = v.sto_items[blockIdx.x];
    const int T = v.T, N = v.N;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = (T + 63) / 64, t0 = lane * K;
    double acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.0;
    double cost = 0.0, dpart = 0.0, viol = 0.0, yterm = 0.0;
    const double absH = c.absHn[it.node];
    for (int s = it.a0 + wv; s < it.a1; s += 4) {           // (wave-uniform trip count per wave)
        const double mc = v.sto_mc[s], pm = v.sto_pmax[s], em = v.sto_emax[s];
        double yE[KMAX], d0[KMAX], c0[KMAX];
        double ysum = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int t = t0 + k;
            const bool ok = k < K && t < T;
            const size_t e = (size_t)s * T + (ok ? t : 0);
            yE[k] = ok ? (ITER ? c.yE[e] : scale * XE[e]) : 0.0;
            d0[k] = ok ? (ITER ? v.D[e] : scale * XD[e]) : 0.0;
            c0[k] = ok ? (ITER ? v.C[e] : scale * XC[e]) : 0.0;
            ysum += yE[k];
        }
        const double incl = wave_prefix(ysum, lane);
        const double total = __shfl(incl, 63);
        double before = incl - ysum;                          // sum of yE over the steps before this lane's first
        double net = 0.0;
        double dn[KMAX], cn[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int t = t0 + k;
            dn[k] = 0.0; cn[k] = 0.0;
            if (k < K && t < T) {
                const double suf = total - before;            // sum_{tau >= t} yE
                before += yE[k];
                const double pi = c.pi[it.node + (size_t)N * t];
                const double rD = mc + pi - suf, rC = mc - pi + suf;
                if (ITER) {
                    const double tau = 1.0 / ((1.0 + absH + (double)(T - t)) * c.w);
                    dn[k] = cclamp(d0[k] - tau * rD, 0.0, pm);
                    cn[k] = cclamp(c0[k] - tau * rC, 0.0, pm);
                    const double bd = 2.0 * dn[k] - d0[k], bc = 2.0 * cn[k] - c0[k];
                    acc[k] += bd - bc;
                    net += bc - bd;
                } else {
                    acc[k] += d0[k] - c0[k];
                    net += c0[k] - d0[k];
                    cost += mc * (d0[k] + c0[k]);
                    dpart += (fmin(rD, 0.0) + fmin(rC, 0.0)) * pm;
                    yterm += fmax(yE[k], 0.0) * em;
                }
            }
        }
        const double inclE = wave_prefix(net, lane);
        double lev = inclE - net;                             // level before this lane's first step
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int t = t0 + k;
            if (k < K && t < T) {
                const size_t e = (size_t)s * T + t;
                if (ITER) {
                    lev += (2.0 * cn[k] - c0[k]) - (2.0 * dn[k] - d0[k]);
                    const double sg = c.w / (2.0 * (double)(t + 1));
                    const double z = yE[k] + sg * lev;
                    const double yn = z - sg * cclamp(z / sg, 0.0, em);
                    v.D[e] = dn[k]; v.C[e] = cn[k];
                    c.yE[e] = yn;
                    c.aD[e] += dn[k]; c.aC[e] += cn[k]; c.aE[e] += yn;
                } else {
                    lev += c0[k] - d0[k];
                    v.E[e] = lev;
                    viol = fmax(viol, fmax(-lev, lev - em));
                }
            }
        }
    }
    // per-item sums over the block's four waves (fixed order)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) red[wv][lane * KMAX + k] = acc[k];
    __syncthreads();
    for (int i = tid; i < 64 * KMAX; i += 256) {
        const int ln = i / KMAX, k = i - ln * KMAX, t = ln * K + k;
        if (k < K && t < T) v.part_sinj[(size_t)blockIdx.x * T + t] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
    if (!ITER) {
        for (int d = 32; d > 0; d >>= 1) {
            cost += __shfl_xor(cost, d); dpart += __shfl_xor(dpart, d); yterm += __shfl_xor(yterm, d);
            viol = fmax(viol, __shfl_xor(viol, d));
        }
        if (lane == 0) { redm[wv][0] = cost; redm[wv][1] = dpart; redm[wv][2] = viol; redm[wv][3] = yterm; }
        __syncthreads();
        if (tid == 0) {
            v.part_scost[blockIdx.x] = (redm[0][0] + redm[1][0]) + (redm[2][0] + redm[3][0]);
            c.m_sto[3 * blockIdx.x + 0] = (redm[0][1] + redm[1][1]) + (redm[2][1] + redm[3][1]);
            c.m_sto[3 * blockIdx.x + 1] = fmax(fmax(redm[0][2], redm[1][2]), fmax(redm[2][2], redm[3][2]));
            c.m_sto[3 * blockIdx.x + 2] = (redm[0][3] + redm[1][3]) + (redm[2][3] + redm[3][3]);
        }
    }
}

// balance and flow rows of one timestep from the nodal sums in v.cons. ITER: the multipliers' steps (balance: free;
// flows: proximal step of the interval's support function) and running sums. !ITER: worst violations and the dual
// objective's terms -yb d_tot - sum |yf| f_max of the candidate scale * (yb, yf).
template <bool ITER>
__global__ __launch_bounds__(256) void kc_dual(CentralView c, const double *yb, const double *yf, double scale)
{
    extern __shared__ double q[];            // N injections | N demands
    __shared__ double red[256];
    const DevView &v = c.v;
    const int tid = threadIdx.x, t = blockIdx.x, N = v.N, L = v.L;
    double *qd = q + N;
    double part = 0.0, dsum = 0.0;
    for (int n = tid; n < N; n += 256) {
        const double dm = v.demand[n + (size_t)N * t];
        const double x = v.cons[n + (size_t)N * t] - dm;
        q[n] = x;
        qd[n] = dm;
        if (!ITER) v.inj[n + (size_t)N * t] = x;
        part += x;
        dsum += dm;
    }
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double bal = red[0];
    __syncthreads();
    red[tid] = dsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double dtot = red[0];
    __syncthreads();
    double fviol = 0.0, fterm = 0.0;
    for (int l = tid; l < L; l += 256) {
        double f = 0.0, fd = 0.0;                // flow of the injections, flow of the demand alone
        for (int n0 = 0; n0 < N; n0 += 8) {
            double h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = n0 + u < N ? v.ptdf[l + (size_t)L * (n0 + u)] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f += h[u] * (n0 + u < N ? q[n0 + u] : 0.0);
                if (!ITER) fd += h[u] * (n0 + u < N ? qd[n0 + u] : 0.0);
            }
        }
        const size_t i = l + (size_t)L * t;
        const double F = v.fmax[l];
        if (ITER) {
            const double sg = c.w * c.sigF[l];
            const double z = c.yf[i] + sg * f;
            const double yn = z - sg * cclamp(z / sg, -F, F);
            c.yf[i] = yn;
            c.af[i] += yn;
        } else {
            v.flow[i] = f;
            fviol = fmax(fviol, fabs(f) - F);
            fterm += fabs(scale * yf[i]) * F + scale * yf[i] * fd;      // the flow rows act on injection = units - demand
        }
    }
    if (ITER) {
        if (tid == 0) {
            const double yn = c.yb[t] + c.w * c.sigB * bal;
            c.yb[t] = yn;
            c.ab[t] += yn;
        }
    } else {
        red[tid] = fterm;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
        const double ft = red[0];
        __syncthreads();
        red[tid] = fviol;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
        if (tid == 0) {
            c.m_dual[3 * t + 0] = fabs(bal);
            c.m_dual[3 * t + 1] = red[0];
            c.m_dual[3 * t + 2] = -scale * yb[t] * dtot - ft;
        }
    }
}

// dst = scale * src (restart to the running average)
__global__ __launch_bounds__(256) void kc_scale_copy(double *dst, const double *src, double scale, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = scale * src[i];
}

void central_launch_price(const CentralView &c, const double *yb, const double *yf, double scale, hipStream_t s)
{
    hipLaunchKernelGGL(kc_price, dim3(c.v.T), dim3(256), (size_t)c.v.L * sizeof(double), s, c, yb, yf, scale);
}

void central_launch_iteration(const CentralView &c, const DevView &vreduce, hipStream_t s)
{
    const DevView &v = c.v;
    central_launch_price(c, c.yb, c.yf, 1.0, s);
    if (v.nGenItems) hipLaunchKernelGGL(kc_gen<true>, dim3(v.nGenItems), dim3(512), 0, s, c, (const double *)nullptr, 1.0);
    if (v.nStoItems) hipLaunchKernelGGL(kc_sto<true>, dim3(v.nStoItems), dim3(256), 0, s, c, (const double *)nullptr, (const double *)nullptr, (const double *)nullptr, 1.0);
    launch_reduce(vreduce, s);
    hipLaunchKernelGGL(kc_dual<true>, dim3(v.T), dim3(256), 2 * (size_t)v.N * sizeof(double), s, c, (const double *)nullptr, (const double *)nullptr, 1.0);
}

void central_launch_metrics(const CentralView &c, const DevView &vreduce, const double *XP, const double *XD, const double *XC, const double *XE,
                            const double *yb, const double *yf, double scale, hipStream_t s)
{
    const DevView &v = c.v;
    central_launch_price(c, yb, yf, scale, s);
    if (v.nGenItems) hipLaunchKernelGGL(kc_gen<false>, dim3(v.nGenItems), dim3(512), 0, s, c, XP, scale);
    if (v.nStoItems) hipLaunchKernelGGL(kc_sto<false>, dim3(v.nStoItems), dim3(256), 0, s, c, XD, XC, XE, scale);
    launch_reduce(vreduce, s);
    hipLaunchKernelGGL(kc_dual<false>, dim3(v.T), dim3(256), 2 * (size_t)v.N * sizeof(double), s, c, yb, yf, scale);
}

void central_launch_scale_copy(double *dst, const double *src, double scale, size_t n, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(kc_scale_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, scale, n);
}

}  // namespace dopf

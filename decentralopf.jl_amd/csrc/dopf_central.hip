// dopf_central.hip — host side of the central reference (dopf_central_solve, include/dopf.h).
//
// Replaces src/opf_central_reference.jl:1-81: sets the LP up from the same dopf_problem the decentral path takes, runs the
// primal-dual iteration of kernels_central.hip in batches, looks at the primal / dual objectives and the worst constraint
// violation of the last iterate and of the running average between batches, restarts from the better one when its
// normalised gap has halved, stops at the requested tolerance.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "dopf_ctx.h"

using namespace dopf;

namespace {

template <class Tp>
int calloc_dev(dopf_ctx *c, Tp **out, size_t n)
{
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(Tp);
    HIPCHK(c, hipMalloc(&p, bytes));
    c->allocs.push_back(p);                 // freed by dopf_destroy
    // (the context's stream does not synchronise with the null stream: the blocking copies that fill some of these arrays
    // must not overtake the zeroing — wait for it here)
    HIPCHK(c, hipMemsetAsync(p, 0, bytes, c->main));
    HIPCHK(c, hipStreamSynchronize(c->main));
    *out = (Tp *)p;
    return DOPF_OK;
}

struct Metrics {
    double pobj, dobj, pinf, gap;
};

}  // namespace

extern "C" int dopf_central_solve(const dopf_problem *p, const dopf_params *q, double tol, int32_t max_iters,
                                  dopf_central_result *res, double *P, double *D, double *C, double *E,
                                  double *system_price, double *nodal_price, double *line_utilization,
                                  double *flow_upper_dual, double *flow_lower_dual)
{
    if (!p || !q || !res || !(tol > 0) || max_iters < 1) return fail(nullptr, DOPF_E_INVALID, "bad argument");
    memset(res, 0, sizeof *res);
    dopf_ctx *c = nullptr;
    dopf_params qq = *q;
    qq.stream = nullptr;
    qq.flags &= ~(DOPF_F_OVERLAP_AGENTS);
    int rc = dopf_create(&c, p, &qq);       // sorted agents, items, node maps, partial-sum arrays, P/D/C/E (zero)
    if (rc) return rc;
    // (callers read dopf_last_error(NULL): the temporary context's message has to outlive it)
    struct Guard { dopf_ctx *c; ~Guard() { keep_error(c); dopf_destroy(c); } } guard{c};
    DeviceGuard dev(c->device);
    const DevView &v = c->v;
    const int N = v.N, L = v.L, T = v.T, G = v.G, S = v.S;
    const size_t NT = (size_t)N * T, LT = (size_t)L * T, GT = (size_t)G * T, ST = (size_t)S * T;

    CentralView cv{};
    cv.v = v;
    cv.w = 1.0;
    cv.sigB = 1.0 / std::max(1, G + 2 * S);
    {   // diagonal step sizes (Pock & Chambolle, alpha = 1): column / row sums of |K|
        std::vector<double> absH(N, 0.0), tauN(N), sigF(L, 0.0);
        std::vector<int> gb(N + 1), sb(N + 1);
        HIPCHK(c, hipMemcpy(gb.data(), v.node_gen_beg, sizeof(int) * (N + 1), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(sb.data(), v.node_sto_beg, sizeof(int) * (N + 1), hipMemcpyDeviceToHost));
        for (int n = 0; n < N; ++n) {
            for (int l = 0; l < L; ++l) absH[n] += std::fabs(p->ptdf[l + (size_t)L * n]);
            tauN[n] = 1.0 / (1.0 + absH[n]);
        }
        for (int l = 0; l < L; ++l) {
            double rs = 0.0;
            for (int n = 0; n < N; ++n) rs += std::fabs(p->ptdf[l + (size_t)L * n]) * ((gb[n + 1] - gb[n]) + 2.0 * (sb[n + 1] - sb[n]));
            sigF[l] = 1.0 / std::max(rs, 1e-300);
        }
        double *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
        if ((rc = calloc_dev(c, &d1, N)) || (rc = calloc_dev(c, &d2, N)) || (rc = calloc_dev(c, &d3, L))) return rc;
        HIPCHK(c, hipMemcpy(d1, tauN.data(), sizeof(double) * N, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(d2, absH.data(), sizeof(double) * N, hipMemcpyHostToDevice));
        if (L) HIPCHK(c, hipMemcpy(d3, sigF.data(), sizeof(double) * L, hipMemcpyHostToDevice));
        cv.tauN = d1; cv.absHn = d2; cv.sigF = d3;
    }
    if ((rc = calloc_dev(c, &cv.yb, T)) || (rc = calloc_dev(c, &cv.yf, LT)) || (rc = calloc_dev(c, &cv.yE, ST)) ||
        (rc = calloc_dev(c, &cv.aP, GT)) || (rc = calloc_dev(c, &cv.aD, ST)) || (rc = calloc_dev(c, &cv.aC, ST)) ||
        (rc = calloc_dev(c, &cv.aE, ST)) || (rc = calloc_dev(c, &cv.ab, T)) || (rc = calloc_dev(c, &cv.af, LT)) ||
        (rc = calloc_dev(c, &cv.pi, NT)) || (rc = calloc_dev(c, &cv.m_gen, v.nGenItems)) ||
        (rc = calloc_dev(c, &cv.m_sto, 3 * (size_t)v.nStoItems)) || (rc = calloc_dev(c, &cv.m_dual, 3 * (size_t)T)))
        return rc;
    DevView vred = v;                        // k_reduce without the slack-sum part: nodal sums and the cost only
    vred.L = 0;
    vred.sliceDual = 0;
    vred.part_T = nullptr;                   // (kc_gen / kc_sto write partial rows, [row][t])
    vred.genRows = 0;                        // (kc_gen writes one partial row per item)

    std::vector<double> mg(v.nGenItems), ms(3 * (size_t)v.nStoItems), md(3 * (size_t)T);
    double dmax = 0.0;
    for (size_t i = 0; i < NT; ++i) dmax = std::max(dmax, std::fabs(p->demand[i]));
    auto metrics = [&](const double *XP, const double *XD, const double *XC, const double *XE, const double *yb, const double *yf,
                       double scale, Metrics &m) -> int {
        central_launch_metrics(cv, vred, XP, XD, XC, XE, yb, yf, scale, c->main);
        double cost = 0.0;
        HIPCHK(c, hipMemcpyAsync(&cost, v.cons + NT, sizeof(double), hipMemcpyDeviceToHost, c->main));
        if (!mg.empty()) HIPCHK(c, hipMemcpyAsync(mg.data(), cv.m_gen, mg.size() * sizeof(double), hipMemcpyDeviceToHost, c->main));
        if (!ms.empty()) HIPCHK(c, hipMemcpyAsync(ms.data(), cv.m_sto, ms.size() * sizeof(double), hipMemcpyDeviceToHost, c->main));
        HIPCHK(c, hipMemcpyAsync(md.data(), cv.m_dual, md.size() * sizeof(double), hipMemcpyDeviceToHost, c->main));
        HIPCHK(c, hipStreamSynchronize(c->main));
        double d = 0.0, inf = 0.0;
        for (double x : mg) d += x;
        for (size_t i = 0; i < ms.size(); i += 3) { d += ms[i] - ms[i + 2]; inf = std::max(inf, ms[i + 1]); }
        for (size_t t = 0; t < (size_t)T; ++t) { inf = std::max(inf, std::max(md[3 * t], md[3 * t + 1])); d += md[3 * t + 2]; }
        m.pobj = cost; m.dobj = d; m.pinf = std::max(inf, 0.0);
        m.gap = std::fabs(cost - d) / (1.0 + std::fabs(cost)) + m.pinf / (1.0 + dmax);
        return DOPF_OK;
    };

    const int batch = 200, max_avg = 4000;
    int it = 0, navg = 0;
    double last_gap = INFINITY;
    Metrics mc{}, ma{};
    bool use_avg = false, done = false;
    while (it < max_iters && !done) {
        const int nb = std::min(batch, max_iters - it);
        for (int k = 0; k < nb; ++k) central_launch_iteration(cv, vred, c->main);
        HIPCHK(c, hipGetLastError());
        it += nb; navg += nb;
        if ((rc = metrics(v.P, v.D, v.C, cv.yE, cv.yb, cv.yf, 1.0, mc))) return rc;
        if ((rc = metrics(cv.aP, cv.aD, cv.aC, cv.aE, cv.ab, cv.af, 1.0 / navg, ma))) return rc;
        use_avg = ma.gap < mc.gap;
        const Metrics &best = use_avg ? ma : mc;
        done = best.gap <= tol;
        if (done || best.gap < 0.5 * last_gap || navg >= max_avg) {
            if (use_avg) {                  // restart from the average
                const double sc = 1.0 / navg;
                central_launch_scale_copy(v.P, cv.aP, sc, GT, c->main);
                central_launch_scale_copy(v.D, cv.aD, sc, ST, c->main);
                central_launch_scale_copy(v.C, cv.aC, sc, ST, c->main);
                central_launch_scale_copy(cv.yE, cv.aE, sc, ST, c->main);
                central_launch_scale_copy(cv.yb, cv.ab, sc, T, c->main);
                central_launch_scale_copy(cv.yf, cv.af, sc, LT, c->main);
            }
            last_gap = best.gap;
            for (auto pr : {std::make_pair(cv.aP, GT), std::make_pair(cv.aD, ST), std::make_pair(cv.aC, ST), std::make_pair(cv.aE, ST),
                            std::make_pair(cv.ab, (size_t)T), std::make_pair(cv.af, LT)})
                if (pr.second) HIPCHK(c, hipMemsetAsync(pr.first, 0, pr.second * sizeof(double), c->main));
            navg = 0;
        }
    }
    // the accepted point is in v.P / v.D / v.C, (yb, yf, yE); its levels, injections and flows come from one more metrics pass
    Metrics mf{};
    if ((rc = metrics(v.P, v.D, v.C, cv.yE, cv.yb, cv.yf, 1.0, mf))) return rc;
    res->objective = mf.pobj; res->dual_objective = mf.dobj; res->primal_infeasibility = mf.pinf; res->gap = mf.gap;
    res->iterations = it; res->converged = mf.gap <= tol ? 1 : 0;
    c->level_from_primal = false;            // (the metrics pass has written the accepted point's levels into v.E)
    if ((rc = dopf_get_primal(c, P, D, C, E))) return rc;
    std::vector<double> yb(T), yf(LT);
    HIPCHK(c, hipMemcpy(yb.data(), cv.yb, sizeof(double) * T, hipMemcpyDeviceToHost));
    if (LT) HIPCHK(c, hipMemcpy(yf.data(), cv.yf, sizeof(double) * LT, hipMemcpyDeviceToHost));
    if (system_price) for (int t = 0; t < T; ++t) system_price[t] = -yb[t];                  // dual.(EB), opf_central_reference.jl:66
    if (nodal_price) {
        // the reference's formula (:71-79): lambda + sum_l (dual FlowUpper + dual FlowLower)[l,t] ptdf[l,n]. Both duals are
        // d objective / d f_max <= 0, i.e. -|yf| whichever limit binds (this is what the script prints; the marginal price
        // of an injection would carry the lower limit's dual with the opposite sign)
        for (int t = 0; t < T; ++t)
            for (int n = 0; n < N; ++n) {
                double pr = -yb[t];
                for (int l = 0; l < L; ++l) pr -= std::fabs(yf[l + (size_t)L * t]) * p->ptdf[l + (size_t)L * n];
                nodal_price[n + (size_t)N * t] = pr;
            }
    }
    if (line_utilization && LT) HIPCHK(c, hipMemcpy(line_utilization, v.flow, sizeof(double) * LT, hipMemcpyDeviceToHost));
    // dual.(FlowUpper), dual.(FlowLower) (:71): the multiplier of |flow| <= max_capacity is positive where the upper limit
    // binds and negative where the lower one does; either dual is d objective / d max_capacity <= 0
    for (size_t i = 0; i < LT; ++i) {
        if (flow_upper_dual) flow_upper_dual[i] = -std::max(yf[i], 0.0);
        if (flow_lower_dual) flow_lower_dual[i] = -std::max(-yf[i], 0.0);
    }
    return DOPF_OK;
}

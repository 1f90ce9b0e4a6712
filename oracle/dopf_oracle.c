/*
 * dopf_oracle.c — CPU oracle (plain C, fp64) for the ADMM consensus-OPF inner loop of
 * rockstaedt/DecentralOPF.jl. TEST INFRASTRUCTURE ONLY — see dopf_oracle.h.
 *
 * Parity status: PINNED against results/TNS_*.csv, results/big_gamma_*.csv and
 * results/wrong_weight_*.csv of the reference (tests/golden/, tests/test_oracle_golden.py).
 *
 * Each function names the reference file:line it restates. Paths are relative to the
 * reference checkout. Nothing here is copied from it: the reference is ~1 kLoC of Julia/JuMP
 * model-building calls; this file is the arithmetic those calls describe.
 */
#include "dopf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static double dmax(double a, double b) { return a > b ? a : b; }

/* in-place Cholesky of the lower triangle of row-major n x n a; returns 0 if positive definite */
static int chol_factor(int n, double *a)
{
    for (int j = 0; j < n; ++j) {
        double d = a[j * n + j];
        for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        a[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = a[i * n + j];
            for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
            a[i * n + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve(int n, const double *l, double *x)
{
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= l[i * n + k] * x[k];
        x[i] = s / l[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= l[k * n + i] * x[k];
        x[i] = s / l[i * n + i];
    }
}

/* LU with partial pivoting of row-major n x n a (in place); returns 0 unless singular */
static int lu_factor(int n, double *a, int *piv)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i) if (fabs(a[i * n + k]) > fabs(a[p * n + k])) p = i;
        piv[k] = p;
        if (a[p * n + k] == 0.0) return -1;
        if (p != k) for (int j = 0; j < n; ++j) { double tmp = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = tmp; }
        for (int i = k + 1; i < n; ++i) {
            double f = a[i * n + k] / a[k * n + k];
            a[i * n + k] = f;
            if (f != 0.0) for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
        }
    }
    return 0;
}

static void lu_solve(int n, const double *a, const int *piv, double *x)
{
    for (int k = 0; k < n; ++k)
        if (piv[k] != k) { double tmp = x[k]; x[k] = x[piv[k]]; x[piv[k]] = tmp; }
    for (int k = 0; k < n; ++k)
        for (int i = k + 1; i < n; ++i) x[i] -= a[i * n + k] * x[k];
    for (int i = n - 1; i >= 0; --i) {
        double v = x[i];
        for (int j = i + 1; j < n; ++j) v -= a[i * n + j] * x[j];
        x[i] = v / a[i * n + i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* generic dense QP: Mehrotra predictor-corrector primal-dual interior point                   */
/* stands in for `optimize!(sub)` = JuMP -> Gurobi barrier, subproblems.jl:86,186              */
/* ------------------------------------------------------------------------------------------ */

static int qp_solve_from(int32_t n, int32_t m, const double *Q, const double *c0, const double *A,
                         const double *b0, const double *lb, const double *ub, double *xout,
                         double *yout, int32_t *iters_out, int scaled_start)
{
    int rc = -1;
    double *c = dalloc(n), *b = dalloc(m), *u = dalloc(n);
    double *x = dalloc(n), *z = dalloc(n), *t = dalloc(n), *s = dalloc(n), *y = dalloc(m);
    double *rd = dalloc(n), *rp = dalloc(m), *M = dalloc((size_t)n * n);
    double *rhs = dalloc(n), *dx = dalloc(n), *dz = dalloc(n), *ds = dalloc(n), *dy = dalloc(m);
    double *dxa = dalloc(n), *dza = dalloc(n), *dsa = dalloc(n);
    /* with equalities the step comes from the augmented system [[M, -A'], [A, 0]] (LU, partial
     * pivoting): variables without curvature that end up strictly inside their box (the storage
     * level E) make M alone numerically singular, so no Schur complement on M. */
    const int nk = n + m;
    double *KK = dalloc(m ? (size_t)nk * nk : 1), *sol = dalloc(nk), *xbest = dalloc(n), *ybest = dalloc(m);
    int *kpiv = (int *)calloc(nk ? nk : 1, sizeof(int));
    int *bounded = (int *)calloc(n ? n : 1, sizeof(int));
    int nb = 0, it = 0;
    if (!c || !b || !u || !x || !z || !t || !s || !y || !rd || !rp || !M || !rhs || !dx || !dz ||
        !ds || !dy || !dxa || !dza || !dsa || !KK || !sol || !kpiv || !bounded || !xbest || !ybest)
        goto done;

    /* shift x = lb + xs so that xs >= 0 */
    double cnorm = 0, bnorm = 0;
    for (int i = 0; i < n; ++i) {
        double v = c0[i];
        for (int j = 0; j < n; ++j) v += Q[i * n + j] * lb[j];
        c[i] = v;
        cnorm = dmax(cnorm, fabs(v));
        u[i] = ub[i] - lb[i];
        bounded[i] = isfinite(u[i]);
        nb += bounded[i];
    }
    for (int r = 0; r < m; ++r) {
        double v = b0[r];
        for (int j = 0; j < n; ++j) v -= A[r * n + j] * lb[j];
        b[r] = v;
        bnorm = dmax(bnorm, fabs(v));
    }
    for (int i = 0; i < n; ++i) {
        if (bounded[i]) {
            if (u[i] <= 0) { u[i] = 0; }            /* degenerate box: keep a sliver interior */
            double w = u[i] > 0 ? u[i] : 1e-9;
            x[i] = 0.5 * w; t[i] = w - x[i]; s[i] = 1.0;
            u[i] = w;
        } else {
            /* second try: start an unbounded variable at the scale its own curvature suggests (the slack of a
             * line that is overloaded by thousands sits far from 10) */
            x[i] = scaled_start ? dmax(10.0, cnorm / dmax(Q[i * n + i], 1e-3)) : 10.0; t[i] = 1.0; s[i] = 0.0;
        }
        z[i] = 1.0;
    }
    if (scaled_start) {
        /* ... and the multipliers so that the dual residual Qx + c - z + s starts at (almost) zero: with z = 1 against a
         * gradient of hundreds the first step blows the complementarity products up by six orders of magnitude */
        for (int i = 0; i < n; ++i) {
            double g = c[i];
            for (int j = 0; j < n; ++j) g += Q[i * n + j] * x[j];
            if (bounded[i]) { if (g >= 0) { z[i] = g + 1.0; s[i] = 1.0; } else { z[i] = 1.0; s[i] = 1.0 - g; } }
            else z[i] = dmax(g, 1.0);
        }
    }
    const double tol_d = 1e-12 * (1.0 + cnorm), tol_p = 1e-12 * (1.0 + bnorm), tol_mu = 1e-15;
    /* "close": the merit the fp64 floor allows scales with the size of the gradient / right-hand side */
    const double close = 1e-6 * (1.0 + cnorm + bnorm);
    double best = INFINITY;
    int stall = 0;

    for (it = 0; it < 200; ++it) {
        double mu = 0, rdn = 0, rpn = 0;
        for (int i = 0; i < n; ++i) {
            double v = c[i] - z[i] + (bounded[i] ? s[i] : 0.0);
            for (int j = 0; j < n; ++j) v += Q[i * n + j] * x[j];
            for (int r = 0; r < m; ++r) v -= A[r * n + i] * y[r];
            rd[i] = v;
            rdn = dmax(rdn, fabs(v));
            mu += x[i] * z[i] + (bounded[i] ? t[i] * s[i] : 0.0);
        }
        mu /= (double)(n + nb);
        for (int r = 0; r < m; ++r) {
            double v = -b[r];
            for (int j = 0; j < n; ++j) v += A[r * n + j] * x[j];
            rp[r] = v;
            rpn = dmax(rpn, fabs(v));
        }
        double merit = rdn + rpn + mu;
        if (!(merit == merit)) break;          /* NaN: the fp64 floor was passed; keep the best */
        if (merit < best) { memcpy(xbest, x, sizeof(double) * n); memcpy(ybest, y, sizeof(double) * (m ? m : 1)); }
        if (rdn <= tol_d && rpn <= tol_p && mu <= tol_mu) { best = merit; break; }
        if (merit < best * 0.999) { best = merit; stall = 0; }
        else if (best < close && ++stall > 6) break;   /* fp64 floor reached */

        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) M[i * n + j] = Q[i * n + j];
            M[i * n + i] += z[i] / x[i] + (bounded[i] ? s[i] / t[i] : 0.0);
        }
        if (m == 0) {
            if (chol_factor(n, M) != 0) { if (best < close) break; goto done; }
        } else {
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < n; ++j) KK[i * nk + j] = M[i * n + j];
                for (int r = 0; r < m; ++r) { KK[i * nk + n + r] = -A[r * n + i]; KK[(n + r) * nk + i] = A[r * n + i]; }
            }
            for (int r = 0; r < m; ++r) for (int q = 0; q < m; ++q) KK[(n + r) * nk + n + q] = 0.0;
            if (lu_factor(nk, KK, kpiv) != 0) { if (best < close) break; goto done; }
        }

        double sigma = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < n; ++i) {
                double rc1 = sigma * mu - x[i] * z[i];
                double rc2 = bounded[i] ? sigma * mu - t[i] * s[i] : 0.0;
                if (pass == 1) {
                    rc1 -= dxa[i] * dza[i];
                    if (bounded[i]) rc2 -= (-dxa[i]) * dsa[i];
                }
                rhs[i] = -rd[i] + rc1 / x[i] - (bounded[i] ? rc2 / t[i] : 0.0);
                dz[i] = rc1; ds[i] = rc2;            /* keep rc for the back-substitution */
            }
            if (m == 0) {
                for (int i = 0; i < n; ++i) dx[i] = rhs[i];
                chol_solve(n, M, dx);
            } else {
                for (int i = 0; i < n; ++i) sol[i] = rhs[i];
                for (int r = 0; r < m; ++r) sol[n + r] = -rp[r];
                lu_solve(nk, KK, kpiv, sol);
                for (int i = 0; i < n; ++i) dx[i] = sol[i];
                for (int r = 0; r < m; ++r) dy[r] = sol[n + r];
            }
            for (int i = 0; i < n; ++i) {
                dz[i] = (dz[i] - z[i] * dx[i]) / x[i];
                ds[i] = bounded[i] ? (ds[i] + s[i] * dx[i]) / t[i] : 0.0;
            }
            double amax = 1.0;
            for (int i = 0; i < n; ++i) {
                if (dx[i] < 0) amax = fmin(amax, -x[i] / dx[i]);
                if (dz[i] < 0) amax = fmin(amax, -z[i] / dz[i]);
                if (bounded[i]) {
                    if (dx[i] > 0) amax = fmin(amax, t[i] / dx[i]);
                    if (ds[i] < 0) amax = fmin(amax, -s[i] / ds[i]);
                }
            }
            if (pass == 0) {
                double mua = 0;
                for (int i = 0; i < n; ++i) {
                    mua += (x[i] + amax * dx[i]) * (z[i] + amax * dz[i]);
                    if (bounded[i]) mua += (t[i] - amax * dx[i]) * (s[i] + amax * ds[i]);
                    dxa[i] = dx[i]; dza[i] = dz[i]; dsa[i] = ds[i];
                }
                mua /= (double)(n + nb);
                double rr = mu > 0 ? mua / mu : 0.0;
                sigma = rr * rr * rr;
            } else {
                double eta = dmax(0.995, 1.0 - mu);
                if (eta > 1.0 - 1e-9) eta = 1.0 - 1e-9;
                double al = fmin(1.0, eta * amax);
                for (int i = 0; i < n; ++i) {
                    x[i] += al * dx[i];
                    z[i] += al * dz[i];
                    /* t is stepped on its own (t = u - x would cancel once t << u) */
                    if (bounded[i]) { t[i] -= al * dx[i]; s[i] += al * ds[i]; }
                }
                for (int r = 0; r < m; ++r) y[r] += al * dy[r];
            }
        }
    }
    if (!(best < close)) goto done;         /* never got close: report failure */
    for (int i = 0; i < n; ++i) xout[i] = lb[i] + xbest[i];
    if (yout) for (int r = 0; r < m; ++r) yout[r] = ybest[r];
    rc = 0;
done:
    if (iters_out) *iters_out = it;
    free(c); free(b); free(u); free(x); free(z); free(t); free(s); free(y); free(rd); free(rp);
    free(M); free(rhs); free(dx); free(dz); free(ds); free(dy); free(dxa); free(dza); free(dsa);
    free(KK); free(sol); free(kpiv); free(bounded); free(xbest); free(ybest);
    return rc;
}

/* Mehrotra from the plain starting point; if that does not get close (seen on random network cases whose line
 * slacks sit thousands away from the start), once more from a starting point scaled to the problem. */
int oracle_qp_solve(int32_t n, int32_t m, const double *Q, const double *c0, const double *A,
                    const double *b0, const double *lb, const double *ub, double *xout,
                    double *yout, int32_t *iters_out)
{
    int rc = qp_solve_from(n, m, Q, c0, A, b0, lb, ub, xout, yout, iters_out, 0);
    if (rc != 0) rc = qp_solve_from(n, m, Q, c0, A, b0, lb, ub, xout, yout, iters_out, 1);
    return rc;
}

/* QP under construction: 1/2 x'Qx + c'x (+const), built from squared affine expressions the way
 * the reference's @expression / @objective calls compose them. */
typedef struct { int n; double *Q, *c; } qp_obj;

/* objective += coef * (sum_k alpha[k] * x[idx[k]] + beta)^2 */
static void qp_add_sq(qp_obj *o, double coef, int k, const int *idx, const double *alpha, double beta)
{
    for (int a = 0; a < k; ++a) {
        o->c[idx[a]] += 2.0 * coef * beta * alpha[a];
        for (int b = 0; b < k; ++b) o->Q[idx[a] * o->n + idx[b]] += 2.0 * coef * alpha[a] * alpha[b];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                    */
/* ------------------------------------------------------------------------------------------ */

struct oracle_ctx {
    int N, L, T, G, S, mode, nthreads;
    int bisect_only;        /* ORACLE_BISECT=1: plain bisection in the exact storage solve (cross-check of the regula falsi) */
    double *demand, *ptdf, *fmax, *gen_mc, *gen_pmax, *sto_mc, *sto_pmax, *sto_emax;
    int *gen_node, *sto_node;
    dopf_params q;
    int A_global;
    /* admm.iteration (admm.jl:29), Convergence.all */
    int iteration, converged;
    /* duals: cur = lambdas[end]; used = the entry the last solve read (lambdas[iteration]) */
    double *lam, *mu, *rho, *lam_used, *mu_used, *rho_used;
    /* last solved primal (= "previous iterate" for the next solve; zeros before the first) */
    double *P, *D, *C, *E;
    double *agentU, *agentK;               /* (G+S) x L x T slack matrices of the last solve */
    /* Result of the last iteration: per-node totals (ResultNode), averages, injection, flows */
    double *ngen, *ndis, *nchg;            /* N x T */
    double *inj, *avgU, *avgK, *flow;      /* N x T, L x T ... */
    double total_cost;
    double res[3];
    /* consensus buffer: ngen | ndis | nchg (N*T each) | sumU | sumK (L*T each) | cost */
    double *cons;
    int64_t ncons;
    /* exact mode scratch: per (n,t) breakpoint tables */
    double *tb_beta, *tb_psi, *tb_slope, *tb_psi0;
    int *tb_m;
    double *c_prev, *s_prev, *f_prev, *price;  /* shared quantities of section 9.1 */
    char err[256];
};

static char g_create_err[256];

const char *oracle_last_error(const oracle_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

void oracle_set_threads(oracle_ctx *ctx, int32_t n) { ctx->nthreads = n < 1 ? 1 : n; }

static double *dupd(const double *src, size_t n)
{
    double *d = dalloc(n);
    if (d && src && n) memcpy(d, src, n * sizeof(double));
    return d;
}

int oracle_create(oracle_ctx **out, const dopf_problem *p, const dopf_params *q, int32_t mode)
{
    if (!out || !p || !q) { snprintf(g_create_err, sizeof g_create_err, "null argument"); return DOPF_E_INVALID; }
    if (p->N < 1 || p->T < 1 || p->L < 0 || p->G < 0 || p->S < 0) {
        snprintf(g_create_err, sizeof g_create_err, "bad sizes N=%d L=%d T=%d G=%d S=%d", p->N, p->L, p->T, p->G, p->S);
        return DOPF_E_INVALID;
    }
    for (int g = 0; g < p->G; ++g)
        if (p->gen_node[g] < 0 || p->gen_node[g] >= p->N) { snprintf(g_create_err, sizeof g_create_err, "gen_node[%d] out of range", g); return DOPF_E_INVALID; }
    for (int s = 0; s < p->S; ++s)
        if (p->sto_node[s] < 0 || p->sto_node[s] >= p->N) { snprintf(g_create_err, sizeof g_create_err, "sto_node[%d] out of range", s); return DOPF_E_INVALID; }
    oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof *c);
    if (!c) return DOPF_E_NOMEM;
    c->N = p->N; c->L = p->L; c->T = p->T; c->G = p->G; c->S = p->S; c->mode = mode; c->nthreads = 1;
    { const char *e = getenv("ORACLE_BISECT"); c->bisect_only = (e && e[0] == '1') ? 1 : 0; }
    c->q = *q;
    c->A_global = q->n_agents_global > 0 ? q->n_agents_global : p->G + p->S;
    const size_t NT = (size_t)p->N * p->T, LT = (size_t)p->L * p->T;
    c->demand = dupd(p->demand, NT);
    c->ptdf = dupd(p->ptdf, (size_t)p->L * p->N);
    c->fmax = dupd(p->f_max, p->L);
    c->gen_mc = dupd(p->gen_mc, p->G); c->gen_pmax = dupd(p->gen_pmax, p->G);
    c->sto_mc = dupd(p->sto_mc, p->S); c->sto_pmax = dupd(p->sto_pmax, p->S); c->sto_emax = dupd(p->sto_emax, p->S);
    c->gen_node = (int *)calloc(p->G ? p->G : 1, sizeof(int));
    c->sto_node = (int *)calloc(p->S ? p->S : 1, sizeof(int));
    for (int g = 0; g < p->G; ++g) c->gen_node[g] = p->gen_node[g];
    for (int s = 0; s < p->S; ++s) c->sto_node[s] = p->sto_node[s];
    c->iteration = 1;                                   /* admm.jl:29 */
    c->lam = dalloc(p->T); c->mu = dalloc(LT); c->rho = dalloc(LT);   /* zeros, admm.jl:34-36 */
    c->lam_used = dalloc(p->T); c->mu_used = dalloc(LT); c->rho_used = dalloc(LT);
    c->P = dalloc((size_t)p->G * p->T);
    c->D = dalloc((size_t)p->S * p->T); c->C = dalloc((size_t)p->S * p->T); c->E = dalloc((size_t)p->S * p->T);
    c->agentU = dalloc((size_t)(p->G + p->S) * LT); c->agentK = dalloc((size_t)(p->G + p->S) * LT);
    c->ngen = dalloc(NT); c->ndis = dalloc(NT); c->nchg = dalloc(NT);
    c->inj = dalloc(NT); c->avgU = dalloc(LT); c->avgK = dalloc(LT); c->flow = dalloc(LT);
    /* injection of "no result yet" is -demand (helpers/results.jl:60-66 + subproblems.jl:49-52) */
    for (size_t i = 0; i < NT; ++i) c->inj[i] = -c->demand[i];
    c->ncons = (int64_t)(3 * NT + 2 * LT + 1);
    c->cons = dalloc((size_t)c->ncons);
    const size_t M2 = 2 * (size_t)p->L;
    c->tb_beta = dalloc(NT * (M2 ? M2 : 1)); c->tb_psi = dalloc(NT * (M2 ? M2 : 1));
    c->tb_slope = dalloc(NT * (M2 + 1)); c->tb_psi0 = dalloc(NT);
    c->tb_m = (int *)calloc(NT, sizeof(int));
    c->c_prev = dalloc(NT); c->s_prev = dalloc(p->T); c->f_prev = dalloc(LT); c->price = dalloc(NT);
    *out = c;
    return DOPF_OK;
}

void oracle_destroy(oracle_ctx *c)
{
    if (!c) return;
    free(c->demand); free(c->ptdf); free(c->fmax); free(c->gen_mc); free(c->gen_pmax);
    free(c->sto_mc); free(c->sto_pmax); free(c->sto_emax); free(c->gen_node); free(c->sto_node);
    free(c->lam); free(c->mu); free(c->rho); free(c->lam_used); free(c->mu_used); free(c->rho_used);
    free(c->P); free(c->D); free(c->C); free(c->E); free(c->agentU); free(c->agentK);
    free(c->ngen); free(c->ndis); free(c->nchg); free(c->inj); free(c->avgU); free(c->avgK);
    free(c->flow); free(c->cons); free(c->tb_beta); free(c->tb_psi); free(c->tb_slope);
    free(c->tb_psi0); free(c->tb_m); free(c->c_prev); free(c->s_prev); free(c->f_prev); free(c->price);
    free(c);
}

/* ------------------------------------------------------------------------------------------ */
/* LITERAL mode: assemble each agent's QP as the reference writes it                           */
/* ------------------------------------------------------------------------------------------ */

/* constant part of injection[n,t] seen by an agent at node `own` whose previous own net
 * contribution to generation/discharge/charge at that node is (pg, pd, pc):
 * subproblems.jl:40-54 (generator) and :130-144 (storage). */
static double inj_const(const oracle_ctx *c, int n, int t, int own, double pg, double pd, double pc)
{
    const int N = c->N;
    double g = c->ngen[n + N * t], d = c->ndis[n + N * t], ch = c->nchg[n + N * t];
    if (n == own) { g -= pg; d -= pd; ch -= pc; }
    return g + d - ch - c->demand[n + N * t];
}

/* add_penalty_terms!, penalty_terms.jl:1-53, weighted as in subproblems.jl:76-80 / :175-179.
 * own variable part of the injection at the agent's node is sum_k av[k]*x[iv[k]] (P, or D - C);
 * iU/iK are the indices of U[l,t], K[l,t] for l = 0. */
static void literal_penalties(const oracle_ctx *c, qp_obj *o, int t, int own, int nv, const int *iv,
                              const double *av, double pg, double pd, double pc, int iU, int iK)
{
    const int N = c->N, L = c->L;
    const double gamma = c->q.gamma, wf = c->q.w_flow;
    int idx[4]; double al[4];
    /* penalty_term_eb[t] = sum(injection[:, t])^2, weight gamma/2 */
    double beta = 0;
    for (int n = 0; n < N; ++n) beta += inj_const(c, n, t, own, pg, pd, pc);
    for (int k = 0; k < nv; ++k) { idx[k] = iv[k]; al[k] = av[k]; }
    qp_add_sq(o, gamma / 2, nv, idx, al, beta);
    for (int l = 0; l < L; ++l) {
        double fl = 0;
        for (int n = 0; n < N; ++n) fl += c->ptdf[l + L * n] * inj_const(c, n, t, own, pg, pd, pc);
        const double h = c->ptdf[l + L * own];
        /* (sum_n ptdf[l,n] injection[n,t] + U[l,t] - f_max[l])^2, weight 10 */
        for (int k = 0; k < nv; ++k) { idx[k] = iv[k]; al[k] = h * av[k]; }
        idx[nv] = iU + l; al[nv] = 1.0;
        qp_add_sq(o, wf, nv + 1, idx, al, fl - c->fmax[l]);
        /* (K[l,t] - sum_n ptdf[l,n] injection[n,t] - f_max[l])^2, weight 10 */
        for (int k = 0; k < nv; ++k) { idx[k] = iv[k]; al[k] = -h * av[k]; }
        idx[nv] = iK + l; al[nv] = 1.0;
        qp_add_sq(o, wf, nv + 1, idx, al, -fl - c->fmax[l]);
        /* (U - avg_U)^2 and (K - avg_K)^2, weight gamma/2 each */
        idx[0] = iU + l; al[0] = 1.0;
        qp_add_sq(o, gamma / 2, 1, idx, al, -c->avgU[l + L * t]);
        idx[0] = iK + l;
        qp_add_sq(o, gamma / 2, 1, idx, al, -c->avgK[l + L * t]);
    }
}

static double node_price(const oracle_ctx *c, const double *lam, const double *mu, const double *rho, int n, int t)
{
    /* lambdas[k][t] + sum_l ptdf[l,node] (mues[k][l,t] - rhos[k][l,t]); subproblems.jl:67-74 */
    double v = lam[t];
    for (int l = 0; l < c->L; ++l) v += c->ptdf[l + c->L * n] * (mu[l + c->L * t] - rho[l + c->L * t]);
    return v;
}

/* optimize_subproblem(generator), subproblems.jl:19-105 */
static int literal_generator(oracle_ctx *c, int g, double *Pnew, double *U, double *K)
{
    const int T = c->T, L = c->L, own = c->gen_node[g];
    const int n = T + 2 * L * T;
    qp_obj o = { n, dalloc((size_t)n * n), dalloc(n) };
    double *lb = dalloc(n), *ub = dalloc(n), *x = dalloc(n);
    const double *prevP = c->P + (size_t)T * g;
    int rc;
    for (int i = 0; i < n; ++i) { lb[i] = 0; ub[i] = INFINITY; }
    for (int t = 0; t < T; ++t) {
        const int iP = t, iU = T + L * t, iK = T + L * T + L * t;
        ub[iP] = c->gen_pmax[g];                                    /* :26 */
        o.c[iP] += c->gen_mc[g] + node_price(c, c->lam, c->mu, c->rho, own, t);     /* :66-75 */
        double one = 1.0;
        literal_penalties(c, &o, t, own, 1, &iP, &one, prevP[t], 0, 0, iU, iK);
        qp_add_sq(&o, c->q.w_prox / 2, 1, &iP, &one, -prevP[t]);    /* :81 */
    }
    rc = oracle_qp_solve(n, 0, o.Q, o.c, NULL, NULL, lb, ub, x, NULL, NULL);
    if (rc == 0) {
        for (int t = 0; t < T; ++t) {
            Pnew[t] = x[t];
            for (int l = 0; l < L; ++l) { U[l + L * t] = x[T + L * t + l]; K[l + L * t] = x[T + L * T + L * t + l]; }
        }
    }
    free(o.Q); free(o.c); free(lb); free(ub); free(x);
    return rc;
}

/* optimize_subproblem(storage), subproblems.jl:107-207 */
static int literal_storage(oracle_ctx *c, int s, double *Dn, double *Cn, double *En, double *U, double *K)
{
    const int T = c->T, L = c->L, own = c->sto_node[s];
    const int n = 3 * T + 2 * L * T;
    qp_obj o = { n, dalloc((size_t)n * n), dalloc(n) };
    double *lb = dalloc(n), *ub = dalloc(n), *x = dalloc(n), *A = dalloc((size_t)T * n), *b = dalloc(T);
    const double *pD = c->D + (size_t)T * s, *pC = c->C + (size_t)T * s;
    int rc;
    for (int i = 0; i < n; ++i) { lb[i] = 0; ub[i] = INFINITY; }
    for (int t = 0; t < T; ++t) {
        const int iD = t, iC = T + t, iE = 2 * T + t, iU = 3 * T + L * t, iK = 3 * T + L * T + L * t;
        ub[iD] = c->sto_pmax[s]; ub[iC] = c->sto_pmax[s]; ub[iE] = c->sto_emax[s];   /* :114-116 */
        const double pr = node_price(c, c->lam, c->mu, c->rho, own, t);
        o.c[iD] += c->sto_mc[s] + pr;                                /* :164-174 */
        o.c[iC] += c->sto_mc[s] - pr;
        int iv[2] = { iD, iC }; double av[2] = { 1.0, -1.0 };
        literal_penalties(c, &o, t, own, 2, iv, av, 0, pD[t], pC[t], iU, iK);
        double one = 1.0;
        qp_add_sq(&o, c->q.w_prox / 2, 1, &iD, &one, -pD[t]);       /* :180 */
        qp_add_sq(&o, c->q.w_prox / 2, 1, &iC, &one, -pC[t]);       /* :181 */
        /* StorageBalance: E[t] == (t == 1 ? 0 : E[t-1]) + C[t] - D[t]   :150-156 */
        A[t * n + iE] = 1.0; A[t * n + iC] = -1.0; A[t * n + iD] = 1.0;
        if (t > 0) A[t * n + iE - 1] = -1.0;
    }
    rc = oracle_qp_solve(n, T, o.Q, o.c, A, b, lb, ub, x, NULL, NULL);
    if (rc == 0) {
        for (int t = 0; t < T; ++t) {
            Dn[t] = x[t]; Cn[t] = x[T + t]; En[t] = x[2 * T + t];
            for (int l = 0; l < L; ++l) { U[l + L * t] = x[3 * T + L * t + l]; K[l + L * t] = x[3 * T + L * T + L * t + l]; }
        }
    }
    free(o.Q); free(o.c); free(lb); free(ub); free(x); free(A); free(b);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* EXACT mode: SURVEY.md section 9.4                                                           */
/* ------------------------------------------------------------------------------------------ */

/* A PTDF entry below this is rounding noise of the matrix inverse (exact value 0): its kinks would sit at
 * |beta| ~ 1e15 and beyond, where Psi(beta) and "beta - 1" lose all precision, while its share of Psi is below
 * 1e-9 w_flow. The exact mode's tables skip such lines (found by scripts/fuzz_parity.py: HIP and the literal mode
 * agreed, the exact mode did not); the literal mode and the flow/slack sums keep every entry. */
#define PTDF_NOISE 1e-12

/* closed-form slacks for a change dlt of the agent's net injection at node n */
static double slackU(const oracle_ctx *c, int l, int t, double h, double dlt)
{
    const double w2 = 2 * c->q.w_flow, g = c->q.gamma;
    double r = c->f_prev[l + c->L * t] + h * dlt - c->fmax[l];
    return dmax(0.0, (g * c->avgU[l + c->L * t] - w2 * r) / (w2 + g));
}
static double slackK(const oracle_ctx *c, int l, int t, double h, double dlt)
{
    const double w2 = 2 * c->q.w_flow, g = c->q.gamma;
    double qv = c->f_prev[l + c->L * t] + h * dlt + c->fmax[l];
    return dmax(0.0, (g * c->avgK[l + c->L * t] + w2 * qv) / (w2 + g));
}

/* Psi_{n,t}(dlt): derivative w.r.t. the agent's net injection of everything in the objective
 * that couples through the network (price, energy-balance penalty, both flow penalties with the
 * slacks eliminated); excludes marginal cost and the prox term. */
static double psi_eval(const oracle_ctx *c, int n, int t, double dlt)
{
    const int L = c->L;
    const double w2 = 2 * c->q.w_flow, g = c->q.gamma;
    double v = c->price[n + c->N * t] + g * (c->s_prev[t] + dlt);
    for (int l = 0; l < L; ++l) {
        const double h = c->ptdf[l + L * n];
        if (fabs(h) < PTDF_NOISE) continue;
        const double f = c->f_prev[l + L * t] + h * dlt, F = c->fmax[l];
        v += w2 * h * ((f + slackU(c, l, t, h, dlt) - F) - (slackK(c, l, t, h, dlt) - f - F));
    }
    return v;
}

static double psi_slope(const oracle_ctx *c, int n, int t, double dlt)
{
    const int L = c->L;
    const double w2 = 2 * c->q.w_flow, g = c->q.gamma, act = g / (w2 + g);
    double v = g;
    for (int l = 0; l < L; ++l) {
        const double h = c->ptdf[l + L * n];
        if (fabs(h) < PTDF_NOISE) continue;
        v += w2 * h * h * ((slackU(c, l, t, h, dlt) > 0 ? act : 1.0) + (slackK(c, l, t, h, dlt) > 0 ? act : 1.0));
    }
    return v;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* breakpoint table of Psi_{n,t}: sorted kinks beta_j, Psi(beta_j), slope on each of the m+1 pieces */
static void build_table(oracle_ctx *c, int n, int t)
{
    const int L = c->L, N = c->N;
    const size_t M2 = 2 * (size_t)L, at = (size_t)n + (size_t)N * t;
    double *beta = c->tb_beta + at * (M2 ? M2 : 1), *psi = c->tb_psi + at * (M2 ? M2 : 1);
    double *slope = c->tb_slope + at * (M2 + 1);
    const double w2 = 2 * c->q.w_flow, g = c->q.gamma;
    int m = 0;
    for (int l = 0; l < L; ++l) {
        const double h = c->ptdf[l + L * n];
        if (fabs(h) < PTDF_NOISE) continue;
        const double f = c->f_prev[l + L * t], F = c->fmax[l];
        beta[m++] = (g * c->avgU[l + L * t] / w2 - f + F) / h;     /* U switches on/off */
        beta[m++] = (-g * c->avgK[l + L * t] / w2 - f - F) / h;    /* K switches on/off */
    }
    qsort(beta, m, sizeof(double), cmp_double);
    for (int j = 0; j < m; ++j) psi[j] = psi_eval(c, n, t, beta[j]);
    if (m == 0) slope[0] = psi_slope(c, n, t, 0.0);
    else {
        slope[0] = psi_slope(c, n, t, beta[0] - 1.0);
        for (int j = 1; j < m; ++j) slope[j] = psi_slope(c, n, t, 0.5 * (beta[j - 1] + beta[j]));
        slope[m] = psi_slope(c, n, t, beta[m - 1] + 1.0);
    }
    c->tb_m[at] = m;
    c->tb_psi0[at] = psi_eval(c, n, t, 0.0);
}

/* exact generator step: P* = clamp(root of mc + Psi(P - P0) + w_prox (P - P0), 0, pmax) */
static double exact_gen_step(const oracle_ctx *c, int n, int t, double mc, double pmax, double P0)
{
    const size_t M2 = 2 * (size_t)c->L, at = (size_t)n + (size_t)c->N * t;
    const double *beta = c->tb_beta + at * (M2 ? M2 : 1), *psi = c->tb_psi + at * (M2 ? M2 : 1);
    const double *slope = c->tb_slope + at * (M2 + 1);
    const int m = c->tb_m[at];
    const double w = c->q.w_prox;
    double dlt;
    if (m == 0) dlt = -(mc + c->tb_psi0[at]) / (slope[0] + w);
    else {
        int lo = 0, hi = m;                 /* first j with psi_j + w beta_j >= -mc */
        while (lo < hi) { int mid = (lo + hi) / 2; if (psi[mid] + w * beta[mid] >= -mc) hi = mid; else lo = mid + 1; }
        const int j = lo, a = j < m ? j : m - 1;
        dlt = beta[a] - (mc + psi[a] + w * beta[a]) / (slope[j] + w);
    }
    return clampd(P0 + dlt, 0.0, pmax);
}

/* min over the box [0,pm]^2 of the strictly convex 2x2 quadratic with gradient
 *   d/dD = a D - b C - rD,  d/dC = a C - b D - rC   (a > b >= 0).
 * Exact: with C "free", D solves a 1-D convex problem (clamp of its stationary point); if the
 * implied C leaves the box, C sits at that bound (monotone contraction argument). */
static void box2(double a, double b, double rD, double rC, double pm, double *D, double *C)
{
    double Df = clampd((a * rD + b * rC) / (a * a - b * b), 0.0, pm);
    double Cf = (rC + b * Df) / a;
    if (Cf < 0.0) { *C = 0.0; *D = clampd(rD / a, 0.0, pm); }
    else if (Cf > pm) { *C = pm; *D = clampd((rD + b * pm) / a, 0.0, pm); }
    else { *C = Cf; *D = Df; }
}

typedef struct { int n, t; double mc, pm, D0, C0; } sto_step;

/* (D,C)(nu) = argmin over the box of the step-t objective minus nu * (C - D); net charge
 * x = C - D is continuous and nondecreasing in nu. */
static void sto_eval(const oracle_ctx *c, const sto_step *st, double nu, double *D, double *C)
{
    const size_t M2 = 2 * (size_t)c->L, at = (size_t)st->n + (size_t)c->N * st->t;
    const double *beta = c->tb_beta + at * (M2 ? M2 : 1), *psi = c->tb_psi + at * (M2 ? M2 : 1);
    const double *slope = c->tb_slope + at * (M2 + 1);
    const int m = c->tb_m[at];
    const double w = c->q.w_prox, q0 = st->D0 - st->C0;
    double anchor_b = 0.0, anchor_p = c->tb_psi0[at], sg = slope[0];
    if (m > 0) {
        /* r(dlt) = dlt - (D(z) - C(z) - q0), z = Psi(dlt) + nu, is increasing: first kink with r >= 0 */
        int lo = 0, hi = m;
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            double z = psi[mid] + nu;
            double Dz = clampd(st->D0 - (st->mc + z) / w, 0.0, st->pm);
            double Cz = clampd(st->C0 - (st->mc - z) / w, 0.0, st->pm);
            if (beta[mid] - (Dz - Cz - q0) >= 0.0) hi = mid; else lo = mid + 1;
        }
        const int j = lo, a = j < m ? j : m - 1;
        anchor_b = beta[a]; anchor_p = psi[a]; sg = slope[j];
    }
    /* on this piece Psi = theta + sg * q with q = D - C */
    const double theta = anchor_p - sg * (anchor_b + q0);
    box2(w + sg, sg, w * st->D0 - st->mc - theta - nu, w * st->C0 - st->mc + theta + nu, st->pm, D, C);
}

/* S_k(nu) of the forward recursion F_t = clamp(F_{t-1} + x_t(nu), 0, emax), F_0 = 0, and the
 * trajectory; returns the largest index <= k whose unclamped level leaves [0, emax] (or -1). */
static int sto_scan(const oracle_ctx *c, const sto_step *st, int k, double emax, double nu,
                    double *Dv, double *Cv, double *Fv, double *Sv, double tol)
{
    double e = 0.0;
    int last = -1;
    for (int t = 0; t <= k; ++t) {
        sto_eval(c, &st[t], nu, &Dv[t], &Cv[t]);
        double sv = e + (Cv[t] - Dv[t]);
        Sv[t] = sv;
        if (sv < -tol || sv > emax + tol) last = t;
        e = clampd(sv, 0.0, emax);
        Fv[t] = e;
    }
    return last;
}

/* exact storage step. Optimality: E_t = F_t(nu_{t+1}) and nu_t = nu_{t+1} unless the unclamped
 * level S_t(nu_{t+1}) leaves [0, emax]; then E_t sits on that bound and nu_t is the root of
 * S_t(nu) = bound. nu_{T+1} = 0 (no terminal value of stored energy). See DESIGN.md. */
static int exact_storage(const oracle_ctx *c, int s, double *Dn, double *Cn, double *En)
{
    const int T = c->T;
    const double emax = c->sto_emax[s], pm = c->sto_pmax[s];
    const double tol = 1e-11 * (1.0 + emax);
    sto_step *st = (sto_step *)calloc(T, sizeof *st);
    double *Dv = dalloc(T), *Cv = dalloc(T), *Fv = dalloc(T), *Sv = dalloc(T);
    for (int t = 0; t < T; ++t) {
        st[t].n = c->sto_node[s]; st[t].t = t; st[t].mc = c->sto_mc[s]; st[t].pm = pm;
        st[t].D0 = c->D[(size_t)T * s + t]; st[t].C0 = c->C[(size_t)T * s + t];
    }
    double nu = 0.0;
    int k = T - 1, rc = 0;
    while (k >= 0) {
        int v = sto_scan(c, st, k, emax, nu, Dv, Cv, Fv, Sv, tol);
        for (int t = v + 1; t <= k; ++t) { Dn[t] = Dv[t]; Cn[t] = Cv[t]; }
        if (v < 0) break;
        const int low = Sv[v] < 0.0;
        const double target = low ? 0.0 : emax;
        /* bracket the root of S_v(nu) = target: S_v is nondecreasing in nu */
        double a = nu, b = nu, step = 1.0;
        int guard = 0;
        if (low) { do { b = nu + step; step *= 2; sto_scan(c, st, v, emax, b, Dv, Cv, Fv, Sv, tol); } while (Sv[v] < target && ++guard < 1100); }
        else     { do { a = nu - step; step *= 2; sto_scan(c, st, v, emax, a, Dv, Cv, Fv, Sv, tol); } while (Sv[v] > target && ++guard < 1100); }
        if (guard >= 1100) { rc = -1; break; }
        /* S_v(a) <= target <= S_v(b). Bracketed regula falsi (Illinois): S_v is piecewise linear, so the secant lands on
         * the root's piece after a few steps; every third step, and whenever the secant point is not strictly inside, the
         * bracket is halved. ORACLE_BISECT=1 (environment) keeps the plain bisection of round 1 as a cross-check. */
        int found = 0;
        if (!c->bisect_only) {
            double fa, fb;
            sto_scan(c, st, v, emax, a, Dv, Cv, Fv, Sv, tol); fa = Sv[v] - target;
            sto_scan(c, st, v, emax, b, Dv, Cv, Fv, Sv, tol); fb = Sv[v] - target;
            const double rtol = 1e-12 * (1.0 + emax);      /* the tolerance of the HIP kernels' root searches */
            if (fabs(fa) <= rtol) { nu = a; found = 1; }
            else if (fabs(fb) <= rtol) { nu = b; found = 1; }
            int side = 0;
            for (int it = 0; it < 200 && !found; ++it) {
                double x = (fb != fa) ? b - fb * (b - a) / (fb - fa) : 0.5 * (a + b);
                if (it % 3 == 2 || !(x > a && x < b)) x = 0.5 * (a + b);
                if (!(x > a && x < b)) break;                       /* adjacent doubles */
                sto_scan(c, st, v, emax, x, Dv, Cv, Fv, Sv, tol);
                const double fx = Sv[v] - target;
                if (fabs(fx) <= rtol) { nu = x; found = 1; break; }
                if (fx < 0.0) { a = x; fa = fx; if (side == -1) fb *= 0.5; side = -1; }
                else          { b = x; fb = fx; if (side == 1) fa *= 0.5; side = 1; }
            }
        }
        if (!found) {
            for (int it = 0; it < 300; ++it) {
                double mid = 0.5 * (a + b);
                if (!(mid > a && mid < b)) break;
                sto_scan(c, st, v, emax, mid, Dv, Cv, Fv, Sv, tol);
                if (Sv[v] < target) a = mid; else b = mid;
            }
            nu = low ? b : a;
        }
        sto_scan(c, st, v, emax, nu, Dv, Cv, Fv, Sv, tol);
        Dn[v] = Dv[v]; Cn[v] = Cv[v];
        k = v - 1;
    }
    double e = 0.0;
    for (int t = 0; t < T; ++t) { e += Cn[t] - Dn[t]; En[t] = e; }
    free(st); free(Dv); free(Cv); free(Fv); free(Sv);
    return rc;
}

/* shared quantities of SURVEY.md section 9.1 from the previous Result */
static void derive_shared(oracle_ctx *c)
{
    const int N = c->N, L = c->L, T = c->T;
    for (int t = 0; t < T; ++t) {
        double s = 0;
        for (int n = 0; n < N; ++n) {
            double v = c->ngen[n + N * t] + c->ndis[n + N * t] - c->nchg[n + N * t] - c->demand[n + N * t];
            c->c_prev[n + N * t] = v;
            s += v;
        }
        c->s_prev[t] = s;
        for (int l = 0; l < L; ++l) {
            double f = 0;
            for (int n = 0; n < N; ++n) f += c->ptdf[l + L * n] * c->c_prev[n + N * t];
            c->f_prev[l + L * t] = f;
        }
        for (int n = 0; n < N; ++n) c->price[n + N * t] = node_price(c, c->lam, c->mu, c->rho, n, t);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* one iteration, split at the consensus sum                                                   */
/* ------------------------------------------------------------------------------------------ */

/* optimize_all_subproblems!, subproblems.jl:1-17 (Jacobi: every agent reads iteration k-1) and the
 * agent loop of Result(...), results.jl:72-106 + update(), helpers/network_elements.jl:1-14 */
int oracle_local_update(oracle_ctx *c)
{
    const int N = c->N, L = c->L, T = c->T, G = c->G, S = c->S;
    const size_t NT = (size_t)N * T, LT = (size_t)L * T;
    if (c->converged) return DOPF_OK;
    double *Pn = dalloc((size_t)G * T), *Dn = dalloc((size_t)S * T), *Cn = dalloc((size_t)S * T), *En = dalloc((size_t)S * T);
    int fail = 0;
    if (c->mode == ORACLE_MODE_EXACT) {
        derive_shared(c);
        char *need = (char *)calloc(N, 1);
        for (int g = 0; g < G; ++g) need[c->gen_node[g]] = 1;
        for (int s = 0; s < S; ++s) need[c->sto_node[s]] = 1;
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic, 4)
        for (int i = 0; i < N * T; ++i) { int n = i % N, t = i / N; if (need[n]) build_table(c, n, t); }
        free(need);
    }
#pragma omp parallel for num_threads(c->nthreads) schedule(dynamic, 16) reduction(| : fail)
    for (int a = 0; a < G + S; ++a) {
        double *U = c->agentU + (size_t)a * LT, *K = c->agentK + (size_t)a * LT;
        if (a < G) {
            const int g = a, n = c->gen_node[g];
            if (c->mode == ORACLE_MODE_LITERAL) fail |= literal_generator(c, g, Pn + (size_t)T * g, U, K) != 0;
            else for (int t = 0; t < T; ++t) {
                double P0 = c->P[(size_t)T * g + t];
                double Pv = exact_gen_step(c, n, t, c->gen_mc[g], c->gen_pmax[g], P0);
                Pn[(size_t)T * g + t] = Pv;
                for (int l = 0; l < L; ++l) {
                    const double h = c->ptdf[l + L * n];
                    U[l + L * t] = slackU(c, l, t, h, Pv - P0);
                    K[l + L * t] = slackK(c, l, t, h, Pv - P0);
                }
            }
        } else {
            const int s = a - G, n = c->sto_node[s];
            double *d = Dn + (size_t)T * s, *ch = Cn + (size_t)T * s, *e = En + (size_t)T * s;
            if (c->mode == ORACLE_MODE_LITERAL) fail |= literal_storage(c, s, d, ch, e, U, K) != 0;
            else {
                fail |= exact_storage(c, s, d, ch, e) != 0;
                for (int t = 0; t < T; ++t) {
                    double dl = (d[t] - ch[t]) - (c->D[(size_t)T * s + t] - c->C[(size_t)T * s + t]);
                    for (int l = 0; l < L; ++l) {
                        const double h = c->ptdf[l + L * n];
                        U[l + L * t] = slackU(c, l, t, h, dl);
                        K[l + L * t] = slackK(c, l, t, h, dl);
                    }
                }
            }
        }
    }
    if (fail) {
        snprintf(c->err, sizeof c->err, "sub-problem solve failed at iteration %d", c->iteration);
        free(Pn); free(Dn); free(Cn); free(En);
        return DOPF_E_SOLVER;
    }
    memcpy(c->P, Pn, sizeof(double) * (size_t)G * T);
    memcpy(c->D, Dn, sizeof(double) * (size_t)S * T);
    memcpy(c->C, Cn, sizeof(double) * (size_t)S * T);
    memcpy(c->E, En, sizeof(double) * (size_t)S * T);
    free(Pn); free(Dn); free(Cn); free(En);

    /* local sums, in agent order (the reference iterates a Dict: any fixed order is faithful) */
    double *cg = c->cons, *cd = cg + NT, *cc = cd + NT, *cU = cc + NT, *cK = cU + LT, *cost = cK + LT;
    memset(c->cons, 0, sizeof(double) * (size_t)c->ncons);
    for (int g = 0; g < G; ++g) {
        const int n = c->gen_node[g];
        double sum = 0;
        for (int t = 0; t < T; ++t) { cg[n + N * t] += c->P[(size_t)T * g + t]; sum += c->P[(size_t)T * g + t]; }
        *cost += sum * c->gen_mc[g];                                      /* results.jl:102-105 */
        for (size_t i = 0; i < LT; ++i) { cU[i] += c->agentU[(size_t)g * LT + i]; cK[i] += c->agentK[(size_t)g * LT + i]; }
    }
    for (int s = 0; s < S; ++s) {
        const int n = c->sto_node[s];
        double sum = 0;
        for (int t = 0; t < T; ++t) {
            cd[n + N * t] += c->D[(size_t)T * s + t];
            cc[n + N * t] += c->C[(size_t)T * s + t];
            sum += c->D[(size_t)T * s + t] + c->C[(size_t)T * s + t];
        }
        *cost += c->sto_mc[s] * sum;                                      /* results.jl:95-97 */
        for (size_t i = 0; i < LT; ++i) { cU[i] += c->agentU[(size_t)(G + s) * LT + i]; cK[i] += c->agentK[(size_t)(G + s) * LT + i]; }
    }
    return DOPF_OK;
}

/* rest of Result(...) results.jl:108-116, update_duals! update_duals.jl:1-39,
 * check_convergence! convergence.jl:1-31 */
int oracle_apply_consensus(oracle_ctx *c)
{
    const int N = c->N, L = c->L, T = c->T;
    const size_t NT = (size_t)N * T, LT = (size_t)L * T;
    if (c->converged) return DOPF_OK;
    const double *cg = c->cons, *cd = cg + NT, *cc = cd + NT, *cU = cc + NT, *cK = cU + LT;
    const double gamma = c->q.gamma;
    memcpy(c->ngen, cg, sizeof(double) * NT);
    memcpy(c->ndis, cd, sizeof(double) * NT);
    memcpy(c->nchg, cc, sizeof(double) * NT);
    c->total_cost = cK[LT];
    for (size_t i = 0; i < NT; ++i) c->inj[i] = -c->demand[i] + cg[i] + cd[i] - cc[i];
    for (size_t i = 0; i < LT; ++i) { c->avgU[i] = 1.0 / c->A_global * cU[i]; c->avgK[i] = 1.0 / c->A_global * cK[i]; }
    memcpy(c->lam_used, c->lam, sizeof(double) * T);
    memcpy(c->mu_used, c->mu, sizeof(double) * LT);
    memcpy(c->rho_used, c->rho, sizeof(double) * LT);
    double r_l = 0, r_m = 0, r_r = 0;
    for (int t = 0; t < T; ++t) {
        double gen = 0, dis = 0, chg = 0, dem = 0;
        for (int n = 0; n < N; ++n) { gen += cg[n + N * t]; dis += cd[n + N * t]; chg += cc[n + N * t]; dem += c->demand[n + N * t]; }
        double ln = c->lam[t] + gamma * (gen + dis - chg - dem);          /* update_duals.jl:8-13 */
        r_l = dmax(r_l, fabs(ln - c->lam[t]));
        c->lam[t] = ln;
        for (int l = 0; l < L; ++l) {
            double f = 0;
            for (int n = 0; n < N; ++n) f += c->ptdf[l + L * n] * c->inj[n + N * t];
            c->flow[l + L * t] = f;                                        /* results.jl:114 */
            const double aU = c->avgU[l + L * t], aK = c->avgK[l + L * t];
            double mn = (c->mu[l + L * t] + gamma * (f + aU - c->fmax[l])) * (aU <= c->q.mask_thr ? 1.0 : 0.0);   /* :18-25 */
            double rn = (c->rho[l + L * t] + gamma * (aK - f - c->fmax[l])) * (aK <= c->q.mask_thr ? 1.0 : 0.0);  /* :30-37 */
            r_m = dmax(r_m, fabs(mn - c->mu[l + L * t]));
            r_r = dmax(r_r, fabs(rn - c->rho[l + L * t]));
            c->mu[l + L * t] = mn;
            c->rho[l + L * t] = rn;
        }
    }
    if (c->iteration != 1) {                                               /* convergence.jl:3 */
        c->res[0] = r_l; c->res[1] = r_m; c->res[2] = r_r;
        c->converged = (r_l < c->q.eps) && (r_m < c->q.eps) && (r_r < c->q.eps);
    }
    if (!c->converged) c->iteration += 1;                                  /* convergence.jl:25-30 */
    return DOPF_OK;
}

int oracle_iterate(oracle_ctx *c, int32_t n_iters, int32_t *iters_done, int32_t *converged)
{
    int done = 0, rc = DOPF_OK;
    for (int i = 0; i < n_iters; ++i) {
        if (c->converged) break;
        if (c->q.max_iters > 0 && c->iteration > c->q.max_iters) break;
        if ((rc = oracle_local_update(c)) != DOPF_OK) break;
        if ((rc = oracle_apply_consensus(c)) != DOPF_OK) break;
        ++done;
    }
    if (iters_done) *iters_done = done;
    if (converged) *converged = c->converged;
    return rc;
}

int64_t oracle_consensus_size(const oracle_ctx *c) { return c->ncons; }
void *oracle_consensus_ptr(oracle_ctx *c) { return c->cons; }
int oracle_sync(oracle_ctx *c, int32_t *iteration, int32_t *converged)
{
    if (iteration) *iteration = c->iteration;
    if (converged) *converged = c->converged;
    return DOPF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* getters / setters                                                                           */
/* ------------------------------------------------------------------------------------------ */

static void cpy(double *dst, const double *src, size_t n) { if (dst && n) memcpy(dst, src, n * sizeof(double)); }

int oracle_get_duals(oracle_ctx *c, double *lambda, double *mu, double *rho)
{
    cpy(lambda, c->lam, c->T); cpy(mu, c->mu, (size_t)c->L * c->T); cpy(rho, c->rho, (size_t)c->L * c->T);
    return DOPF_OK;
}
int oracle_get_duals_used(oracle_ctx *c, double *lambda, double *mu, double *rho)
{
    cpy(lambda, c->lam_used, c->T); cpy(mu, c->mu_used, (size_t)c->L * c->T); cpy(rho, c->rho_used, (size_t)c->L * c->T);
    return DOPF_OK;
}
int oracle_get_primal(oracle_ctx *c, double *P, double *D, double *C, double *E)
{
    cpy(P, c->P, (size_t)c->G * c->T); cpy(D, c->D, (size_t)c->S * c->T);
    cpy(C, c->C, (size_t)c->S * c->T); cpy(E, c->E, (size_t)c->S * c->T);
    return DOPF_OK;
}
int oracle_get_consensus(oracle_ctx *c, double *injection, double *avg_U, double *avg_K, double *line_util, double *total_cost)
{
    cpy(injection, c->inj, (size_t)c->N * c->T); cpy(avg_U, c->avgU, (size_t)c->L * c->T);
    cpy(avg_K, c->avgK, (size_t)c->L * c->T); cpy(line_util, c->flow, (size_t)c->L * c->T);
    if (total_cost) *total_cost = c->total_cost;
    return DOPF_OK;
}
int oracle_get_residuals(oracle_ctx *c, double *a, double *b, double *r, int32_t *iteration)
{
    if (a) *a = c->res[0];
    if (b) *b = c->res[1];
    if (r) *r = c->res[2];
    if (iteration) *iteration = c->iteration;
    return DOPF_OK;
}
/* get_nodal_price, helpers/network_elements.jl:16-25 */
int oracle_get_nodal_price(oracle_ctx *c, int32_t which, double *out)
{
    const double *lam = which ? c->lam : c->lam_used, *mu = which ? c->mu : c->mu_used, *rho = which ? c->rho : c->rho_used;
    for (int t = 0; t < c->T; ++t)
        for (int n = 0; n < c->N; ++n) {
            double v = lam[t];
            for (int l = 0; l < c->L; ++l) v += (mu[l + c->L * t] + rho[l + c->L * t]) * c->ptdf[l + c->L * n];
            out[n + c->N * t] = v;
        }
    return DOPF_OK;
}
int oracle_get_agent_slacks(oracle_ctx *c, int32_t agent, double *U, double *K)
{
    if (agent < 0 || agent >= c->G + c->S) return DOPF_E_INVALID;
    const size_t LT = (size_t)c->L * c->T;
    cpy(U, c->agentU + (size_t)agent * LT, LT); cpy(K, c->agentK + (size_t)agent * LT, LT);
    return DOPF_OK;
}

int oracle_set_state(oracle_ctx *c, const double *P, const double *D, const double *C,
                     const double *avg_U, const double *avg_K, const double *lambda,
                     const double *mu, const double *rho, int32_t iteration)
{
    const int N = c->N, T = c->T;
    const size_t NT = (size_t)N * T, LT = (size_t)c->L * T;
    if (iteration < 1) return DOPF_E_INVALID;
    if (P) memcpy(c->P, P, sizeof(double) * (size_t)c->G * T);
    if (D) memcpy(c->D, D, sizeof(double) * (size_t)c->S * T);
    if (C) memcpy(c->C, C, sizeof(double) * (size_t)c->S * T);
    if (avg_U) memcpy(c->avgU, avg_U, sizeof(double) * LT);
    if (avg_K) memcpy(c->avgK, avg_K, sizeof(double) * LT);
    if (lambda) memcpy(c->lam, lambda, sizeof(double) * T);
    if (mu) memcpy(c->mu, mu, sizeof(double) * LT);
    if (rho) memcpy(c->rho, rho, sizeof(double) * LT);
    /* re-derive the per-node totals and E from the primal state (single-shard view) */
    memset(c->ngen, 0, sizeof(double) * NT); memset(c->ndis, 0, sizeof(double) * NT); memset(c->nchg, 0, sizeof(double) * NT);
    for (int g = 0; g < c->G; ++g) for (int t = 0; t < T; ++t) c->ngen[c->gen_node[g] + N * t] += c->P[(size_t)T * g + t];
    for (int s = 0; s < c->S; ++s) {
        double e = 0;
        for (int t = 0; t < T; ++t) {
            c->ndis[c->sto_node[s] + N * t] += c->D[(size_t)T * s + t];
            c->nchg[c->sto_node[s] + N * t] += c->C[(size_t)T * s + t];
            e += c->C[(size_t)T * s + t] - c->D[(size_t)T * s + t];
            c->E[(size_t)T * s + t] = e;
        }
    }
    for (size_t i = 0; i < NT; ++i) c->inj[i] = -c->demand[i] + c->ngen[i] + c->ndis[i] - c->nchg[i];
    c->iteration = iteration;
    c->converged = 0;
    return DOPF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* calculate_ptdf, helpers/ptdf.jl:1-41                                                        */
/* ------------------------------------------------------------------------------------------ */

int oracle_calculate_ptdf(int32_t N, int32_t L, const int32_t *from, const int32_t *to,
                          const double *sus, int32_t slack, double *out)
{
    if (N < 1 || L < 0 || slack < 0 || slack >= N) return DOPF_E_INVALID;
    const int R = N - 1;
    double *inc = dalloc((size_t)L * N), *Bn = dalloc((size_t)N * N), *aug = dalloc((size_t)R * 2 * R + 1), *Binv = dalloc((size_t)N * N);
    for (int l = 0; l < L; ++l) { inc[l * N + from[l]] = 1.0; inc[l * N + to[l]] = -1.0; }   /* ptdf.jl:17-21 */
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double v = 0;
            for (int l = 0; l < L; ++l) v += inc[l * N + i] * sus[l] * inc[l * N + j];       /* Bn = A' B A */
            Bn[i * N + j] = v;
        }
    /* invert Bn without the slack row/column (Gauss-Jordan, partial pivoting) */
    int *map = (int *)calloc(N, sizeof(int));
    for (int i = 0, k = 0; i < N; ++i) if (i != slack) map[k++] = i;
    for (int i = 0; i < R; ++i) {
        for (int j = 0; j < R; ++j) aug[i * 2 * R + j] = Bn[map[i] * N + map[j]];
        for (int j = 0; j < R; ++j) aug[i * 2 * R + R + j] = (i == j);
    }
    int rc = DOPF_OK;
    for (int col = 0; col < R && rc == DOPF_OK; ++col) {
        int piv = col;
        for (int r = col + 1; r < R; ++r) if (fabs(aug[r * 2 * R + col]) > fabs(aug[piv * 2 * R + col])) piv = r;
        if (fabs(aug[piv * 2 * R + col]) < 1e-300) { rc = DOPF_E_INVALID; break; }
        if (piv != col) for (int j = 0; j < 2 * R; ++j) { double tmp = aug[col * 2 * R + j]; aug[col * 2 * R + j] = aug[piv * 2 * R + j]; aug[piv * 2 * R + j] = tmp; }
        double d = aug[col * 2 * R + col];
        for (int j = 0; j < 2 * R; ++j) aug[col * 2 * R + j] /= d;
        for (int r = 0; r < R; ++r) if (r != col) {
            double f = aug[r * 2 * R + col];
            if (f != 0.0) for (int j = 0; j < 2 * R; ++j) aug[r * 2 * R + j] -= f * aug[col * 2 * R + j];
        }
    }
    if (rc == DOPF_OK) {
        for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) Binv[map[i] * N + map[j]] = aug[i * 2 * R + R + j];
        for (int l = 0; l < L; ++l)
            for (int n = 0; n < N; ++n) {
                double v = 0;
                for (int k = 0; k < N; ++k) v += sus[l] * inc[l * N + k] * Binv[k * N + n];  /* PTDF = Bl * B_inv */
                out[l + L * n] = v;
            }
    }
    free(inc); free(Bn); free(aug); free(Binv); free(map);
    return rc;
}

/*
 * dopf_oracle.h — CPU oracle for the ADMM consensus-OPF inner loop.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (decentralopf.jl_amd/, libdopf_hip.so)
 * may include, link or call this. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED. Both modes reproduce the reference's shipped trajectories
 * results/TNS_*.csv, results/big_gamma_*.csv, results/wrong_weight_*.csv (see
 * tests/golden/ and tests/test_oracle_golden.py) and the thesis tables quoted in BASELINE.md.
 *
 * Two modes, same driver around them:
 *   ORACLE_MODE_LITERAL (0): every agent's QP is assembled term by term exactly as
 *       src/optimization/subproblems.jl:19-207 and src/optimization/penalty_terms.jl:1-53 write
 *       it (variables P|U|K resp. D|C|E|U|K, the squared penalty expressions, the storage
 *       balance equality) and handed to a generic dense primal-dual interior-point QP solver —
 *       the stand-in for JuMP -> Gurobi barrier (third-party, un-pinned, not in /root/reference;
 *       SURVEY.md section 8c). Every sub-QP is strictly convex in its decision variables, so
 *       the minimiser is unique and any exact QP method is a valid oracle.
 *   ORACLE_MODE_EXACT   (1): the slack-eliminated exact solve of SURVEY.md section 9.4
 *       (piecewise-linear monotone root per generator-timestep; price-threshold recursion over
 *       the state of charge per storage). Used where the literal mode is too slow, and as the
 *       timed CPU baseline. It is pinned by the golden files AND by the literal mode.
 *
 * The API mirrors include/dopf.h one to one (prefix oracle_ instead of dopf_), host memory only.
 */
#ifndef DOPF_ORACLE_H
#define DOPF_ORACLE_H

#include "../include/dopf.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MODE_LITERAL 0
#define ORACLE_MODE_EXACT   1

typedef struct oracle_ctx oracle_ctx;

int  oracle_create(oracle_ctx **out, const dopf_problem *p, const dopf_params *q, int32_t mode);
void oracle_destroy(oracle_ctx *ctx);
const char *oracle_last_error(const oracle_ctx *ctx);
void oracle_set_threads(oracle_ctx *ctx, int32_t n_threads); /* OpenMP over agents; 1 = serial */

int oracle_iterate(oracle_ctx *ctx, int32_t n_iters, int32_t *iters_done, int32_t *converged);
int oracle_local_update(oracle_ctx *ctx);
int oracle_apply_consensus(oracle_ctx *ctx);
int64_t oracle_consensus_size(const oracle_ctx *ctx);
void *oracle_consensus_ptr(oracle_ctx *ctx);   /* host pointer */
int oracle_sync(oracle_ctx *ctx, int32_t *iteration, int32_t *converged);

int oracle_get_duals(oracle_ctx *ctx, double *lambda, double *mu, double *rho);
int oracle_get_duals_used(oracle_ctx *ctx, double *lambda, double *mu, double *rho);
int oracle_get_primal(oracle_ctx *ctx, double *P, double *D, double *C, double *E);
int oracle_get_consensus(oracle_ctx *ctx, double *injection, double *avg_U, double *avg_K,
                         double *line_util, double *total_cost);
int oracle_get_residuals(oracle_ctx *ctx, double *lam_res, double *mu_res, double *rho_res,
                         int32_t *iteration);
int oracle_get_nodal_price(oracle_ctx *ctx, int32_t which, double *out);
int oracle_set_state(oracle_ctx *ctx, const double *P, const double *D, const double *C,
                     const double *avg_U, const double *avg_K,
                     const double *lambda, const double *mu, const double *rho, int32_t iteration);
/* Per-agent slack matrices of the last solve (ResultGenerator.U/K, ResultStorage.U/K,
 * src/structures/results.jl:1-17); agent index: generators 0..G-1 then storages G..G+S-1. */
int oracle_get_agent_slacks(oracle_ctx *ctx, int32_t agent, double *U /*L*T*/, double *K /*L*T*/);

/* calculate_ptdf, src/helpers/ptdf.jl:1-41. from/to are 0-based node ids; out is L x N col-major. */
int oracle_calculate_ptdf(int32_t N, int32_t L, const int32_t *from, const int32_t *to,
                          const double *susceptance, int32_t slack, double *out);

/* The generic QP solver of the literal mode, exported so tests can pin it on hand-made QPs:
 *   min 1/2 x'Qx + c'x  s.t.  A x = b,  lb <= x <= ub  (ub may be +inf; lb finite).
 * Q is n x n row-major symmetric PSD, A is m x n row-major. Returns 0 on success. */
int oracle_qp_solve(int32_t n, int32_t m, const double *Q, const double *c, const double *A,
                    const double *b, const double *lb, const double *ub, double *x,
                    double *y /*m, may be NULL*/, int32_t *iters /*may be NULL*/);

#ifdef __cplusplus
}
#endif
#endif

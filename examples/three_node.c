/* The reference's shipped case (src/cases/three_node.jl: 3 nodes, 3 lines, 4 generators, 1 storage, 2 timesteps)
 * driven through the C ABI of include/dopf.h exactly as src/opf_admm_decentral.jl drives the Julia code:
 *   admm = ADMM(0.3, nodes, generators, storages, lines); run!(admm); get_nodal_price(admm.iteration)
 * Build:  gcc -O2 -Iinclude examples/three_node.c -o three_node -Ldecentralopf.jl_amd/csrc -ldopf_hip -Wl,-rpath,$PWD/decentralopf.jl_amd/csrc
 * Expected on an MI355X: "converged after 476 iterations, total cost 14034.51" (thesis Table 16; central LP: 14035). */
#include <stdio.h>
#include <stdlib.h>
#include "dopf.h"

int main(void)
{
    enum { N = 3, L = 3, T = 2, G = 4, S = 1 };
    /* src/cases/three_node.jl: node demands [n + N*t] (column-major like a Julia Matrix); N3 is the slack node */
    const double demand[N * T] = {10, 50, 120, 250, 70, 200};
    /* calculate_ptdf(nodes, lines) (helpers/ptdf.jl) for L1 = N2-N1 (b=1), L2 = N3-N1 (b=1), L3 = N2-N3 (b=2): [l + L*n] */
    const double ptdf[L * N] = {-0.4, -0.6, 0.4, 0.2, -0.2, 0.8, 0.0, 0.0, 0.0};
    const double f_max[L] = {20, 45, 70};
    const double gen_mc[G] = {3, 4, 30, 50}, gen_pmax[G] = {80, 120, 300, 120};     /* pv, wind, coal, gas */
    const int32_t gen_node[G] = {0, 1, 2, 0};
    const double sto_mc[S] = {1}, sto_pmax[S] = {10}, sto_emax[S] = {20};            /* battery at N1 */
    const int32_t sto_node[S] = {0};

    dopf_problem p = {N, L, T, G, S, demand, ptdf, f_max, gen_mc, gen_pmax, gen_node, sto_mc, sto_pmax, sto_emax, sto_node};
    dopf_params q;
    dopf_default_params(&q);            /* gamma 0.3, flow weight 10, prox weight 1, eps 1e-3, mask 1e-2 */
    q.max_iters = 10000;
    dopf_ctx *ctx = NULL;
    if (dopf_create(&ctx, &p, &q) != DOPF_OK) { fprintf(stderr, "dopf_create: %s\n", dopf_last_error(NULL)); return 1; }
    int32_t done = 0, conv = 0;
    if (dopf_iterate(ctx, 10000, &done, &conv) != DOPF_OK) { fprintf(stderr, "dopf_iterate: %s\n", dopf_last_error(ctx)); return 1; }
    double inj[N * T], aU[L * T], aK[L * T], flow[L * T], cost = 0, price[N * T];
    dopf_get_consensus(ctx, inj, aU, aK, flow, &cost);
    dopf_get_nodal_price(ctx, 0, price);
    double r[3]; int32_t iteration = 0;
    dopf_get_residuals(ctx, &r[0], &r[1], &r[2], &iteration);
    printf("%s after %d iterations, total cost %.2f\n", conv ? "converged" : "not converged", iteration, cost);
    for (int t = 0; t < T; ++t) printf("t=%d nodal prices %.3f %.3f %.3f\n", t + 1, price[0 + N * t], price[1 + N * t], price[2 + N * t]);
    dopf_destroy(ctx);
    return conv ? 0 : 2;
}

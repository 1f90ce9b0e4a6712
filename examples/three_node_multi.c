/* The same run as examples/three_node.c, sharded: dopf_multi_* cuts the five units into n shards, one context per shard,
 * and the library sums the consensus vector over the shards every iteration — what a Julia `ccall` host gets with
 * ADMM(...; n_gpus = n) in decentralopf.jl_amd/julia/DecentralOPFHip.jl. No launcher, no collective library:
 * DOPF_F_COMM_P2P is the peer exchange (direct stores into the other shards' memory + sequence flags, include/dopf.h).
 *   usage: three_node_multi [n_shards = 2]
 * With fewer GPUs than shards the shards share device 0 (the exchange does not care where a peer's memory lives), so this
 * runs on a one-GPU box too; with n GPUs pass devices = NULL to dopf_multi_create and every shard gets its own.
 * Build:  gcc -O2 -Iinclude examples/three_node_multi.c -o three_node_multi -Ldecentralopf.jl_amd/csrc -ldopf_hip -Wl,-rpath,$PWD/decentralopf.jl_amd/csrc
 * Expected: "converged after 476 iterations, total cost 14034.51 (2 shards)" — the single-context run, shard count aside. */
#include <stdio.h>
#include <stdlib.h>
#include "dopf.h"

int main(int argc, char **argv)
{
    enum { N = 3, L = 3, T = 2, G = 4, S = 1, MAXSH = 4 };
    const int n_shards = argc > 1 ? atoi(argv[1]) : 2;
    if (n_shards < 1 || n_shards > MAXSH) { fprintf(stderr, "1..%d shards\n", MAXSH); return 1; }
    const double demand[N * T] = {10, 50, 120, 250, 70, 200};                       /* src/cases/three_node.jl, [n + N*t] */
    const double ptdf[L * N] = {-0.4, -0.6, 0.4, 0.2, -0.2, 0.8, 0.0, 0.0, 0.0};    /* calculate_ptdf: [l + L*n] */
    const double f_max[L] = {20, 45, 70};
    const double gen_mc[G] = {3, 4, 30, 50}, gen_pmax[G] = {80, 120, 300, 120};
    const int32_t gen_node[G] = {0, 1, 2, 0};
    const double sto_mc[S] = {1}, sto_pmax[S] = {10}, sto_emax[S] = {20};
    const int32_t sto_node[S] = {0};
    dopf_problem p = {N, L, T, G, S, demand, ptdf, f_max, gen_mc, gen_pmax, gen_node, sto_mc, sto_pmax, sto_emax, sto_node};
    dopf_params q;
    dopf_default_params(&q);
    q.max_iters = 10000;
    q.flags |= DOPF_F_COMM_P2P;
    const int32_t devices[MAXSH] = {0, 0, 0, 0};      /* all shards on device 0; NULL = shard i on device i */
    dopf_multi *m = NULL;
    if (dopf_multi_create(&m, &p, &q, n_shards, devices) != DOPF_OK) { fprintf(stderr, "dopf_multi_create: %s\n", dopf_multi_last_error(NULL)); return 1; }
    int32_t done = 0, conv = 0;
    if (dopf_multi_iterate(m, 10000, &done, &conv) != DOPF_OK) { fprintf(stderr, "dopf_multi_iterate: %s\n", dopf_multi_last_error(m)); return 1; }
    /* replicated state (duals, consensus, prices) from shard 0; primal rows of all shards in the caller's order */
    dopf_ctx *c0 = dopf_multi_ctx(m, 0);
    double inj[N * T], aU[L * T], aK[L * T], flow[L * T], cost = 0, price[N * T], P[T * G], D[T * S], C[T * S], E[T * S];
    dopf_get_consensus(c0, inj, aU, aK, flow, &cost);
    dopf_get_nodal_price(c0, 0, price);
    dopf_multi_get_primal(m, P, D, C, E);
    double r[3]; int32_t iteration = 0;
    dopf_get_residuals(c0, &r[0], &r[1], &r[2], &iteration);
    printf("%s after %d iterations, total cost %.2f (%d shards)\n", conv ? "converged" : "not converged", iteration, cost, (int)dopf_multi_size(m));
    for (int t = 0; t < T; ++t) printf("t=%d nodal prices %.3f %.3f %.3f | coal %.2f battery level %.2f\n", t + 1, price[0 + N * t], price[1 + N * t],
                                       price[2 + N * t], P[t + T * 2], E[t]);
    dopf_multi_destroy(m);
    return conv ? 0 : 2;
}

#!/usr/bin/env python3
"""bench.py — ADMM consensus-OPF iterations on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2] [--gamma G]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ADMM iteration (all agent x-updates + consensus + dual update + stop test) over one
synthetic N-agent x T-timestep grid that is resident in HBM before the timed region starts.
Prints ONE JSON line on rank 0:
  metric/value   agent-subproblem-updates per second, whole job (= agents x iterations / s)
  roofline       generator x-update kernel: algorithmic bytes per launch / its HIP-event duration
  cpu_baseline   the oracle's exact mode (a "port", oracle/dopf_oracle.c) on the host cores, bounded sample
Workloads (BASELINE.json `configs`): config1 = 1k gens + 100 storages x 24; config2 = 50k agents x 96
(default: the largest single-GPU configuration); config4 = 1M agents x 24; config3 = 118-node/186-line
synthetic network, 100k agents x 168. With --gpus N every rank holds one such grid (weak scaling) and
the per-iteration consensus sum is one RCCL all-reduce.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "config1": (1, "synthetic 1k generators + 100 storages, 24 timesteps (BASELINE configs[1])"),
    "config2": (2, "synthetic 50k agents (45455 gen + 4545 storage), 96 timesteps, copper plate (BASELINE configs[2])"),
    "config3": (3, "synthetic 118-node/186-line network, 100k agents, 168 timesteps (BASELINE configs[3], graph is synthetic)"),
    "config4": (4, "synthetic 1M agents (909091 gen + 90909 storage), 24 timesteps, copper plate (BASELINE configs[4])"),
    # BASELINE configs[3] is an 8-GPU configuration: this is one GPU's eighth of its agents on the full network
    # (with --gpus 8 every rank owns such a share: the configuration itself)
    "config3-share": (3, "one GPU's share (12.5k of 100k agents) of the synthetic 118-node/186-line network, 168 timesteps "
                         "(BASELINE configs[3] is an 8-GPU run; graph is synthetic)"),
}
SHARE = {"config3-share": 0.125}


def algorithmic_bytes(G, S, T, N, L):
    """per generator update 16T+20 B, per storage update 40T+28 B, shared
    consensus data (N+5L+2)*8T B once per GPU per iteration."""
    gen = G * (16 * T + 20)
    sto = S * (40 * T + 28)
    shared = (N + 5 * L + 2) * 8 * T
    return gen, sto, shared


def cpu_baseline(pp, gamma, w_flow=10.0, budget_s=20.0):
    """Oracle (exact mode) on the host cores: a bounded number of iterations of the SAME problem."""
    import numpy as np  # noqa: F401
    from decentralopf_jl_amd import _capi
    import __graft_entry__ as ge
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build()
    from oracle import binding as ob
    api = ob.OracleApi(ge.ORACLE_LIB)
    cores = os.cpu_count() or 1
    e = _capi.Engine(api, params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0), mode=1, **pp.engine_kwargs())
    ob.set_threads(e, cores)
    t0 = time.perf_counter()
    e.iterate(1)
    t1 = time.perf_counter() - t0
    n = 1
    extra = int(max(0, min(50, (budget_s - t1) // max(t1, 1e-6))))
    if extra > 0:
        t0 = time.perf_counter()
        e.iterate(extra)
        t1 += time.perf_counter() - t0
        n += extra
    A = pp.G + pp.S
    return {"value": A * n / t1, "unit": "agent-updates/s", "cores": cores, "kind": "port",
            "sample": f"{n} ADMM iteration(s) of the same workload from the zero state, oracle exact mode, "
                      f"OpenMP over agents ({t1:.1f} s)",
            "iters_per_sec": n / t1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; marks the line invalid)")
    ap.add_argument("--gamma", type=float, default=None, help="ADMM penalty (BASELINE's rho); default 1/A (convergent)")
    ap.add_argument("--w-flow", type=float, default=None, help="weight of the flow-consensus terms (reference: 10); default 10 without lines, 0.3/A with")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-iters", type=int, default=32)
    ap.add_argument("--overlap", action="store_true", help="fork the storage kernel onto a side stream")
    ap.add_argument("--flags", type=int, default=0, help="DOPF_F_* bits (include/dopf.h), e.g. 16 = separate generator/storage launches")
    ap.add_argument("--no-also", action="store_true", help="skip the short side runs of the other single-GPU workloads")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; nccl (= RCCL) is the product path, gloo only to "
                                                      "rehearse the multi-rank logic on a box with fewer GPUs than ranks")
    ap.add_argument("--force-sharded", action="store_true", help="debug: drive the sharded (all-reduce) path even on one rank")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON record): libraries that chat on stdout (RCCL prints a
    # version banner at communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import dopf_pkg
    pkg = dopf_pkg.load()
    from decentralopf_jl_amd import _capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank %= max(1, torch.cuda.device_count())       # (ranks > GPUs only happens in a rehearsal on a small box)
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    idx, desc = WORKLOADS[args.workload]
    # weak scaling: every rank owns one full grid of the workload (own seed), demand adds up
    base = synth.baseline_config(idx, scale=args.scale * SHARE.get(args.workload, 1.0))
    if world > 1:
        cfg = dict(base.meta)
        pp = synth.synthetic_case(cfg["n_gen"], cfg["n_sto"], cfg["T"], N=cfg["N"], L=cfg["L"], seed=synth.SEED + rank)
        if cfg["L"] > 0:          # one network for everybody: rank 0's
            pp.ptdf, pp.f_max = base.ptdf, base.f_max * world
        own_demand = pp.demand.copy()
        dem = torch.tensor(pp.demand, dtype=torch.float64, device="cuda")
        dist.all_reduce(dem)
        pp.demand = dem.cpu().numpy()
    else:
        pp = base
    A_local = pp.G + pp.S
    A_global = A_local * world
    gamma = args.gamma if args.gamma is not None else 1.0 / A_global
    # flow-consensus weight: the reference's literal 10 with lines makes every agent undo the whole line violation on
    # its own, an all-on/all-off 2-cycle for more than a few dozen agents whatever gamma is; it has to shrink with the
    # number of agents like gamma does (0.3/A converges on the synthetic networks tried). Irrelevant on a copper plate.
    w_flow = args.w_flow if args.w_flow is not None else (10.0 if pp.L == 0 else 0.3 / A_global)

    if not sharded:
        eng = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank,
                                                                        flags=args.flags | (_capi.F_OVERLAP_AGENTS if args.overlap else 0)),
                           **pp.engine_kwargs())
        step = lambda n: eng.iterate(n)
        sync = lambda: eng.sync()
    else:
        from decentralopf_jl_amd.sharded import ShardedADMM
        # every rank already holds its own grid: a 1-way "shard" of its local problem, global agent count
        # passed explicitly; the all-reduce runs over all ranks on the engine's own stream
        sh = ShardedADMM(pp, 0, 1, gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank, n_agents_global_override=A_global)
        st, tens = sh.stream, sh._tensor

        def _all_reduce():                # (ShardedADMM.step makes the engine's stream current around its loop)
            dist.all_reduce(tens, op=dist.ReduceOp.SUM)
        sh._all_reduce = _all_reduce
        eng = sh.engine
        step = lambda n: sh.step(n)
        sync = lambda: sh.sync()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step(args.warmup)
    sync()
    barrier()
    t0 = time.perf_counter()
    step(args.steps)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    it_after, _ = sync()
    assert it_after == 1 + args.warmup + args.steps, (it_after, args.warmup, args.steps)
    fails = eng.solver_failures()

    # Per-kernel durations, live, HIP events on the stream the kernels run on. The timed region above replays
    # hipGraphs (no place for events), so the SAME iterations — W warm-up, then K — are run again on a fresh engine,
    # the K launched kernel by kernel with an event pair around each: `timing` = mean over iterations W..W+K-1,
    # `steady` = the last `timed_iters` of them (no cold-start or structure-change iterations left in there).
    # With N > 1 ranks the replay is rank 0's own grid as a single-GPU problem (its own demand, gamma = 1/A_local): the
    # kernels one GPU runs per iteration, without the collective.
    timing = steady = None
    if rank == 0:
        if world > 1:
            import copy
            ppr = copy.copy(pp)
            ppr.demand = own_demand
            g_r, wf_r = 1.0 / A_local, (w_flow if pp.L == 0 else 0.3 / A_local)
            if pp.L > 0:
                ppr.f_max = pp.f_max / world
        else:
            ppr, g_r, wf_r = pp, gamma, w_flow
        er = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=g_r, w_flow=wf_r, eps=0.0, device=local_rank, flags=args.flags),
                          **ppr.engine_kwargs())
        er.iterate(args.warmup)
        tail = max(1, min(args.timed_iters, args.steps))
        parts, left = [], args.steps - tail
        while left > 0:
            n = min(left, 4096)
            parts.append(er.iterate_timed(n))
            left -= n
        steady = er.iterate_timed(tail)
        parts.append(steady)
        tot = sum(p_["iters"] for p_ in parts)
        timing = {k: (sum(p_[k] * p_["iters"] for p_ in parts) / tot if k.endswith("_ms") else steady[k]) for k in steady}
        timing["iters"] = tot
        er.close()

    if rank == 0:
        gen_b, sto_b, shared_b = algorithmic_bytes(pp.G, pp.S, pp.T, pp.N, pp.L)
        out = {
            "metric": "agent_subproblem_updates_per_sec", "value": A_global * args.steps / dt,
            "unit": "agent-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "description": desc, "agents_per_gpu": A_local,
                       "generators_per_gpu": pp.G, "storages_per_gpu": pp.S, "timesteps": pp.T,
                       "nodes": pp.N, "lines": pp.L, "gamma": gamma, "w_flow": w_flow, "w_prox": 1.0,
                       "parallelism": f"agents sharded x{world}, 1 all-reduce of {(pp.N + 2 * pp.L) * pp.T + 1} f64 per iteration"
                       if world > 1 else "single GPU"},
            "iters_per_sec": args.steps / dt,
            "updates_per_sec_per_gpu": A_local * args.steps / dt,
            "solver_failures": int(fails),
        }
        if args.scale != 1.0:
            out["invalid"] = "scaled-down workload (debug run)"
        if timing is not None:
            peak = 8000.0        # GB/s, HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling
            # an event pair costs a fixed few microseconds even with nothing between (empty_ms): net it out
            k_ms = max(timing["gen_ms"] - timing["empty_ms"], 1e-6)
            fused = bool(timing.get("agents_fused"))
            pair = pp.L == 0 and pp.T % 2 == 0
            if fused:       # ONE launch does every x-update: generators and storages
                kname, alg_b = "k_agents", gen_b + sto_b + shared_b
            else:
                kname, alg_b = ("k_gen_update_pair" if pair else "k_gen_update"), gen_b + shared_b
            ach = alg_b / (k_ms * 1e-3) / 1e9
            traffic = None       # HBM bytes per launch from the committed PMC passes (scripts/profile.sh)
            pmc_file = os.path.join(ROOT, "profiles", "r01_pmc.json")
            if os.path.exists(pmc_file):
                recs = json.load(open(pmc_file)).get(args.workload, {})
                rec = next((r for k, r in sorted(recs.items()) if k.startswith(kname) and r.get("FETCH_SIZE") is not None), None)
                if rec:        # gfx950: FETCH_SIZE counts half of a streaming read (MI355X_MICROARCH.md, HBM); unit KiB
                    traffic = (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
            ks_ms = max(steady["gen_ms"] - steady["empty_ms"], 1e-6)
            out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": peak,
                               "unit": "GB/s", "frac": ach / peak, "traffic": traffic,
                               "algorithmic_bytes_per_launch": alg_b,
                               "kernel_ms": k_ms, "kernel_ms_with_event_overhead": timing["gen_ms"],
                               "window": f"mean over the timed region's iterations ({args.warmup}..{args.warmup + args.steps - 1} from the zero "
                                         "state, replayed with events): includes the iterations in which storages fall back to the cold scan",
                               "steady_state": {"kernel_ms": ks_ms, "achieved": alg_b / (ks_ms * 1e-3) / 1e9,
                                                "frac": alg_b / (ks_ms * 1e-3) / 1e9 / peak,
                                                "window": f"last {steady['iters']} iterations of that region"}}
            if traffic is not None:
                out["roofline"]["traffic_GBps"] = traffic / (k_ms * 1e-3) / 1e9
                out["roofline"]["traffic_frac_of_peak"] = out["roofline"]["traffic_GBps"] / peak
            bs = 256 if fused else 512
            rows = bs // (pp.T // 2) if pair else 1
            row_skip = pair and not (args.flags & _capi.F_NO_ROW_SKIP) and max(rows, -(-pp.G // 2048)) >= 8 * rows    # as dopf_create decides
            if fused:
                out["roofline"]["what"] = ("all x-updates of an iteration in one launch: generator blocks stream P (HBM bound), storage "
                                           "blocks run the warm-started SoC solve (latency/VALU bound) on the same CUs")
            if row_skip:
                if not fused:
                    out["roofline"]["kernel"] = "k_gen_update_pair_skip"
                out["roofline"]["note"] = ("rows of P that sit on a bound for all timesteps and provably stay there are neither read nor "
                                           "written (bit-identical results), so the launch moves fewer bytes than the 16T+20 B per-update "
                                           "model: `achieved`/`frac` (algorithmic bytes / time) can exceed what `traffic` shows moved")
            out["kernels_ms"] = {k: v for k, v in timing.items() if k.endswith("_ms")}
            out["kernels_ms_steady_state"] = {k: v for k, v in steady.items() if k.endswith("_ms")}
            out["agents_fused"] = fused
            if not fused:
                s_ms = max(timing["sto_ms"] - timing["empty_ms"], 1e-6)
                out["storage_kernel"] = {"kernel": "k_sto (warm start, then the cold scan for what it left over)" if pp.L == 0 else "k_sto_warm + k_sto_update", "bound": "fp64 VALU (segmented Newton + certificate; scan fallback), not HBM",
                                         "algorithmic_bytes_per_launch": sto_b, "kernel_ms": s_ms,
                                         "achieved_GBps": sto_b / s_ms * 1e-6}
            if fused and world == 1 and not args.force_sharded:
                # the two halves of k_agents on their own (separate launches, DOPF_F_NO_FUSE), steady state: the generator
                # sweep is the HBM-bound part, the storage solve the latency-bound one
                ex = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank,
                                                                               flags=args.flags | _capi.F_NO_FUSE), **pp.engine_kwargs())
                ex.iterate(args.warmup + args.steps - steady["iters"])
                tx = ex.iterate_timed(steady["iters"])
                ex.close()
                g_ms, s_ms2 = max(tx["gen_ms"] - tx["empty_ms"], 1e-6), max(tx["sto_ms"] - tx["empty_ms"], 1e-6)
                out["roofline"]["parts_as_separate_launches"] = {
                    "k_gen_update_pair": {"kernel_ms": g_ms, "algorithmic_bytes_per_launch": gen_b + shared_b,
                                          "achieved": (gen_b + shared_b) / g_ms * 1e-6, "frac": (gen_b + shared_b) / g_ms * 1e-6 / peak},
                    "k_sto (warm start + cold scan)": {"kernel_ms": s_ms2, "algorithmic_bytes_per_launch": sto_b,
                                                       "achieved": sto_b / s_ms2 * 1e-6, "frac": sto_b / s_ms2 * 1e-6 / peak},
                    "window": f"iterations {args.warmup + args.steps - steady['iters']}..{args.warmup + args.steps - 1}, as the steady state above"}
            whole = (gen_b + sto_b + shared_b) / (dt / args.steps) / 1e9
            out["whole_iteration_GBps"] = whole
            out["whole_iteration_frac_of_peak"] = whole / peak
        if not sharded:
            # the other half of BASELINE's metric: wall time until every |dual change| < 1e-3, from the zero state
            budget = int(max(64, min(100000, 10.0 * args.steps / dt)))     # at most ~10 s of iterations
            e2 = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=1e-3, max_iters=budget, device=local_rank, flags=args.flags),
                              **pp.engine_kwargs())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done2, conv2 = 0, False
            while not conv2 and done2 < budget:        # short slices: (almost) no launches after the stop test fires
                d_, conv2 = e2.iterate(min(32, budget - done2))
                done2 += d_
                if d_ == 0:
                    break
            t2 = time.perf_counter() - t0
            out["time_to_1e-3_residual"] = {"seconds": t2 if conv2 else None, "iterations": done2, "converged": bool(conv2),
                                            "iteration_cap": budget, "total_cost": e2.get_consensus()[4]}
            # BASELINE's third target: converged objective within 1e-3 of the central optimum (fixtures: HiGHS solves of
            # the same seed-stable cases, tests/golden/make_synthetic_optima.py)
            opt_file = os.path.join(ROOT, "tests", "golden", "synthetic_optima.json")
            if conv2 and args.scale == 1.0 and os.path.exists(opt_file):
                opt = json.load(open(opt_file)).get(args.workload)
                if opt and (opt["G"], opt["S"], opt["T"]) == (pp.G, pp.S, pp.T) and args.w_flow is None:
                    out["time_to_1e-3_residual"]["central_lp_optimum"] = opt["objective"]
                    out["time_to_1e-3_residual"]["relative_gap_to_central_lp"] = \
                        abs(out["time_to_1e-3_residual"]["total_cost"] - opt["objective"]) / opt["objective"]
            e2.close()
        if not sharded and not args.no_also and args.scale == 1.0:
            # the other BASELINE configurations that fit one GPU, same engine, short runs (reported, not the metric)
            also = []
            for wl in ("config1", "config4", "config2", "config3-share"):
                if wl == args.workload:
                    continue
                ppx = synth.baseline_config(WORKLOADS[wl][0], scale=SHARE.get(wl, 1.0))
                Ax = ppx.G + ppx.S
                ex = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / Ax, w_flow=10.0 if ppx.L == 0 else 0.3 / Ax,
                                                                               eps=0.0, device=local_rank),
                                  **ppx.engine_kwargs())
                ex.iterate(args.warmup)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ex.iterate(200)
                tx = time.perf_counter() - t0
                gb, sb, shb = algorithmic_bytes(ppx.G, ppx.S, ppx.T, ppx.N, ppx.L)
                also.append({"workload": wl, "agents": Ax, "timesteps": ppx.T, "iters_per_sec": 200 / tx,
                             "agent_updates_per_sec": Ax * 200 / tx, "ms_per_step": 1e3 * tx / 200,
                             "whole_iteration_GBps": (gb + sb + shb) / (tx / 200) / 1e9})
                ex.close()
            out["also"] = also
        if not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pp, gamma, w_flow)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

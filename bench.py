#!/usr/bin/env python3
"""bench.py — ADMM consensus-OPF iterations on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2] [--gamma G]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ADMM iteration (all agent x-updates + consensus + dual update + stop test) over one
synthetic N-agent x T-timestep grid that is resident in HBM before the timed region starts.
Prints ONE JSON line on rank 0:
  metric/value   agent-subproblem-updates per second, whole job (= agents x iterations / s)
  roofline       the dominant kernel: algorithmic bytes per launch / its duration over the timed region (`frac`) and in steady
                 state. On copper plates the whole iteration is ONE launch (consensus sum, dual step and stop test run in the
                 launch's last block), and the duration is the per-iteration time of the graph replay itself — `frac` is then the
                 whole-iteration fraction; otherwise HIP events net of the event overhead that makes the kernels of an
                 iteration add up to that per-iteration time. `traffic` = HBM bytes per launch from the committed PMC passes
                 (`traffic_source` names the file); `whole_iteration` = all kernels and gaps
  cpu_baseline   the oracle's exact mode (a "port", oracle/dopf_oracle.c) on the host cores, bounded sample
Workloads (BASELINE.json `configs`): config1 = 1k gens + 100 storages x 24; config2 = 50k agents x 96
(default: the largest single-GPU configuration); config4 = 1M agents x 24; config3 = 118-node/186-line
synthetic network, 100k agents x 168.

--gpus N > 1 without a torch.distributed.run environment: this process starts the N ranks itself (children,
one per GPU, before anything here touches the GPU) and relays rank 0's line. Every rank holds one full grid
(weak scaling); the per-iteration sum of the consensus vector over the ranks is done inside the library: `--comm p2p`
the peer exchange (one kernel of the iteration graph stores the vector into every peer's memory over xGMI and adds the
copies in rank order), `--comm lib` an RCCL all-reduce issued by the library, `--comm torch` the older path where
torch.distributed issues it between two library calls. `--comm auto` (default) tries them in that order: a transport
that raises, leaves the ranks with different duals, fails the host-side check of the summed injections or fails in the
timed region is given up in place and named in the line (`comm.transports_given_up`).
"""
import argparse
import json
import os
import subprocess
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
    # HBM stress beyond the 256 MiB Infinity Cache: P alone is 384 MB
    "config4x2": (5, "synthetic 1M agents, 48 timesteps, copper plate: config4 with twice the horizon, so that the generator "
                     "array (384 MB) no longer fits the 256 MiB Infinity Cache"),
}
SHARE = {"config3-share": 0.125}
PEAK_GBPS = 8000.0        # HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling


def pmc_iteration_traffic(wl_):
    """HBM bytes of ONE iteration (all kernels of the chain) from the committed PMC passes, or None"""
    for tag in ("r04", "r03", "r02"):
        f_ = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
        if os.path.exists(f_):
            recs = {k: r for k, r in json.load(open(f_)).get(wl_, {}).items()
                    if k.startswith(("k_agents", "k_net_agents", "k_gen_update", "k_sto", "k_reduce", "k_dual_price_small<true", "k_dual_price_t1024<true",
                                     "k_tables", "k_slack")) and r.get("FETCH_SIZE") is not None and r.get("WRITE_SIZE") is not None}
            if recs:
                return sum((2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0 for r in recs.values()), f"profiles/{tag}_pmc.json"
    return None, None


def make_problem(synth, wl, scale=1.0):
    idx = WORKLOADS[wl][0]
    if idx == 5:
        a = int(1000000 * scale)
        return synth.synthetic_case(a - a // 11, a // 11, 48)
    return synth.baseline_config(idx, scale=scale * SHARE.get(wl, 1.0))


def algorithmic_bytes(G, S, T, N, L):
    """per generator update 16T+20 B, per storage update 40T+28 B, shared
    consensus data (N+5L+2)*8T B once per GPU per iteration."""
    gen = G * (16 * T + 20)
    sto = S * (40 * T + 28)
    shared = (N + 5 * L + 2) * 8 * T
    return gen, sto, shared


def bytes_moved(G, S, T, N, L):
    """What the kernels of this build actually move of the model above: the storage level E = cumsum(C - D) is not stored by
    the solve any more (nothing on the path reads it back; dopf_get_primal rebuilds it on request) — 32T+28 B per storage
    update instead of 40T+28. (Row skipping moves fewer generator bytes still: that is `traffic`, measured.)"""
    gen, sto, shared = algorithmic_bytes(G, S, T, N, L)
    return gen, S * (32 * T + 28), shared


def valu_roofline(workload, kernel_prefix, kernel_ms):
    """The VALU bound of a kernel next to its HBM bound: vector instructions per launch x 4 cycles (the issue rate of a wave64
    fp64 instruction on a SIMD, measured 4.0x for every kernel here) / (1024 SIMDs x 2.4 GHz) = the time the chip's vector
    pipes need for them at full occupancy and maximum clock; `frac` = that / the kernel's duration. Instruction counts from the
    committed counter passes (profiles/<tag>_valu.json: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ..., scripts/prof_pmc.sh)."""
    for tag in ("r04", "r03"):
        f_ = os.path.join(ROOT, "profiles", f"{tag}_valu.json")
        if not os.path.exists(f_):
            continue
        recs = json.load(open(f_)).get(workload, {})
        rec = next((r for k, r in sorted(recs.items()) if k.startswith(kernel_prefix) and r.get("SQ_INSTS_VALU")), None)
        if rec:
            n = rec["SQ_INSTS_VALU"]
            t_min_ms = n * 4.0 / (1024 * 2.4e9) * 1e3
            return {"bound": "valu", "valu_instructions_per_launch": n, "valu_instructions_per_wave": n / max(rec.get("SQ_WAVES", 1.0), 1.0),
                    "cycles_per_instruction": 4.0, "simds": 1024, "clock_GHz": 2.4, "min_ms": t_min_ms, "kernel_ms": kernel_ms,
                    "frac": t_min_ms / kernel_ms if kernel_ms else None,
                    "source": f"profiles/{tag}_valu.json (counter passes of `bench.py --workload {workload}`, mean per launch — not measured in this run)"}
    return None


def cpu_baseline(pp, gamma, w_flow, synth, budget_s=14.0):
    """Oracle (exact mode) on the host cores: all cores on a bounded number of iterations of the SAME problem, and one
    core on a 1/25 cut of it (same generator, horizon and penalty; a full iteration on one core takes minutes)."""
    from decentralopf_jl_amd import _capi
    from oracle import binding as ob
    import __graft_entry__ as ge
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build()
    api = ob.OracleApi(ge.ORACLE_LIB)
    # the process's CPU share, not the machine's thread count (a one-GPU lease of the pool is 16 CPUs of a 256-thread host)
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)

    def run(problem, threads, budget):
        e = _capi.Engine(api, params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0), mode=ob.MODE_EXACT,
                         **problem.engine_kwargs())
        ob.set_threads(e, threads)
        t0 = time.perf_counter()
        e.iterate(1)
        t1 = time.perf_counter() - t0
        n = 1
        extra = int(max(0, min(50, (budget - t1) // max(t1, 1e-6))))
        if extra > 0:
            t0 = time.perf_counter()
            e.iterate(extra)
            t1 += time.perf_counter() - t0
            n += extra
        e.close()
        return n, t1
    A = pp.G + pp.S
    cut = None
    if pp.meta.get("n_gen"):
        cut = synth.synthetic_case(max(1, pp.G // 25), max(1, pp.S // 25), pp.T, N=pp.N, L=pp.L, seed=pp.meta.get("seed", synth.SEED))
        run(cut, cores, 1.5)          # untimed: the OpenMP team's first second in a process runs far below its steady rate
    n, t1 = run(pp, cores, budget_s)
    out = {"value": A * n / t1, "unit": "agent-updates/s", "cores": cores, "kind": "port",
           "sample": f"{n} ADMM iteration(s) of the same workload from the zero state, oracle exact mode, "
                     f"OpenMP over agents ({t1:.1f} s)",
           "iters_per_sec": n / t1,
           "note": "a reported baseline, not the target: the reference's own JuMP/Gurobi path cannot be timed (no Julia, no licence); "
                   "the GPU/CPU ratio says nothing about kernel quality, roofline.frac does"}
    if cut is not None:
        n1, t11 = run(cut, 1, budget_s / 2)
        out["one_core"] = {"value": (cut.G + cut.S) * n1 / t11, "unit": "agent-updates/s", "cores": 1,
                           "sample": f"{n1} iteration(s) of a 1/25 cut of the workload ({cut.G}+{cut.S} agents, same horizon and "
                                     f"penalty), one thread ({t11:.1f} s)"}
    return out


def self_launch(args, argv, child_cmd=None, min_budget=20.0):
    """--gpus N without a launcher: start the N ranks as children (this process never touches the GPU), relay
    rank 0's JSON line. A run that hangs is killed at the deadline and retried on the next communication mode.
    child_cmd(mode, port, rest) -> argv of the child (tests substitute stub children; default: torch.distributed.run on this
    file); min_budget: floor of one attempt's share of --launch-timeout."""
    import socket
    # auto: the ranks try the library communicator and fall back to torch in place when it raises; a hang is this
    # watchdog's business, and the second attempt then goes straight to torch
    modes = [args.comm] if args.comm != "auto" else ["auto", "torch"]
    last = ""
    lost = []                           # attempts this watchdog had to end: they go into the final line (comm.transports_given_up)
    t_end = time.time() + args.launch_timeout
    os.environ.setdefault("DOPF_XCHG_TIMEOUT_MS", "5000")     # a lost peer ends a rank's exchange kernels after 5 s, not 20
    rest, skip = [], False
    for a in argv:                      # the children get the mode of the attempt: drop --comm X / --comm=X
        if skip:
            skip = False
        elif a == "--comm":
            skip = True
        elif not a.startswith("--comm="):
            rest.append(a)
    for mode in modes:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        if child_cmd is not None:
            cmd = child_cmd(mode, port, rest)
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
                  rest + [f"--comm={mode}"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   DOPF_BENCH_LOST_ATTEMPTS=json.dumps(lost))
        # the whole run has ONE budget (--launch-timeout): the first attempt may use 60 % of what is left, so that the
        # fallback attempt (torch only) still fits under the caller's own limit
        left = t_end - time.time()
        budget = max(min_budget, left * (0.6 if mode != modes[-1] else 1.0))
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, env=env, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=budget)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, 9)          # exactly the process group started above
            except ProcessLookupError:
                pass
            p.wait()
            last = f"mode {mode}: no result within {budget:.0f} s"
            lost.append({"mode": mode, "why": f"killed at deadline ({budget:.0f} s)"})
            print(f"bench.py: {last}; trying the next mode", file=sys.stderr)
            continue
        lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            sys.stdout.write(lines[-1] + "\n")
            sys.stdout.flush()
            return 0
        last = f"mode {mode}: exit code {p.returncode}"
        lost.append({"mode": mode, "why": f"exit signal {-p.returncode}" if p.returncode < 0 else f"exit code {p.returncode}"})
        print(f"bench.py: {last}; trying the next mode", file=sys.stderr)
    print(f"bench.py: multi-GPU run failed ({last})", file=sys.stderr)
    return 1


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
    ap.add_argument("--no-side", action="store_true", help="only the timed region and the kernel replay: no time-to-residual run, no central "
                                                           "reference solve, no side workloads (profiling passes)")
    ap.add_argument("--split-total", action="store_true",
                    help="N > 1: the workload's agents are SPLIT over the ranks (BASELINE configs[3]/[4] name totals: 100k / 1M agents on 8 GPUs) "
                         "instead of one full grid per rank; the line then says scaling = strong")
    ap.add_argument("--comm", default="auto", choices=["auto", "p2p", "lib", "lib-graph", "torch"],
                    help="N > 1: how the consensus vector is summed over the ranks. p2p = the library's peer exchange (one kernel "
                         "per iteration, direct stores into the peers' memory); lib = the library's own RCCL communicator, plain "
                         "launches without host synchronisation (lib-graph: the collective captured in the iteration hipGraph); "
                         "torch = torch.distributed between two library calls. auto = p2p, then lib, then torch: a transport that "
                         "raises or fails its checks is given up in place")
    ap.add_argument("--backend", default="nccl", help="--comm torch only: torch.distributed backend; nccl (= RCCL) is the product path, "
                                                      "gloo rehearses the multi-rank logic on a box with fewer GPUs than ranks")
    ap.add_argument("--force-sharded", action="store_true", help="debug: drive the sharded (all-reduce) path even on one rank")
    ap.add_argument("--launch-timeout", type=float, default=150.0,
                    help="self-launched multi-GPU run: budget in seconds for ALL attempts together (the first may use 60 %% of it)")
    args = ap.parse_args()
    if args.no_side:
        args.no_also = args.no_cpu_baseline = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # the driver's N = 1 form of the command with N > 1: be the launcher (no GPU call has happened in this process)
        raise SystemExit(self_launch(args, sys.argv[1:]))
    args.gpus = world

    # stdout carries exactly ONE line (the JSON record): libraries that chat on stdout (RCCL prints a
    # version banner at communicator creation) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import dopf_pkg
    dopf_pkg.load()
    from decentralopf_jl_amd import _capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank %= max(1, torch.cuda.device_count())       # (ranks > GPUs only happens in a rehearsal on a small box)
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    # transports to try, in order. Started by a launcher with --comm auto, a transport that raises (on any rank) or leaves
    # the ranks with different duals is given up IN PLACE and the next one is set up: the run still ends in a JSON line.
    # (A transport that hangs can only be handled from outside: self_launch's watchdog.)
    modes = (["p2p", "lib", "torch"] if args.comm == "auto" else [args.comm]) if sharded else ["single"]
    data_group = None
    if sharded:
        import datetime
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        if modes[0] in ("p2p", "lib", "lib-graph") or args.backend != "nccl":
            # control plane only (unique id, demand sum, barriers, max of the times): gloo on host tensors.
            # The data path — the per-iteration consensus sum — is the library's own RCCL communicator.
            dist.init_process_group("gloo" if modes[0] in ("p2p", "lib", "lib-graph") else args.backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=max(30.0, min(120.0, 0.5 * args.launch_timeout))))
            ctl_dev = "cpu"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            ctl_dev = "cuda"
    else:
        ctl_dev = "cpu"

    desc = WORKLOADS[args.workload][1]
    # weak scaling: every rank owns one full grid of the workload (own seed), demand adds up
    base = make_problem(synth, args.workload, args.scale)
    own_demand = None
    split_total = args.split_total and world > 1
    if split_total:
        # BASELINE's configuration itself: its agents in `world` contiguous slices, one per rank (demand stays whole)
        pp = base.shard(rank, world)
        own_demand = np.asarray(base.demand, dtype=np.float64) / world      # (for rank 0's single-GPU kernel replay)
    elif world > 1:
        cfg = dict(base.meta)
        pp = synth.synthetic_case(cfg["n_gen"], cfg["n_sto"], cfg["T"], N=cfg["N"], L=cfg["L"], seed=synth.SEED + rank)
        if cfg["L"] > 0:          # one network for everybody: rank 0's
            pp.ptdf, pp.f_max = base.ptdf, base.f_max * world
        own_demand = pp.demand.copy()
        dem = torch.tensor(pp.demand, dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(dem)
        pp.demand = dem.cpu().numpy()
    else:
        pp = base
    A_local = pp.G + pp.S
    A_global = (base.G + base.S) if split_total else A_local * world
    gamma = args.gamma if args.gamma is not None else 1.0 / A_global
    # flow-consensus weight: the reference's literal 10 with lines makes every agent undo the whole line violation on
    # its own, an all-on/all-off 2-cycle for more than a few dozen agents whatever gamma is; it has to shrink with the
    # number of agents like gamma does (0.3/A converges on the synthetic networks tried). Irrelevant on a copper plate.
    w_flow = args.w_flow if args.w_flow is not None else (10.0 if pp.L == 0 else 0.3 / A_global)
    comm_info = None

    def set_up(mode):
        """-> (engine, step(n), sync(), close()) for one transport. Two halves: everything a rank does on its own (engine, export of
        its handle / unique id) comes first and is voted on; the host collectives that carry the handles run only when EVERY rank
        got that far — a rank that failed alone would otherwise sit in a different collective than its peers until the
        host channel's timeout."""
        if mode == "single":
            e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank,
                                                                          flags=args.flags | (_capi.F_OVERLAP_AGENTS if args.overlap else 0)),
                             **pp.engine_kwargs())
            return e, e.iterate, e.sync, e.close
        if mode in ("p2p", "lib", "lib-graph"):
            e, mine, err = None, None, ""
            try:
                fl = args.flags | (_capi.F_COMM_GRAPH if mode == "lib-graph" else 0)
                e = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank, flags=fl,
                                                                              n_agents_global=A_global), **pp.engine_kwargs())
                # p2p: the ranks' receive areas are mapped into each other (hipIpc handles over the host channel), the sum is one
                # kernel of the iteration graph; lib: the library joins the ranks (RCCL) and owns the all-reduce
                mine = e.xchg_export(world) if mode == "p2p" else (e.comm_unique_id() if rank == 0 else b"")
            except Exception as ex:                 # noqa: BLE001
                err = f"{type(ex).__name__}: {ex}"
            if not all_ranks(not err):
                if e is not None:
                    e.close()
                raise RuntimeError(err or "another rank failed before the handles were exchanged")
            if mode == "p2p":
                hs = [None] * world
                if world > 1:
                    dist.all_gather_object(hs, mine)
                else:
                    hs = [mine]
                e.xchg_init(world, rank, hs)
            else:
                box = [mine if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(box, src=0)
                e.comm_init(world, rank, box[0])
            return e, e.iterate, e.sync, e.close
        from decentralopf_jl_amd.sharded import ShardedADMM
        # every rank already holds its own grid: a 1-way "shard" of its local problem, global agent count
        # passed explicitly; the all-reduce runs over all ranks on the engine's own stream
        nonlocal data_group
        if ctl_dev == "cpu" and args.backend == "nccl" and data_group is None:
            data_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # control plane stays on gloo
        sh = ShardedADMM(pp, 0, 1, gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank, n_agents_global_override=A_global)
        tens = sh._tensor

        def _all_reduce():                # (ShardedADMM.step makes the engine's stream current around its loop)
            dist.all_reduce(tens, op=dist.ReduceOp.SUM, group=data_group)
        sh._all_reduce = _all_reduce
        return sh.engine, sh.step, sh.sync, sh.engine.close

    def all_ranks(ok):
        if dist is None or world == 1:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    _wm = {}

    def clock_warm(seconds=0.25):
        """Bring the GPU out of its idle power state before a measurement: after the host-side set-up of a problem (numpy,
        uploads) the card sits at its lowest clocks and needs ~100 ms of load to ramp; the latency-bound kernel chains here
        ran up to 4x slower in that window, run to run. Untimed, touches no solver state."""
        if "a" not in _wm:
            _wm["a"] = torch.randn(2048, 2048, device="cuda")
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            for _ in range(20):
                _wm["b"] = _wm["a"] @ _wm["a"]
            torch.cuda.synchronize()

    def sums_are_right(eng):
        """The consensus sum itself, independently of the transport: every rank adds up its own units' injections on the host,
        the host channel (gloo) adds the ranks, and the result must be the injection every rank's device holds."""
        P, D, C, _E = eng.get_primal()
        loc = np.zeros((pp.N, pp.T))
        np.add.at(loc, np.asarray(pp.gen_node, dtype=np.int64), P)
        np.add.at(loc, np.asarray(pp.sto_node, dtype=np.int64), D - C)
        tot = torch.tensor(loc, dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tot)
        want = tot.cpu().numpy() - np.asarray(pp.demand, dtype=np.float64).reshape(pp.N, pp.T)
        got = eng.get_consensus()[0]
        return float(np.abs(got - want).max()) <= 1e-9 * max(1.0, float(np.abs(want).max()))

    clock_warm()
    side_errors = []                # errors of side solves and solver failures: reported in the line, never fatal
    comm_mode, given_up, dt = None, [], None
    try:                            # attempts the launcher's watchdog had to end (killed at the deadline / died on a signal)
        given_up += json.loads(os.environ.get("DOPF_BENCH_LOST_ATTEMPTS", "[]"))
    except ValueError:
        pass
    for mode in modes:
        # one transport: set-up, warm-up, checks, then the timed region; whatever goes wrong on any rank, all ranks move on
        why, closer = "", None
        try:
            eng, step, sync, closer = set_up(mode)
            step(args.warmup)
            sync()
        except Exception as e:                      # noqa: BLE001 — whatever the transport raised
            why = f"{type(e).__name__}: {e}"
        if not all_ranks(not why):
            why = why or "another rank failed"
        elif world > 1:
            # every rank must hold the same duals: they are computed redundantly from the summed vector. A sum that did
            # not happen (or happened out of order) shows up here, before anything is timed.
            lam_here = torch.tensor(eng.get_duals()[0], dtype=torch.float64, device=ctl_dev)
            lam_all = [torch.zeros_like(lam_here) for _ in range(world)]
            dist.all_gather(lam_all, lam_here)
            if any(not torch.equal(lam_all[0], x) for x in lam_all[1:]):
                why = f"duals differ across ranks after {args.warmup} iterations — the consensus sum did not do its job"
            elif not all_ranks(sums_are_right(eng)):
                why = "the summed injections are not the sum of the ranks' injections"
        if not why:
            try:
                barrier()
                t0 = time.perf_counter()
                try:
                    step(args.steps)
                except _capi.DopfError as e:        # DOPF_E_SOLVER: the iterations ran, a storage root search hit its cap
                    if "sub-problem" not in str(e):
                        raise
                    side_errors.append(f"timed region: {e}")
                sync()
                barrier()
                dt = time.perf_counter() - t0
            except Exception as e:                  # noqa: BLE001
                why = f"in the timed region: {type(e).__name__}: {e}"
            if not all_ranks(not why):
                why = why or "another rank failed in the timed region"
        if not why:
            comm_mode = mode
            break
        given_up.append({"mode": mode, "why": why})
        print(f"bench.py rank {rank}: transport {mode} given up ({why})", file=sys.stderr)
        if closer is not None:
            try:
                closer()
            except Exception:                       # noqa: BLE001
                pass
    if comm_mode is None:
        raise SystemExit(f"rank {rank}: no transport worked: {given_up}")
    use_lib_comm = comm_mode in ("lib", "lib-graph", "p2p")
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    it_after, _ = sync()
    assert it_after == 1 + args.warmup + args.steps, (it_after, args.warmup, args.steps)
    fails = eng.solver_failures()
    if use_lib_comm:
        w_, r_, g_ = eng.comm_info()
        comm_info = {"transport": ("peer exchange: one kernel per iteration stores the rank's consensus vector into every peer's "
                                   "memory and adds the copies in rank order (dopf_xchg_*)") if comm_mode == "p2p" else
                                  "RCCL all-reduce issued by libdopf_hip (dopf_comm_init)",
                     "mode": comm_mode, "world": w_, "captured_in_hipgraph": bool(g_)}
    elif sharded:
        comm_info = {"transport": f"torch.distributed ({args.backend}) all-reduce between dopf_local_update and dopf_apply_consensus",
                     "world": world, "captured_in_hipgraph": False}
    if comm_info is not None and given_up:
        # in-place fallbacks of this run AND attempts the launcher's watchdog had to end (a retry must not hide an abort)
        comm_info["transports_given_up"] = given_up

    # Per-kernel durations, live, HIP events on the stream the kernels run on. The timed region above replays
    # hipGraphs (no place for events), so the SAME iterations — W warm-up, then K — are run again on a fresh engine,
    # the K launched kernel by kernel with an event pair around each: `timing` = mean over iterations W+1..W+K,
    # `steady` = `timed_iters` iterations after iteration max(W+K, 200) (no cold-start or structure-change iterations
    # left in there, whatever W and K are).
    # With N > 1 ranks the replay is rank 0's own grid as a single-GPU problem (its own demand, gamma = 1/A_local): the
    # kernels one GPU runs per iteration, without the collective.
    timing = steady = dev_it_ms = None
    steady_from = 0
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
        clock_warm()
        er.iterate(args.warmup)
        parts, left = [], args.steps
        while left > 0:
            n = min(left, 4096)
            parts.append(er.iterate_timed(n))
            left -= n
        tot = sum(p_["iters"] for p_ in parts)
        timing = {k: (sum(p_[k] * p_["iters"] for p_ in parts) / tot if k.endswith("_ms") else parts[-1][k]) for k in parts[-1]}
        timing["iters"] = tot
        steady_from = max(args.warmup + args.steps, 200)
        if steady_from > args.warmup + args.steps:
            er.iterate(steady_from - args.warmup - args.steps)
        steady = er.iterate_timed(max(1, args.timed_iters))
        er.close()
        # ... and once more through the graphs with an event in front of the first launch and one behind the last
        # (DOPF_F_TIME_CALLS): the device-side per-iteration time of the timed region's iterations — the wall-clock
        # figure above also carries the host's launch latency and the status read-back of the call, which on a short
        # region (20 iterations) is a tenth of it
        et = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=g_r, w_flow=wf_r, eps=0.0, device=local_rank,
                                                                       flags=args.flags | _capi.F_TIME_CALLS), **ppr.engine_kwargs())
        clock_warm()
        et.iterate(args.warmup)
        et.iterate(args.steps)
        dev_it_ms = et.last_call_ms() / args.steps if et.last_call_ms() > 0 else None
        et.close()

    if rank == 0:
        gen_b, sto_b, shared_b = algorithmic_bytes(pp.G, pp.S, pp.T, pp.N, pp.L)
        sto_mv = bytes_moved(pp.G, pp.S, pp.T, pp.N, pp.L)[1]
        out = {
            "metric": "agent_subproblem_updates_per_sec", "value": A_global * args.steps / dt,
            "unit": "agent-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if split_total else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "description": desc, "agents_per_gpu": A_local,
                       "generators_per_gpu": pp.G, "storages_per_gpu": pp.S, "timesteps": pp.T,
                       "nodes": pp.N, "lines": pp.L, "gamma": gamma, "w_flow": w_flow, "w_prox": 1.0,
                       "parallelism": (f"agents sharded x{world} ({'the configuration split over the ranks' if split_total else 'one full grid per rank'}), "
                                       f"1 all-reduce of {(pp.N + 2 * pp.L) * pp.T + 1} f64 per iteration") if world > 1 else "single GPU",
                       "agents_total": A_global},
            "iters_per_sec": args.steps / dt,
            "updates_per_sec_per_gpu": A_local * args.steps / dt,
            "solver_failures": int(fails),
            "timed_region": f"iterations {args.warmup + 1}..{args.warmup + args.steps} from the zero state (the run starts cold: the first "
                            "~15 iterations re-shape every storage's contact structure and cost 1.5-3x a settled iteration)",
        }
        if comm_info:
            out["comm"] = comm_info
        if args.scale != 1.0:
            out["invalid"] = "scaled-down workload (debug run)"
        it_ms = 1e3 * dt / args.steps
        whole_b = gen_b + sto_b + shared_b
        if timing is not None:
            fused = bool(timing.get("agents_fused"))
            tail = bool(timing.get("tail_fused"))
            pair = pp.L == 0 and pp.T % 2 == 0
            bs = 256 if fused else 512
            rows = bs // (pp.T // 2) if pair else 1
            row_skip = pair and not (args.flags & _capi.F_NO_ROW_SKIP) and max(rows, -(-pp.G // 2048)) >= 8 * rows    # as dopf_create decides
            if fused:       # ONE launch does every x-update: generators and storages
                kname, alg_b = (("k_agents_l" if timing.get("sto_lean") else "k_agents") if pp.L == 0 else "k_net_agents"), whole_b
            else:
                kname = ("k_gen_update_pair_skip" if row_skip else "k_gen_update_pair") if pair else "k_gen_update"
                alg_b = gen_b + shared_b
            # Kernel durations. An event pair adds a fixed cost to what it brackets (`empty_ms` is an upper bound of it: part
            # of an empty pair's cost overlaps with a real kernel's own launch). The overhead used here is the one that makes
            # the kernels of an iteration ADD UP to the per-iteration time of the graph replay (no events in there): live,
            # no constant. rocprofv3's per-kernel durations agree with it (profiles/).
            launched = ["gen_ms"] + ([] if fused else ["sto_ms"]) + (["slack_ms"] if pp.L > 0 and not timing.get("quiet") else []) + \
                       (["tables_ms"] if pp.L > 0 and timing["tables_ms"] > 1.5 * timing["empty_ms"] else []) + \
                       ([] if tail else ([] if timing.get("slack_in_dual") else ["reduce_ms"]) + ["dual_ms"])

            def overhead(tm, per_iter_ms):
                return min(tm["empty_ms"], max(0.0, (sum(tm[k] for k in launched) - per_iter_ms) / len(launched)))
            # per-iteration time of the graph replay on the device (events around the call's launches); rank 0's own grid when N > 1
            ref_it_ms = dev_it_ms if dev_it_ms else it_ms
            ov = overhead(timing, ref_it_ms)
            one_launch = fused and tail
            # one launch per iteration: the per-iteration time of the graph replay IS the launch (plus its one kernel boundary)
            k_ms = ref_it_ms if one_launch else max(timing["gen_ms"] - ov, 1e-6)
            traffic, traffic_source = None, None       # HBM bytes per launch from the committed PMC passes (scripts/profile.sh)
            for tag in ("r04", "r03", "r02", "r01"):
                pmc_file = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
                if traffic is None and os.path.exists(pmc_file):
                    recs = json.load(open(pmc_file)).get(args.workload, {})
                    rec = next((r for k, r in sorted(recs.items()) if k.startswith(kname) and r.get("FETCH_SIZE") is not None), None)
                    if rec:        # gfx950: FETCH_SIZE counts half of a streaming read (MI355X_MICROARCH.md, HBM); unit KiB
                        traffic = (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0
                        traffic_source = (f"profiles/{tag}_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the default command "
                                          f"(`python bench.py --workload {args.workload}`), mean per launch — not measured in this run")
            # steady state: its own per-iteration time is not measured with the graph; same overhead as the timed region
            ks_ms = max(steady["gen_ms"] - ov, 1e-6)
            if one_launch:
                ks_ms = max(steady["iter_ms"] - (timing["iter_ms"] - ref_it_ms), 1e-6)
            # Row skipping moves fewer bytes than the 16T+20 B/update model: algorithmic bytes / time would exceed what the
            # memory system did (and the 8 TB/s peak). The roofline fraction is then taken from the MEASURED traffic.
            basis_b, basis = alg_b, "algorithmic bytes (SURVEY.md 8d model)"
            if row_skip and traffic is not None:
                basis_b, basis = traffic, ("measured HBM traffic (PMC): the kernel skips rows that provably stay on a bound, so the "
                                           "model's bytes are not moved")
            ach = basis_b / (k_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": PEAK_GBPS,
                               "unit": "GB/s", "frac": ach / PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                               "bytes_basis": basis,
                               "algorithmic_bytes_per_launch": alg_b,
                               "algorithmic_bytes_moved": alg_b - (sto_b - sto_mv) if fused else alg_b,
                               "frac_of_bytes_moved": (alg_b - (sto_b - sto_mv) if fused else alg_b) / (k_ms * 1e-3) / 1e9 / PEAK_GBPS,
                               "kernel_ms": k_ms,
                               "kernel_ms_basis": ("device-side per-iteration time of the timed region's graph replay (events around the call's launches, "
                                                   "DOPF_F_TIME_CALLS): the iteration is this ONE launch (x-updates, consensus sum, dual step, stop test) "
                                                   "plus its kernel boundary") if one_launch else
                                                  ("HIP events on the kernels' stream over the timed region's iterations (replayed), net of the event "
                                                   "overhead that makes the iteration's kernels add up to the graph replay's device-side per-iteration time"),
                               "device_ms_per_iteration": dev_it_ms,
                               "kernel_ms_events_raw": timing["gen_ms"], "event_overhead_ms": ov, "empty_event_pair_ms": timing["empty_ms"],
                               "window": f"the timed region: iterations {args.warmup + 1}..{args.warmup + args.steps} from the zero state",
                               "steady_state": {"kernel_ms": ks_ms, "achieved": basis_b / (ks_ms * 1e-3) / 1e9,
                                                "frac": basis_b / (ks_ms * 1e-3) / 1e9 / PEAK_GBPS,
                                                "window": f"{steady['iters']} iterations from iteration {steady_from + 1} on"},
                               "whole_iteration": {"bytes": whole_b, "ms": it_ms, "achieved": whole_b / it_ms * 1e-6,
                                                   "frac": whole_b / it_ms * 1e-6 / PEAK_GBPS,
                                                   "what": "algorithmic bytes of one iteration / per-iteration time of the timed region "
                                                           "(all launches and the gaps between them)"}}
            vr = valu_roofline(args.workload, kname, k_ms)
            if vr:
                out["roofline"]["valu"] = vr
            if row_skip:
                out["roofline"]["algorithmic_GBps"] = alg_b / (k_ms * 1e-3) / 1e9
                trf_all, src_all = pmc_iteration_traffic(args.workload)
                if trf_all is not None:      # (the model's bytes are not moved: what the memory system did, all kernels of the iteration)
                    out["roofline"]["whole_iteration"].update({"hbm_traffic": trf_all, "hbm_traffic_source": src_all,
                                                               "hbm_traffic_frac": trf_all / it_ms * 1e-6 / PEAK_GBPS})
            if fused:
                out["roofline"]["what"] = ("all x-updates of an iteration in one launch: generator blocks " +
                                           ("stream P (HBM bound)" if pp.L == 0 else "sweep P against the nodes' Psi tables") + ", storage "
                                           "blocks run the active-set SoC solve (fp64 VALU bound) on the same CUs" +
                                           ("; its last block adds up the blocks' sums and runs the dual step and the stop test" if tail else ""))
            out["kernels_ms"] = {k: v for k, v in timing.items() if k.endswith("_ms")}
            out["kernels_ms_steady_state"] = {k: v for k, v in steady.items() if k.endswith("_ms")}
            out["agents_fused"], out["tail_fused"] = fused, tail
            if not fused:
                s_ms = max(timing["sto_ms"] - ov, 1e-6)
                out["storage_kernel"] = {"kernel": ("k_sto_l" if timing.get("sto_lean") else "k_sto") + " (active-set solve; scan kernel for what it leaves over" +
                                                   ("; carries the iteration's tail block)" if tail else ")") if pp.L == 0 else "k_sto_warm + k_sto_update",
                                         "bound": "fp64 VALU (segmented Newton + certificate), not HBM",
                                         "algorithmic_bytes_per_launch": sto_b, "algorithmic_bytes_moved": sto_mv, "kernel_ms": s_ms,
                                         "achieved_GBps": sto_b / s_ms * 1e-6, "frac_of_hbm_peak": sto_b / s_ms * 1e-6 / PEAK_GBPS,
                                         "valu": valu_roofline(args.workload, "k_sto_l" if timing.get("sto_lean") else "k_sto", s_ms)}
            if fused and world == 1 and not args.force_sharded and not args.no_side:
                # the two halves of k_agents on their own (separate launches, DOPF_F_NO_FUSE), steady state: the generator
                # sweep is the HBM-bound part, the storage solve the VALU-bound one
                ex = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=0.0, device=local_rank,
                                                                               flags=args.flags | _capi.F_NO_FUSE | _capi.F_NO_TAIL_FUSE), **pp.engine_kwargs())
                clock_warm()
                ex.iterate(steady_from)
                tx = ex.iterate_timed(steady["iters"])
                ex.close()
                g_ms, s_ms2 = max(tx["gen_ms"] - ov, 1e-6), max(tx["sto_ms"] - ov, 1e-6)
                out["roofline"]["parts_as_separate_launches"] = {
                    "k_gen_update_pair": {"kernel_ms": g_ms, "algorithmic_bytes_per_launch": gen_b + shared_b,
                                          "achieved": (gen_b + shared_b) / g_ms * 1e-6, "frac": (gen_b + shared_b) / g_ms * 1e-6 / PEAK_GBPS},
                    "k_sto": {"kernel_ms": s_ms2, "algorithmic_bytes_per_launch": sto_b,
                              "achieved": sto_b / s_ms2 * 1e-6, "frac": sto_b / s_ms2 * 1e-6 / PEAK_GBPS},
                    "window": "as the steady state above; partial rows + k_reduce + dual kernel (DOPF_F_NO_TAIL_FUSE)"}
            out["whole_iteration_GBps"] = whole_b / it_ms * 1e-6
            out["whole_iteration_frac_of_peak"] = whole_b / it_ms * 1e-6 / PEAK_GBPS
        if not sharded and not args.no_side:
            try:
                # the other half of BASELINE's metric: wall time until every |dual change| < 1e-3, from the zero state
                budget = int(max(64, min(100000, 10.0 * args.steps / dt)))     # at most ~10 s of iterations
                e2 = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=gamma, w_flow=w_flow, eps=1e-3, max_iters=budget, device=local_rank, flags=args.flags),
                                  **pp.engine_kwargs())
                clock_warm()
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
                # the central reference solved on the device (dopf_central_solve = src/opf_central_reference.jl as a first-order LP
                # solve): the same target without a host LP
                t0 = time.perf_counter()
                cr = _capi.central_solve(_capi.hip_api(), tol=1e-7, max_iters=100000,
                                         params=_capi.default_params(device=local_rank), **pp.engine_kwargs())
                tcr = time.perf_counter() - t0
                out["central_reference_on_device"] = {"objective": cr["objective"], "dual_objective": cr["dual_objective"],
                                                      "primal_infeasibility": cr["primal_infeasibility"], "gap": cr["gap"],
                                                      "iterations": cr["iterations"], "converged": cr["converged"], "seconds": tcr}
                if conv2 and cr["converged"]:
                    out["time_to_1e-3_residual"]["relative_gap_to_device_central_reference"] = \
                        abs(out["time_to_1e-3_residual"]["total_cost"] - cr["objective"]) / cr["objective"]
            except _capi.DopfError as e:
                side_errors.append(f"time-to-residual / central reference: {e}")
        if not sharded and not args.no_also and args.scale == 1.0:
            # the other BASELINE configurations that fit one GPU, same engine, short runs (reported, not the metric)
            also = []
            for wl in ("config1", "config4", "config2", "config3-share", "config3", "config4x2"):
                if wl == args.workload:
                    continue
                ppx = make_problem(synth, wl)
                Ax = ppx.G + ppx.S
                ex = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0 / Ax, w_flow=10.0 if ppx.L == 0 else 0.3 / Ax,
                                                                               eps=0.0, device=local_rank),
                                  **ppx.engine_kwargs())
                wx = max(args.warmup, 48)
                clock_warm()
                ex.iterate(wx)
                torch.cuda.synchronize()
                nx = 200 if wl != "config3" else 100
                t0 = time.perf_counter()
                ex.iterate(nx)
                tx = time.perf_counter() - t0
                gb, sb, shb = algorithmic_bytes(ppx.G, ppx.S, ppx.T, ppx.N, ppx.L)
                skips = wl in ("config4", "config4x2")      # (row skipping: the model's bytes are not moved — see hbm_traffic_* below)
                key = "model_bytes" if skips else "whole_iteration"
                also.append({"workload": wl, "agents": Ax, "timesteps": ppx.T, "iters_per_sec": nx / tx,
                             "agent_updates_per_sec": Ax * nx / tx, "ms_per_step": 1e3 * tx / nx,
                             key + "_GBps": (gb + sb + shb) / (tx / nx) / 1e9,
                             key + "_frac_of_peak": (gb + sb + shb) / (tx / nx) / 1e9 / PEAK_GBPS,
                             "window": f"iterations {wx + 1}..{wx + nx}"})
                if skips:
                    also[-1]["model_bytes_note"] = ("SURVEY 8d's bytes per update x updates / time: NOT a bandwidth — the generator sweep skips "
                                                    "rows parked on a bound, so these bytes are not moved; hbm_traffic_frac_of_peak is the measured one")
                if wl in ("config4", "config4x2"):
                    # row skipping moves fewer bytes than the model; config4x2's arrays (384 MB of P) do not fit the 256 MiB Infinity
                    # Cache: there "fraction of HBM peak" means HBM. Bytes from the committed PMC passes, time from this run.
                    trf, src = pmc_iteration_traffic(wl)
                    if trf is not None:
                        also[-1]["hbm_traffic_per_iteration"] = trf
                        also[-1]["hbm_traffic_frac_of_peak"] = trf / (tx / nx) / 1e9 / PEAK_GBPS
                        also[-1]["hbm_traffic_source"] = src + " (all kernels of an iteration; not measured in this run)"
                ex.close()
            # BASELINE configs[1] names "rho = 1.0" (rho there = the reference's gamma): the iteration RATE with that literal
            # penalty (it does not converge at this size: Jacobi with a fixed prox weight)
            pp1 = make_problem(synth, "config1")
            ex = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=1.0, eps=1e-3, device=local_rank), **pp1.engine_kwargs())
            clock_warm()
            ex.iterate(48)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d1, c1 = ex.iterate(400)
            tx = time.perf_counter() - t0
            also.append({"workload": "config1, literal gamma = 1.0 (BASELINE's rho)", "agents": pp1.G + pp1.S, "timesteps": pp1.T,
                         "iters_per_sec": d1 / tx, "agent_updates_per_sec": (pp1.G + pp1.S) * d1 / tx, "ms_per_step": 1e3 * tx / max(d1, 1),
                         "converged": bool(c1), "lambda_residual_after_448": ex.get_residuals()[0],
                         "note": "oscillates: every agent answers the same imbalance, aggregate gain ~ A*gamma/(1+gamma)"})
            ex.close()
            out["also"] = also
            # BASELINE configs[4] asks for a penalty sweep on the 1M-agent grid: iterations / seconds to the 1e-3 residual
            # (the same on the headline workload: where the bench's gamma = 1/A sits in its convergent range)
            def penalty_sweep(ppx, values, cap):
                Ax, rows = ppx.G + ppx.S, []
                for m in values:
                    ex_ = _capi.Engine(_capi.hip_api(), params=_capi.default_params(gamma=m / Ax, eps=1e-3, max_iters=cap, device=local_rank),
                                       **ppx.engine_kwargs())
                    ex_.iterate(0)
                    clock_warm()
                    t0_ = time.perf_counter()
                    dn, cv = 0, False
                    while not cv and dn < cap:
                        d_, cv = ex_.iterate(min(64, cap - dn))
                        dn += d_
                        if d_ == 0:
                            break
                    tx_ = time.perf_counter() - t0_
                    rows.append({"gamma_times_A": m, "iterations": dn, "converged": bool(cv), "seconds": tx_, "iters_per_sec": dn / tx_})
                    ex_.close()
                return {"agents": Ax, "timesteps": ppx.T, "stop_test": "all |dual change| < 1e-3", "iteration_cap": cap, "rows": rows}
            out["config4_penalty_sweep"] = penalty_sweep(make_problem(synth, "config4"), (0.3, 1.0, 1.5, 3.0), 2500)
            if args.workload == "config2":
                out["config2_penalty_sweep"] = penalty_sweep(pp, (0.3, 1.0, 1.5, 2.0), 2000)
        if side_errors:
            out["errors"] = side_errors
        if not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pp, gamma, w_flow, synth)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        eng.close()              # (communicator teardown while every rank is still there)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

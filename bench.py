#!/usr/bin/env python3
"""bench.py — agent-replans/s of the HIP hot path on BASELINE.json's circle-exchange workload.

One "step" = one replan round of the local shard: ONE launch of the fused kernel (separating planes +
exact MIQP for every local agent) and, for N > 1, ONE RCCL all-gather of the new plans.

Workload (config.workload): BASELINE configs[1] at N=1 — 64 agents, circular exchange, empty environment,
H = 10, agent_agile_config.yaml weights/limits. For N GPUs the swarm has 64*N agents, 64 per GPU (weak
scaling; --agents overrides, e.g. --agents 1024 with --gpus 8 is BASELINE configs[3]).

Inputs are produced by SIMULATION, not drawn from a distribution: during the untimed
set-up the swarm is flown in closed loop with the device solver; the inputs of rounds
[--first-round, --first-round + warmup + steps) — when the agents converge on the centre and many separating
planes are active — are kept resident in HBM and replayed, one recorded round per step.

Prints ONE JSON line (rank 0). `roofline.achieved` = algorithmic bytes per launch (SURVEY.md section 8d
formula x agents per launch) / mean kernel duration measured with HIP events on the launch stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(n_rob, N, P, rbar):
    """fp64 bytes one agent-replan must touch."""
    return ((n_rob - 1) * N * 3 * 8 + (N + 1) * 3 * 8 + 9 * 8 + N * 6 * 8 + P * rbar * 4 * 8
            + (N + 1) * 9 * 8 + N * 3 * 8 + P)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--agents", type=int, default=0, help="total agents (default 64 per GPU)")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--first-round", type=int, default=25, help="first recorded closed-loop round")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="wall-clock budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache", default="", help="npz file with the recorded rounds: written after the closed-loop "
                    "set-up if missing, loaded instead of flying the swarm if present (profiling runs: only the "
                    "timed launches remain in the process)")
    ap.add_argument("--no-event-pass", action="store_true", help="skip the second (HIP-event) pass")
    ap.add_argument("--radius", type=float, default=0.0, help="circle radius [m]. Default: 22 m x number of GPUs when the "
                    "swarm is the default 64 agents per GPU (the ring density of BASELINE configs[1] at every N); with "
                    "--agents, max(22, agents / 2 pi) (chord >= 1 m, SURVEY.md section 8d)")
    ap.add_argument("--scenario", choices=("circle", "lanes"), default="circle", help="circle: the antipodal exchange "
                    "of BASELINE configs[1] (the bench line). lanes: a line formation (32 lanes wide, stacked in z) flying "
                    "through a pillar forest; corridors come from the voxel decomposition (next row f2). Single GPU.")
    ap.add_argument("--host-reference", action="store_true", help="generate the reference trajectories of the set-up "
                    "flight on the host (csrc/swarm_host.cpp) instead of with the f1 device kernel (hdsm_reference)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI, the product path) or gloo "
                    "(testing the multi-rank flow on a box with fewer GPUs than ranks: ranks share devices and the "
                    "all-gather is staged through the host)")
    ap.add_argument("--round-offset", type=int, default=-1, help="index of the first TIMED recorded round "
                    "(default = warmup). A profiling run uses --warmup 0 --round-offset 10 to launch exactly the "
                    "rounds the default run times")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank = local_rank % max(1, torch.cuda.device_count()) if args.dist_backend == "gloo" else local_rank
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    host_staged = world > 1 and args.dist_backend != "nccl"

    def all_gather_dev(full, shard):
        """one all-gather of the shard into the full buffer (RCCL on device memory; host-staged under gloo)"""
        if not host_staged:
            dist.all_gather_into_tensor(full, shard)
        else:
            f = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(f, shard.cpu())
            full.copy_(f)

    from multi_agent_pkgs_amd import lib, swarm
    from multi_agent_pkgs_amd.params import agile_params

    N = args.horizon
    prm = agile_params(N, max_rows_static=18)
    P, RS = prm.poly_hor, prm.max_rows_static
    n_rob = args.agents if args.agents > 0 else 64 * world
    radius = args.radius if args.radius > 0 else (22.0 * world if args.agents <= 0 else max(22.0, n_rob / (2 * np.pi)))
    first, n_local = swarm.shard_range(n_rob, rank, world)
    per = (n_rob + world - 1) // world
    K, W = args.steps, args.warmup
    off = args.round_offset if args.round_offset >= 0 else W
    n_rec = max(K + W, off + K)

    solver = lib.Solver(prm, max(n_local, 1), n_rob, device=dev.index)
    stream = torch.cuda.current_stream()

    # ---------------------------------------------------------------- set-up: closed-loop flight, recording
    def allgather_np(local):
        t = torch.from_numpy(np.ascontiguousarray(local)).to(dev)
        full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        all_gather_dev(full, t)
        return full.cpu().numpy()

    def solve_np(inp, plans, has):
        return solver.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"],
                             inp["A"], inp["b"], plans, has)

    cfg = swarm.default_swarm_config()
    rec, fails, total_rounds = [], 0, args.first_round + n_rec
    cache = (args.cache + f".rank{rank}" if world > 1 else args.cache) if args.cache else ""
    keys = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
    if cache and os.path.exists(cache):
        z = np.load(cache)
        assert int(z["n_rob"]) == n_rob and int(z["N"]) == N and z["state"].shape[0] >= n_rec
        assert "radius" not in z or abs(float(z["radius"]) - radius) < 1e-9
        rec = [{k: z[k][r] for k in keys} for r in range(n_rec)]
        fails = int(z["fails"])
    else:
        from multi_agent_pkgs_amd.params import agile_ref_config
        rcfg = agile_ref_config()

        def ref_dev(ids, path, n_path, plans, has):  # row f1 on the device: removes the host's O(n_rob^2 N) step
            full, _, pv = solver.reference(rcfg, ids, path, n_path, plans, has)
            return full, pv

        starts = goals = None
        if args.scenario == "lanes":
            assert world == 1, "--scenario lanes is a single-GPU workload"
            n_y = min(n_rob, 32)
            assert n_rob % n_y == 0
            starts, goals, occ, occ_origin = swarm.lane_forest_scenario(n_y, n_rob // n_y, seed=7)
        loop = swarm.SwarmLoop(prm, cfg, n_rob, rank=rank, world=world, solve=solve_np,
                               allgather=allgather_np if world > 1 else None, radius=radius,
                               reference=None if args.host_reference else ref_dev, starts=starts, goals=goals)
        if args.scenario == "lanes":
            loop.shard.set_world(occ, occ_origin)
        for r in range(total_rounds):
            out = loop.step(record=rec if r >= args.first_round else None)
            if r >= args.first_round:
                fails += int((out["status"] == 2).sum())
        if cache:
            np.savez(cache, n_rob=n_rob, N=N, fails=fails, radius=radius,
                     **{k: np.stack([x[k] for x in rec]) for k in keys})

    def stack(key, dtype):
        return torch.from_numpy(np.ascontiguousarray(np.stack([x[key] for x in rec]), dtype=dtype)).to(dev)

    d_agent = stack("agent_id", np.int32)
    d_state, d_ref = stack("state", np.float64), stack("ref", np.float64)
    d_npoly, d_nrows = stack("n_poly", np.int32), stack("n_rows", np.int32)
    d_A, d_b = stack("A", np.float64), stack("b", np.float64)
    d_plans, d_has = stack("plans", np.float64), stack("has_plan", np.uint8)
    rows_mean = float(np.mean([x["n_rows"][x["n_rows"] > 0].mean() for x in rec]))
    # outputs: the shard of the NEXT round's plans buffer, gathered into a full buffer when N > 1
    d_traj = torch.zeros((per, N + 1, 9), dtype=torch.float64, device=dev)
    d_ctrl = torch.zeros((per, N, 3), dtype=torch.float64, device=dev)
    d_used = torch.zeros((per, P), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(per, dtype=torch.int32, device=dev)
    d_obj = torch.zeros(per, dtype=torch.float64, device=dev)
    d_next = torch.zeros((world * per, N + 1, 9), dtype=torch.float64, device=dev)

    def step(r):
        solver.replan_device(d_agent[r], d_state[r], d_ref[r], d_npoly[r], d_nrows[r], d_A[r], d_b[r],
                             d_plans[r], d_has[r], d_traj[:n_local], d_ctrl[:n_local], d_used[:n_local],
                             d_status[:n_local], d_obj[:n_local], stream=stream)
        if world > 1:
            all_gather_dev(d_next, d_traj)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- timed region (the contract)
    for r in range(off - W, off):
        step(r)
    barrier()
    t0 = time.perf_counter()
    for r in range(off, off + K):
        step(r)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if not host_staged else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------------------------------------------------------- second pass: per-launch kernel time
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for k, r in enumerate(range(off, off + K) if not args.no_event_pass else []):
        ev[k][0].record(stream)
        solver.replan_device(d_agent[r], d_state[r], d_ref[r], d_npoly[r], d_nrows[r], d_A[r], d_b[r],
                             d_plans[r], d_has[r], d_traj[:n_local], d_ctrl[:n_local], d_used[:n_local],
                             d_status[:n_local], d_obj[:n_local], stream=stream)
        ev[k][1].record(stream)
    torch.cuda.synchronize()
    kern_ms = (np.array([a.elapsed_time(b) for a, b in ev]) if not args.no_event_pass
               else np.full(K, elapsed / K * 1e3))
    stats = solver.last_stats(n_local)

    # ---------------------------------------------------------------- third pass: the host-buffer entry point
    # hdsm_replan (host pointers: H2D of the inputs, kernel, D2H of the outputs, synchronous) on the same rounds.
    # Reported next to the line, never as `value`.
    host_ms = None
    if world == 1 and not args.no_event_pass:
        t1 = time.perf_counter()
        for r in range(off, off + K):
            solve_np(rec[r], rec[r]["plans"], rec[r]["has_plan"])
        host_ms = (time.perf_counter() - t1) / K * 1e3

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU restatement (oracle/hdsm_oracle.c) on the SAME recorded rounds, farmed over all host cores:
        # one task = one recorded round (its agents solved one after the other by one thread, mirroring the
        # reference's one-process-per-agent, Threads=1 deployment); tasks repeat until ~cpu_seconds of work.
        from concurrent.futures import ThreadPoolExecutor
        from oracle import pyoracle as orc
        orc.lib()
        cores = os.cpu_count() or 1

        def one(x):
            orc.replan(prm, x["agent_id"], x["state"], x["ref"], x["n_poly"], x["n_rows"], x["A"], x["b"],
                       x["plans"], x["has_plan"], n_threads=1)
            return n_local

        sample = rec[off:off + K]
        t1 = time.perf_counter()
        one(sample[0])
        per_task = max(time.perf_counter() - t1, 1e-4)
        reps = int(min(max(1, args.cpu_seconds * cores / (per_task * len(sample))), 200))
        tasks = sample * reps
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=cores) as ex:
            done = sum(ex.map(one, tasks))
        dt_cpu = time.perf_counter() - t1
        cpu = {"value": done / dt_cpu, "unit": "agent-replans/s", "cores": cores, "kind": "port",
               "sample": f"{len(sample)} recorded rounds x {n_local} agents x {reps} repeats of the same "
                         f"workload on the CPU restatement (oracle/hdsm_oracle.c, not Gurobi), "
                         f"one round per thread, {cores} threads",
               "seconds": dt_cpu, "per_core_replans_per_s": done / dt_cpu / min(cores, len(tasks))}

    if rank == 0:
        value = n_rob * K / elapsed
        B = algorithmic_bytes(n_rob, N, P, rows_mean)
        mean_ms = float(kern_ms.mean())
        achieved = B * n_local / (mean_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "agent QP-replans/sec", "value": value, "unit": "agent-replans/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{n_rob} agents circular exchange (R = {radius:g} m), empty env" if args.scenario == "circle"
                                    else f"{n_rob} agents in line formation through a lane forest (corridors by voxel "
                                         f"decomposition, <= {int(max(x['n_rows'].max() for x in rec))} static rows)") + f", H={N}, "
                                   f"poly_hor={P}, closed-loop rounds {args.first_round}..{total_rounds - 1} replayed",
                       "agents": n_rob, "agents_per_gpu": n_local, "horizon": N, "poly_hor": P,
                       "parallelism": f"agents sharded over {world} GPU(s), one all-gather per round"},
            "p50_solve_latency_ms": float(np.percentile(kern_ms, 50)),
            "p95_solve_latency_ms": float(np.percentile(kern_ms, 95)),
            "kernel_ms_mean": mean_ms,
            "host_buffer_path": None if host_ms is None else {
                "ms_per_round": host_ms, "agent_replans_per_s": n_rob / (host_ms * 1e-3),
                "what": "hdsm_replan with host pointers (PCIe-inclusive: H2D inputs, kernel, D2H outputs, sync)"},
            "failed_instances_recorded": fails,
            "solver_stats_last_round": {"qp_iters_max": int(stats["qp_iters"].max()),
                                        "qp_iters_mean": float(stats["qp_iters"].mean()),
                                        "nodes_max": int(stats["nodes"].max()),
                                        "sweeps_max": int(stats["sweeps"].max()),
                                        "staged_rows_max": int(stats["cand"].max())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_replan": B, "kernel": "k_replan"},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""GPU tests of the ABI 1.1 additions: execution knobs in hdsm_params, limit flags and the optional wall-clock budget,
stream ordering on a handle, argument checks of the host-pointer reference entry point, sweep statistics, and the
RCCL publish / exchange entry points (single-rank communicator here; world_size 2 runs in tests/test_distributed.py on gloo
for the host mirror and on the driver's multi-GPU node for RCCL itself)."""
import numpy as np
import pytest

import problems
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config

pytestmark = pytest.mark.gpu
ARG_KEYS = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")


@pytest.fixture(scope="module")
def hdsm():
    from multi_agent_pkgs_amd import lib
    return lib


def test_execution_knobs_never_change_the_answer(hdsm, oracle):
    """presweep / branch_rule / stage_radius / threads / prefilter / duo settings through hdsm_params (not the environment):
    same statuses and trajectories as the oracle for every combination tried."""
    base = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(base, 48, seed=77, spacing=1.2, turn=True, narrow=True)
    args = [sn[k] for k in ARG_KEYS]
    o = oracle.replan(base, *args, n_threads=8)
    combos = [dict(), dict(presweep=1), dict(presweep=2), dict(branch_rule=1), dict(stage_radius=0.2),
              dict(threads_per_instance=64), dict(prefilter_min_agents=1), dict(prefilter_min_agents=-1),
              dict(duo_min_instances=1), dict(duo_min_instances=-1)]
    for kw in combos:
        prm = agile_params(10, max_rows_static=18, **kw)
        g = hdsm.Solver(prm, 48, 48).replan(*args)
        assert (g["status"] == o["status"]).all(), kw
        ok = o["status"] != 2
        assert np.abs(g["traj"] - o["traj"])[ok].max() < 1e-7, kw
    with pytest.raises(hdsm.HdsmError) as e:
        hdsm.Solver(agile_params(10, presweep=3), 4, 4)
    assert e.value.code == hdsm.HDSM_ERR_BAD_ARG
    with pytest.raises(hdsm.HdsmError):
        hdsm.Solver(agile_params(10, time_limit_s=-1.0), 4, 4)


def test_limit_flags_and_time_limit(hdsm):
    """Node / iteration budgets and the optional wall-clock budget (Gurobi TimeLimit, AC:952) are reported per instance through
    hdsm_last_sweep_stats flags; an instance stopped without incumbent leaves its outputs untouched."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 32, seed=5, narrow=True, turn=True, chamfer=True)
    args = [sn[k] for k in ARG_KEYS]
    full = hdsm.Solver(prm, 32, 32)
    g0 = full.replan(*args)
    assert (full.last_sweep_stats(32)["flags"] == 0).all()
    assert g0["qp_iters"].max() > 6
    # iteration budget
    sol = hdsm.Solver(agile_params(10, max_rows_static=18, max_qp_iters=4, warm_start=False), 32, 32)
    out = dict(traj=np.full((32, 11, 9), 7.0), ctrl=np.full((32, 10, 3), 7.0), used=np.zeros((32, 4), np.uint8),
               status=np.zeros(32, np.int32), obj=np.full(32, 7.0))
    g = sol.replan(*args, out=out)
    fl = sol.last_sweep_stats(32)["flags"]
    hit = g0["qp_iters"] > 12
    assert hit.any() and (fl[hit] & hdsm.HDSM_FLAG_ITER_LIMIT).all()
    bad = g["status"] == 2
    assert bad.any() and (g["traj"][bad] == 7.0).all() and (g["obj"][bad] == 7.0).all()
    # node budget
    sol = hdsm.Solver(agile_params(10, max_rows_static=18, max_nodes=1, warm_start=False), 32, 32)
    g = sol.replan(*args)
    fl = sol.last_sweep_stats(32)["flags"]
    deep = g0["nodes"] > 1
    assert deep.any() and (fl[deep] & hdsm.HDSM_FLAG_NODE_LIMIT).all() and not (fl[~deep] & hdsm.HDSM_FLAG_NODE_LIMIT).any()
    # wall-clock budget: 50 ns is spent before the first iteration of anything that has to iterate at all
    sol = hdsm.Solver(agile_params(10, max_rows_static=18, time_limit_s=5e-8, warm_start=False), 32, 32)
    g = sol.replan(*args)
    fl = sol.last_sweep_stats(32)["flags"]
    timed = (fl & hdsm.HDSM_FLAG_TIME_LIMIT) != 0
    assert timed.sum() >= 16 and (g["status"][timed] != 0).all() and (g["status"][~timed] == g0["status"][~timed]).all()
    # a generous budget changes nothing
    sol = hdsm.Solver(agile_params(10, max_rows_static=18, time_limit_s=0.08), 32, 32)
    g = sol.replan(*args)
    assert (g["status"] == g0["status"]).all() and np.abs(g["traj"] - g0["traj"]).max() < 1e-9
    assert (sol.last_sweep_stats(32)["flags"] == 0).all()


def test_launches_on_different_streams_are_ordered(hdsm):
    """Two calls on one handle on different streams, no host synchronisation in between: the second must wait for the first
    (they share branch-and-bound snapshots, warm-start sets and the prefilter records). Results = the serial ones."""
    import torch
    prm = agile_params(10, max_rows_static=18)
    dev = torch.device("cuda", 0)
    sns = [problems.swarm_snapshot(prm, 300, seed=s, spacing=1.1, turn=True, narrow=True) for s in (11, 12)]
    ref = []
    for sn in sns:  # serial, fresh handles
        ref.append(hdsm.Solver(prm, 300, 300).replan(*[sn[k] for k in ARG_KEYS]))
    dt = dict(agent_id=torch.int32, state=torch.float64, ref=torch.float64, n_poly=torch.int32, n_rows=torch.int32,
              A=torch.float64, b=torch.float64, plans=torch.float64, has_plan=torch.uint8)
    for trial in range(3):
        sol = hdsm.Solver(agile_params(10, max_rows_static=18, warm_start=False), 300, 300)
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        outs = []
        torch.cuda.synchronize()
        for sn, st in zip(sns, streams):
            d = {k: torch.from_numpy(np.ascontiguousarray(sn[k])).to(dev).to(dt[k]).contiguous() for k in ARG_KEYS}
            o = dict(traj=torch.zeros((300, 11, 9), dtype=torch.float64, device=dev), ctrl=torch.zeros((300, 10, 3), dtype=torch.float64, device=dev),
                     used=torch.zeros((300, 4), dtype=torch.uint8, device=dev), status=torch.zeros(300, dtype=torch.int32, device=dev),
                     obj=torch.zeros(300, dtype=torch.float64, device=dev))
            outs.append((d, o))
        torch.cuda.synchronize()
        for (d, o), st in zip(outs, streams):
            sol.replan_device(*[d[k] for k in ARG_KEYS], o["traj"], o["ctrl"], o["used"], o["status"], o["obj"], stream=st)
        torch.cuda.synchronize()
        for (d, o), r in zip(outs, ref):
            assert (o["status"].cpu().numpy() == r["status"]).all()
            ok = r["status"] != 2
            assert np.abs(o["traj"].cpu().numpy() - r["traj"])[ok].max() < 1e-7


def test_reference_host_argument_checks_and_sweep_stats(hdsm):
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 16, seed=3)
    sol = hdsm.Solver(prm, 16, 16)
    path = np.zeros((16, 3, 3))
    path[:, 1] = [5.0, 0, 0]
    ids = np.arange(16, dtype=np.int32)
    for bad in (0, 4):
        n_path = np.full(16, 2, np.int32)
        n_path[5] = bad
        with pytest.raises(hdsm.HdsmError) as e:
            sol.reference(agile_ref_config(), ids, path, n_path, sn["plans"], sn["has_plan"])
        assert e.value.code == hdsm.HDSM_ERR_BAD_ARG
    sol.reference(agile_ref_config(), ids, path, np.full(16, 2, np.int32), sn["plans"], sn["has_plan"])
    # sweep statistics: without the prefilter every sweep loads (n_rob) x N positions and no sphere record
    import os
    sol = hdsm.Solver(agile_params(10, max_rows_static=18, prefilter_min_agents=-1), 16, 16)
    g = sol.replan(*[sn[k] for k in ARG_KEYS])
    st = sol.last_sweep_stats(16)
    if "HDSM_BOUNDS_MIN" not in os.environ:   # (the environment overrides hdsm_params: scripts force the prefilter on with it)
        assert (st["sphere_records"] == 0).all() and (st["pairs"] == g["sweeps"] * 16 * 10).all()
    sol2 = hdsm.Solver(agile_params(10, max_rows_static=18, prefilter_min_agents=1), 16, 16)
    g2 = sol2.replan(*[sn[k] for k in ARG_KEYS])
    st2 = sol2.last_sweep_stats(16)
    assert (st2["sphere_records"] == g2["sweeps"] * 16).all() and (st2["pairs"] <= st["pairs"]).all()


def test_publish_and_exchange_single_rank(hdsm):
    """hdsm_publish_device + hdsm_exchange_device on a one-rank RCCL communicator: the all-gather is the identity, the has_plan
    flags travel inside the records (NaN sentinel) and come out as bytes, padding agents are published without a plan."""
    import torch
    prm = agile_params(10, max_rows_static=18)
    dev = torch.device("cuda", 0)
    sol = hdsm.Solver(prm, 8, 8)
    comm = hdsm.Comm(sol, hdsm.comm_unique_id(), 0, 1)
    assert (comm.rank, comm.world) == (0, 1)
    per, n_local = 8, 6
    rng = np.random.default_rng(0)
    traj = torch.from_numpy(rng.normal(size=(per, 11, 9))).to(dev)
    has = torch.tensor([1, 0, 1, 1, 0, 1, 1, 1], dtype=torch.uint8, device=dev)
    local = torch.zeros((per, 11, 9), dtype=torch.float64, device=dev)
    full = torch.zeros((per, 11, 9), dtype=torch.float64, device=dev)
    has_all = torch.full((per,), 9, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()
    comm.publish_device(traj, has, local, n_local=n_local, stream=st)
    comm.exchange_device(local, full, has_all, stream=st)
    torch.cuda.synchronize()
    want = np.array([1, 0, 1, 1, 0, 1, 0, 0], np.uint8)
    assert (has_all.cpu().numpy() == want).all()
    f, t = full.cpu().numpy(), traj.cpu().numpy()
    assert np.array_equal(f[want == 1], t[want == 1])
    assert np.isnan(f[want == 0][:, 0, 0]).all() and (f[want == 0].reshape(4, -1)[:, 1:] == 0).all()
    comm.close()


def test_cpp_sharded_loop_runs_one_rank_through_rccl(tmp_path):
    """examples/sharded_loop.cpp: a C++ host driving reference -> replan -> publish -> exchange on one stream through the C ABI,
    with a (one-rank) RCCL communicator built from a unique id handed over in a file."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "sharded_loop")
    assert os.path.exists(exe), "examples/sharded_loop not built (__graft_entry__.build())"
    r = subprocess.run([exe, "0", "1", str(tmp_path / "uid"), "32", "50"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "agents with a plan 32" in r.stdout


def test_mip_gap_accepts_what_gurobi_accepts(hdsm, oracle):
    """hdsm_params.mip_gap (Gurobi's MIPGap, 1e-4 by default there): with a gap the search may stop on an assignment whose
    objective is within gap * |J| of the optimum and needs no more nodes than the exact search; with gap 0 it is exact."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 64, seed=21, narrow=True, turn=True, chamfer=True, spacing=1.4)
    args = [sn[k] for k in ARG_KEYS]
    exact = hdsm.Solver(agile_params(10, max_rows_static=18, warm_start=False), 64, 64).replan(*args)
    for gap in (1e-4, 5e-2):
        g = hdsm.Solver(agile_params(10, max_rows_static=18, warm_start=False, mip_gap=gap), 64, 64).replan(*args)
        assert (g["status"] == exact["status"]).all()
        ok = exact["status"] == 0
        assert (g["obj"][ok] <= exact["obj"][ok] * (1 + gap) + 1e-6).all() and (g["obj"][ok] >= exact["obj"][ok] - 1e-6).all()
        assert (g["nodes"] <= exact["nodes"]).all()
    assert exact["nodes"].max() > 1


def test_kernel_timing_measures_the_solver_kernel_alone(hdsm):
    """hdsm_set_kernel_timing / hdsm_last_kernel_ms: off by default (asking is an error), a positive duration after a launch,
    below the wall-clock time of the synchronous host-pointer call that contains it."""
    import time
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 64, seed=5, spacing=1.5)
    args = [sn[k] for k in ARG_KEYS]
    sol = hdsm.Solver(prm, 64, 64)
    sol.replan(*args)
    with pytest.raises(hdsm.HdsmError):
        sol.last_kernel_ms()
    sol.set_kernel_timing(True)
    t0 = time.perf_counter()
    sol.replan(*args)
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms = sol.last_kernel_ms()
    assert 0.0 < ms < wall_ms
    sol.set_kernel_timing(False)
    with pytest.raises(hdsm.HdsmError):
        sol.last_kernel_ms()


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (the RCCL exchange with world > 1)")
def test_cpp_sharded_loop_two_ranks_over_rccl(tmp_path):
    """examples/sharded_loop.cpp with TWO ranks, one GPU each: the unique id travels through a file, every round ends with one
    ncclAllGather of the published plans (hdsm_exchange_device); both ranks must see all agents' plans and finish the flight."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "sharded_loop")
    uid = str(tmp_path / "uid")
    procs = []
    for rank in range(2):
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([exe, str(rank), "2", uid, "64", "50"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("agents with a plan 64" in o for o in outs), outs


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs (the RCCL exchange with world > 1)")
def test_bench_self_launches_two_ranks_and_reports_rccl_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts its two ranks itself, rank 0 prints ONE JSON line with
    rccl_ranks = 2 (what ncclCommCount says) and the secondary weak-scaling record; the device-resident loop runs with the
    two-rank communicator as well (hdsm_dswarm_round -> hdsm_exchange_device)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--agents", "128",
                        "--first-round", "20", "--repeats", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    z = json.loads(lines[0])
    assert z["n_gpus"] == 2 and z["rccl_ranks"] == 2 and z["scaling"] == "strong" and z["value"] > 0
    w = z["weak_scaling_record"]
    assert w["scaling"] == "weak" and w["agents_per_gpu"] == 1024 and w["agents"] == 2048 and w["value"] > 0
    # the device-resident loop over the two-rank communicator runs by default (last, under a watchdog that cannot cost the line)
    assert z["device_resident_loop"]["ms_per_round"] > 0 and z["device_resident_loop"]["ranks"] == 2
    # the exchange alone (events around hdsm_exchange_device) and every rank's own time for the timed steps
    assert z["exchange_ms_p50"] > 0 and z["exchange_ms"]["samples"] >= 5 and z["exchange_ms"]["bytes_per_rank"] == 64 * 11 * 9 * 8
    assert len(z["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in z["ms_per_step_per_rank"])
    assert max(z["ms_per_step_per_rank"]) <= z["ms_per_step"] * 1.5


def _run_bench_two_ranks(extra, env_extra=None, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--agents", "128",
                        "--first-round", "20", "--repeats", "1", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu_host_staged_exchange_carries_every_multi_gpu_key():
    """The N > 1 flow of bench.py end to end on a box with ONE GPU (what every box of this build has been): `--gpus 2 --dist-backend
    gloo` starts two ranks that share device 0, shards the swarm, runs the timed region with the barrier / max-over-ranks contract
    and a host-staged all-gather in place of hdsm_exchange_device. Every key the multi-GPU line carries must be there and sane
    (`rccl_ranks` null because no RCCL communicator exists; the exchange figures are labelled as the flow check they are) — so that
    the first run on a multi-GPU node can only differ in the exchange itself."""
    z = _run_bench_two_ranks(["--dist-backend", "gloo"], {"HDSM_BENCH_WEAK_PER_GPU": "64"})
    assert z["n_gpus"] == 2 and z["scaling"] == "strong" and z["value"] > 0 and z["steps"] == 5 and z["warmup"] == 2
    assert z["rccl_ranks"] is None and z["exchange_error"] is None and "gloo" in z["exchange"]
    assert z["config"]["agents"] == 128 and z["config"]["agents_per_gpu"] == 64
    assert len(z["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in z["ms_per_step_per_rank"])
    assert max(z["ms_per_step_per_rank"]) <= z["ms_per_step"] * 1.5
    assert z["exchange_ms_p50"] > 0 and z["exchange_ms"]["backend"] == "gloo-host-staged" and z["exchange_ms"]["bytes_per_rank"] == 64 * 11 * 9 * 8
    w = z["weak_scaling_record"]
    assert w["scaling"] == "weak" and w["agents_per_gpu"] == 64 and w["agents"] == 128 and w["value"] > 0 and z["weak_scaling_value"] == w["value"]
    assert "gloo" in w["exchange"]
    assert z["roofline"]["frac"] > 0 and z["kernel_ms_mean"] > 0 and z["failed_instances_timed_rounds"] >= 0


@pytest.mark.skipif(_device_count() != 1, reason="two ranks on ONE device: the box must have exactly one GPU")
def test_bench_survives_a_communicator_that_cannot_be_created():
    """`python bench.py --gpus 2` on a one-GPU box: both ranks land on device 0 and RCCL refuses the communicator (duplicate GPU,
    ncclInvalidUsage — or never answers: the creation runs under a budget). The ranks agree on the failure, fall back to the
    host-staged exchange and rank 0 still prints the ONE JSON line, with `exchange_error` saying what happened — a dead rank and
    no line is what this used to produce."""
    import torch  # noqa: F401
    z = _run_bench_two_ranks(["--no-weak-record"], {"HDSM_BENCH_COMM_BUDGET_S": "60", "HDSM_BENCH_SAME_DEVICE": "1"})
    assert z["n_gpus"] == 2 and z["value"] > 0 and z["rccl_ranks"] is None
    err = z["exchange_error"]
    assert err is not None and len(err["ranks_failed"]) >= 1 and err["first_error"]
    assert "gloo" in z["exchange"] and z["weak_scaling_record"] is None
    assert len(z["ms_per_step_per_rank"]) == 2


def test_two_ros_nodes_exchange_traj_full_on_the_in_memory_bus():
    """ros/hdsm_agent_node.cpp (compiled against the API-shaped rclcpp of tests/ros_shim): two nodes in one process, each hosting
    four agents of an eight-agent ring, timers fired in lock step; every plan a node knows about the OTHER node's agents arrived
    as a multi_agent_planner_msgs/Trajectory on <topic>_<id>/traj_full (AC:46-48, 610-677)."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "ros_shim", "two_nodes")
    assert os.path.exists(exe), "tests/ros_shim/two_nodes not built (__graft_entry__.build())"
    r = subprocess.run([exe, "15"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"topics (\d+) published (\d+) delivered (\d+)", r.stdout)
    topics, pub, dlv = (int(x) for x in m.groups())
    kinds = {k: int(v) for k, v in re.findall(r"kind (\w+) (\d+)", r.stdout)}
    # 8 agents x 8 topics (AC:42-86: traj_full, traj, traj_ref, path, traj_hist, polyhedra, seeds, position); only traj_full has a
    # subscriber (the other node)
    assert topics == 64 and set(kinds) == {"traj_full", "traj", "traj_ref", "path", "traj_hist", "polyhedra", "seeds", "position"}, r.stdout
    assert kinds["traj_full"] == dlv and kinds["traj_full"] >= 8 * 14, r.stdout
    for k in ("traj_ref", "path", "traj_hist", "polyhedra", "seeds", "position"):
        assert kinds[k] == 8 * 15, (k, r.stdout)
    assert kinds["traj"] >= 8 * 14, r.stdout
    assert "node 0: remote plans known 4 of 4, rounds 15" in r.stdout and "node 1: remote plans known 4 of 4, rounds 15" in r.stdout, r.stdout
    # ComputeYawAngle: agent 0 starts at angle 0 of the ring and flies towards -x (yaw -> +-pi), agent 2 at angle pi/2 towards -y
    y0, y2 = (float(x) for x in re.search(r"yaw agent0 (\S+) agent2 (\S+)", r.stdout).groups())
    assert abs(y0) > 1.0 and -2.2 < y2 < -0.5, r.stdout


def test_staging_overflow_in_a_shared_cu_kernel_is_rescued(hdsm, oracle, monkeypatch):
    """The kernels that share a CU have a fraction of the staging rows of the one-per-CU kernel. A dense H = 10 neighbourhood
    (64 agents 1 m apart, all within reach over the horizon) fills the 256 rows of the four-per-CU kernel (HDSM_QUAD_MIN=1 picks it
    for this small batch) with violated rows alone — 7 of the 64 instances in the CPU execution of the kernel source. Host buffers
    (hdsm_replan): the instances that overflowed are solved again at once with the large staging area — the oracle's answers come
    back. Device pointers (hdsm_replan_device, nothing to wait for): the first launch reports them honestly
    (HDSM_FLAG_STAGING_OVERFLOW, never a wrong optimum); once the handle has seen the flag the following launches carry the rescue
    pass. (Until round 6 this test used a 320-row instantiation of the H = 15 kernel that existed only for it: see hdsm_api.hip,
    CMAX_DUO48.)"""
    import torch
    import problems
    from multi_agent_pkgs_amd.params import make_params
    from test_gpu_fuzz import K
    prm = make_params(n_hor=10, max_rows_static=18, poly_hor=4)
    n_rob = 64
    sn = problems.swarm_snapshot(prm, n_rob, seed=4242, spacing=1.0, narrow=False, turn=False, chamfer=False, absent_frac=0, speed=(0.0, 3.0))
    args = [sn[k] for k in K]
    bounded = prm.copy()
    bounded.max_nodes, bounded.max_qp_iters = 100000, 1000000
    o = oracle.replan(bounded, *args, n_threads=32)
    again = np.where(o["status"] == 1)[0]
    if len(again):
        big = prm.copy()
        big.max_nodes, big.max_qp_iters = 500000, 100000000
        o2 = oracle.replan(big, *[sn[k][again] if k not in ("plans", "has_plan") else sn[k] for k in K], n_threads=32, search=1)
        for k in ("traj", "ctrl", "status", "obj"):
            o[k][again] = o2[k]
    assert (o["status"] != 1).all() and (o["status"] == 0).sum() >= 16
    monkeypatch.setenv("HDSM_DUO_MIN", "1")
    monkeypatch.setenv("HDSM_TRI_MIN", "1")
    monkeypatch.setenv("HDSM_QUAD_MIN", "1")
    sol = hdsm.Solver(prm, n_rob, n_rob)
    sol_dev = hdsm.Solver(prm, n_rob, n_rob)
    for k_ in ("HDSM_DUO_MIN", "HDSM_TRI_MIN", "HDSM_QUAD_MIN"):
        monkeypatch.delenv(k_)
    g = sol.replan(*args)
    assert (g["status"] == o["status"]).all() and (sol.last_sweep_stats(n_rob)["flags"] & 8 == 0).all()
    ok = o["status"] == 0
    assert np.abs(g["traj"] - o["traj"])[ok].max() < 1e-6
    # the asynchronous entry point
    dev = torch.device("cuda", 0)
    dt = dict(agent_id=torch.int32, state=torch.float64, ref=torch.float64, n_poly=torch.int32, n_rows=torch.int32,
              A=torch.float64, b=torch.float64, plans=torch.float64, has_plan=torch.uint8)
    d = {k: torch.from_numpy(np.ascontiguousarray(sn[k])).to(dev).to(dt[k]).contiguous() for k in K}
    N, P = prm.n_hor, prm.poly_hor
    out = dict(traj=torch.zeros((n_rob, N + 1, 9), dtype=torch.float64, device=dev), ctrl=torch.zeros((n_rob, N, 3), dtype=torch.float64, device=dev),
               used=torch.zeros((n_rob, P), dtype=torch.uint8, device=dev), status=torch.zeros(n_rob, dtype=torch.int32, device=dev),
               obj=torch.zeros(n_rob, dtype=torch.float64, device=dev))
    flagged = []
    for call in range(2):
        sol_dev.replan_device(*[d[k] for k in K], out["traj"], out["ctrl"], out["used"], out["status"], out["obj"])
        torch.cuda.synchronize()
        st, fl = out["status"].cpu().numpy(), sol_dev.last_sweep_stats(n_rob)["flags"]
        flagged.append(int((fl & 8 != 0).sum()))
        wrong = (st != o["status"]) & (fl & 8 == 0)
        assert not wrong.any(), (call, np.where(wrong)[0].tolist())             # an answer without the flag is the oracle's
        same = (st == 0) & (o["status"] == 0) & (fl & 8 == 0)
        assert np.abs(out["traj"].cpu().numpy() - o["traj"])[same].max() < 1e-6
    assert flagged[0] > 0 and flagged[1] == 0 and (st == o["status"]).all(), flagged


def test_device_swarm_accepts_an_empty_trailing_shard(hdsm):
    """n_rob = 5 on 4 ranks: ceil split 2 + 2 + 1 + 0 — swarm.shard_range gives rank 3 first_id = n_rob and no agent. The device loop
    must take that shard (it only relays the exchange); a shard that is not a block of the split is still refused."""
    from multi_agent_pkgs_amd import swarm
    prm = agile_params(10, max_rows_static=18)
    cfg = swarm.default_swarm_config()
    n_rob, world = 5, 4
    assert [swarm.shard_range(n_rob, r, world) for r in range(world)] == [(0, 2), (2, 2), (4, 1), (5, 0)]
    for rank in range(world):
        first, n_local = swarm.shard_range(n_rob, rank, world)
        loop = swarm.SwarmLoop(prm, cfg, n_rob, rank=rank, world=world, solve=None, allgather=lambda x: x)
        sol = hdsm.Solver(prm, max(n_local, 1), n_rob)
        dsw = swarm.DeviceSwarm(loop.shard, sol, world_size=world)
        assert dsw.per == 2
        dsw.close()
    bad = swarm.SwarmShard(prm, cfg, n_rob, 1, np.zeros((2, 3)), np.ones((2, 3)))   # first_id 1 is no multiple of per = 2
    with pytest.raises(hdsm.HdsmError) as e:
        swarm.DeviceSwarm(bad, hdsm.Solver(prm, 2, n_rob), world_size=world)
    assert e.value.code == hdsm.HDSM_ERR_BAD_ARG


def test_registered_host_arrays_get_the_same_answers_delivered_by_the_device(hdsm):
    """hdsm_host_register: with the caller's arrays page-locked, hdsm_replan moves the inputs by DMA and the device writes the
    results straight into the output arrays — the same numbers as with pageable arrays, and the arrays of instances without a
    solution are left untouched (include/hdsm.h, HDSM_NO_SOLUTION)."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 96, seed=5, spacing=0.9, turn=True, narrow=True)  # tight: some instances have no solution
    args = [np.ascontiguousarray(sn[k]).copy() for k in ARG_KEYS]
    sol = hdsm.Solver(prm, 96, 96)
    plain = sol.replan(*args)
    assert (plain["status"] == 2).any() and (plain["status"] == 0).any()
    sol.reset_warm_start()
    out = dict(traj=np.full((96, 11, 9), 7.5), ctrl=np.full((96, 10, 3), 7.5), used=np.full((96, prm.poly_hor), 9, dtype=np.uint8),
               status=np.full(96, -1, dtype=np.int32), obj=np.full(96, 7.5))
    pinned = [a for a in args if a.nbytes] + list(out.values())
    for a in pinned:
        hdsm.host_register(a)
    try:
        got = sol.replan(*args, out=out)
    finally:
        for a in pinned:
            hdsm.host_unregister(a)
    assert (got["status"] == plain["status"]).all()
    ok = plain["status"] != 2
    # (two runs of the kernel stage the neighbour rows in a different order: equal to rounding, not bit for bit)
    assert np.abs(got["traj"][ok] - plain["traj"][ok]).max() < 1e-9 and np.abs(got["ctrl"][ok] - plain["ctrl"][ok]).max() < 1e-8
    assert np.array_equal(got["used"][ok], plain["used"][ok]) and np.allclose(got["obj"][ok], plain["obj"][ok], rtol=1e-10, atol=0)
    assert (got["traj"][~ok] == 7.5).all() and (got["ctrl"][~ok] == 7.5).all() and (got["used"][~ok] == 9).all() and (got["obj"][~ok] == 7.5).all()
    with pytest.raises(hdsm.HdsmError):
        hdsm.host_unregister(np.zeros(8))  # never registered


def test_an_array_that_runs_past_its_registered_range_never_reaches_the_fetch_kernel(hdsm):
    """hdsm_replan decided "registered" from the base pointer alone, so an array that ran past its registered range was
    handed to k_fetch / k_deliver, which read or wrote beyond the mapping: a GPU memory fault and a dead process. The library now records
    the ranges registered through hdsm_host_register and takes the kernel paths only for arrays that lie INSIDE one with every byte the
    call touches. Here only the first half of each large array is registered: the call takes the copy path, where the HIP runtime itself
    refuses a copy that straddles a registration (a clean HDSM_ERR_DEVICE, no fault) or performs it; afterwards the process and the
    handle are alive and a call with properly registered arrays uses the direct paths again."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 64, seed=9, spacing=1.1, turn=True)
    args = [np.ascontiguousarray(sn[k]).copy() for k in ARG_KEYS]
    sol = hdsm.Solver(prm, 64, 64)
    plain = sol.replan(*args)
    sol.reset_warm_start()
    out = dict(traj=np.zeros((64, 11, 9)), ctrl=np.zeros((64, 10, 3)), used=np.zeros((64, prm.poly_hor), dtype=np.uint8),
               status=np.full(64, -1, dtype=np.int32), obj=np.zeros(64))
    import ctypes as C
    lib = hdsm.load()
    lib.hdsm_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.hdsm_host_unregister.argtypes = [C.c_void_p]
    halves = [a for a in args + list(out.values()) if a.nbytes >= 8192]   # (whole pages: half of the array is at least a page)
    assert len(halves) >= 4
    for a in halves:
        assert lib.hdsm_host_register(C.c_void_p(a.ctypes.data), a.nbytes // 2) == 0
    ok = plain["status"] != 2
    try:
        try:
            got = sol.replan(*args, out=out)
            assert (got["status"] == plain["status"]).all() and np.abs(got["traj"][ok] - plain["traj"][ok]).max() < 1e-9
        except hdsm.HdsmError as e:
            assert e.code == hdsm.HDSM_ERR_DEVICE      # refused by the runtime's copy, reported — not a fault
    finally:
        for a in halves:
            assert lib.hdsm_host_unregister(C.c_void_p(a.ctypes.data)) == 0
    # the handle is alive: whole arrays registered -> the direct paths, same numbers
    sol.reset_warm_start()
    pinned = [a for a in args if a.nbytes] + list(out.values())
    for a in pinned:
        hdsm.host_register(a)
    try:
        got = sol.replan(*args, out=out)
    finally:
        for a in pinned:
            hdsm.host_unregister(a)
    assert (got["status"] == plain["status"]).all() and np.abs(got["traj"][ok] - plain["traj"][ok]).max() < 1e-9

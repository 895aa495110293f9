"""What one rank of an N-GPU run executes, measured on ONE GPU: the closed loop of 64*N agents is flown and recorded
(bench.py --cache), then only the first 64 instances of every recorded round are replayed and timed (kernel only, no
all-gather). usage: python scripts/gpu_shard_probe.py 2 4 8"""
import os, sys, subprocess, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_agent_pkgs_amd import lib
from multi_agent_pkgs_amd.params import agile_params

for world in [int(x) for x in sys.argv[1:]] or [8]:
    n_rob = 64 * world
    cache = f"{ROOT}/gpurun_out/shard_cache_{n_rob}.npz"
    if os.path.exists(cache):
        os.remove(cache)
    subprocess.check_call([sys.executable, "bench.py", "--no-cpu-baseline", "--agents", str(n_rob), "--cache", cache,
                           "--no-event-pass"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    z = np.load(cache)
    prm = agile_params(10, max_rows_static=18)
    sol = lib.Solver(prm, 64, n_rob)
    dev = torch.device("cuda", 0)
    keys = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b")
    R = z["state"].shape[0]
    d = {k: torch.from_numpy(np.ascontiguousarray(z[k][:, :64])).to(dev) for k in keys}
    plans, has = torch.from_numpy(z["plans"]).to(dev), torch.from_numpy(z["has_plan"]).to(dev)
    out = [torch.zeros((64, 11, 9), dtype=torch.float64, device=dev), torch.zeros((64, 10, 3), dtype=torch.float64, device=dev),
           torch.zeros((64, 4), dtype=torch.uint8, device=dev), torch.zeros(64, dtype=torch.int32, device=dev),
           torch.zeros(64, dtype=torch.float64, device=dev)]
    st = torch.cuda.current_stream()
    def go(r):
        sol.replan_device(*[d[k][r] for k in keys], plans[r], has[r], *out, stream=st)
    for r in range(10):
        go(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(10, R):
        go(r)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (R - 10) * 1e3
    print(f"N={world}: n_rob={n_rob}, 64 instances per rank: {ms:.4f} ms per round (kernel only) -> "
          f"{n_rob / (ms * 1e-3):.0f} agent-replans/s if the all-gather were free")

#!/usr/bin/env python3
"""The instances of a launch timeline (gpurun_out/timeline.bin) by the number of neighbour sweeps they made: how long they are, how many
active-set runs they needed (two runs and three sweeps: a verification sweep found rows — one run and three sweeps: the staging radius
overflowed the staging area and was tightened), what the sweeps and the time outside operations cost.
usage: python scripts/timeline_sweeps.py [timeline.bin] [ROUNDS=20]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "timeline.bin")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
raw = np.fromfile(path, dtype=np.int64)
blocks, i = [], 0
while i < len(raw):
    w = {0x54494D454C494E45: 16, 0x54494D454C494E32: 24, 0x54494D454C494E33: 32}[int(raw[i])]
    n = int(raw[i + 1]); blocks.append(raw[i + 2:i + 2 + n * w].reshape(n, w)); i += 2 + n * w
ev = [b for b in blocks[-rounds:] if b.shape[1] >= 32]
out = {"launches": []}
for li, b in enumerate(ev):
    dur = (b[:, 1] - b[:, 0]) * 0.01
    rec = {"launch": li, "slowest_us": float(dur.max())}
    for sw in sorted(set(int(x) for x in b[:, 6])):
        m = b[:, 6] == sw
        rec["sweeps=%d" % sw] = {"n": int(m.sum()), "dur_mean": round(float(dur[m].mean()), 1), "dur_max": round(float(dur[m].max()), 1), "runs_mean": round(float(b[m, 21].mean()), 2),
                                 "ops_mean": round(float(b[m, 4].mean()), 1), "seq_sweep_us_mean": round(float(b[m, 19].mean() * 0.01), 1), "run_us_mean": round(float(b[m, 20].mean() * 0.01), 1),
                                 "warm_us_mean": round(float(b[m, 17].mean() * 0.01), 1), "staged_mean": round(float(b[m, 7].mean()), 0), "status2": int((b[m, 10] == 2).sum())}
    j = int(np.argmax(dur))
    rec["slowest"] = {"us": float(dur[j]), "ops": int(b[j, 4]), "warm_ops": int(b[j, 18]), "sweeps": int(b[j, 6]), "runs": int(b[j, 21]), "seq_sweep_us": float(b[j, 19] * 0.01), "run_us": float(b[j, 20] * 0.01),
                      "warm_us": float(b[j, 17] * 0.01), "setup_us": float(b[j, 16] * 0.01), "leaf_us": float(b[j, 22] * 0.01), "tail_us": float(b[j, 23] * 0.01), "staged": int(b[j, 7]), "q": int(b[j, 13]), "status": int(b[j, 10])}
    out["launches"].append(rec)
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""List-scheduling replay of recorded k_replan launches (gpurun_out/timeline.bin, written by the -DHDSM_TIMELINE build through
scripts/gpu_timeline.sh): for each of the last ROUNDS launches — the event pass of bench.py, i.e. the timed rounds — the span
seen by the instances, the slowest instance, total instance time / resident slots, and the makespan a greedy dispatcher with
512 slots (2 workgroups per CU) reaches for several launch orders, all keys taken from the PREVIOUS launch:
  index order | previous iterations (round-2 first version) | previous duration, failed first | the shipped key (duration + 9
  units per active row, failed first) | the true durations (ideal longest-first) | lower bound max(slowest, total / slots).

usage: python scripts/replay_launch_order.py [timeline.bin] [ROUNDS=20] [out.json]"""
import heapq
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "timeline.bin")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
raw = np.fromfile(path, dtype=np.int64)
blocks, i = [], 0
while i < len(raw):
    assert raw[i] in (0x54494D454C494E45, 0x54494D454C494E32, 0x54494D454C494E33)  # "TIMELINE": 16 entries per instance, "TIMELIN2": 24, "TIMELIN3": 32
    w = {0x54494D454C494E45: 16, 0x54494D454C494E32: 24, 0x54494D454C494E33: 32}[int(raw[i])]
    n = int(raw[i + 1])
    blocks.append(raw[i + 2:i + 2 + n * w].reshape(n, w))  # begin, end (10 ns ticks), block, hw id, iters, nodes, sweeps, staged, flags, cold, status, pairs, spheres, q
    i += 2 + n * w
ev = blocks[-rounds - 1:]
SLOTS = int(os.environ.get("SLOTS", "768"))  # resident workgroups: 3 per CU (k_replan_tri), 2 per CU = 512 (k_replan_duo)


def makespan(order, dur):
    h = [0.0] * SLOTS
    heapq.heapify(h)
    end = 0.0
    for k in order:
        t = heapq.heappop(h) + dur[k]
        end = max(end, t)
        heapq.heappush(h, t)
    return end


rows = []
for r in range(1, len(ev)):
    cur, prv = ev[r], ev[r - 1]
    dur = (cur[:, 1] - cur[:, 0]) * 0.01
    pd, pst, pit, pq = (prv[:, 1] - prv[:, 0]) * 0.01, prv[:, 10], prv[:, 4], prv[:, 13]
    key_ship = np.where(pst == 2, 255, np.clip(np.floor(pd / 0.64) + 9 * pq, 0, 254))
    rows.append({
        "span_us": float((cur[:, 1].max() - cur[:, 0].min()) * 0.01), "slowest_us": float(dur.max()), "slowest_iters": int(cur[dur.argmax(), 4]),
        "total_over_slots_us": float(dur.sum() / SLOTS), "bound_us": float(max(dur.max(), dur.sum() / SLOTS)),
        "replay_index_order": makespan(np.arange(len(dur)), dur), "replay_prev_iters": makespan(np.argsort(-pit, kind="stable"), dur),
        "replay_prev_duration_failed_first": makespan(np.argsort(-np.where(pst == 2, 1e9, pd), kind="stable"), dur),
        "replay_shipped_key": makespan(np.argsort(-key_ship, kind="stable"), dur), "replay_true_durations": makespan(np.argsort(-dur), dur)})
mean = {k: float(np.mean([x[k] for x in rows])) for k in rows[0]}
out = {"what": __doc__.split("\n\n")[0], "slots": SLOTS, "rounds": len(rows), "mean_over_rounds": mean, "per_round": rows}
print(json.dumps({"mean_over_rounds": {k: round(v, 1) for k, v in mean.items()}}, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

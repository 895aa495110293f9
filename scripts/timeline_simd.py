#!/usr/bin/env python3
"""Where the solver wavefronts of a launch sat (gpurun_out/timeline.bin of scripts/gpu_timeline.sh): per CU, the SIMDs of the
instances' wave 0 (HW_REG_HW_ID read by thread 0), and whether an instance whose wave 0 shares its SIMD with another instance's
wave 0 pays for it (time per regular operation against the number of solver wavefronts on the same SIMD).

usage: python scripts/timeline_simd.py [timeline.bin] [ROUNDS=20]"""
import collections
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
    w = {0x54494D454C494E45: 16, 0x54494D454C494E32: 24, 0x54494D454C494E33: 32}[int(raw[i])]
    n = int(raw[i + 1])
    blocks.append(raw[i + 2:i + 2 + n * w].reshape(n, w))
    i += 2 + n * w
ev = [b for b in blocks[-rounds:] if b.shape[1] >= 24]
pattern = collections.Counter()
pair_pattern = collections.Counter()  # (SIMD of wave 0, SIMD of wave 1) of an instance
share_hist = collections.Counter()
rows = []
for b in ev:
    hw = b[:, 3] & 0xFFFFFFFF
    xcc = (b[:, 3] >> 32) & 15
    simd = (hw >> 4) & 3
    slot = hw & 15
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    dur = (b[:, 1] - b[:, 0]) * 0.01
    wit = b[:, 18]
    reg = b[:, 4] - wit
    run = b[:, 20] * 0.01
    warm = b[:, 17] * 0.01
    if b.shape[1] >= 32:
        for j in range(len(b)):
            pair_pattern[(int(simd[j]), int((b[j, 30] >> 4) & 3), int(slot[j]), int(b[j, 30] & 15))] += 1
    by_cu = collections.defaultdict(list)
    for j in range(len(b)):
        by_cu[int(key[j])].append(j)
    if b is ev[-1]:  # a few CUs of the last launch in full: (workgroup index, SIMD / slot of wave 0, SIMD / slot of wave 1, xcc)
        examples = [[[int(b[j, 2]), int(simd[j]), int(slot[j]), int((b[j, 30] >> 4) & 3), int(b[j, 30] & 15), int(xcc[j])] for j in sorted(js, key=lambda j: b[j, 2])]
                    for k, js in list(sorted(by_cu.items()))[:6]]
    for k, js in by_cu.items():
        pattern[tuple(sorted(int(simd[j]) for j in js))] += 1
        for j in js:
            # solver wavefronts of OTHER instances on this SIMD that are alive for at least half of this instance's life
            t0, t1 = b[j, 0], b[j, 1]
            mates = [m for m in js if m != j and simd[m] == simd[j]]
            ov = sum(max(0, min(t1, b[m, 1]) - max(t0, b[m, 0])) for m in mates) / max(1, t1 - t0)
            share_hist[len(mates)] += 1
            rows.append((len(mates), ov, float(dur[j]), float(reg[j]), float(run[j]), float(wit[j]), float(warm[j]), int(slot[j]), int(simd[j])))
R = np.array(rows)
out = {"launches": len(ev), "cus_seen": int(sum(pattern.values()) / max(1, len(ev))),
       "simd_patterns_of_wave0_per_cu (top 8)": [[list(k), v] for k, v in pattern.most_common(8)],
       "instances_by_solver_mates_on_same_simd": dict(sorted(share_hist.items())),
       "(simd wave0, simd wave1, slot wave0, slot wave1) (top 16)": [[list(k), v] for k, v in pair_pattern.most_common(16)]}
# time per regular operation / per warm operation by the time-weighted number of mates
for name, num, den in (("us_per_regular_op", 4, 3), ("us_per_warm_op", 6, 5)):
    res = {}
    for lo, hi in ((0.0, 0.05), (0.05, 0.5), (0.5, 1.0), (1.0, 1.5), (1.5, 9.0)):
        m = (R[:, 1] >= lo) & (R[:, 1] < hi) & (R[:, den] >= 4)
        if m.sum() > 10:
            A = np.vstack([np.ones(m.sum()), R[m, den]]).T
            c = np.linalg.lstsq(A, R[m, num], rcond=None)[0]
            res["overlap %.2f..%.2f" % (lo, hi)] = {"n": int(m.sum()), "const_us": round(float(c[0]), 3), "per_op_us": round(float(c[1]), 3)}
    out[name + "_by_mate_overlap"] = res
# the slowest instance of every launch: operations against what is left in the working set at the end
B = ev
slow_rows = []
for b in B:
    j = int(np.argmax(b[:, 1] - b[:, 0]))
    slow_rows.append((float((b[j, 1] - b[j, 0]) * 0.01), int(b[j, 4]), int(b[j, 18]), int(b[j, 13]), int(b[j, 6]), int(b[j, 7]), int(b[j, 10])))
out["slowest_instance_per_launch (us, operations, warm operations, final working set, sweeps, staged rows, status)"] = slow_rows
allb = np.vstack(B)
long_ = (allb[:, 1] - allb[:, 0]) * 0.01 > 50.0
out["instances_longer_than_50us"] = {"n": int(long_.sum()), "operations_mean": float(allb[long_, 4].mean()), "warm_operations_mean": float(allb[long_, 18].mean()),
                                      "final_working_set_mean": float(allb[long_, 13].mean()), "status_counts": {int(k): int(v) for k, v in zip(*np.unique(allb[long_, 10], return_counts=True))}}
print(json.dumps(out, indent=1))
for e in examples:
    print(e)

#!/usr/bin/env python3
"""What an instance costs, from a launch timeline (gpurun_out/timeline.bin of scripts/gpu_timeline.sh): least-squares fit of the
instance durations of the last ROUNDS launches against their work counters, and the launch anatomy (span, slowest instance).

usage: python scripts/timeline_fit.py [timeline.bin] [ROUNDS=20] [label]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "timeline.bin")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
label = sys.argv[3] if len(sys.argv) > 3 else ""
raw = np.fromfile(path, dtype=np.int64)
blocks, i = [], 0
while i < len(raw):
    assert raw[i] in (0x54494D454C494E45, 0x54494D454C494E32, 0x54494D454C494E33)  # "TIMELINE": 16 entries per instance, "TIMELIN2": 24, "TIMELIN3": 32
    w = {0x54494D454C494E45: 16, 0x54494D454C494E32: 24, 0x54494D454C494E33: 32}[int(raw[i])]
    n = int(raw[i + 1])
    blocks.append(raw[i + 2:i + 2 + n * w].reshape(n, w))
    i += 2 + n * w
ev = blocks[-rounds:]
X, Y, spans, slow, late = [], [], [], [], []
for b in ev:
    dur = (b[:, 1] - b[:, 0]) * 0.01
    start = (b[:, 0] - b[:, 0].min()) * 0.01
    X.append(np.vstack([np.ones(len(dur)), b[:, 4], b[:, 6], b[:, 11] / 128.0, b[:, 10] == 2]).T.astype(float))
    Y.append(dur)
    spans.append(float((start + dur).max()))
    slow.append(float(dur.max()))
    late.append(float(start[np.argmax(start + dur)]))
X, Y = np.vstack(X), np.concatenate(Y)
coef = np.linalg.lstsq(X, Y, rcond=None)[0]
out = {"label": label, "rounds": len(ev), "fit_us": {"const": coef[0], "per_operation": coef[1], "per_sweep": coef[2], "per_128_pairs": coef[3], "no_solution": coef[4]},
       "fit_rms_us": float(np.sqrt(np.mean((X @ coef - Y) ** 2))), "mean_span_us": float(np.mean(spans)), "mean_slowest_us": float(np.mean(slow)),
       "mean_instance_us": float(Y.mean()), "rounds_set_by_a_late_starter": int(sum(1 for t in late if t > 5.0)), "operations_mean": float(X[:, 1].mean())}
if ev[0].shape[1] >= 24:  # phases of an instance (10-ns ticks): set-up, warm start (+ its operations), sweeps, active-set runs, leaf tests
    B = np.vstack(ev)
    dur = (B[:, 1] - B[:, 0]) * 0.01
    setup, warm, wit, sweep, run, runs, leaf = (B[:, k] * (1.0 if k in (18, 21) else 0.01) for k in (16, 17, 18, 19, 20, 21, 22))
    tail = B[:, 23] * 0.01  # read-back, outputs, warm-start store (part of "rest")
    reg = B[:, 4] - wit  # regular operations
    has_w, has_r = wit > 0, reg > 0
    offs = np.cumsum([0] + [b.shape[0] for b in ev[:-1]])
    slowest = np.array([np.argmax((b[:, 1] - b[:, 0])) + offs[k] for k, b in enumerate(ev)])
    def fit(x, y):
        A = np.vstack([np.ones(len(x)), x]).T
        return [round(float(v), 3) for v in np.linalg.lstsq(A, y, rcond=None)[0]]
    out["phases_us_mean"] = {"setup": setup.mean(), "warm_start": warm.mean(), "sweeps": sweep.mean(), "runs": run.mean(), "leaf": leaf.mean(),
                             "rest": (dur - setup - warm - sweep - run - leaf).mean(), "tail(in rest)": tail.mean(), "total": dur.mean()}
    out["phases_us_slowest"] = {"setup": setup[slowest].mean(), "warm_start": warm[slowest].mean(), "sweeps": sweep[slowest].mean(), "runs": run[slowest].mean(),
                                "leaf": leaf[slowest].mean(), "rest": (dur - setup - warm - sweep - run - leaf)[slowest].mean(), "total": dur[slowest].mean(),
                                "warm_ops": wit[slowest].mean(), "regular_ops": reg[slowest].mean(), "runs_n": runs[slowest].mean()}
    if B.shape[1] >= 32:
        sub = {"setup_stage": B[:, 24], "setup_map": B[:, 25], "setup_sums": B[:, 26], "tail_rollout": B[:, 27], "tail_outputs": B[:, 28], "tail_store": B[:, 29]}
        out["setup_and_tail_us_mean"] = {k: float(v.mean() * 0.01) for k, v in sub.items()}
        out["setup_and_tail_us_slowest"] = {k: float(v[slowest].mean() * 0.01) for k, v in sub.items()}
    out["warm_start_fit_us(const, per_op)"] = fit(wit[has_w], warm[has_w]) if has_w.any() else None
    out["runs_fit_us(const, per_regular_op)"] = fit(reg[has_r], run[has_r]) if has_r.any() else None
    out["run_without_operation_us"] = float(run[~has_r & (runs > 0)].mean()) if (~has_r & (runs > 0)).any() else None
print(json.dumps(out, default=lambda v: round(float(v), 3)))

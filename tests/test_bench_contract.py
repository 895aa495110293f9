"""CPU checks of bench.py's bookkeeping (no GPU): the algorithmic-bytes formula against SURVEY.md section 8d."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey_table():
    b = _bench()
    # cfg1 2.6 KB, cfg2 17.7 KB, cfg3 65.4 KB, cfg4 248.1 KB, cfg5 1.479 MB
    assert b.algorithmic_bytes(1, 10, 4, 6) == 0 + 264 + 552 + 768 + 1036
    assert b.algorithmic_bytes(64, 10, 4, 6) == 15120 + 2620
    assert b.algorithmic_bytes(256, 10, 4, 18) == 61200 + 264 + 552 + 2304 + 1036
    assert b.algorithmic_bytes(1024, 10, 4, 6) == 245520 + 2620
    assert b.algorithmic_bytes(4096, 15, 4, 18) == 1474200 + 384 + 792 + 2304 + 1516
    assert b.HBM_PEAK_GBS == 8000.0


def test_gpus_n_without_a_launcher_starts_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset must not die on an assertion: it re-launches itself under
    torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 — the command line the driver uses."""
    import subprocess
    import sys
    import pytest
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_kernel_source_identity_is_stable():
    b = _bench()
    assert b.kernel_source_sha16() == b.kernel_source_sha16() and len(b.kernel_source_sha16()) == 16

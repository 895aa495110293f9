"""The built gfx950 code objects of libhdsm.so, read on the CPU: no solver kernel may use scratch memory.

Round 3 removed every scratch access from the solver kernels (hoisted addresses spilled across the active-set run), round 5 brought
20 B/lane back unnoticed (a loop-invariant 16-byte zero hoisted in front of the tree loop and spilled: +1.9 MB of HBM writes per
launch). The kernel descriptors in the library say what the hardware will be asked for (`.private_segment_fixed_size`), so the
regression is caught here, without a GPU and without recompiling."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "multi_agent_pkgs_amd", "libhdsm.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernel_descriptors(tmp_path):
    """{kernel name: {field: int}} of every gfx950 code object bundled into the library."""
    objcopy, bundler, readelf = (os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))
    for t in (objcopy, bundler, readelf):
        if not os.path.exists(t):
            pytest.skip("ROCm LLVM tools not installed: " + t)
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([objcopy, "--dump-section", ".hip_fatbin=" + fat, LIB, str(tmp_path / "unused.so")])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    out = {}
    for k, st in enumerate(starts):
        part = str(tmp_path / ("bundle%d.bin" % k))
        open(part, "wb").write(blob[st:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        co = str(tmp_path / ("dev%d.co" % k))
        subprocess.check_call([bundler, "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([readelf, "--notes", co], text=True)
        cur = None
        for ln in notes.splitlines():
            m = re.match(r"\s*\.(\w+):\s+(\S+)\s*$", ln)
            if not m:
                continue
            if m.group(1) == "name" and m.group(2).startswith("_Z"):
                cur = out.setdefault(m.group(2), {})
            elif cur is not None and m.group(1) in ("private_segment_fixed_size", "vgpr_count", "sgpr_spill_count", "vgpr_spill_count",
                                                    "group_segment_fixed_size"):
                cur[m.group(1)] = int(m.group(2))
    return out


def test_no_solver_kernel_uses_scratch(tmp_path):
    if not os.path.exists(LIB):
        pytest.skip("libhdsm.so not built")
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no ROCm toolchain")
    desc = _kernel_descriptors(tmp_path)
    solver = {k: v for k, v in desc.items() if "k_replan" in k}
    # every launch shape DESIGN.md lists must be in the library
    for shape in ("8k_replanILi32ELi1536ELi256", "8k_replanILi48ELi1024ELi256", "12k_replan_duoILi32ELi768ELi256", "12k_replan_triILi32ELi384ELi128",
                  "13k_replan_quadILi32ELi256ELi128", "14k_replan_duo48ILi48E"):
        assert any(shape in k for k in solver), shape
    bad = {k: v["private_segment_fixed_size"] for k, v in solver.items() if v.get("private_segment_fixed_size", 0) != 0}
    assert not bad, "solver kernels with scratch (bytes per lane): %r" % bad
    # the kernels that share a CU must fit the register budget of their residency: 2 waves per SIMD -> 256 registers
    for k, v in solver.items():
        if any(t in k for t in ("k_replan_quad", "k_replan_tri", "k_replan_duoILi32")):
            assert v["vgpr_count"] <= 256, (k, v)
    # the device-corridor and swarm kernels were scratch-free on their hot paths too; record what the others use
    others = {k: v.get("private_segment_fixed_size", 0) for k, v in desc.items() if "k_replan" not in k}
    assert len(others) >= 10

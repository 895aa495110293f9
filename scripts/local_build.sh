#!/bin/bash
# Local gate before any gpurun call: rebuild the product library and the CPU emulation, fail on ANY compiler error.
# usage: bash scripts/local_build.sh && gpurun ...
cd "$(dirname "$0")/.."
before=$(md5sum multi_agent_pkgs_amd/libhdsm.so 2>/dev/null | cut -c1-12)
out=$(make -C multi_agent_pkgs_amd/csrc 2>&1; make -C tests/wave_emu 2>&1)
if echo "$out" | grep -qE "error|Error [0-9]"; then
  echo "$out" | grep -E "error" | head -10
  echo "BUILD FAILED - not going to the GPU"; exit 1
fi
after=$(md5sum multi_agent_pkgs_amd/libhdsm.so | cut -c1-12)
spill=$(make -C multi_agent_pkgs_amd/csrc resource-usage 2>&1 | grep -E "ScratchSize" | grep -vE ": 0 " | wc -l)
echo "build ok: libhdsm.so $before -> $after, kernels with scratch: $spill"

#!/bin/bash
# round 6, first contact of seq_attn16.hip with the hardware: operand layout probe, parity against the oracle, launch times next to
# the two-kernel path (FDMI_FUSE_ATTN=0) and the round-5 kernel (2)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6a
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6a
hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma16_layout_probe.hip -o /tmp/mfma16_probe 2>/dev/null && /tmp/mfma16_probe 2>&1 | tee $O/probe.log
VERBOSE=1 timeout 900 python scripts/round6/sa16_check.py 2>&1 | tail -60 | tee $O/sa16_check.log
for rep in 1 2; do
for fa in 1 2 0; do
  TAG="c2 fuse_attn=$fa" FDMI_FUSE_ATTN=$fa timeout 200 python scripts/kernel_times.py 2>&1 | tail -1
done
done 2>&1 | tee $O/times.log

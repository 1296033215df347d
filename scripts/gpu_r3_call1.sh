#!/bin/bash
# round 3, call 1: co-issue probe 2 + same-box A/B of the three candidate patches (c1, c12, c123, c3)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
hipcc -O3 --offload-arch=gfx950 scripts/probes/coissue2_probe.hip -o /tmp/coissue2_probe && timeout 120 /tmp/coissue2_probe > $OUT/coissue2.log 2>&1
tail -5 $OUT/coissue2.log
VARIANTS=". c1 c12 c123 c3" CHECK_VARIANTS="c123" DEBUG_CFGS="released mini" ROUNDS=2 bash scripts/gpu_ab.sh

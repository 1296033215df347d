#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/probes/mall_probe.hip -o /tmp/mall_probe && timeout 200 /tmp/mall_probe 2>&1 | tee $OUT/mall_probe.log
echo "== done"

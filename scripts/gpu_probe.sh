#!/bin/bash
# micro-probes (scripts/probes/*.hip): compile on the box, run, log under gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for p in ${PROBES:-dma_probe}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/probes/$p.hip -o /tmp/$p || exit 1
  timeout 300 /tmp/$p ${PROBE_ARGS:-} 2>&1 | tee gpurun_out/$p.log
done

#!/bin/bash
# round 4, call B: attention anatomy (stamps, ablations, split variants), ingest probe v2, C3 per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ingest probe"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/probes/ingest_probe.hip -o /tmp/ingest_probe && timeout 200 /tmp/ingest_probe 1.9 2>&1 | tee $OUT/ingest_probe.log
echo "== stamps (default library)"
timeout 300 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/stamps.log; grep -A26 "== attention" $OUT/stamps.log | head -60
echo "== A/B"
: > $OUT/ab.log
for r in 1 2; do
  for v in r3attn . bp57 bp68 bp59 bp48 bp38 nocomp nomem; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib TAG="$v" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
echo "== C3 per-kernel times"
for v in r3attn .; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib TAG="[$v]" timeout 300 python scripts/c3_times.py 2>&1 | grep -E "c3|c2" | tee -a $OUT/c3_times.log
done
echo "== done"

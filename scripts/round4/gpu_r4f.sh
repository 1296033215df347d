#!/bin/bash
# round 4, call F: exclusive-store probe; GEMM TAIL instantiations selected at launch (A/B with the env override)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ingest probe"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/probes/ingest_probe.hip -o /tmp/ingest_probe && timeout 200 /tmp/ingest_probe 1.9 2>&1 | tee $OUT/ingest_probe.log | grep -E "EXCLUSIVE|burst stores|whole k-loop|fragment reads \+ MFMAs"
echo "== pytest selection"
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "packed or c3 or smoke or gemm or graph" 2>&1 | tail -4 | tee $OUT/pytest_sel.log
echo "== A/B (C2: the host picks TAIL = 0)"
: > $OUT/ab.log
for r in 1 2 3 4; do
  for t in default 0 1; do
    if [ $t = default ]; then unset FDMI_GEMM_TAIL; else export FDMI_GEMM_TAIL=$t; fi
    TAG="tail=$t" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
unset FDMI_GEMM_TAIL
echo "== C3 per-kernel times (packed rows: the host picks TAIL = 1)"
: > $OUT/c3_times.log
for r in 1 2 3; do
for t in 0 default; do
  if [ $t = default ]; then unset FDMI_GEMM_TAIL; else export FDMI_GEMM_TAIL=$t; fi
  TAG="[tail=$t]" timeout 300 python scripts/c3_times.py 2>&1 | grep -E "c3|c2" | tee -a $OUT/c3_times.log
done
done
echo "== done"

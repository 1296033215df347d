#!/bin/bash
# round 4, call C: attention v2 (copies / ctx store in H1) + GEMM tail slices: parity, stamps, A/B, C3 per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ingest probe (tail part)"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/probes/ingest_probe.hip -o /tmp/ingest_probe && timeout 200 /tmp/ingest_probe 1.9 2>&1 | tee $OUT/ingest_probe.log | tail -9
echo "== stage dumps (default library)"
for cfg in small mini released ragged long; do timeout 200 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|ctx|h_out|eps"; done | tee $OUT/debug_img.log
echo "== pytest selection"
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "forward_relative_key or long_sequences or c3 or packed or released or smoke or c1 or gemm or wide or graph or strided or streamed" 2>&1 | tail -8 | tee $OUT/pytest_sel.log
echo "== stamps (default library)"
timeout 300 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/stamps.log; grep -A14 "== attention" $OUT/stamps.log | head -16
echo "== A/B"
: > $OUT/ab.log
for r in 1 2 3; do
  for v in r3attn . kearly0 v2bp511 v2bp48 v2bp26; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib TAG="$v" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
echo "== C3 per-kernel times"
: > $OUT/c3_times.log
for r in 1 2; do
for v in notail .; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib TAG="[$v]" timeout 300 python scripts/c3_times.py 2>&1 | grep -E "c3|c2" | tee -a $OUT/c3_times.log
done
done
echo "== done"

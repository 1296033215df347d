#!/bin/bash
# round 4, call A: ingest probe, correctness of the reworked attention kernel, same-box A/B against the round-3 kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | head -3
echo "== ingest probe"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/probes/ingest_probe.hip -o /tmp/ingest_probe && timeout 120 /tmp/ingest_probe 1.9 2>&1 | tee $OUT/ingest_probe.log
echo "== stage dumps (default library)"
for cfg in small mini released ragged long; do timeout 200 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|ctx|h_out|eps"; done | tee $OUT/debug_img.log
echo "== pytest selection"
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "forward_relative_key or long_sequences or c3 or packed or released or smoke or c1" 2>&1 | tail -8 | tee $OUT/pytest_sel.log
echo "== A/B"
: > $OUT/ab.log
for r in 1 2 3; do
  for v in r3attn . bp410 bp58 bp711 lock; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib TAG="$v" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
echo "== A/B at the C3 chunk shapes (padded rows)"
for v in r3attn .; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib TAG="$v L=101" B=512 L=101 timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
done
echo "== done"

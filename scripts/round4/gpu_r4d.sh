#!/bin/bash
# round 4, call D: head sizes 64 / 96 / 128, attention placement variants, GEMM tail with the straight-line loader
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest selection"
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "head_sizes or huggingface or forward_relative_key or long_sequences or packed or c3 or smoke" 2>&1 | tail -12 | tee $OUT/pytest_sel.log
echo "== A/B"
: > $OUT/ab.log
for r in 1 2 3; do
  for v in r3attn v1 v1k v1ilp . v2ilp notail; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib TAG="$v" timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
echo "== stamps (default library)"
timeout 300 python scripts/stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/stamps.log; grep -A14 "== attention" $OUT/stamps.log | head -16
echo "== C3 per-kernel times"
: > $OUT/c3_times.log
for r in 1 2; do
for v in notail .; do
  lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
  FDMI_LIB=$lib TAG="[$v]" timeout 300 python scripts/c3_times.py 2>&1 | grep -E "c3|c2" | tee -a $OUT/c3_times.log
done
done
echo "== done"

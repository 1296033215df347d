#!/bin/bash
# 256-row-tile GEMM (gemm_img256.hip) against the 128-row kernel: stage parity, then per-kernel times, same box
#   MASK=17 CFGS="released mini" ROUNDS=2 bash scripts/gpu_256.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
MASK=${MASK:-17}
: > $OUT/g256.log
for cfg in ${CFGS:-released}; do
  FDMI_GEMM_256=$MASK timeout 300 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|max|Error|error" | sed "s/^/[256 $cfg] /" | tee -a $OUT/g256.log
done
if [ -n "${PYTEST_K:-}" ]; then
  FDMI_GEMM_256=$MASK timeout 900 python -m pytest tests -q -m gpu -x -k "$PYTEST_K" -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT/g256.log
fi
for r in $(seq 1 ${ROUNDS:-2}); do
  FDMI_GEMM_256=0 TAG=base timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/g256.log
  FDMI_GEMM_256=$MASK TAG=t256_$MASK timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/g256.log
done
echo "== done"

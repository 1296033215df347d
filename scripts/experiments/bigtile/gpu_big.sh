#!/bin/bash
# 256-row-tile GEMM (gemm_big.hip) against the 128-row kernel: stage parity, then per-kernel times, same box
#   MASK=17 CFGS="released mini" ROUNDS=2 bash scripts/gpu_big.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
MASK=${MASK:-17}
: > $OUT/big.log
for cfg in ${CFGS:-released}; do
  FDMI_GEMM_BIG=$MASK timeout 300 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|max|h_out|eps|Error|error" | sed "s/^/[big $cfg] /" | tee -a $OUT/big.log
done
if [ -n "${PYTEST_K:-}" ]; then
  FDMI_GEMM_BIG=$MASK timeout 900 python -m pytest tests -q -m gpu -x -k "$PYTEST_K" -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT/big.log
fi
for r in $(seq 1 ${ROUNDS:-2}); do
  FDMI_GEMM_BIG=0 TAG=base timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/big.log
  FDMI_GEMM_BIG=$MASK TAG=big$MASK timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/big.log
done
echo "== done"

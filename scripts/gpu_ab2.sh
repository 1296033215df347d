#!/bin/bash
# parity of the default library (stage check), then per-kernel times of library variants, alternating
#   CFGS="released mini" VARIANTS="base ." ROUNDS=2 bash scripts/gpu_ab2.sh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/ab2.log
for cfg in ${CFGS:-released}; do
  timeout 300 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|max|Error|error" | sed "s/^/[$cfg] /" | tee -a $OUT/ab2.log
done
if [ -n "${PYTEST_K:-}" ]; then
  timeout 900 python -m pytest tests -q -m gpu -x -k "$PYTEST_K" -p no:cacheprovider 2>&1 | tail -5 | tee -a $OUT/ab2.log
fi
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${VARIANTS:-base .}; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib TAG="$v" timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab2.log
  done
done
echo "== done"

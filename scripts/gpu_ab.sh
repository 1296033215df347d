#!/bin/bash
# same-box A/B of library variants (python -m foldingdiff_amd.build <variant> DEFINE...): per-kernel times, alternating
#   VARIANTS="nl1 . nl4" ROUNDS=3 bash scripts/gpu_ab.sh       ("." = the default library)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/ab.log
if [ -n "${DEBUG_CFGS:-}" ]; then
  for v in ${CHECK_VARIANTS:-.}; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    for cfg in $DEBUG_CFGS; do FDMI_LIB=$lib timeout 300 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|h_out|eps" | sed "s/^/[$v] /"; done
  done | tee $OUT/ab_check.log
fi
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in ${VARIANTS:-.}; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib TAG="$v" timeout 300 python scripts/kernel_times.py 2>&1 | tail -1 | tee -a $OUT/ab.log
  done
done
if [ -n "${BENCH_VARIANTS:-}" ]; then
  for v in $BENCH_VARIANTS; do
    lib=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so
    FDMI_LIB=$lib timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'backbones/s', round(r['value'],2), 'ms/step', round(r['whole_step']['ms_per_timestep'],3))" | tee -a $OUT/ab.log
  done
fi
echo "== done"

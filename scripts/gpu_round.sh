#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel-trace stats.  Outputs -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6
nproc; lscpu | grep "Model name"
echo "== build"; python -m foldingdiff_amd.build 2>&1 | tail -2
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider 2>&1 | tail -60 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 2>&1 | tail -3 | tee $OUT/bench.log
if [ "${WITH_UNFUSED:-1}" = "1" ]; then
  timeout 600 python bench.py --steps 1 --warmup 1 --fuse-ln 0 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_unfused.log
fi
if [ "${WITH_PROF:-1}" = "1" ]; then
  echo "== rocprofv3"
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 0 --timesteps ${PROF_T:-200} --no-cpu-baseline > $OLDPWD/$OUT/prof_run.log 2>&1
  cd $OLDPWD
  tail -2 $OUT/prof_run.log
  find $OUT/prof -name "*kernel_stats*" | head; 
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-220
  # keep only the small summaries
  find $OUT/prof -name "*kernel_trace*" -size +20M -delete
fi
echo "== done"

#!/bin/bash
# One gpurun call: parity tests, smoke, bench, rocprofv3 kernel-trace stats.  Outputs -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocminfo"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6
nproc; lscpu | grep "Model name"
# which class of box is this (the pool has two: DESIGN.md section 4)?  clocks / power / temperature before and after the bench
smi() { /opt/rocm/bin/rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "junction|memory\)|sclk|mclk|fclk|Power \(W\)" | sed "s/^/[smi $1] /"; }
smi start | tee $OUT/smi.log
echo "== build"; python -m foldingdiff_amd.build 2>&1 | tail -2
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -q -m gpu --durations=8 -p no:cacheprovider 2>&1 | tail -60 | tee $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 ${BENCH_ARGS:-} 2>&1 | tail -3 | tee $OUT/bench.log
smi after-bench | tee -a $OUT/smi.log
if [ -n "${BENCH2_ARGS:-}" ]; then
  env ${BENCH2_ENV:-X=1} timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 $BENCH2_ARGS 2>&1 | tail -1 | tee $OUT/bench2.log
fi
if [ -n "${BENCH3_ARGS:-}" ]; then
  env ${BENCH3_ENV:-X=1} timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 $BENCH3_ARGS 2>&1 | tail -1 | tee $OUT/bench3.log
fi
if [ "${WITH_PROF:-1}" = "1" ]; then
  echo "== rocprofv3 kernel-trace --stats (same command as the bench, T=${PROF_T:-200})"
  R=$PWD
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 0 --timesteps ${PROF_T:-200} --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths ${BENCH_ARGS:-} > $R/$OUT/prof_run.log 2>&1
  tail -1 $R/$OUT/prof_run.log | cut -c1-400
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 --pmc $c (eager launches, T=4)"
    FDMI_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --timesteps 4 --profile-every 0 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths ${BENCH_ARGS:-} > $R/$OUT/pmc_$c.log 2>&1
    tail -1 $R/$OUT/pmc_$c.log | cut -c1-200
  done
  cd $R
  python scripts/pmc_summary.py $OUT > $OUT/prof_summary.txt 2>&1
  cat $OUT/prof_summary.txt | cut -c1-200
  find $OUT -name "*kernel_trace.csv" -size +8M -delete
  find $OUT -name "*.db" -delete
fi
if [ -n "${PMC_EXTRA:-}" ]; then
  R=$PWD; cd /tmp; i=0
  IFS=';' read -ra SETS <<< "$PMC_EXTRA"
  for set in "${SETS[@]}"; do
    i=$((i+1)); echo "== rocprofv3 --pmc $set"
    FDMI_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pmcx_$i -o pmc -- python $R/bench.py --steps 1 --warmup 0 --timesteps 3 --profile-every 0 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths ${BENCH_ARGS:-} > $R/$OUT/pmcx_$i.log 2>&1
    tail -1 $R/$OUT/pmcx_$i.log | cut -c1-160
  done
  cd $R
  python scripts/pmc_summary.py $OUT > $OUT/prof_summary.txt 2>&1
  grep -E "gemm_img|attn_img" $OUT/prof_summary.txt | grep -v "^#" | cut -c1-170
  find $OUT -name "*kernel_trace.csv" -size +8M -delete
fi
echo "== done"

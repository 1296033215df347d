#!/bin/bash
# rocprofv3 summaries of the exact-fp32 mode (kernel-trace stats + FETCH/WRITE PMC passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m foldingdiff_amd.build 2>&1 | tail -1
python bench.py --steps 1 --warmup 1 --precision f32 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_f32.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_f32 -o bench -- python $R/bench.py --steps 1 --warmup 0 --timesteps 200 --precision f32 --no-cpu-baseline > $R/$OUT/prof_f32_run.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  FDMI_NO_GRAPH=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$OUT/pmcf32_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --timesteps 4 --profile-every 0 --precision f32 --no-cpu-baseline > $R/$OUT/pmcf32_$c.log 2>&1
done
cd $R
rm -rf $OUT/prof $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcx_*
python scripts/pmc_summary.py $OUT > $OUT/prof_summary_f32.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +8M -delete
grep -E "gemm_f32|attn_f32" $OUT/prof_summary_f32.txt | cut -c1-150

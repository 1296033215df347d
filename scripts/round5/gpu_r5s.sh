#!/bin/bash
# round 5: scalar-instruction diet of the fused kernel's stage top (variant presalu = the tree before it): bit-identity, then burst
# and sustained same-box A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5s
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5s
timeout 600 python scripts/round5/sa_check.py 2>&1 | tail -4 | tee $O/sa_check.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "fused_projection or end_to_end or history" 2>&1 | tail -2 | tee $O/pytest_sel.log
for rep in 1 2; do
for v in . presalu; do
  TAG="c2 fused $v" FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done
done 2>&1 | tee $O/ab.log
for rep in 1 2; do
for v in . presalu; do
  FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths --no-traffic 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step']/1000,3), round(d['kernels']['qkv_attention_fused']['avg_ms']*1000,1))"
done
done 2>&1 | tee $O/bench_ab.log

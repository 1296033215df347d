#!/bin/bash
# round 5: the fused kernel with its first / last iteration as code of their own (FDMI_SA_EDGES=1, default) against one loop body
# for everything (variant noedges): bit-identity checks, then same-box timing A/B
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5m
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5m
timeout 600 python scripts/round5/sa_check.py > $O/sa_check.log 2>&1; tail -16 $O/sa_check.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "fused_projection or end_to_end_vs_reference or full_size_c2 or history" 2>&1 | tail -4 | tee $O/pytest_sel.log
for rep in 1 2; do
for v in . noedges; do
  TAG="c2 fused $v" FDMI_FUSE_ATTN=1 FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time.*head_update_wrap=[0-9.]* //"
done
done > $O/ab.log 2>&1
cat $O/ab.log
for v in . noedges; do
  FDMI_LIB=$PWD/foldingdiff_amd/_lib/$v/libfdmi.so timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths --no-traffic 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernels'].get('qkv_attention_fused'))"
done 2>&1 | tee $O/bench_ab.log

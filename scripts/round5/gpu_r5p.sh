#!/bin/bash
# round 5, final tree: fused (default) against the two-kernel path on one box (whole step and the kernels), the sampler bit-identity
# test after the graph-validity fix, and a second sample of the headline
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5p
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5p
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -q -p no:cacheprovider -m gpu -k "fused_projection or end_to_end or history or graph or rccl or wrap_on_the_device" 2>&1 | tail -3 | tee $O/pytest_sel.log
for rep in 1 2; do
for fa in 1 0; do
  TAG="c2 fuse_attn=$fa" FDMI_FUSE_ATTN=$fa timeout 200 python scripts/kernel_times.py 2>&1 | tail -1 | sed "s/embed_ln_time=[0-9.]* //"
done
done 2>&1 | tee $O/ab.log
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths 2>&1 | tail -1 > $O/bench_short.json
python -c "import json; d=json.load(open('$O/bench_short.json')); print(d['value'], d['ms_per_step'], d['kernels']['qkv_attention_fused']['avg_ms'], d['roofline']['traffic'])"

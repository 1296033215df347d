#!/bin/bash
# sustained same-box A/B: the 16-row kernel (FDMI_FUSE_ATTN=1) against the round-5 kernel (2) and the two-kernel path (0)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r6o
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r6o
for rep in 1 2; do
for fa in 1 2; do
  FDMI_FUSE_ATTN=$fa timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths --no-traffic 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_attn=$fa', round(d['value'],2), 'backbones/s', round(d['ms_per_step']/1000,3), 'ms/step', {k: round(v['avg_ms']*1000,1) for k,v in d['kernels'].items()})"
done
done 2>&1 | tee $O/bench_ab.log

#!/bin/bash
# round 5, final tree: SUSTAINED same-box A/B (bench.py, 1 warm-up + 1 timed pass of T = 1000 each) of the fused projection + attention
# kernel against the two-kernel path -- kernel_times.py times a burst of a few steps at boost clocks, the bench seven seconds of load
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5q
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5q
for rep in 1 2; do
for fa in 1 0; do
  FDMI_FUSE_ATTN=$fa timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra --no-user-paths --no-traffic 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('fuse_attn=$fa', round(d['value'],2), 'backbones/s', round(d['ms_per_step']/1000,3), 'ms/step', {n: round(v['avg_ms']*1000,1) for n,v in k.items()})"
done
done 2>&1 | tee $O/bench_ab.log
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tee -a $O/bench_ab.log

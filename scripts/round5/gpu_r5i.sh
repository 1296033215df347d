#!/bin/bash
# round 5: the new GPU tests (fused kernel, wrap KAT, N2 on the device, C2-size step, RCCL skip) + bench with in-run traffic
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r5i
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -x -q -k "fused_projection or wrap_on_the_device or sample_end_to_end or full_size_c2 or two_gpu_rccl or c3_manuscript or reconstruct or strided" > gpurun_out/r5i/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r5i/pytest.log
grep -v "HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r5i/pytest.log | tail -15
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-c5-extra > gpurun_out/r5i/bench.log 2>&1
tail -1 gpurun_out/r5i/bench.log | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('value', r['value'], 'ms/step', r['ms_per_step'])
print('roofline', json.dumps(r['roofline'])[:900])
print('small_batch', r['extras'].get('small_batch'))
print('nerf', r['extras'].get('nerf'))
print('c3', {k: v['value'] for k, v in r['extras']['c3'].items() if isinstance(v, dict) and 'value' in v})
print('host', {k: v['value'] for k, v in r['extras']['host_entry'].items() if isinstance(v, dict)})
"

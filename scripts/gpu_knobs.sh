#!/bin/bash
# A/B of environment knobs on the per-kernel timings of a short bench (T=200): one line per configuration.
# usage: bash scripts/gpu_knobs.sh "X=1" "FDMI_LN_DBG=1" ...      (results -> gpurun_out/knobs.log)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --steps 1 --warmup 1 --timesteps 200 --profile-every 50 --no-cpu-baseline --no-exact-f32 2>/dev/null | tail -1 |
    python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$cfg', 'ms/timestep=%.3f' % j['whole_step']['ms_per_timestep'], ' '.join('%s=%.1f' % (k.replace('gemm_','g_'), v['avg_ms']*1e3) for k,v in j['kernels'].items()))
" | tee -a gpurun_out/knobs.log
done

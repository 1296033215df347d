#!/bin/bash
# quick iteration: stage dumps (correctness), cycle stamps, a 1-pass bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in ${DEBUG_CFGS:-small released long}; do timeout 300 python scripts/debug_img.py $cfg 2>&1 | grep -E "==|ctx|h_out|eps|v  "; done | tee $OUT/debug_img.log
if [ -n "${PYTEST_K:-}" ]; then timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "$PYTEST_K" 2>&1 | tail -8 | tee $OUT/pytest_sel.log; fi
timeout 300 python scripts/stamps.py > $OUT/stamps.log 2>&1
python - <<'PY'
import re
lines=open('gpurun_out/stamps.log').read().splitlines()
sec=None
for ln in lines:
    if ln.startswith('=='): sec=ln; print(ln[:60]); cnt=0; continue
    f=ln.split()
    if 'wave' in ln: print(ln); cnt=0; continue
    if len(f)>=8 and f[0].isdigit():
        cnt+=1
        if cnt in (3,4) or (f[5]!='0' and 'GEMM' in (sec or '')) or ('attention' in (sec or '') and cnt<=6): print(ln)
PY
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 ${BENCH_ARGS:-} 2>&1 | tail -1 > $OUT/bench.log
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench.log').read())
print('backbones/s', round(r['value'],2), 'ms/step', round(r['whole_step']['ms_per_timestep'],3))
for k,v in r['kernels'].items(): print(f"  {k:18s} {v['avg_ms']*1e3:8.1f} us  {v['tflops']:7.1f} TF  {v['gbs']:7.0f} GB/s")
PY
echo "== done"

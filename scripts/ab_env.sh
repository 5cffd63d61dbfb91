#!/bin/bash
# usage (GPU box): scripts/ab_env.sh VAR "v1 v2 ..." [bench args]  -- times bench.py under each value of an environment variable
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > gpurun_out/env_${var}_$v.log 2>&1 || echo FAILED
  python - <<PY
import json
d=json.loads(open('gpurun_out/env_${var}_$v.log').read().strip().split('\n')[-1])
print('$var=$v', 'step', d['ms_per_step'], 'call', d['roofline']['kernel_ms'])
PY
done

#!/bin/bash
# usage (GPU box): scripts/ab.sh [bench args]  -- times every ab/*.so variant of libf3d_hip.so with bench.py.
# The loader is pointed at the variant through F3D_LIBRARY: the product's own library file is never touched.
for lib in ab/*.so; do
  F3D_LIBRARY=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > gpurun_out/ab_$(basename $lib .so).log 2>&1 || echo "FAILED $lib"
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$(basename $lib .so).log').read().strip().split('\n')[-1])
print('$(basename $lib .so)', 'step', d['ms_per_step'], 'call', d['roofline']['kernel_ms'], 'sort', d['roofline']['sort_ms'])
PY
done

#!/bin/bash
# usage (GPU box): scripts/ab_extras.sh  -- streaming-kernel timings of every ab/*.so variant of libf3d_hip.so (loader pointed at the
# variant through F3D_LIBRARY; the product's library file is never touched)
for lib in ab/*.so; do
  F3D_LIBRARY=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-merge > gpurun_out/abx_$(basename $lib .so).log 2>&1 || echo "FAILED $lib"
  python - <<PY
import json
d=json.loads(open('gpurun_out/abx_$(basename $lib .so).log').read().strip().split('\n')[-1])
print('$(basename $lib .so)', d['ms_per_step'], {k.split(' ')[0]: v['ms'] for k, v in d['streaming_kernels'].items() if 'ms' in v})
PY
done

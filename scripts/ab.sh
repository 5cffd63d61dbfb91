#!/bin/bash
# usage (GPU box): scripts/ab.sh [bench args]  -- times every ab/*.so variant of libf3d_hip.so with bench.py
pkg="3d-point-cloud-segmentation-using-2d-img-segmentation_amd/f3d"
cp $pkg/libf3d_hip.so /tmp/libf3d_keep.so
for lib in ab/*.so; do
  cp $lib $pkg/libf3d_hip.so
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > gpurun_out/ab_$(basename $lib .so).log 2>&1 || echo "FAILED $lib"
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$(basename $lib .so).log').read().strip().split('\n')[-1])
print('$(basename $lib .so)', 'step', d['ms_per_step'], 'call', d['roofline']['kernel_ms'], 'sort', d['roofline']['sort_ms'])
PY
done
cp /tmp/libf3d_keep.so $pkg/libf3d_hip.so

#!/bin/bash
# usage (GPU box): scripts/ab_extras.sh  -- streaming-kernel timings of every ab/*.so variant of libf3d_hip.so
pkg="3d-point-cloud-segmentation-using-2d-img-segmentation_amd/f3d"
cp $pkg/libf3d_hip.so /tmp/libf3d_keep.so
for lib in ab/*.so; do
  cp $lib $pkg/libf3d_hip.so
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/abx_$(basename $lib .so).log 2>&1 || echo "FAILED $lib"
  python - <<PY
import json
d=json.loads(open('gpurun_out/abx_$(basename $lib .so).log').read().strip().split('\n')[-1])
print('$(basename $lib .so)', d['ms_per_step'], {k.split(' ')[0]: v['ms'] for k, v in d['streaming_kernels'].items()})
PY
done
cp /tmp/libf3d_keep.so $pkg/libf3d_hip.so

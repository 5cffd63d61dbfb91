#!/bin/bash
# usage: scripts/build_variant.sh <name> [-DFLAG ...]   -- libf3d_hip.so with extra flags for f3d_fuse.hip, written to ab/<name>.so
# (only f3d_fuse.hip is recompiled; scripts/ab.sh then times every ab/*.so on the GPU box)
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/3d-point-cloud-segmentation-using-2d-img-segmentation_amd/csrc"
mkdir -p "$root/ab" "$src/_build_var"
make -s -C "$src" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -I"$root/include" -I"$src" "$@" \
    -c "$src/f3d_fuse.hip" -o "$src/_build_var/$name.o" || exit 1
objs=$(ls "$src"/_build/*.o | grep -v f3d_fuse.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $objs "$src/_build_var/$name.o" -o "$root/ab/$name.so" -Wl,-soname,libf3d_hip.so -Wl,-rpath,/opt/rocm/lib

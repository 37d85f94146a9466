#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: the whole library with extra compile flags into variants/libronk_<name>.so (A/B runs
# through RONK_LIB_PATH; variants/ is git-ignored but travels with gpurun)
set -e
NAME=$1; shift
B=build_$NAME; mkdir -p $B variants
OBJS=""
for f in tile_kernels_wl tile_kernels_r4 tile_kernels_mont tile_kernels_mont_feat tile_kernels_mont_mul tile_kernels tile_kernels_cfg tile_kernels_half tile_kernels_feat tile_kernels_mul small_kernels ronk_core ronk_plan ronk_callers ronk_dist ronk_msm; do
  OBJS="$OBJS $B/$f.o"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude "$@" -c ronkathon_amd/csrc/$f.hip -o $B/$f.o ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libronk_$NAME.so $OBJS
ls -la variants/libronk_$NAME.so

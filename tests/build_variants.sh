#!/bin/bash
# builds experimental variants of the kernel library: tests/build_variants.sh name "-DFOO=1 ..." ...
# REFILL_ONLY=1: the flags only concern refill.hip -- kernels.hip is not rebuilt, the shipped kernels.o is linked
set -e
cd "$(dirname "$0")/../rayca_amd/csrc"
mkdir -p variants
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result"
[ -f host_scene.o ] || g++ $COMMON -c host_scene.cpp -o host_scene.o -pthread
[ -f bvh_build.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON -c bvh_build.hip -o bvh_build.o
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  if [ -n "$REFILL_ONLY" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $flags -c refill.hip -o variants/$name.refill.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC kernels.o variants/$name.refill.o bvh_build.o host_scene.o -o variants/librayca_$name.so -lpthread
    rm variants/$name.refill.o
  else
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $flags -c kernels.hip -o variants/$name.o &
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $flags -c refill.hip -o variants/$name.refill.o &
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/$name.o variants/$name.refill.o bvh_build.o host_scene.o -o variants/librayca_$name.so -lpthread
    rm variants/$name.o variants/$name.refill.o
  fi
  echo built $name
done

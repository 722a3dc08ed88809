#!/bin/bash
# A library variant whose device code goes through an assembly post-pass (tests/asm_fix.py):
#   tests/build_asm_variant.sh name "<extra hipcc flags>" <fix> [<fix> ...]
# device: hipcc -S -> asm_fix.py -> assemble -> lld -> bundle; host: the same source with that bundle embedded.
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../rayca_amd/csrc"
mkdir -p variants
LLVM=/opt/rocm/lib/llvm/bin
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result"
[ -f host_scene.o ] || g++ $COMMON -c host_scene.cpp -o host_scene.o -pthread
[ -f bvh_build.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON -c bvh_build.hip -o bvh_build.o
unit() {
  u=$1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $flags --cuda-device-only -S $u.hip -o variants/$name.$u.s
  python3 ../../tests/asm_fix.py variants/$name.$u.s variants/$name.$u.fixed.s "${@:2}"
  $LLVM/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c variants/$name.$u.fixed.s -o variants/$name.$u.dev.o
  $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o variants/$name.$u.co variants/$name.$u.dev.o
  $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=variants/$name.$u.co -output=variants/$name.$u.hipfb
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $COMMON $flags --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang variants/$name.$u.hipfb -c $u.hip -o variants/$name.$u.o
  rm -f variants/$name.$u.s variants/$name.$u.fixed.s variants/$name.$u.dev.o variants/$name.$u.co variants/$name.$u.hipfb
}
unit kernels "$@" &
unit refill "$@" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/$name.kernels.o variants/$name.refill.o bvh_build.o host_scene.o -o variants/librayca_$name.so -lpthread
rm -f variants/$name.kernels.o variants/$name.refill.o
echo built $name

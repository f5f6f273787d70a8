#!/bin/bash
# usage: tools/build_variant.sh <tag> [-DFLAG ...] — libptamd_<tag>.so = the library with kernels.hip (+ renderer.hip) compiled under extra flags
# (experiments only; select with PTAMD_LIB=platinum_amd/csrc/libptamd_<tag>.so; tools/ab.sh benches them)
set -e
tag=$1; shift
cd "$(dirname "$0")/../platinum_amd/csrc"
make -s all
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c kernels.hip -o /tmp/kernels_$tag.o   # same flags as the Makefile's kernels.o
/opt/rocm/bin/hipcc $F "$@" -c renderer.hip -o /tmp/renderer_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libptamd_$tag.so /tmp/renderer_$tag.o /tmp/kernels_$tag.o multi_device.o lbvh.o scene_io.o scene_gltf.o scene_image.o scene_jpeg.o -lz -ldl -lpthread
echo built libptamd_$tag.so

#!/bin/bash
# usage: build_variant_n16.sh NAME "-DN16_STAMPS ..."  -> csrc/variants/libpfdyn_NAME.so: pf_n16.hip alone recompiled with the extra
# flags (seconds), the other objects of the current build reused
set -e
cd "$(dirname "$0")"
mkdir -p variants
make -s pf_kernels.o pf_train.o pf_rg.o pf_host.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=8 $2 -c pf_n16.hip -o variants/n_$1.o
hipcc -shared -fPIC --offload-arch=gfx950 pf_kernels.o pf_train.o pf_rg.o variants/n_$1.o pf_host.o -o variants/libpfdyn_$1.so
rm -f variants/n_$1.o

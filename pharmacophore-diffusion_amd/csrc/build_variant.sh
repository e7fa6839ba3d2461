#!/bin/bash
# usage: build_variant.sh NAME "-DPF_CH=4 ..."   -> csrc/variants/libpfdyn_NAME.so  (kernel A/B experiments)
set -e
cd "$(dirname "$0")"
mkdir -p variants
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -c pf_kernels.hip -o variants/k_$1.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed $2 -c pf_train.hip -o variants/t_$1.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 $2 -c pf_rg.hip -o variants/r_$1.o
[ -f pf_host.o ] || make pf_host.o
hipcc -shared -fPIC --offload-arch=gfx950 variants/k_$1.o variants/t_$1.o variants/r_$1.o pf_host.o -o variants/libpfdyn_$1.so
rm -f variants/k_$1.o variants/t_$1.o variants/r_$1.o

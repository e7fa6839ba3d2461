#!/bin/bash
# usage: build_variant.sh NAME "-DPFT_STAMPS ..."   -> csrc/variants/libpfdyn_NAME.so  (kernel A/B experiments, stamp builds)
# Every kernel file is recompiled with the extra flags; pf_host.o is reused.
set -e
cd "$(dirname "$0")"
mkdir -p variants
KP="-mllvm -amdgpu-kernarg-preload-count=8"
RG="-mllvm -amdgpu-mfma-vgpr-form=1"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $KP $2 -c pf_kernels.hip -o variants/k_$1.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed $2 -c pf_train.hip -o variants/t_$1.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $RG $KP $2 -c pf_rg.hip -o variants/r_$1.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $RG $KP $2 -c pf_n16.hip -o variants/n_$1.o
[ -f pf_host.o ] || make pf_host.o
hipcc -shared -fPIC --offload-arch=gfx950 variants/k_$1.o variants/t_$1.o variants/r_$1.o variants/n_$1.o pf_host.o -o variants/libpfdyn_$1.so
rm -f variants/k_$1.o variants/t_$1.o variants/r_$1.o variants/n_$1.o

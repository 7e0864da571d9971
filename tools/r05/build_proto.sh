#!/bin/bash
# Prototype build: the library with the experimental entry points of chol_panel.hip (GPRY_PROTOTYPES) -> tools/r05/libgpry_hip_proto.so
# (git-ignored; travels to the GPU box).  Run from the repository root after `make -C gpry_amd/csrc`.
set -e
cd "$(dirname "$0")/../../gpry_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 \
    -DGPRY_PROTOTYPES -c chol_panel.hip -o /tmp/chol_panel_proto.o
OBJS=$(ls *.o | grep -v chol_panel.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/r05/libgpry_hip_proto.so $OBJS /tmp/chol_panel_proto.o -L/opt/rocm/lib -lrccl -ldl -Wl,-rpath,/opt/rocm/lib
echo built tools/r05/libgpry_hip_proto.so

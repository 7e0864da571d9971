#!/bin/bash
# Diagnostic build: the library with s_memtime stamps in the panel step (GPRY_PANEL_STAMPS) -> tools/r05/libgpry_hip_stamps.so
# (git-ignored; travels to the GPU box).  Run from the repository root after `make -C gpry_amd/csrc`.
set -e
cd "$(dirname "$0")/../../gpry_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 \
    -DGPRY_PANEL_STAMPS -c chol_panel.hip -o /tmp/chol_panel_stamps.o
OBJS=$(ls *.o | grep -v chol_panel.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/r05/libgpry_hip_stamps.so $OBJS /tmp/chol_panel_stamps.o -L/opt/rocm/lib -lrccl -ldl -Wl,-rpath,/opt/rocm/lib
echo built tools/r05/libgpry_hip_stamps.so

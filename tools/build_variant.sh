#!/bin/bash
# Build a variant of libvidi_hip.so that differs from the product library in ONE translation unit compiled with extra defines
# (same ABI; for same-box A/Bs through tools/ab_*.py or VIDI_HIP_LIB).  usage: tools/build_variant.sh <name> <source.hip> [-DFOO ...]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -c "from vidi_amd.build import build; build(verbose=False)"
extra=""
case $src in attn_self.hip|attn_cross.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; attn_self_rm.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans";; esac
obj=/tmp/vidi_variant_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra "$@" -c vidi_amd/csrc/$src -o $obj
objs=$(ls vidi_amd/csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vidi_amd/libvidi_hip_${name}.so $objs $obj
echo vidi_amd/libvidi_hip_${name}.so
# a variant that does not load (e.g. a kernel whose host launch stub was dropped) would only show on the GPU box: check here
python -c "import ctypes, sys; ctypes.CDLL(sys.argv[1])" vidi_amd/libvidi_hip_${name}.so

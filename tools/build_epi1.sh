#!/bin/bash
# libvidi_hip_epi1.so: the product library with the GEMM translation units compiled with -DVIDI_W4_EPI2=0 (epilogue form 1 everywhere:
# the round-3 epilogue I/O), for a same-box A/B of the prefill through VIDI_HIP_LIB (tools/gpu_round.sh abepi)
set -e
cd "$(dirname "$0")/.."
python -c "from vidi_amd.build import build; build(verbose=False)"
objs=""
for src in gemm_w4_bf16 gemm_w4_f16 gemm_w4_modes gemm_w4_lnf gemm_w4_patch; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DVIDI_W4_EPI2=0 -c vidi_amd/csrc/$src.hip -o /tmp/vidi_epi1_$src.o &
  objs="$objs /tmp/vidi_epi1_$src.o"
done
wait
rest=$(ls vidi_amd/csrc/build/*.o | grep -v "/gemm_w4_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vidi_amd/libvidi_hip_epi1.so $rest $objs
python -c "import ctypes, sys; ctypes.CDLL(sys.argv[1])" vidi_amd/libvidi_hip_epi1.so
echo vidi_amd/libvidi_hip_epi1.so

#!/bin/bash
# Operand ablations of bf16_dcn_patchw_kernel (csrc/bf16_dcn_patch.hip, -DDPW_ABL=k: timing only, wrong results): one library per k
# next to the product library, then the single-layer bench on each.  Run on the GPU box from the repo root:
#   bash tools/bf16_dcn_patchw_ablate.sh [k ...]      (default: 0 1 2 4 8 6 7)
cd m3dssd_amd/csrc || exit 1
KS=${@:-0 1 2 4 8 6 7}
for k in $KS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -DDPW_ABL=$k -c bf16_dcn_patch.hip -o build/abl_dcn_patch_$k.o || exit 1
  OBJS=$(ls build/*.o | grep -v "trace_\|abl_\|/bf16_dcn_patch.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/libm3dssd_hip_abl$k.so $OBJS build/abl_dcn_patch_$k.o || exit 1
done
cd ../..
for k in $KS; do
  echo "== DPW_ABL=$k"; M3D_HIP_LIB=m3dssd_amd/csrc/build/libm3dssd_hip_abl$k.so python tools/bf16_dcn_bench.py --patchw 1.5 2>&1 | grep -v amdgpu.ids
done

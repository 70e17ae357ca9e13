#!/bin/bash
# Upper bounds for operand-delivery changes of the fp16x2 GEMM (results of the ablated builds are garbage; only the time counts):
# nox = no activation loads / staging stores in the pipelined loop, noxw = additionally no weight LDS-DMA.  Random and all-zero operands.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
AB=$PWD/roreg_amd/csrc/ab
{
for lib in "" $AB/libroreg_hip_gemm_nox.so $AB/libroreg_hip_gemm_noxw.so "" $AB/libroreg_hip_gemm_nox.so $AB/libroreg_hip_gemm_noxw.so; do
  echo "== lib: ${lib:-in-tree}"
  ROREG_HIP_LIB=$lib python tools/gemm_power_probe.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r03/gemm_ablation.log 2>&1
cat gpurun_out/r03/gemm_ablation.log

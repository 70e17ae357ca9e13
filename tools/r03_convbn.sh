#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_fourier.py tests/test_hip_kernels.py tests/test_hip_pipeline.py -m gpu -x -q > gpurun_out/r03/pytest_convbn.log 2>&1; echo "tests rc $?"; tail -2 gpurun_out/r03/pytest_convbn.log
{
for rep in 1 2; do
echo "== base (BN parameters / row scales as global loads inside convert)"; ROREG_HIP_LIB=$PWD/roreg_amd/libroreg_hip_base.so timeout 300 python tools/et_conv_power_probe.py 131072 2>&1 | grep operands
echo "== staged in LDS once per tile"; timeout 300 python tools/et_conv_power_probe.py 131072 2>&1 | grep operands
done
echo "== outputs"; ROREG_HIP_LIB=$PWD/roreg_amd/libroreg_hip_base.so timeout 300 python tools/et_conv_once.py 2>&1 | tail -2; timeout 300 python tools/et_conv_once.py 2>&1 | tail -2
} > gpurun_out/r03/conv_bn_ab.txt 2>&1
cat gpurun_out/r03/conv_bn_ab.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "== checksum word layout"; timeout 300 python tools/gemm_checksum.py 2>&1 | grep checksum
echo "== checksum plane layout + LDS-DMA"; ROREG_GEMM_XDMA=1 timeout 300 python tools/gemm_checksum.py 2>&1 | grep checksum
echo "== probe word layout"; timeout 300 python tools/gemm_power_probe.py 2>&1 | grep operands
echo "== probe XDMA"; ROREG_GEMM_XDMA=1 timeout 300 python tools/gemm_power_probe.py 2>&1 | grep operands
echo "== probe word layout"; timeout 300 python tools/gemm_power_probe.py 2>&1 | grep operands
echo "== probe XDMA"; ROREG_GEMM_XDMA=1 timeout 300 python tools/gemm_power_probe.py 2>&1 | grep operands
} > gpurun_out/r03/xdma_ab.log 2>&1
cat gpurun_out/r03/xdma_ab.log

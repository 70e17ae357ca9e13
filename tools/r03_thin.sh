#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
for B in 61440; do
echo "== default (LDS-DMA for 32->256)"; timeout 300 python tools/time_gemm_small.py $B 2>&1 | grep "B="
echo "== ROREG_GEMM_XDMA=0"; ROREG_GEMM_XDMA=0 timeout 300 python tools/time_gemm_small.py $B 2>&1 | grep "B="
echo "== ROREG_TILE_M128=1"; ROREG_TILE_M128=1 timeout 300 python tools/time_gemm_small.py $B 2>&1 | grep "B="
done
} > gpurun_out/r03/thin_gemm.txt 2>&1
cat gpurun_out/r03/thin_gemm.txt

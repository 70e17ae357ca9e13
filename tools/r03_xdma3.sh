#!/bin/bash
# LDS-DMA GEMM as the default: whole gpu suite (default), fourier + pipeline + fullsize tests with the switch off, bench both ways
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_xdma_default.log 2>&1; echo "suite rc $?"; tail -2 gpurun_out/r03/pytest_gpu_xdma_default.log
ROREG_GEMM_XDMA=0 timeout 1500 python -m pytest tests/test_hip_fourier.py tests/test_hip_pipeline.py tests/test_hip_fullsize.py -m gpu -x -q > gpurun_out/r03/pytest_gpu_xdma_off.log 2>&1; echo "suite(off) rc $?"; tail -2 gpurun_out/r03/pytest_gpu_xdma_off.log
for m in 0 1 0 1; do
  ROREG_GEMM_XDMA=$m timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_v3_xdma$m.json 2> gpurun_out/r03/bench_v3_xdma$m.err; echo "bench xdma=$m rc $?"
  python - <<PY
import json
j = json.load(open('gpurun_out/r03/bench_v3_xdma$m.json'))
print('xdma', $m, j['value'], j['ms_per_step'], j['value_all_local_transforms'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['accuracy']['inlier_ratio'], j['accuracy']['registration_recall_pointdsc'])
PY
done

#!/bin/bash
# LDS-DMA activation path (ROREG_GEMM_XDMA=1) against the default: the new bitwise test, the whole gpu suite under the switch, bench + kernel trace both ways
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_fourier.py -m gpu -x -q -k plane_layout > gpurun_out/r03/pytest_xdma_unit.log 2>&1; echo "unit rc $?"; tail -3 gpurun_out/r03/pytest_xdma_unit.log
ROREG_GEMM_XDMA=1 timeout 1500 python -m pytest tests/test_hip_fourier.py tests/test_hip_pipeline.py -m gpu -x -q > gpurun_out/r03/pytest_gpu_xdma.log 2>&1; echo "suite(xdma) rc $?"; tail -3 gpurun_out/r03/pytest_gpu_xdma.log
for m in 0 1; do
  ROREG_GEMM_XDMA=$m timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_xdma$m.json 2> gpurun_out/r03/bench_xdma$m.err; echo "bench xdma=$m rc $?"
done
for m in 0 1; do
  rm -rf gpurun_out/r03/kt_xdma$m
  ROREG_GEMM_XDMA=$m timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/kt_xdma$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r03/kt_xdma$m.line 2> gpurun_out/r03/kt_xdma$m.err
  db=$(find gpurun_out/r03/kt_xdma$m -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db gpurun_out/r03/kt_xdma$m.txt > /dev/null
  rm -rf gpurun_out/r03/kt_xdma$m
done
python - <<'PY'
import json
for m in (0, 1):
    j = json.load(open(f'gpurun_out/r03/bench_xdma{m}.json'))
    print('xdma', m, j['value'], j['ms_per_step'], j['value_all_local_transforms'], j['roofline']['frac'], j['roofline'].get('achieved'), j['accuracy'])
PY
head -30 gpurun_out/r03/kt_xdma0.txt; head -30 gpurun_out/r03/kt_xdma1.txt

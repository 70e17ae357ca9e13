#!/bin/bash
# ft_nonlin with pair-slot addressing: unit tests, kernel timing, whole gpu suite, bench both GEMM paths
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_fourier.py -m gpu -x -q > gpurun_out/r03/pytest_ft_unit.log 2>&1; echo "fourier tests rc $?"; tail -3 gpurun_out/r03/pytest_ft_unit.log
timeout 600 python tools/time_ft.py > gpurun_out/r03/time_ft.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03/time_ft.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_ft.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r03/pytest_gpu_ft.log
for m in 0 1; do
  ROREG_GEMM_XDMA=$m timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_ft_xdma$m.json 2> gpurun_out/r03/bench_ft_xdma$m.err; echo "bench xdma=$m rc $?"
done
python - <<'PY'
import json
for m in (0, 1):
    j = json.load(open(f'gpurun_out/r03/bench_ft_xdma{m}.json'))
    print('xdma', m, j['value'], j['ms_per_step'], j['value_all_local_transforms'], j['roofline']['frac'], j['accuracy'])
PY

#!/bin/bash
# round 3, kernel A/B: GEMM with 16-byte epilogue stores vs round 2's (bitwise + time), ET stencil conv LDS order, then the suite + bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
OLD=$PWD/roreg_amd/csrc/ab/libroreg_hip_gemm_r02.so
{
echo "== gemm checksum (new)"; python tools/gemm_checksum.py
echo "== gemm checksum (round-2 GEMM)"; ROREG_HIP_LIB=$OLD python tools/gemm_checksum.py
echo "== time_gemm new"; python tools/time_gemm.py 65536 2>&1 | grep -v "max err"
echo "== time_gemm round-2 GEMM"; ROREG_HIP_LIB=$OLD python tools/time_gemm.py 65536 2>&1 | grep -v "max err"
echo "== time_gemm new again"; python tools/time_gemm.py 65536 2>&1 | grep "fp16x2"
echo "== ET conv LDS order"; python tools/et_conv_lds_order.py
} > gpurun_out/r03/kernels_ab.log 2>&1
cat gpurun_out/r03/kernels_ab.log
timeout 1200 python -m pytest tests/test_hip_fourier.py tests/test_hip_kernels.py tests/test_hip_fullsize.py -m gpu -x -q > gpurun_out/r03/pytest_subset.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r03/pytest_subset.log
timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r03/bench_k1.json 2> gpurun_out/r03/bench_k1.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03/bench_k1.json'))
print('value', j['value'], 'all', j['value_all_local_transforms'], 'roofline', j['roofline']['avg_launch_ms'], j['roofline']['frac'], 'accuracy', j['accuracy']['inlier_ratio'], j['accuracy']['registration_recall_pointdsc'], j['accuracy']['rotation_error_deg'])
print('transforms', j['transforms']['ms_per_step'], j['config']['phase_ms_one_synchronised_pass_of_secondary_scene'])
PY

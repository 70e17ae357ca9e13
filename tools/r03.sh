#!/bin/bash
# Round 3 GPU-box recipes, one sub-command per former one-off script (tools/r03_<name>.sh -> bash tools/r03.sh <name>); kept as the record of how
# the profiles/r03_* files were produced (bodies unindented: they contain here-documents).  Round 4 recipes: tools/r04.sh.
cmd=$1; shift
case $cmd in
configs2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1500 python tools/time_configs.py 16 60 > gpurun_out/r03/time_configs.txt 2>&1; echo "configs rc $?"; grep -v amdgpu.ids gpurun_out/r03/time_configs.txt | tail -8
timeout 900 python bench.py --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > gpurun_out/r03/bench_bf16.json 2> gpurun_out/r03/bench_bf16.err; echo "bf16 rc $?"
timeout 900 python bench.py --steps 5 --warmup 2 --pair-lists uniform --no-cpu-baseline --no-secondary > gpurun_out/r03/bench_uniform.json 2> gpurun_out/r03/bench_uniform.err; echo "uniform rc $?"
python - <<'PY'
import json
for n in ('bf16', 'uniform'):
    j = json.load(open(f'gpurun_out/r03/bench_{n}.json'))
    print(n, j['value'], j.get('value_all_local_transforms'), j['accuracy']['registration_recall_pointdsc'], j.get('rd_rm_leg', {}))
PY
;;
final)
# round 3 final validation at HEAD: build check, smoke, full gpu suite, the driver's bench command, 3-rank shared-GPU run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r03/pytest_gpu_final.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err; echo "bench rc $?"
ROREG_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 3 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r03/bench_shared3_final.err | grep "^{" > gpurun_out/r03/bench_shared3_final.json; echo "bench3 rc $?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03/bench_final.json')); k=json.load(open('gpurun_out/r03/bench_shared3_final.json'))
print('final', j['value'], j['ms_per_step'], j['value_all_local_transforms'], j['roofline']['frac'], j['roofline']['traffic'], j['accuracy']['inlier_ratio'], j['accuracy']['registration_recall_pointdsc'])
print('shared3', k['n_gpus'], k['value'], k['config']['eqv_transfers_per_step'], k['config']['cloud_extractions_per_rank'], k['accuracy']==j['accuracy'])
PY
;;
first)
# round 3, first GPU call: the gpu test suite, the default bench line, the multi-rank control flow on one GPU (3 ranks over gloo)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r03/smoke.log 2>&1; echo "smoke rc $?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc $?"; tail -c 600 gpurun_out/r03/bench_default.err
ROREG_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 3 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_shared3.json 2> gpurun_out/r03/bench_shared3.err; echo "bench3 rc $?"; tail -c 600 gpurun_out/r03/bench_shared3.err
head -c 1500 gpurun_out/r03/bench_default.json
;;
ft)
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
;;
ft2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for r in 1 2 3 4 6; do echo "== rounds $r"; ROREG_FT_ROUNDS=$r timeout 600 python tools/time_ft.py 2>&1 | grep "f16x2"; done > gpurun_out/r03/time_ft_rounds.txt 2>&1
cat gpurun_out/r03/time_ft_rounds.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_ft.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r03/pytest_gpu_ft.log
;;
kernels)
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
;;
pipeline)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_pipeline.py -m gpu -x -q > gpurun_out/r03/pytest_pipeline.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest_pipeline.log
for mode in piped plain piped plain; do
  if [ $mode = plain ]; then export ROREG_NO_PIPELINE=1; else unset ROREG_NO_PIPELINE; fi
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r03/bench_$mode.json 2> gpurun_out/r03/bench_$mode.err; echo "bench $mode rc $?"
  python -c "
import json; j=json.load(open('gpurun_out/r03/bench_$mode.json')); print('$mode', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'])"
done
unset ROREG_NO_PIPELINE
bash tools/gaps_of_bench.sh > gpurun_out/r03/gaps_piped.txt 2>&1; head -12 gpurun_out/r03/gaps_piped.txt
;;
profile_all)
# round 3: full gpu suite, default bench line, kernel trace + PMC passes (tools/profile_r03.sh), scaling estimate
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
COMMIT=${1:-unknown}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_full.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest_gpu_full.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_line_steps20_warmup5.json 2> gpurun_out/r03/bench_line.err; echo "bench rc $?"
bash tools/profile_r03.sh 3dmatch-full f16x2 $COMMIT > gpurun_out/r03/profile.log 2>&1; echo "profile rc $?"; tail -5 gpurun_out/r03/profile.log
bash tools/gaps_of_bench.sh > gpurun_out/r03/bench_gpu_idle.txt 2>&1
timeout 900 python tools/scaling_estimate.py 2 banded > gpurun_out/r03/scaling_estimate_banded.txt 2>&1; cat gpurun_out/r03/scaling_estimate_banded.txt | grep "^N="
timeout 900 python tools/scaling_estimate.py 2 uniform > gpurun_out/r03/scaling_estimate_uniform.txt 2>&1; cat gpurun_out/r03/scaling_estimate_uniform.txt | grep "^N="
;;
profile_configs)
# round 3: kernel traces of BASELINE configs[3]/[4]'s path (RD + RM leg) and of the all-local-transforms mode; bench lines for bf16 storage and uniform pair lists
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
cd /tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r03/kt_rdrm
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/kt_rdrm -- python3 tools/time_configs.py 16 60 --only RD+RM+yohoo > gpurun_out/r03/kt_rdrm.log 2>&1
db=$(find gpurun_out/r03/kt_rdrm -name '*.db' | head -1); python3 tools/rocprof_summary.py $db gpurun_out/r03/rd_rm_config_kernel_trace.txt > /dev/null; find gpurun_out/r03/kt_rdrm -name '*.db' -delete
tail -2 gpurun_out/r03/kt_rdrm.log
timeout 900 python tools/time_configs.py 16 60 > gpurun_out/r03/time_configs.txt 2>&1; cat gpurun_out/r03/time_configs.txt | grep "pairs/s"
timeout 600 python bench.py --steps 5 --warmup 2 --dtype bf16 --no-cpu-baseline > gpurun_out/r03/bench_line_bf16.json 2> /dev/null; echo "bf16 rc $?"
timeout 600 python bench.py --steps 5 --warmup 2 --pair-lists uniform --no-cpu-baseline --no-secondary > gpurun_out/r03/bench_line_uniform.json 2> /dev/null; echo "uniform rc $?"
python - <<'PY'
import json
for n in ('bf16','uniform'):
    j=json.load(open(f'gpurun_out/r03/bench_line_{n}.json')); print(n, j['value'], j.get('value_all_local_transforms'), (j.get('accuracy') or {}).get('registration_recall_pointdsc'))
PY
;;
soak)
# round 3 robustness sweep: size fuzz vs the oracle, engine vs stage classes over 48 combinations, non-finite inputs, poisoned workspaces
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1200 python tools/fuzz_sizes_vs_oracle.py > gpurun_out/r03/fuzz_sizes.log 2>&1; echo "fuzz rc $?"; tail -3 gpurun_out/r03/fuzz_sizes.log
timeout 1500 python tools/soak_engine_vs_stages.py > gpurun_out/r03/soak.log 2>&1; echo "soak rc $?"; tail -3 gpurun_out/r03/soak.log
timeout 900 python tools/nan_robustness.py > gpurun_out/r03/nan.log 2>&1; echo "nan rc $?"; tail -3 gpurun_out/r03/nan.log
ROREG_POISON_EMPTY=1 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_poison.log 2>&1; echo "poison pytest rc $?"; tail -3 gpurun_out/r03/pytest_poison.log
timeout 1200 python tools/scaling_estimate.py 3 banded > gpurun_out/r03/scaling_estimate_banded_final.txt 2>&1; grep "^N=" gpurun_out/r03/scaling_estimate_banded_final.txt
timeout 1200 python tools/scaling_estimate.py 3 uniform > gpurun_out/r03/scaling_estimate_uniform_final.txt 2>&1; grep "^N=" gpurun_out/r03/scaling_estimate_uniform_final.txt
;;
soak2)
# robustness sweep with the LDS-DMA GEMM default: size fuzz vs the oracle, engine vs stage classes, non-finite inputs, poisoned workspaces, 3-rank shared-GPU run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 1200 python tools/fuzz_sizes_vs_oracle.py > gpurun_out/r03/fuzz_sizes.log 2>&1; echo "fuzz rc $?"; tail -3 gpurun_out/r03/fuzz_sizes.log
timeout 1500 python tools/soak_engine_vs_stages.py > gpurun_out/r03/soak.log 2>&1; echo "soak rc $?"; tail -3 gpurun_out/r03/soak.log
timeout 900 python tools/nan_robustness.py > gpurun_out/r03/nan.log 2>&1; echo "nan rc $?"; tail -3 gpurun_out/r03/nan.log
ROREG_POISON_EMPTY=1 timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_poison.log 2>&1; echo "poison pytest rc $?"; tail -3 gpurun_out/r03/pytest_poison.log
ROREG_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 3 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/r03/bench_shared3_final.err | grep "^{" > gpurun_out/r03/bench_shared3_final.json; echo "bench3 rc $?"
python -c "
import json
k=json.load(open('gpurun_out/r03/bench_shared3_final.json'))
print('shared3', k['n_gpus'], k['value'], k['config']['eqv_transfers_per_step'], k['config']['cloud_extractions_per_rank'], k['accuracy'])"
;;
split)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_pipeline.py -m gpu -x -q -k "run_plan or run_scenes" > gpurun_out/r03/pytest_split.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest_split.log
timeout 1200 python tools/scaling_estimate.py 3 banded > gpurun_out/r03/scaling_estimate_banded_v2.txt 2>&1; grep "^N=" gpurun_out/r03/scaling_estimate_banded_v2.txt
timeout 1200 python tools/scaling_estimate.py 3 uniform > gpurun_out/r03/scaling_estimate_uniform_v2.txt 2>&1; grep "^N=" gpurun_out/r03/scaling_estimate_uniform_v2.txt
;;
split2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_pipeline.py -m gpu -x -q -k "run_plan or run_scenes" 2>&1 | tail -2
for mj in 4 1 2 4 1; do
  echo "== ROREG_PLAN_MIN_JOBS=$mj"; ROREG_PLAN_MIN_JOBS=$mj timeout 900 python tools/scaling_estimate.py 3 banded --worlds=1,8 2>&1 | grep "^N="
done
;;
thin)
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
;;
xdma)
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
;;
xdma2)
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
;;
xdma3)
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
;;
*) echo "usage: bash tools/r03.sh {configs2 | final | first | ft | ft2 | kernels | pipeline | profile_all | profile_configs | soak | soak2 | split | split2 | thin | xdma | xdma2 | xdma3}"; exit 2 ;;
esac

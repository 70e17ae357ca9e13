#!/bin/bash
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

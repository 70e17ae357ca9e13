#!/bin/bash
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

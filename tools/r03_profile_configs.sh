#!/bin/bash
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

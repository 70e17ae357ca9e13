#!/bin/bash
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

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_pipeline.py -m gpu -x -q -k "run_plan or run_scenes" > gpurun_out/r03/pytest_split.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest_split.log
timeout 1200 python tools/scaling_estimate.py 3 banded > gpurun_out/r03/scaling_estimate_banded_v2.txt 2>&1; grep "^N=" gpurun_out/r03/scaling_estimate_banded_v2.txt
timeout 1200 python tools/scaling_estimate.py 3 uniform > gpurun_out/r03/scaling_estimate_uniform_v2.txt 2>&1; grep "^N=" gpurun_out/r03/scaling_estimate_uniform_v2.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_pipeline.py -m gpu -x -q -k "run_plan or run_scenes" 2>&1 | tail -2
for mj in 4 1 2 4 1; do
  echo "== ROREG_PLAN_MIN_JOBS=$mj"; ROREG_PLAN_MIN_JOBS=$mj timeout 900 python tools/scaling_estimate.py 3 banded --worlds=1,8 2>&1 | grep "^N="
done

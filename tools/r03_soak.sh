#!/bin/bash
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

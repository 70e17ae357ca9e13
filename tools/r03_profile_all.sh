#!/bin/bash
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

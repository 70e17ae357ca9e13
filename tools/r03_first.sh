#!/bin/bash
# round 3, first GPU call: the gpu test suite, the default bench line, the multi-rank control flow on one GPU (3 ranks over gloo)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r03/smoke.log 2>&1; echo "smoke rc $?"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench rc $?"; tail -c 600 gpurun_out/r03/bench_default.err
ROREG_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 3 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03/bench_shared3.json 2> gpurun_out/r03/bench_shared3.err; echo "bench3 rc $?"; tail -c 600 gpurun_out/r03/bench_shared3.err
head -c 1500 gpurun_out/r03/bench_default.json

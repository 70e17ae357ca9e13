#!/bin/bash
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

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for r in 1 2 3 4 6; do echo "== rounds $r"; ROREG_FT_ROUNDS=$r timeout 600 python tools/time_ft.py 2>&1 | grep "f16x2"; done > gpurun_out/r03/time_ft_rounds.txt 2>&1
cat gpurun_out/r03/time_ft_rounds.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_ft.log 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r03/pytest_gpu_ft.log

# usage (on the GPU box): bash tools/probe/gpu_tests.sh <tag> [pytest args...]   -> gpurun_out/r06/tests_<tag>.log
tag=$1; shift
mkdir -p gpurun_out/r06
timeout 3000 python3 -m pytest tests -m gpu -q "$@" > gpurun_out/r06/tests_$tag.log 2>&1
echo "rc=$?"
tail -25 gpurun_out/r06/tests_$tag.log

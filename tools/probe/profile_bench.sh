# usage (on the GPU box): bash tools/probe/profile_bench.sh <tag> [bench args...]   -> gpurun_out/kt_<tag>.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/kt_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > gpurun_out/kt_$tag.log 2>&1
db=$(find gpurun_out/kt_$tag -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db gpurun_out/kt_$tag.txt > /dev/null
tail -1 gpurun_out/kt_$tag.log | cut -c1-300
find gpurun_out/kt_$tag -name '*.db' -delete

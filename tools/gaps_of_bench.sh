# GPU idle time of bench.py's steps from a kernel trace (tools/rocprof_gaps.py).  Usage on the GPU box: bash tools/gaps_of_bench.sh [bench args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace -d /tmp/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" > /tmp/kt_line.json 2> /tmp/kt.err
db=$(find /tmp/kt -name '*.db' | head -1)
python3 tools/rocprof_gaps.py $db ${GAP_SECONDS:-4.8}

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_sk
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_sk -- python3 tools/time_sinkhorn.py > gpurun_out/kt_sk.log 2>&1
db=$(find gpurun_out/kt_sk -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db gpurun_out/kt_sk.txt > /dev/null
find gpurun_out/kt_sk -name '*.db' -delete
head -12 gpurun_out/kt_sk.txt | cut -c1-150

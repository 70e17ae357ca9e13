cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_rm
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_rm -- python3 tools/time_match_ot.py > gpurun_out/kt_rm.log 2>&1
db=$(find gpurun_out/kt_rm -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db gpurun_out/kt_rm.txt > /dev/null
find gpurun_out/kt_rm -name '*.db' -delete
tail -2 gpurun_out/kt_rm.log

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_rdc
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_rdc -- python3 tools/time_configs.py 16 60 --only RD+mutual+yohoo > gpurun_out/kt_rdc.log 2>&1
db=$(find gpurun_out/kt_rdc -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db gpurun_out/kt_rdc.txt > /dev/null
find gpurun_out/kt_rdc -name '*.db' -delete
tail -2 gpurun_out/kt_rdc.log

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_rmc
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_rmc -- python3 tools/time_configs.py 16 60 --only RD+RM+yohoo > gpurun_out/kt_rmc.log 2>&1
db=$(find gpurun_out/kt_rmc -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db gpurun_out/kt_rmc.txt > /dev/null
find gpurun_out/kt_rmc -name '*.db' -delete
tail -2 gpurun_out/kt_rmc.log

# usage (GPU box): bash tools/probe/kt_rd_rm_quick.sh <tag>   -> gpurun_out/r06/rd_rm_quick_<tag>_kernel_trace.txt (env is inherited: A/B switches)
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT; rm -rf $OUT/ktq_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/ktq_$tag -- python3 bench.py --pipeline rd_rm --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $OUT/ktq_$tag.log 2>&1
db=$(find $OUT/ktq_$tag -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db $OUT/rd_rm_quick_${tag}_kernel_trace.txt > /dev/null
find $OUT/ktq_$tag -name '*.db' -delete
tail -1 $OUT/ktq_$tag.log | cut -c1-100
head -24 $OUT/rd_rm_quick_${tag}_kernel_trace.txt | cut -c1-150

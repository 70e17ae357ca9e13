# usage (on the GPU box): bash tools/probe/profile_rd_rm_full.sh [tag]  -> gpurun_out/r06/{rd_rm_k5000_<tag>_kernel_trace.txt, _gaps.txt, _line.json}
# RoReg's own pipeline (--RD --RM --ET yohoo --keynum 5000) on the full benchmark shape as the timed region itself
tag=${1:-a}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
OUT=gpurun_out/r06; mkdir -p $OUT
ARGS="--pipeline rd_rm --steps 2 --warmup 1 --no-secondary --no-cpu-baseline"
timeout 900 python3 bench.py $ARGS > $OUT/rd_rm_k5000_${tag}_line.json 2> $OUT/rd_rm_k5000_${tag}_line.err
cut -c1-400 $OUT/rd_rm_k5000_${tag}_line.json
rm -rf $OUT/kt_rd_rm_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_rd_rm_$tag -- python3 bench.py $ARGS > $OUT/kt_rd_rm_$tag.log 2>&1
db=$(find $OUT/kt_rd_rm_$tag -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db $OUT/rd_rm_k5000_${tag}_kernel_trace.txt > /dev/null
python3 tools/rocprof_gaps.py $db 8 detail > $OUT/rd_rm_k5000_${tag}_gaps.txt
find $OUT/kt_rd_rm_$tag -name '*.db' -delete
tail -1 $OUT/kt_rd_rm_$tag.log | cut -c1-300
head -12 $OUT/rd_rm_k5000_${tag}_kernel_trace.txt
cat $OUT/rd_rm_k5000_${tag}_gaps.txt

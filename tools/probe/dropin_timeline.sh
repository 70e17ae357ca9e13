# usage (on the GPU box): bash tools/probe/dropin_timeline.sh [tag]   -> gpurun_out/r06/dropin_timeline_<tag>.txt
tag=${1:-a}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 600 python3 tools/probe/dropin_timeline.py > $OUT/dropin_timeline_${tag}_plain.txt 2>&1
grep -v Registering $OUT/dropin_timeline_${tag}_plain.txt | tail -12
rm -rf $OUT/tl_$tag
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/tl_$tag -- python3 tools/probe/dropin_timeline.py > $OUT/tl_$tag.log 2>&1
grep "pairs/s\|host marks" $OUT/tl_$tag.log
db=$(find $OUT/tl_$tag -name '*.db' | head -1)
python3 tools/probe/dropin_timeline_report.py $db > $OUT/dropin_timeline_$tag.txt 2>&1
find $OUT/tl_$tag -name '*.db' -delete
cat $OUT/dropin_timeline_$tag.txt

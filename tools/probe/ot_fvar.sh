# usage: bash tools/probe/ot_fvar.sh <size> <pairs> <fvar...>   -> of_iter_kernel's mean duration per ROREG_OT_FVAR
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp ROREG_TS_RECOMPUTE_ONLY=1; OUT=gpurun_out/r05; mkdir -p $OUT
size=$1; pairs=$2; shift 2
for v in "$@"; do
  export ROREG_OT_FVAR=$v
  rm -rf $OUT/kt_x; rocprofv3 --kernel-trace --stats -d $OUT/kt_x -- python3 tools/time_sinkhorn.py $size $pairs > $OUT/kt_x.log 2>&1
  db=$(find $OUT/kt_x -name '*.db' | head -1); python3 tools/rocprof_summary.py $db $OUT/x.txt > /dev/null
  echo "== FVAR $v"; grep -E "of_iter" $OUT/x.txt | cut -c1-60
done
rm -rf $OUT/kt_x

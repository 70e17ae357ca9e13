cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp ROREG_TS_RECOMPUTE_ONLY=1; OUT=gpurun_out/r05; mkdir -p $OUT
for lag in 0 1 2 3 4 6 8; do
  export ROREG_OT_LAG=$lag
  rm -rf $OUT/kt_x; rocprofv3 --kernel-trace --stats -d $OUT/kt_x -- python3 tools/time_sinkhorn.py 2500 100 > $OUT/kt_x.log 2>&1
  db=$(find $OUT/kt_x -name '*.db' | head -1); python3 tools/rocprof_summary.py $db $OUT/x.txt > /dev/null
  echo "== LAG $lag"; grep -E "of_iter" $OUT/x.txt | cut -c1-60
done
rm -rf $OUT/kt_x

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp ROREG_TS_RECOMPUTE_ONLY=1; OUT=gpurun_out/r05; mkdir -p $OUT
for cfg in "2500x2500 100" "2500x1250 100" "2500x630 100" "2500x300 100" "2500x5000 50"; do
  set -- $cfg
  rm -rf $OUT/kt_x; rocprofv3 --kernel-trace --stats -d $OUT/kt_x -- python3 tools/time_sinkhorn.py $1 $2 > $OUT/kt_x.log 2>&1
  db=$(find $OUT/kt_x -name '*.db' | head -1); python3 tools/rocprof_summary.py $db $OUT/x.txt > /dev/null
  echo "== $cfg"; grep -E "of_iter|of_update_cols" $OUT/x.txt | cut -c1-120; tail -1 $OUT/kt_x.log
done
rm -rf $OUT/kt_x

# usage (on the GPU box): bash tools/probe/profile_all.sh   -> gpurun_out/final/{kt_<mode>.txt, pmc.txt, pmc.json}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
for mode in f16x2 bf16x3 f32; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --gemm $mode > $OUT/kt_$mode.log 2>&1
  db=$(find $OUT/kt_$mode -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db $OUT/kt_$mode.txt > /dev/null
  find $OUT/kt_$mode -name '*.db' -delete
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$mode/g$i -- python3 bench.py $ARGS --gemm $mode > $OUT/pmc_${mode}_g$i.log 2>&1
  done
done
python3 tools/pmc_summary.py $OUT/pmc.txt $OUT/pmc.json f16x2=$OUT/pmc_f16x2 bf16x3=$OUT/pmc_bf16x3 f32=$OUT/pmc_f32
find $OUT -name '*agent_info.csv' -delete
du -sh $OUT

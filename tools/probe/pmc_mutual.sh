# usage (GPU box): bash tools/probe/pmc_mutual_full.sh  -> gpurun_out/r06/mutual_pmc.txt
# PMC passes (one counter group each; FETCH_SIZE and WRITE_SIZE alone) over RoReg's own pipeline on the full benchmark shape:
#   python3 bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT; rm -rf $OUT/pmc_mutual
ARGS="--steps 1 --warmup 0 --no-secondary --no-cpu-baseline"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_mutual/g$i -- python3 bench.py $ARGS > $OUT/pmc_mutual_g$i.log 2>&1
  tail -1 $OUT/pmc_mutual_g$i.log | cut -c1-120
done
python3 tools/pmc_kernel_means.py $OUT/pmc_mutual > $OUT/mutual_pmc.txt
rm -rf $OUT/pmc_mutual
python3 tools/pmc_table.py $OUT/mutual_pmc.txt 26

# usage (GPU box): bash tools/probe/pmc_des2r.sh  -> gpurun_out/r06/des2r_pmc.txt   (the R_indicator kernel alone, both forms, one counter group per pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
for split in 0 1 2; do
  export ROREG_DES2R_SPLIT=$split
  rm -rf $OUT/pmc_des2r_$split
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_des2r_$split/g$i -- python3 tools/probe/des2r_ab.py > $OUT/pmc_des2r_${split}_g$i.log 2>&1
    tail -1 $OUT/pmc_des2r_${split}_g$i.log | cut -c1-160
  done
  python3 tools/pmc_kernel_means.py $OUT/pmc_des2r_$split > $OUT/des2r_pmc_$split.txt
  rm -rf $OUT/pmc_des2r_$split
  grep -A22 "^des2r" $OUT/des2r_pmc_$split.txt | head -24
done

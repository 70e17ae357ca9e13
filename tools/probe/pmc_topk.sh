# usage (GPU box): bash tools/probe/pmc_topk.sh  -> gpurun_out/r06/topk_pmc_<packed>.txt   (the top-k search alone, plain and packed insertion)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
for packed in 0 1; do
  export ROREG_TOPK_PACKED=$packed
  rm -rf $OUT/pmc_topk_$packed
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_topk_$packed/g$i -- python3 tools/probe/topk_ab.py > $OUT/pmc_topk_${packed}_g$i.log 2>&1
  done
  python3 tools/pmc_kernel_means.py $OUT/pmc_topk_$packed > $OUT/topk_pmc_$packed.txt
  rm -rf $OUT/pmc_topk_$packed
  grep -A24 "^topk_dot_mfma_kernel" $OUT/topk_pmc_$packed.txt | head -56
done

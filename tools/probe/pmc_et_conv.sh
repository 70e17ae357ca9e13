# kernel trace + counter passes of the ET trunk convolution alone (${ET_PROBE:-tools/probe/et_conv_once.py}): gpurun_out/r05/et_conv_pmc.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=gpurun_out/r05; mkdir -p $OUT
rm -rf $OUT/pmc_et; i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_et/g$i -- python3 ${ET_PROBE:-tools/probe/et_conv_once.py} > $OUT/pmc_et_g$i.log 2>&1 || tail -3 $OUT/pmc_et_g$i.log
done
python3 tools/pmc_kernel_means.py $OUT/pmc_et > $OUT/et_conv_pmc.txt
grep -A24 "group_conv_split" $OUT/et_conv_pmc.txt | head -40
rm -rf $OUT/pmc_et

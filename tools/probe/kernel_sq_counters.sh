# SQ wait / issue / LDS counters of one kernel (three rocprofv3 --pmc passes).  Usage on the GPU box:
#   bash tools/probe/kernel_sq_counters.sh <kernel-name-substring> <script.py> [args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=$1; shift
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p1 -o p --output-format csv -- python3 "$@" > /tmp/l1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d /tmp/p2 -o p --output-format csv -- python3 "$@" > /tmp/l2 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_LDS_ADDR_CONFLICT -d /tmp/p3 -o p --output-format csv -- python3 "$@" > /tmp/l3 2>&1
python3 tools/pmc_kernel_means.py /tmp/p1 $K; python3 tools/pmc_kernel_means.py /tmp/p2 $K; python3 tools/pmc_kernel_means.py /tmp/p3 $K

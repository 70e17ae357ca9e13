# SQ wait / issue / LDS counters of the big fp16 x 2 irrep GEMM (three rocprofv3 --pmc passes).  Usage on the GPU box: bash tools/gemm_sq_counters.sh [random|zeros]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=${1:-random}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p1 -o p --output-format csv -- python3 tools/gemm_once.py $K > /tmp/l1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d /tmp/p2 -o p --output-format csv -- python3 tools/gemm_once.py $K > /tmp/l2 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES -d /tmp/p3 -o p --output-format csv -- python3 tools/gemm_once.py $K > /tmp/l3 2>&1
tail -2 /tmp/l1 /tmp/l2 /tmp/l3 | grep -i "error\|invalid" | head
python3 tools/pmc_kernel_means.py /tmp/p1; python3 tools/pmc_kernel_means.py /tmp/p2; python3 tools/pmc_kernel_means.py /tmp/p3

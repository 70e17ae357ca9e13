run() { echo "--- $*"; env "$@" timeout 300 python tools/gemm_persist_ab.py 2 2>&1 | grep -E "B=61440|B=14464|B=1000|ALL|MISM"; }
run ROREG_X=1

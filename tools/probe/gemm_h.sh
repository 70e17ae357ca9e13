export FORMS=0,2 SHAPES="512,256,61440,0,0;512,256,61440,1,0;256,512,61440,1,0;256,512,61440,0,0;512,512,61440,0,0"
echo "--- zeros"; ROREG_AB_ZEROS=1 timeout 300 python tools/gemm_persist_ab.py 2 2>&1 | grep -E "B=61440"
echo "--- random"; timeout 300 python tools/gemm_persist_ab.py 2 2>&1 | grep -E "B=61440"

for which in prev new prev new; do
  if [ $which = prev ]; then export ROREG_HIP_LIB=$PWD/tools/probe/libroreg_hip_prev.so; else unset ROREG_HIP_LIB; fi
  echo "== $which"; timeout 300 python tools/et_conv_power_probe.py 131072 2>&1 | grep operands
done
unset ROREG_HIP_LIB
timeout 1500 python -m pytest tests/test_hip_fourier.py tests/test_hip_rm.py tests/test_hip_kernels.py -m gpu -x -q 2>&1 | tail -3

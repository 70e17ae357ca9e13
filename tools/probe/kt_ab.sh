#!/bin/bash
# Kernel-trace A/B of two builds of the library on ONE box (the bench's step time moves by +-5 ms between runs; a kernel's own time does not):
#   cp roreg_amd/libroreg_hip.so roreg_amd/libroreg_hip_base.so; <edit a kernel>; make -C roreg_amd/csrc
#   gpurun -- bash tools/probe/kt_ab.sh '<kernel name regex>'
# prints, for new / base / new / base, the matching kernels' (total ms, average us) over the three steps of a short bench and the trace's total.
pat=${1:-irrep_gemm_xdma_kernel<1>}
export TMPDIR=/tmp
for v in new base new base; do
  L=$PWD/roreg_amd/libroreg_hip.so; [ $v = base ] && L=$PWD/roreg_amd/libroreg_hip_base.so
  rm -rf /tmp/kt_$v
  ROREG_HIP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/kt_$v.json 2>/dev/null
  db=$(find /tmp/kt_$v -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db /tmp/kt_$v.txt > /dev/null
  echo "$v: $(grep -E "$pat" /tmp/kt_$v.txt | head -3 | awk '{printf "%s ms (%s us)  ", $2, $3}') | all kernels $(head -1 /tmp/kt_$v.txt | awk '{print $(NF-1)}') ms"
done

#!/bin/bash
# Kernel-trace A/B of one environment switch on ONE box:  gpurun -- bash tools/probe/kt_env_ab.sh VAR '<kernel name regex>'
# prints, for VAR=1 / 0 / 1 / 0, the matching kernels' (total ms, average us) over the three steps of a short bench, the trace's total and the bench value.
var=$1; pat=${2:-irrep_gemm_xdma}
export TMPDIR=/tmp
for v in 1 0 1 0; do
  rm -rf /tmp/kt_$v
  env $var=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/kt_$v.json 2>/dev/null
  db=$(find /tmp/kt_$v -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db /tmp/kt_$v.txt > /dev/null
  echo "$var=$v: $(grep -E "$pat" /tmp/kt_$v.txt | head -3 | awk '{printf "%s ms (%s us)  ", $2, $3}') | all kernels $(head -1 /tmp/kt_$v.txt | awk '{print $(NF-1)}') ms | $(python3 -c "import json;r=json.loads(open('/tmp/kt_$v.json').read().strip().splitlines()[-1]);print(round(r['value'],1), r['accuracy']['registration_recall_pointdsc'])")"
done

# same-box A/B of two builds of the library: bash tools/probe/bench_ab_lib.sh <prev.so> [reps]
PREV=$1; REPS=${2:-2}
for r in $(seq $REPS); do
  for which in prev new; do
    if [ $which = prev ]; then export ROREG_HIP_LIB=$PWD/$PREV; else unset ROREG_HIP_LIB; fi
    timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$which', j['value'], 'ms/step', j['ms_per_step'], 'contract', j.get('value_contract_complete'), 'frac', j['roofline']['frac'], 'avg ms', j['roofline']['avg_launch_ms'])"
  done
done

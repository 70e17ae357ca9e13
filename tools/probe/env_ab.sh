# usage: bash tools/probe/env_ab.sh VAR valueA valueB [pipeline]   -- alternating bench runs on one box under two values of an environment switch
VAR=$1; A=$2; B=$3; PL=${4:-rd_rm}
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v timeout 600 python3 bench.py --pipeline $PL --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$VAR=$v $PL', 'pairs/s %.1f' % j['value'], 'ms/step %.1f' % j['ms_per_step'])"
  done
done

# A/B of run_scenes' first-round prefill (ROREG_PIPELINE_PREFILL = extra scenes started in the first round), both pipelines, alternating runs on one box
for rep in 1 2; do
  for pf in 0 1 2; do
    for pl in rd_rm mutual; do
      ROREG_PIPELINE_PREFILL=$pf timeout 600 python3 bench.py --pipeline $pl --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('prefill $pf $pl', 'pairs/s %.1f' % j['value'], 'ms/step %.1f' % j['ms_per_step'])"
    done
  done
done

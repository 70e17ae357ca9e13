# A/B of the chain kernel's tile height (ROREG_LC2_WAVES=4: 128-row tiles, 4 wavefronts; default: 256-row tiles / 8 wavefronts where they fit) on the rd_rm pipeline
for rep in 1 2; do
  for w in 4 0; do
    ROREG_LC2_WAVES=$w timeout 600 python3 bench.py --pipeline rd_rm --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('lc2 waves $w (0 = default)', 'pairs/s %.1f' % j['value'], 'ms/step %.1f' % j['ms_per_step'])"
  done
done

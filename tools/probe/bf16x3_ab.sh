# Same-box A/B of the strictly 24-bit matrix-core mode (--gemm bf16x3) across rounds: each tree's OWN bench.py and library, alternating.
# Trees: roreg_amd/csrc/ab/r04, r05 = `git archive` of the commits that closed rounds 4 and 5 (400cf12, f38ac11), built in place; head = this tree.
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for t in r04 r05 head; do
    if [ $t = head ]; then d=$R; else d=$R/roreg_amd/csrc/ab/$t; fi
    cd $d
    timeout 600 python3 bench.py --gemm bf16x3 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('$t', 'pairs/s %.1f' % j['value'], 'ms/step %.1f' % j['ms_per_step'], 'gemm avg ms %.3f' % r['avg_launch_ms'], 'launches', r['launches'], 'transforms ms/step', (j.get('transforms') or {}).get('ms_per_step'))"
  done
done

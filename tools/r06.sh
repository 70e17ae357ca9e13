#!/bin/bash
# Round 6 GPU-box recipes (run through gpurun from the repo root): bash tools/r06.sh <cmd> [args]
#   final                  -> the driver's bench command (python bench.py --steps 20 --warmup 5) -> gpurun_out/r06/bench_final.json + a summary
#   rdrm [tag]             -> kernel trace + gap profile of RoReg's own pipeline on the full shape (tools/probe/profile_rd_rm_full.sh)
#   pmc_rdrm               -> PMC passes of the same (tools/probe/pmc_rd_rm_full.sh)
#   mutual [tag]           -> kernel trace + idle gaps of the driver's (mutual-matcher) timed region
#   forced                 -> the RCCL path on one GPU (--force-collectives)
#   tests [pytest args]    -> the GPU test suite -> gpurun_out/r06/tests_<tag>.log
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06; mkdir -p $OUT
export TMPDIR=/tmp
cmd=$1; shift
case $cmd in
final)
  timeout 2400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc $?"
  python3 - "$OUT/bench_final.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); c = j['config']
print('value', j['value'], 'ms/step', j['ms_per_step'], 'contract', j.get('value_contract_complete'), 'bf16x3', j.get('value_bf16x3'), j.get('value_contract_complete_bf16x3'),
      'frac', j['roofline']['frac'], 'avg ms', j['roofline']['avg_launch_ms'], 'ft ms', (j.get('transforms') or {}).get('ms_per_step'))
print('rd_rm_k5000', j.get('value_rd_rm_k5000'), 'contract', j.get('value_rd_rm_k5000_contract_complete'), 'all iterations', j.get('value_rd_rm_k5000_all_sinkhorn_iterations'),
      'roofline_rd_rm frac', (j.get('roofline_rd_rm') or {}).get('frac'))
print('sinkhorn', json.dumps(c['rd_rm_k5000']['sinkhorn']))
print('stages', c['rd_rm_k5000']['stage_ms_one_synchronised_pass_rank0'])
d = c.get('dropin_leg') or {}
print('dropin', {k: (round(v['pairs_per_s'], 1) if isinstance(v, dict) and 'pairs_per_s' in v else None) for k, v in d.items() if k in ('stages', 'engine', 'engine_yohoc', 'engine_rd_rm', 'no_files')}, d.get('files'))
y = c.get('yohoc_leg') or {}
print('yohoc', {k: round(v['pairs_per_s'], 1) for k, v in y.items() if isinstance(v, dict)})
print('legs', c.get('rd_rm_leg_pairs_per_s'), c.get('rd_rm_leg_pairs_per_s_bf16'), c.get('rd_rm_leg_k5000_pairs_per_s'), c.get('rd_rm_leg_k5000_sinkhorn_ms_per_pair'), 'fmr/ir/rr', c.get('fmr'), c.get('ir'), c.get('rr'))
print('cpu_baseline', (j.get('cpu_baseline') or {}).get('value'), (j.get('cpu_baseline') or {}).get('cores'))
PY
  ;;
rdrm) bash tools/probe/profile_rd_rm_full.sh ${1:-final} ;;
pmc_rdrm) bash tools/probe/pmc_rd_rm_full.sh ;;
mutual)
  tag=${1:-final}
  rm -rf $OUT/kt_b
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_line_under_kernel_trace_$tag.json 2> $OUT/kt_b.err
  db=$(find $OUT/kt_b -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db $OUT/bench_kernel_trace_$tag.txt > /dev/null
  python3 tools/rocprof_gaps.py $db 3.0 > $OUT/bench_gpu_idle_$tag.txt 2>&1; head -6 $OUT/bench_gpu_idle_$tag.txt
  rm -rf $OUT/kt_b
  head -14 $OUT/bench_kernel_trace_$tag.txt | cut -c1-170 ;;
forced)
  timeout 900 python3 bench.py --force-collectives --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_forced.json 2> $OUT/bench_forced.err
  python3 -c "import json; j=json.load(open('$OUT/bench_forced.json')); print('forced', j['value'], j['config'].get('forced_collectives'), j['config'].get('backend'), j['config'].get('eqv_bytes_moved_per_step'))" ;;
tests)
  bash tools/probe/gpu_tests.sh ${TAG:-final} "$@" ;;
esac

#!/bin/bash
# Round 5 GPU-box recipes (run through gpurun from the repo root): bash tools/r05.sh <cmd> [args]
#   tests "<pytest -k expr>" [file]   -> gpurun_out/r05/pytest_<tag>.log
#   ot5000                            -> Sinkhorn at 5000 x 5000: cooperating workgroups vs the two-pass form (timing + kernel trace)
#   bench [tag] [extra args]          -> the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05; mkdir -p $OUT
export TMPDIR=/tmp
cmd=$1; shift
kt() {   # kt <name> <python args...>: kernel trace + stats summary -> $OUT/<name>_kernel_trace.txt
  local name=$1; shift
  rm -rf $OUT/kt_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -- python3 "$@" > $OUT/kt_$name.log 2>&1
  local db=$(find $OUT/kt_$name -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db $OUT/${name}_kernel_trace.txt > /dev/null
  rm -rf $OUT/kt_$name
}
pmc() {  # pmc <name> <python args...>: one pass per counter group -> $OUT/<name>_pmc.txt (per kernel+grid means)
  local name=$1; shift
  rm -rf $OUT/pmc_$name; local i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name/g$i -- python3 "$@" > $OUT/pmc_${name}_g$i.log 2>&1
  done
  python3 tools/pmc_kernel_means.py $OUT/pmc_$name > $OUT/${name}_pmc.txt
  rm -rf $OUT/pmc_$name
}
case $cmd in
tests)
  tag=$(echo "$1" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40)
  timeout 3000 python -m pytest ${2:-tests} -m gpu -x -q -s -k "$1" > $OUT/pytest_$tag.log 2>&1; echo "pytest rc $?"; grep -E "^\[|passed|failed|Error|error" $OUT/pytest_$tag.log | tail -40 ;;
ot5000)
  echo "--- cooperating workgroups"; timeout 600 python tools/time_sinkhorn.py 5000 1 4 13 26 2>&1 | tail -4
  echo "--- two-pass form (ROREG_OT_COOP=0)"; ROREG_OT_COOP=0 timeout 600 python tools/time_sinkhorn.py 5000 13 26 2>&1 | tail -2
  echo "--- 2500"; timeout 600 python tools/time_sinkhorn.py 2500 30 100 2>&1 | tail -2
  kt sinkhorn_5000 tools/time_sinkhorn.py 5000 26
  grep -E "of_|calls" $OUT/sinkhorn_5000_kernel_trace.txt | cut -c1-160 ;;
bench)
  tag=${1:-head}; shift
  timeout 1500 python bench.py --steps 20 --warmup 5 "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "bench rc $?"
  python3 - "$OUT/bench_$tag.json" <<'PY'
import json, sys
j=json.load(open(sys.argv[1])); c=j['config']
print('value', j['value'], 'ms/step', j['ms_per_step'], 'contract', j.get('value_contract_complete'), 'bf16x3', j.get('value_contract_complete_bf16x3'), j.get('value_bf16x3'), 'frac', j['roofline']['frac'], 'avg ms', j['roofline']['avg_launch_ms'],
      'ft ms', (j.get('transforms') or {}).get('ms_per_step'), 'rd_rm', c.get('rd_rm_leg_pairs_per_s'), c.get('rd_rm_leg_pairs_per_s_bf16'), 'k5000', c.get('rd_rm_leg_k5000_pairs_per_s'), c.get('rd_rm_leg_k5000_sinkhorn_ms_per_pair'), 'rr', c.get('rr'))
PY
  ;;
*) echo "unknown command $cmd"; exit 2 ;;
esac

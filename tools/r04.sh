#!/bin/bash
# Round 4 GPU-box recipes (one script, sub-commands; run through gpurun from the repo root):
#   bash tools/r04.sh tests "<pytest -k expression>"      -> gpurun_out/r04/pytest_<tag>.log
#   bash tools/r04.sh forced                               -> bench.py --force-collectives (RCCL path on one GPU) -> gpurun_out/r04/bench_forced.json
#   bash tools/r04.sh config4 [tag]                        -> kernel trace + PMC passes of BASELINE configs[3]'s path (RD + RM + yohoo, 16 clouds / 60 pairs)
#   bash tools/r04.sh bench [tag] [extra bench args]       -> the driver's bench command -> gpurun_out/r04/bench_<tag>.json
#   bash tools/r04.sh profile [tag]                        -> kernel trace + PMC passes of the bench command (profiles/r04_*)
#   bash tools/r04.sh ot [tag]                             -> kernel trace + PMC passes of 30 stacked 2500 x 2500 Sinkhorn problems (tools/time_sinkhorn.py)
#   bash tools/r04.sh final                                -> build, smoke, full gpu suite, bench, forced bench
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04; mkdir -p $OUT
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
ot)
  tag=${1:-ot}
  kt sinkhorn_$tag tools/time_sinkhorn.py 2500 30
  grep -E "of_|ot_fused|calls" $OUT/sinkhorn_${tag}_kernel_trace.txt | cut -c1-150
  rm -rf $OUT/pmc_sk; i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_sk/g$i -- python3 tools/time_sinkhorn.py 2500 30 > $OUT/pmc_sk_g$i.log 2>&1 || tail -3 $OUT/pmc_sk_g$i.log
  done
  python3 tools/pmc_kernel_means.py $OUT/pmc_sk > $OUT/sinkhorn_${tag}_pmc.txt
  grep -E "of_iter|of_update|ot_fused" -A14 $OUT/sinkhorn_${tag}_pmc.txt | head -80
  rm -rf $OUT/pmc_sk ;;
tests)
  tag=$(echo "$1" | tr -c 'a-zA-Z0-9' '_' | cut -c1-40)
  timeout 2400 python -m pytest tests -m gpu -x -q -k "$1" > $OUT/pytest_$tag.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_$tag.log ;;
forced)
  timeout 900 python bench.py --force-collectives --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_forced.json 2> $OUT/bench_forced.err; echo "forced rc $?"; tail -3 $OUT/bench_forced.err
  timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_plain.json 2> $OUT/bench_plain.err; echo "plain rc $?"
  python3 - <<'PY'
import json
f=json.load(open('gpurun_out/r04/bench_forced.json')); p=json.load(open('gpurun_out/r04/bench_plain.json'))
print('forced', f['value'], f['config']['forced_collectives'], f['config']['eqv_bytes_moved_per_step'], f['config']['eqv_transfers_per_step'], f['config'].get('backend'))
print('plain ', p['value'], p['config']['forced_collectives'], p['config']['eqv_bytes_moved_per_step'])
PY
  ;;
config4)
  tag=${1:-head}
  kt rd_rm_config_$tag tools/time_configs.py 16 60 --only RD+RM+yohoo; tail -2 $OUT/kt_rd_rm_config_$tag.log
  pmc rd_rm_config_$tag tools/time_configs.py 16 60 --only RD+RM+yohoo
  head -60 $OUT/rd_rm_config_${tag}_kernel_trace.txt | cut -c1-200 ;;
bench)
  tag=${1:-head}; shift
  timeout 1200 python bench.py --steps 20 --warmup 5 "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "bench rc $?"
  python3 - "$OUT/bench_$tag.json" <<'PY'
import json, sys
j=json.load(open(sys.argv[1])); c=j['config']
print('value', j['value'], 'ms/step', j['ms_per_step'], 'all', j.get('value_all_local_transforms'), 'bf16x3', j.get('value_bf16x3'), 'frac', j['roofline']['frac'], 'avg ms', j['roofline']['avg_launch_ms'],
      'ft ms', (j.get('transforms') or {}).get('ms_per_step'), 'rd_rm', c.get('rd_rm_leg_pairs_per_s'), c.get('rd_rm_leg_pairs_per_s_bf16'), 'rr', c.get('rr'))
PY
  ;;
profile)
  tag=${1:-head}
  ARGS="--no-cpu-baseline --no-secondary"
  rm -rf $OUT/kt_bench_$tag
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_bench_$tag -- python3 bench.py --steps 2 --warmup 1 $ARGS > $OUT/bench_line_under_kernel_trace_$tag.json 2> $OUT/kt_bench_$tag.err
  db=$(find $OUT/kt_bench_$tag -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db $OUT/bench_kernel_trace_$tag.txt > /dev/null
  python3 tools/rocprof_gaps.py $db 3.1 > $OUT/bench_gpu_idle_$tag.txt 2>&1       # (the two timed steps: the last ~3.1 s of the trace)
  rm -rf $OUT/kt_bench_$tag
  i=0; rm -rf $OUT/pmc_bench_$tag
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_bench_$tag/g$i -- python3 bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_bench_${tag}_g$i.log 2>&1
  done
  ROREG_GIT_COMMIT=$tag python3 tools/pmc_summary.py $OUT/bench_pmc_$tag.txt $OUT/irrep_gemm_pmc_$tag.json 3dmatch-full:f16x2=$OUT/pmc_bench_$tag | tail -5
  rm -rf $OUT/pmc_bench_$tag
  head -30 $OUT/bench_kernel_trace_$tag.txt | cut -c1-220 ;;
final)
  python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_gpu_final.log
  timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc $? lines $(wc -l < $OUT/bench_final.json)"
  timeout 900 python bench.py --force-collectives --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_forced_final.json 2> $OUT/bench_forced_final.err; echo "forced rc $? lines $(wc -l < $OUT/bench_forced_final.json)"
  ROREG_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 3 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_shared3_final.json 2> $OUT/bench_shared3_final.err; echo "shared3 rc $? lines $(wc -l < $OUT/bench_shared3_final.json)"
  python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r04/bench_final.json')); f=json.load(open('gpurun_out/r04/bench_forced_final.json')); k=json.load(open('gpurun_out/r04/bench_shared3_final.json'))
c=j['config']
print('final', j['value'], j['ms_per_step'], j['value_all_local_transforms'], j.get('value_bf16x3'), j['roofline']['frac'], j['roofline']['traffic'], c.get('rd_rm_leg_pairs_per_s'), c.get('rd_rm_leg_sinkhorn_ms_per_pair'), c.get('rr'))
print('forced', f['value'], f['config']['forced_collectives'], f['config']['eqv_bytes_moved_per_step'], f['config'].get('backend'))
print('shared3', k['n_gpus'], k['value'], k['config']['eqv_transfers_per_step'], k['config']['cloud_extractions_per_rank'], k['accuracy'] == j['accuracy'])
PY
  ;;
*) echo "unknown sub-command $cmd"; exit 2 ;;
esac
du -sh $OUT | tail -1

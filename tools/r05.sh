#!/bin/bash
# Round 5 GPU-box recipes (run through gpurun from the repo root): bash tools/r05.sh <cmd> [args]
#   tests "<pytest -k expr>" [file]   -> gpurun_out/r05/pytest_<tag>.log
#   ot5000                            -> Sinkhorn at 5000 x 5000: cooperating workgroups vs the two-pass form (timing + kernel trace)
#   bench [tag] [extra args]          -> the driver's bench command
#   gemmab                            -> the big GEMM launched per tile vs as persistent workgroups (bitwise? ms per launch), store-rate probe
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
  echo "--- two passes per iteration (default beyond 2559 target points)"; timeout 600 python tools/time_sinkhorn.py 5000 1 4 13 26 52 2>&1 | tail -5
  echo "--- cooperating workgroups (ROREG_OT_COOP=1)"; ROREG_OT_COOP=1 timeout 600 python tools/time_sinkhorn.py 5000 13 26 52 2>&1 | tail -3
  echo "--- 2500"; timeout 600 python tools/time_sinkhorn.py 2500 30 100 2>&1 | tail -2
  kt sinkhorn_5000 tools/time_sinkhorn.py 5000 26
  grep -E "of_|calls" $OUT/sinkhorn_5000_kernel_trace.txt | cut -c1-160 ;;
otpmc)   # otpmc <tag> <size> <pairs>: kernel trace + Sinkhorn-specific counter passes of tools/time_sinkhorn.py
  tag=${1:-ot}; size=${2:-2500}; pairs=${3:-100}
  export ROREG_TS_RECOMPUTE_ONLY=1
  kt sinkhorn_$tag tools/time_sinkhorn.py $size $pairs
  grep -E "of_|calls" $OUT/sinkhorn_${tag}_kernel_trace.txt | cut -c1-150
  rm -rf $OUT/pmc_sk; i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_sk/g$i -- python3 tools/time_sinkhorn.py $size $pairs > $OUT/pmc_sk_g$i.log 2>&1 || tail -3 $OUT/pmc_sk_g$i.log
  done
  python3 tools/pmc_kernel_means.py $OUT/pmc_sk > $OUT/sinkhorn_${tag}_pmc.txt
  grep -E "of_iter" -A16 $OUT/sinkhorn_${tag}_pmc.txt | head -60
  rm -rf $OUT/pmc_sk ;;
config4)   # config4 [tag] [keynum]: kernel trace (+ idle gaps) of BASELINE configs[3]'s path (RD + RM + yohoo, 16 clouds / 60 pairs)
  tag=${1:-head}; kn=${2:-2500}
  rm -rf $OUT/kt_c4
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_c4 -- python3 tools/time_configs.py 16 60 --only RD+RM+yohoo --keynum $kn > $OUT/kt_c4.log 2>&1; tail -2 $OUT/kt_c4.log
  db=$(find $OUT/kt_c4 -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db $OUT/rd_rm_config_${tag}_kernel_trace.txt > /dev/null
  python3 tools/rocprof_gaps.py $db 0.5 > $OUT/rd_rm_config_${tag}_gaps.txt 2>&1; tail -15 $OUT/rd_rm_config_${tag}_gaps.txt
  rm -rf $OUT/kt_c4
  head -45 $OUT/rd_rm_config_${tag}_kernel_trace.txt | cut -c1-190 ;;
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
profile)   # profile [tag]: kernel trace (+ idle gaps) of the driver's bench command, short
  tag=${1:-head}
  rm -rf $OUT/kt_b
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_line_under_kernel_trace_$tag.json 2> $OUT/kt_b.err
  db=$(find $OUT/kt_b -name '*.db' | head -1)
  python3 tools/rocprof_summary.py $db $OUT/bench_kernel_trace_$tag.txt > /dev/null
  python3 tools/rocprof_gaps.py $db 3.0 > $OUT/bench_gpu_idle_$tag.txt 2>&1; head -6 $OUT/bench_gpu_idle_$tag.txt
  rm -rf $OUT/kt_b
  head -16 $OUT/bench_kernel_trace_$tag.txt | cut -c1-170 ;;
pmcbench)   # pmcbench <tag>: four PMC passes of one bench step -> $OUT/bench_pmc_<tag>.txt, $OUT/irrep_gemm_pmc_<tag>.json (roofline.traffic / mfma_busy_fraction)
  tag=${1:-head}
  ARGS="--no-cpu-baseline --no-secondary"
  i=0; rm -rf $OUT/pmc_bench_$tag
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_bench_$tag/g$i -- python3 bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_bench_${tag}_g$i.log 2>&1
  done
  ROREG_GIT_COMMIT=$tag python3 tools/pmc_summary.py $OUT/bench_pmc_$tag.txt $OUT/irrep_gemm_pmc_$tag.json 3dmatch-full:f16x2=$OUT/pmc_bench_$tag | tail -5
  rm -rf $OUT/pmc_bench_$tag ;;
final)      # final: build, smoke, full gpu suite, the driver's bench command, forced-collectives bench, 3 ranks on one GPU
  python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_final.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_gpu_final.log
  timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "bench rc $? lines $(wc -l < $OUT/bench_final.json)"
  timeout 900 python bench.py --force-collectives --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_forced_final.json 2> $OUT/bench_forced_final.err; echo "forced rc $? lines $(wc -l < $OUT/bench_forced_final.json)"
  ROREG_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 3 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_shared3_final.json 2> $OUT/bench_shared3_final.err; echo "shared3 rc $? lines $(wc -l < $OUT/bench_shared3_final.json)"
  python3 - <<'PY'
import json
j=json.load(open('gpurun_out/r05/bench_final.json')); f=json.load(open('gpurun_out/r05/bench_forced_final.json')); k=json.load(open('gpurun_out/r05/bench_shared3_final.json'))
c=j['config']
print('final', j['value'], j['ms_per_step'], j['value_contract_complete'], j.get('value_contract_complete_bf16x3'), j.get('value_bf16x3'), j['roofline']['frac'], j['roofline']['traffic'], c.get('rd_rm_leg_pairs_per_s'), c.get('rd_rm_leg_k5000_pairs_per_s'), c.get('rd_rm_leg_k5000_sinkhorn_ms_per_pair'), c.get('rr'))
print('forced', f['value'], f['config']['forced_collectives'], f['config']['eqv_bytes_moved_per_step'], f['config'].get('backend'))
print('shared3', k['n_gpus'], k['value'], k['config']['eqv_transfers_per_step'], k['config']['cloud_extractions_per_rank'], k['accuracy'] == j['accuracy'])
PY
  ;;
gemmab)
  timeout 600 python tools/gemm_persist_ab.py 3 2>&1 | grep -v amdgpu.ids
  echo "--- zero operands"; ROREG_AB_ZEROS=1 timeout 600 python tools/gemm_persist_ab.py 2 2>&1 | grep "B=61440"
  hipcc --offload-arch=gfx950 -O3 tools/probe/store_rate.hip -o /tmp/store_rate 2>/dev/null && timeout 120 /tmp/store_rate ;;
*) echo "unknown command $cmd"; exit 2 ;;
esac

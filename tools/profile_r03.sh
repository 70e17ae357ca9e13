# usage (on the GPU box): bash tools/profile_r03.sh [workload] [mode] [git commit]   -> gpurun_out/r03/{kt_<wl>_<mode>.txt, pmc_<wl>_<mode>.{txt,json}, bench_line_*.json}
# kernel trace of the bench command, then the PMC passes (each counter group in a run of its own, no tracing), as MI355X_MICROARCH.md prescribes.
WL=${1:-3dmatch-full}; MODE=${2:-f16x2}; export ROREG_GIT_COMMIT=${3:-unknown}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03
mkdir -p $OUT
ARGS="--no-cpu-baseline --no-secondary --workload $WL --gemm $MODE"
rm -rf $OUT/kt_${WL}_$MODE
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/kt_${WL}_$MODE -- python3 bench.py --steps 2 --warmup 1 $ARGS > $OUT/bench_line_under_kernel_trace_${WL}_$MODE.json 2> $OUT/kt_${WL}_$MODE.err
db=$(find $OUT/kt_${WL}_$MODE -name '*.db' | head -1)
python3 tools/rocprof_summary.py $db $OUT/kt_${WL}_$MODE.txt > /dev/null
find $OUT/kt_${WL}_$MODE -name '*.db' -delete
i=0
rm -rf $OUT/pmc_${WL}_$MODE
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${WL}_$MODE/g$i -- python3 bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc_${WL}_${MODE}_g$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT/pmc_${WL}_$MODE.txt $OUT/pmc_${WL}_$MODE.json $WL:$MODE=$OUT/pmc_${WL}_$MODE | tail -5
find $OUT -name '*agent_info.csv' -delete
find $OUT -name '*counter_collection.csv' -delete
du -sh $OUT

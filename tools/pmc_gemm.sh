cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE WRITE_SIZE SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcg/$n -- python3 tools/time_gemm.py 40000 > gpurun_out/pmcg_$n.log 2>&1
done
python3 - <<'PY'
import glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcg/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'irrep_gemm' in r['Kernel_Name']:
            k = 'split' if 'split' in r['Kernel_Name'] else 'f32'
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print(f'   {c:32s} {sum(v)/len(v):18.1f}  n={len(v)}')
PY

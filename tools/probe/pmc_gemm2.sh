cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcg
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcg/g$i -- python3 tools/time_gemm.py 20000 > gpurun_out/pmcg_g$i.log 2>&1
  tail -2 gpurun_out/pmcg_g$i.log | cut -c1-200
done
python3 - <<'PY'
import glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcg/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'irrep_gemm' in r['Kernel_Name']:
            k = ('fp16x2' if '32, 2>' in r['Kernel_Name'] else 'bf16x3') if 'split' in r['Kernel_Name'] else 'f32'
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]; print(f'   {c:32s} {sum(v)/len(v):18.1f}  n={len(v)}')
PY

# usage (GPU box): bash tools/probe/pmc_gemm.sh  -> per-kernel PMC averages of tools/time_gemm.py (both GEMM kernels)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcg
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmcg/g$i -- python3 tools/time_gemm.py 20000 > gpurun_out/pmcg_g$i.log 2>&1
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

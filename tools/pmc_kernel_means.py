"""Mean of every PMC counter over the launches of one kernel (rocprofv3 --pmc csv output).  Usage: pmc_kernel_means.py DIR [kernel-substring]"""
import sys, csv, glob, collections
pat = sys.argv[2] if len(sys.argv) > 2 else 'irrep_gemm_split_kernel'
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]; print(f'{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}')

"""Per-kernel means of every PMC counter (rocprofv3 --pmc csv output, one pass per counter group).
Usage: pmc_kernel_means.py DIR [kernel-substring]      (no substring: every kernel, heaviest first)
Derived per launch: hbm_side_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md: wide coalesced reads are
tallied at half their bytes; Infinity-Cache hits included), lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 * 1024)."""
import collections
import csv
import glob
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ''


def short(name):
    parts = name.split('(anonymous namespace)::')
    n = parts[1] if len(parts) > 1 else parts[0]
    return n.split('(')[0][:72]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
mean = lambda v: sum(v) / len(v)
keys = sorted(acc, key=lambda k: -sum(acc[k].get('GRBM_GUI_ACTIVE', [0])))
print('# rocprofv3 --pmc, one pass per counter group; per-launch means (n = launches seen by that pass)')
for k in keys:
    v = acc[k]
    n = max(len(x) for x in v.values())
    print(f'{k}  launches={n}')
    for c in sorted(v):
        print(f'    {c:28s} {mean(v[c]):18.1f}')
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
        print(f'    {"hbm_side_bytes":28s} {(2 * mean(v["FETCH_SIZE"]) + mean(v["WRITE_SIZE"])) * 1024:18.0f}')
    if 'SQ_LDS_BANK_CONFLICT' in v and mean(v.get('SQ_LDS_IDX_ACTIVE', [0])) > 0:
        print(f'    {"lds_conflict":28s} {mean(v["SQ_LDS_BANK_CONFLICT"]) / mean(v["SQ_LDS_IDX_ACTIVE"]):18.3f}')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v and mean(v['GRBM_GUI_ACTIVE']) > 0:
        print(f'    {"mfma_busy":28s} {mean(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (mean(v["GRBM_GUI_ACTIVE"]) / 8.0 * 1024.0):18.3f}')

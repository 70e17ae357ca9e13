"""Summarise rocprofv3 --pmc CSVs (one pass per counter group) per kernel+grid into a text table and, for the dominant
kernel, the per-launch HBM traffic figure bench.py reports as roofline.traffic.
usage: python tools/pmc_summary.py <out.txt> <out.json> <counter_collection.csv>..."""
import collections
import csv
import json
import sys


def short(name):
    parts = name.split('(anonymous namespace)::')
    n = parts[1] if len(parts) > 1 else parts[0]
    return n.split('(')[0][:48]


def main():
    out_txt, out_json, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[(short(r['Kernel_Name']), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    lines = ['# rocprofv3 --pmc (separate passes per counter group) of: python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline',
             '# per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE counts half of a wide coalesced read, see MI355X_MICROARCH.md)']
    keys = sorted(agg, key=lambda k: -sum(agg[k].get('SQ_WAVE_CYCLES', [0])))
    for k in keys:
        v = agg[k]
        if not any(s in k[0] for s in ('irrep_gemm', 'group_conv', 'ft_nonlin', 'nn_search', 'des2r', 'ransac', 'refine')):
            continue
        lines.append(f'{k[0]}  grid={k[1]}  dispatches={max(len(x) for x in v.values())}')
        for c in sorted(v):
            lines.append(f'    {c:28s} {sum(v[c]) / len(v[c]):18.1f}')
    big = [k for k in agg if k[0].startswith('irrep_gemm') and k[1] >= 4000000 and 'FETCH_SIZE' in agg[k]]
    res = {}
    if big:
        fetch = sum(sum(agg[k]['FETCH_SIZE']) for k in big); nf = sum(len(agg[k]['FETCH_SIZE']) for k in big)
        write = sum(sum(agg[k].get('WRITE_SIZE', [0])) for k in big); nw = max(1, sum(len(agg[k].get('WRITE_SIZE', [])) for k in big))
        f_kib = fetch / nf; w_kib = write / nw
        res = {'kernel': 'irrep_gemm_kernel<32> (GF 256->512 / 512->256 launches of bench.py, 40000 keypoints)',
               'FETCH_SIZE_KiB_per_launch': f_kib, 'WRITE_SIZE_KiB_per_launch': w_kib,
               'hbm_bytes_per_launch': (2.0 * f_kib + w_kib) * 1024.0,
               'note': 'traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (wide coalesced reads '
                       'are tallied at half their bytes); Infinity-Cache hits are included in FETCH_SIZE, so this is an upper bound on HBM bytes'}
        lines.append('')
        lines.append(json.dumps(res))
    open(out_txt, 'w').write('\n'.join(lines) + '\n')
    json.dump(res, open(out_json, 'w'), indent=1)
    print('\n'.join(lines[:60]))


if __name__ == '__main__':
    main()

"""Summarise rocprofv3 --pmc CSVs (one pass per counter group) per kernel+grid into a text table and, for the dominant
kernel of each matrix-core mode, the per-launch HBM traffic figure bench.py reports as roofline.traffic.
usage: python tools/pmc_summary.py <out.txt> <out.json> <workload>:<mode>=<dir> [...]     (mode: f16x2 | bf16x3 | f32)
Each <dir> holds the counter_collection.csv files of the passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
--no-secondary --workload <workload> --gemm <mode>`.  The dominant-kernel figures are taken over exactly the launches bench.py prices in
`roofline`: the instantiations tagged BIG = 1 (C * O = 256 * 512: GF's 256->512 and 512->256 layers)."""
import collections
import csv
import glob
import json
import re, sys

KEEP = ('irrep_gemm', 'group_conv', 'ft_nonlin', 'nn_search', 'des2r', 'ransac', 'refine', 'gf_finalize', 'et_gather', 'match_prepare')


def short(name):
    parts = name.split('(anonymous namespace)::')
    n = parts[1] if len(parts) > 1 else parts[0]
    return n.split('(')[0][:56]


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(short(r['Kernel_Name']), int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def main():
    out_txt, out_json = sys.argv[1], sys.argv[2]
    lines = ['# rocprofv3 --pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE in passes of their own), of:',
             '#   python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --gemm <mode>',
             '# per-dispatch averages; FETCH_SIZE/WRITE_SIZE in KiB as reported (gfx950: FETCH_SIZE tallies a wide coalesced read at half',
             '# its bytes, MI355X_MICROARCH.md section HBM -- see the calibration line of each mode).']
    import hashlib, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # bench.py quotes `roofline.traffic` from this file only while csrc/fourier.hip still hashes to kernel_source_sha16
    res = {'hbm_bytes_per_launch': {}, 'detail': {}, 'git_commit': os.environ.get('ROREG_GIT_COMMIT'),
           # the conditions of the profiled command (bench.py refuses to quote the figure for a run under other ones); ROREG_PMC_CONDITIONS = JSON overrides
           'conditions': dict({'kpts': 5000, 'dtype': 'fp32', 'gpus': 1, 'pair_lists': 'banded', 'xdma': os.environ.get('ROREG_GEMM_XDMA', '1') != '0', 'mfma16': os.environ.get('ROREG_GEMM_MFMA16', '1') != '0'},
                              **json.loads(os.environ.get('ROREG_PMC_CONDITIONS', '{}'))),
           'kernel_source_sha16': hashlib.sha256(open(os.path.join(here, 'roreg_amd', 'csrc', 'fourier.hip'), 'rb').read()).hexdigest()[:16]}
    for spec in sys.argv[3:]:
        mode, d = spec.split('=')
        gm = mode.split(':')[-1]
        agg = load(d)
        lines.append('')
        lines.append(f'## mode {mode}')
        keys = sorted(agg, key=lambda k: -sum(agg[k].get('GRBM_GUI_ACTIVE', agg[k].get('SQ_WAVE_CYCLES', [0]))))
        for k in keys:
            v = agg[k]
            if not any(s in k[0] for s in KEEP):
                continue
            lines.append(f'{k[0]}  grid={k[1]}  dispatches={max(len(x) for x in v.values())}')
            for c in sorted(v):
                lines.append(f'    {c:28s} {sum(v[c]) / len(v[c]):18.1f}')
        # calibration of the read counter on a kernel with a known byte count: gf_finalize reads [B,32,60] f32 once (rows = grid/256*4)
        cal = None
        for k in agg:
            if k[0].startswith('gf_finalize') and 'FETCH_SIZE' in agg[k]:
                rows = k[1] // 64                                 # one wavefront per keypoint
                known = rows * 32 * 60 * 4 / 1024.0
                f = sum(agg[k]['FETCH_SIZE']) / len(agg[k]['FETCH_SIZE'])
                if cal is None or rows > cal[0]:
                    cal = (rows, known, f)
        if cal:
            lines.append(f'calibration: gf_finalize on {cal[0]} keypoints reads {cal[1]:.0f} KiB; FETCH_SIZE reports {cal[2]:.0f} KiB -> factor {cal[1] / cal[2]:.3f}')
        kern = 'irrep_gemm_kernel' if gm == 'f32' else 'irrep_gemm_'
        def big_tag(name):          # irrep_gemm_split_kernel<CT, NP, WO, BIG[, PIPE]> (fourth template argument) or irrep_gemm_xdma_kernel<BIG> / irrep_gemm_xdma16_kernel<BIG>
            m = re.search(r'irrep_gemm_split_kernel<\s*\d+,\s*\d+,\s*\d+,\s*(\d+)', name) or re.search(r'irrep_gemm_xdma(?:16)?_kernel<\s*(\d+)', name)
            return bool(m) and m.group(1) == '1'
        tagged = (lambda k: k[1] >= 4000000) if gm == 'f32' else (lambda k: big_tag(k[0]))
        big = [k for k in agg if k[0].startswith(kern) and tagged(k) and 'FETCH_SIZE' in agg[k] and 'WRITE_SIZE' in agg[k]]
        if gm != 'f32' and big:
            kern = sorted({k[0].split('<')[0] for k in big})[-1]
        if big:
            nf = sum(len(agg[k]['FETCH_SIZE']) for k in big); nw = sum(len(agg[k]['WRITE_SIZE']) for k in big)
            f_kib = sum(sum(agg[k]['FETCH_SIZE']) for k in big) / nf
            w_kib = sum(sum(agg[k]['WRITE_SIZE']) for k in big) / nw
            factor = 2.0
            res['hbm_bytes_per_launch'][mode] = (factor * f_kib + w_kib) * 1024.0
            busy = [k for k in agg if k[0].startswith(kern) and tagged(k) and 'SQ_VALU_MFMA_BUSY_CYCLES' in agg[k] and 'SQ_BUSY_CYCLES' in agg[k]]
            clk = [k for k in agg if k[0].startswith(kern) and tagged(k) and 'GRBM_GUI_ACTIVE' in agg[k]]
            extra = {}
            if busy:
                mf = sum(sum(agg[k]['SQ_VALU_MFMA_BUSY_CYCLES']) for k in busy); n_b = sum(len(agg[k]['SQ_VALU_MFMA_BUSY_CYCLES']) for k in busy)
                extra['SQ_VALU_MFMA_BUSY_CYCLES_per_launch'] = mf / n_b
            if clk:
                ga = sum(sum(agg[k]['GRBM_GUI_ACTIVE']) for k in clk); n_c = sum(len(agg[k]['GRBM_GUI_ACTIVE']) for k in clk)
                extra['GRBM_GUI_ACTIVE_per_launch'] = ga / n_c
                if busy:
                    extra['mfma_busy_fraction'] = (mf / n_b) / ((ga / n_c) / 8.0 * 1024.0)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            res['detail'][mode] = {'kernel': kern + (': the BIG-tagged launches (C*O = 256*512) = the population of roofline.launches' if gm != 'f32' else '<32>: grids >= 4M threads'),
                                   'launches_counted': nf, **extra,
                                   'FETCH_SIZE_KiB_per_launch': f_kib, 'WRITE_SIZE_KiB_per_launch': w_kib, 'read_factor': factor,
                                   'calibration_factor_measured_on_gf_finalize': (cal[1] / cal[2]) if cal else None}
    res['note'] = ('traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (wide coalesced reads are '
                   'tallied at half their bytes); Infinity-Cache hits are included in FETCH_SIZE, so this is an upper bound on HBM bytes')
    lines.append('')
    lines.append(json.dumps(res))
    open(out_txt, 'w').write('\n'.join(lines) + '\n')
    json.dump(res, open(out_json, 'w'), indent=1)
    print('\n'.join(lines[-12:]))


if __name__ == '__main__':
    main()

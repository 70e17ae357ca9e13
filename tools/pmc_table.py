"""usage: python tools/pmc_table.py <pmc_kernel_means output> [n]   -- one line per kernel: share of the GPU time of the run, LDS bank-conflict ratio,
matrix-pipe / vector-pipe / LDS-index-pipe busy fractions (of GRBM_GUI_ACTIVE per XCD; 1024 SIMDs, 256 CUs), HBM-side TB/s (at 2.1 GHz under the counters)."""
import sys
cur = None; rows = {}
for line in open(sys.argv[1]):
    if line[0] not in ' #\n' and 'launches=' in line:
        cur = line.split('  launches=')[0]; rows[cur] = {'n': int(line.split('launches=')[1])}
    elif cur and line.startswith('    '):
        p = line.split()
        if len(p) == 2:
            try: rows[cur][p[0]] = float(p[1])
            except ValueError: pass
out = []
for k, r in rows.items():
    g = r.get('GRBM_GUI_ACTIVE')
    if not g: continue
    out.append((g * r['n'], k[:64], r['n'], r.get('lds_conflict', 0), r.get('mfma_busy', 0), r.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / (g / 8),
                r.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / (g / 8), r.get('hbm_side_bytes', 0) / (g / 8 / 2.1e9) / 1e12))
tot = sum(o[0] for o in out)
print(f"{'kernel':64s} {'n':>5s} {'%':>5s} {'ldsconf':>7s} {'mfma':>5s} {'valu':>5s} {'ldsbusy':>7s} {'TB/s':>5s}")
for o in sorted(out, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f'{o[1]:64s} {o[2]:5d} {100 * o[0] / tot:5.1f} {o[3]:7.3f} {o[4]:5.2f} {o[5]:5.2f} {o[6]:7.2f} {o[7]:5.2f}')

"""GPU idle time inside a kernel trace (rocprofv3 rocpd database): gaps between the end of one kernel and the start of the next, by size class.
usage: python tools/rocprof_gaps.py <results.db> [last_seconds] [detail]     (only the kernels of the last so many seconds of the trace;
`detail`: every gap >= 1 ms with its position in the window and the three kernels on either side)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
s_col = 'start' if 'start' in cols else [c for c in cols if 'start' in c][0]
e_col = 'end' if 'end' in cols else [c for c in cols if 'end' in c][0]
rows = db.execute(f'select {s_col}, {e_col}, name from kernels order by {s_col}').fetchall()
last = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
t1 = rows[-1][1]
rows = [r for r in rows if (t1 - r[0]) / 1e9 <= last]
busy = sum(r[1] - r[0] for r in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
end = rows[0][1]; prev = rows[0][2]
for s, e, n in rows[1:]:
    if s > end: gaps.append((s - end, prev[:44] + ' -> ' + n[:44]))
    if e > end: end = e; prev = n
idle = sum(g for g, _ in gaps)
print(f'{len(rows)} kernels over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms ({100 * idle / span:.2f} %)')
for lo, hi in ((0, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)):
    g = [x for x, _ in gaps if lo <= x < hi]
    print(f'   gaps {lo / 1e3:>6.0f}..{hi / 1e3:<8.0f} us: {len(g):6d}, {sum(g) / 1e6:8.2f} ms')
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for g, n in gaps:
    if g >= 200e3: agg[n][0] += 1; agg[n][1] += g
print('   gaps >= 200 us by (previous kernel -> next kernel):')
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f'      {c:4d} x, {t / 1e6:8.2f} ms  {n}')
if len(sys.argv) > 3 and sys.argv[3] == 'detail':
    print('   gaps >= 1 ms, in time order (offset from the window start; the kernels before | after):')
    end = rows[0][1]; last_i = 0
    t0 = rows[0][0]
    for i in range(1, len(rows)):
        s_, e_, n_ = rows[i]
        if s_ > end and s_ - end >= 1e6:
            before = ' < '.join(r[2].split('(')[0].split('::')[-1][:28] for r in rows[max(0, last_i - 2):last_i + 1])
            after = ' > '.join(r[2].split('(')[0].split('::')[-1][:28] for r in rows[i:i + 3])
            print(f'      at {(end - t0) / 1e6:9.1f} ms: {(s_ - end) / 1e6:6.2f} ms   {before}  |  {after}')
        if e_ > end:
            end = e_; last_i = i

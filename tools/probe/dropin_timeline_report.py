"""usage: python tools/probe/dropin_timeline_report.py <results.db>   -- cuts a rocprofv3 kernel + memory-copy trace into windows at idle
stretches >= 300 ms and prints, per window: span, kernel-busy time, the copies by direction (bytes, busy union, how much of it lies
under kernels), the first kernel's offset, and every stretch >= 2 ms with no kernel running (with the copies active inside it)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
def pick(sub):
    c = [t for t in tables if t == sub] or [t for t in tables if sub in t.lower() and 'rocpd_' not in t.lower()] or [t for t in tables if sub in t.lower()]
    return c[0] if c else None
kt, ct = pick('kernels'), pick('memory_copies') or pick('memory_copy') or pick('memcpy')
def cols(t): return [r[1] for r in db.execute(f'pragma table_info({t})')]
kc = cols(kt)
ks, ke = ('start' if 'start' in kc else [c for c in kc if 'start' in c][0]), ('end' if 'end' in kc else [c for c in kc if 'end' in c][0])
K = db.execute(f'select {ks}, {ke}, name from {kt} order by {ks}').fetchall()
C = []
if ct:
    cc = cols(ct)
    print('copy table', ct, cc)
    cs, ce = ('start' if 'start' in cc else [c for c in cc if 'start' in c][0]), ('end' if 'end' in cc else [c for c in cc if 'end' in c][0])
    name = 'name' if 'name' in cc else None
    size = [c for c in cc if c in ('size', 'bytes')] or [c for c in cc if 'size' in c or 'bytes' in c]
    C = db.execute(f"select {cs}, {ce}, {name or 'NULL'}, {size[0] if size else 0} from {ct} order by {cs}").fetchall()
else:
    print('no memory copy table among', tables)
ev = sorted([(s, e, 'K', n, 0) for s, e, n in K] + [(s, e, 'C', str(n), b or 0) for s, e, n, b in C])
wins, cur, end = [], [], None
for x in ev:
    if end is not None and x[0] - end >= 300e6:
        wins.append(cur); cur = []
    cur.append(x); end = x[1] if end is None else max(end, x[1])
wins.append(cur)
def union(iv):
    tot, e0 = 0, None
    out = []
    for s, e in sorted(iv):
        if e0 is None or s > e0:
            out.append([s, e]); e0 = e
        elif e > e0:
            out[-1][1] = e; e0 = e
    return out
def overlap(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: tot += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
for w, win in enumerate(wins):
    t0, t1 = win[0][0], max(x[1] for x in win)
    ku = union([(s, e) for s, e, k, _, _ in win if k == 'K'])
    kbusy = sum(e - s for s, e in ku)
    nk = sum(1 for x in win if x[2] == 'K')
    print(f'\nwindow {w}: {(t1 - t0) / 1e6:.1f} ms, {nk} kernels busy {kbusy / 1e6:.1f} ms' + (f', first kernel at {(ku[0][0] - t0) / 1e6:.1f} ms, last ends {(ku[-1][1] - t0) / 1e6:.1f} ms' if ku else ''))
    by = {}
    for s, e, k, n, b in win:
        if k == 'C':
            by.setdefault(n, []).append((s, e, b))
    for n, v in by.items():
        u = union([(s, e) for s, e, _ in v])
        print(f'   copies {n[:40]:40s}: {len(v):5d}, {sum(b for _, _, b in v) / 1e9:7.3f} GB, busy {sum(e - s for s, e in u) / 1e6:7.1f} ms (under kernels {overlap(u, ku) / 1e6:7.1f}), first at {(u[0][0] - t0) / 1e6:.1f} last ends {(u[-1][1] - t0) / 1e6:.1f} ms')
    if w and ku:
        prev = t0
        for s, e in ku + [[t1, t1]]:
            if s - prev >= 2e6:
                inside = [(n[:24], round((max(cs, prev) - t0) / 1e6, 1), round((min(ce, s) - max(cs, prev)) / 1e6, 2)) for cs, ce, k, n, b in win if k == 'C' and min(ce, s) > max(cs, prev)]
                last = [x[3].split('(')[0].split('::')[-1][:30] for x in win if x[2] == 'K' and x[1] <= prev + 1][-1:]
                nxt = [x[3].split('(')[0].split('::')[-1][:30] for x in win if x[2] == 'K' and x[0] >= s][:1]
                tot_c = sum(c[2] for c in inside)
                print(f'   no kernel {(prev - t0) / 1e6:8.1f} .. {(s - t0) / 1e6:8.1f} ms ({(s - prev) / 1e6:6.1f} ms)  {last} | {nxt}   copies inside: {len(inside)}, {tot_c:.1f} ms')
            prev = max(prev, e)

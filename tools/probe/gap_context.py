"""For the idle gaps >= `min_us` of the last `last_s` seconds of a rocprofv3 kernel trace: the two kernels before and after each gap."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); last = float(sys.argv[2]); min_us = float(sys.argv[3])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
s_col = 'start' if 'start' in cols else [c for c in cols if 'start' in c][0]
e_col = 'end' if 'end' in cols else [c for c in cols if 'end' in c][0]
rows = db.execute(f'select {s_col}, {e_col}, name from kernels order by {s_col}').fetchall()
t1 = rows[-1][1]; rows = [r for r in rows if (t1 - r[0]) / 1e9 <= last]
end = rows[0][1]
for i in range(1, len(rows)):
    s, e, n = rows[i]
    if s - end >= min_us * 1e3:
        ctx = [rows[j][2][:40] for j in range(max(0, i - 3), min(len(rows), i + 3))]
        print(f'{(s - end) / 1e3:8.0f} us | ' + ' | '.join(ctx[:3]) + '  ==>  ' + ' | '.join(ctx[3:]))
    end = max(end, e)

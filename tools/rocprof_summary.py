"""Summarise a rocprofv3 rocpd database (kernel trace) into the text table committed under profiles/.
usage: python tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    rows = c.execute('select name, grid_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, count(*), sum(duration), avg(duration), '
                     'min(duration), max(duration) from kernels group by name, grid_x order by sum(duration) desc').fetchall()
    total = sum(r[7] for r in rows)
    lines = [f'# rocprofv3 --kernel-trace --stats summary ({sys.argv[1]}); total kernel time {total/1e6:.3f} ms',
             f'{"calls":>6} {"total_ms":>10} {"avg_us":>10} {"min_us":>10} {"max_us":>10} {"pct":>6} {"grid":>9} {"lds":>7} {"vgpr":>5} {"agpr":>5} {"sgpr":>5}  kernel']
    for (name, grid, lds, vg, ag, sg, n, tot, avg, mn, mx) in rows:
        short = name if len(name) < 110 else name[:107] + '...'
        lines.append(f'{n:6d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f} {grid:9d} {lds:7d} {vg:5d} {ag:5d} {sg:5d}  {short}')
    out = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out)
    print(out)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd .db output): count, average / minimum / total duration.
usage: scripts/kstats.py gpurun_out/prof/x_results.db [out.csv]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute('select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc').fetchall()
    lines = ['kernel,calls,avg_us,min_us,total_ms']
    for name, cnt, avg, mn, tot in rows:
        lines.append(f'"{name}",{cnt},{avg / 1e3:.2f},{mn / 1e3:.2f},{tot / 1e6:.3f}')
    text = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text)
    for name, cnt, avg, mn, tot in rows[:30]:
        print(f'{name[:100]:100s} n={cnt:5d} avg={avg / 1e3:9.1f}us min={mn / 1e3:9.1f}us tot={tot / 1e6:9.3f}ms')


if __name__ == '__main__':
    main()

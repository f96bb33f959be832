'''Summarise a rocprofv3 rocpd sqlite trace: per-kernel count / total / avg duration.'''
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f'# total kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches')
print('name,calls,total_ms,avg_us,min_us,max_us,pct')
for name, n, s, a, mn, mx in rows[:40]:
    short = re.sub(r'\(.*', '', name)[:90]
    print(f'"{short}",{n},{s/1e6:.2f},{a/1e3:.1f},{mn/1e3:.1f},{mx/1e3:.1f},{100*s/tot:.1f}')

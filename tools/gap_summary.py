import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select start, end, name from kernels order by start").fetchall()
# the measured block = everything after the largest idle gap
gaps = [(rows[i + 1][0] - rows[i][1], i) for i in range(len(rows) - 1)]
g, i = max(gaps)
blk = rows[i + 1:]
span = blk[-1][1] - blk[0][0]
busy = sum(e - s for s, e, _ in blk)
idle = [blk[k + 1][0] - blk[k][1] for k in range(len(blk) - 1)]
print(f'kernels {len(blk)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {100*(span-busy)/span:.1f} %  '
      f'mean gap {sum(idle)/len(idle)/1e3:.2f} us  gaps > 10 us: {sum(1 for d in idle if d > 10000)}')

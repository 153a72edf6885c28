"""Timeline of the last N ms of a rocprofv3 run (kernel + memory-copy trace, the .db output): one line per DMA and per
kernel longer than a threshold, times relative to the first event shown.  usage: rocpd_timeline.py run_results.db [ms] [min_us]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
span = float(sys.argv[2]) if len(sys.argv) > 2 else 13.5
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
ks = list(cur.execute("select start,end,name,queue_id from kernels"))
ms = list(cur.execute("select start,end,name,size,queue_id from memory_copies"))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|gbx::|^void ", "", n)
    return n.split("(")[0][:60]
ev = [(k[0], k[1], "K q%s %s" % (k[3], short(k[2]))) for k in ks]
ev += [(m[0], m[1], "M q%s %s %d" % (m[4], m[2].replace("MEMORY_COPY_", ""), m[3])) for m in ms]
tend = max(e[1] for e in ev)
ev = sorted(e for e in ev if e[0] > tend - span * 1e6)
t0 = ev[0][0]
for a, b, name in ev:
    d = (b - a) / 1e3
    if name[0] == "K" and d < min_us and "unpack" not in name and "sort" not in name and "classify" not in name: continue
    print("%8.1f %8.1f %8.1f %s" % ((a - t0) / 1e3, (b - t0) / 1e3, d, name))

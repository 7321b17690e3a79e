"""Kernel timeline of a rocprofv3 --kernel-trace results.db: one line per dispatch (start ms, end ms, queue, name),
table-building and calibration kernels left out.  python benchmarks/prof_timeline.py results.db > timeline.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = cur.execute("select %s, start, end, %s from kernels order by start" % (name_col, qcol)).fetchall()
t0 = rows[0][1]
for n, s, e, q in rows:
    short = n.split("(")[0].replace("void ", "").replace("bp::", "")
    if short.startswith(("k_fb_", "k_spin", "k_point_mul", "k_add_two_mul", "k_field_op", "k_gsum")):
        continue
    print("%10.3f %10.3f %8.3f q=%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, short))

"""Overlap analysis of a rocprofv3 --kernel-trace results.db: busy time (union of kernel intervals) against the
sum of kernel durations, per queue, plus a coarse timeline of one window."""
import sqlite3
import sys


def main(path, t_from_ms=None, t_len_ms=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute("select %s, start, end, %s from kernels order by start" % (name_col, qcol or "0")).fetchall()
    t0 = rows[0][1]
    total = sum(e - s for _, s, e, _ in rows)
    union, cur_s, cur_e = 0, None, None
    for _, s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print("kernels %d  sum %.1f ms  union(busy) %.1f ms  span %.1f ms  columns %s" % (
        len(rows), total / 1e6, union / 1e6, (rows[-1][2] - t0) / 1e6, cols))
    if t_from_ms is not None:
        lo = t0 + int(t_from_ms * 1e6)
        hi = lo + int(t_len_ms * 1e6)
        for n, s, e, q in rows:
            if e >= lo and s <= hi:
                print("%9.3f %9.3f  q=%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, n[:50]))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], float(a[2]) if len(a) > 2 else None, float(a[3]) if len(a) > 3 else None)

"""Summarise a rocprofv3 --kernel-trace results.db: per-kernel calls / total / avg / share."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-60s %8s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for n, c, t, a, mn, mx in rows:
        print("%-60s %8d %12.1f %12.1f %12.1f %12.1f %6.1f" % (n[:60], c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                              100.0 * t / total))


if __name__ == "__main__":
    main(sys.argv[1])

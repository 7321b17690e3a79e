"""GPU occupancy inside the steady state of a kernel trace (rocprofv3 --kernel-trace results.db): takes the window
between the `skip`-th and the last launch of k_verdict (one per chunk and pass), and prints the span, the union of
kernel intervals (GPU busy with at least one kernel), the summed kernel durations and the per-kernel totals there."""
import sqlite3
import sys


def main(path, skip=8):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    verd = [e for n, s, e in rows if "k_verdict" in n]
    lo, hi = verd[skip], verd[-1]
    inside = [(n, max(s, lo), min(e, hi)) for n, s, e in rows if e > lo and s < hi]
    span = hi - lo
    union, cs, ce = 0, None, None
    for _, s, e in inside:
        if ce is None or s > ce:
            if ce is not None:
                union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    union += ce - cs
    total = sum(e - s for _, s, e in inside)
    print("window %.1f ms (%d k_verdict launches inside)  busy(union) %.1f ms = %.1f %%  sum of kernels %.1f ms = %.2f x span"
          % (span / 1e6, len(verd) - skip - 1, union / 1e6, 100.0 * union / span, total / 1e6, total / span))
    per = {}
    for n, s, e in inside:
        k = n.split("(")[0][:44]
        per[k] = per.get(k, 0) + (e - s)
    for k, t in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-46s %9.1f ms  %5.1f %% of span" % (k, t / 1e6, 100.0 * t / span))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)

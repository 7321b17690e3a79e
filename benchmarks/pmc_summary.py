"""Per-kernel sums of a rocprofv3 --pmc results.db (counter value per dispatch, averaged per launch)."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(grid_size) "
                       "from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    print("%-50s %-14s %8s %16s %16s %12s" % ("kernel", "counter", "launches", "sum", "avg/launch", "avg grid"))
    for k, c, n, s, a, g in rows:
        print("%-50s %-14s %8d %16.1f %16.1f %12.0f" % (k[:50], c, n, s, a, g))


if __name__ == "__main__":
    main(sys.argv[1])

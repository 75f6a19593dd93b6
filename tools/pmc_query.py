"""Print per-kernel averages of every counter in a rocprofv3 --pmc results db (rocpd sqlite)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
for r in db.execute("select kernel_name, counter_name, avg(value), count(*), avg(end-start) from counters_collection "
                    "where kernel_name like ? group by kernel_name, counter_name", (pat,)):
    print(f"{r[0][:60]:60s} {r[1]:28s} avg={r[2]:.6g} n={r[3]} dur_us={r[4] / 1e3:.1f}")

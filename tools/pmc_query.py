"""Print per-kernel PMC averages from a rocprofv3 --pmc output directory."""
import sqlite3, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    v = [t for t in tabs if t == "counters_collection"]
    if not v:
        print("no counters_collection view; have", tabs[:12]); continue
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
    q = f"select substr({kn},1,46), counter_name, count(*), avg(value) from counters_collection group by {kn}, counter_name order by 4 desc limit 40"
    for r in c.execute(q): print("  %-48s %-11s n=%4d avg %.1f" % r)

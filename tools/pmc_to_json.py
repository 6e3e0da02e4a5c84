"""rocprofv3 --pmc output dirs -> per-kernel mean counter values as JSON.
usage: pmc_to_json.py out.json label1=dir1 label2=dir2 ..."""
import glob, json, sqlite3, sys
out = {}
for arg in sys.argv[2:]:
    label, d = arg.split("=", 1)
    for f in glob.glob(d + "/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            continue
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
        for name, cname, cnt, avg in c.execute(
                f"select {kn}, counter_name, count(*), avg(value) from counters_collection group by {kn}, counter_name"):
            out.setdefault(label, {}).setdefault(name.split("(")[0][:100], {})[cname] = dict(n=cnt, mean=avg)
        # per-dispatch values in launch order (kernels of one name may be launched with different shapes)
        dcol = "dispatch_id" if "dispatch_id" in cols else None
        if dcol:
            for name, cname, did, val in c.execute(
                    f"select {kn}, counter_name, {dcol}, sum(value) from counters_collection group by {kn}, counter_name, {dcol} order by {dcol}"):
                out[label][name.split("(")[0][:100]][cname].setdefault("per_dispatch", []).append(val)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])

"""rocpd sqlite (rocprofv3 --kernel-trace --stats) -> markdown table; usage: prof_summary.py <dir> <out.md> "<title>" "<command>"."""
import glob, sqlite3, sys
d, out, title, cmd = sys.argv[1:5]
rows = []
for f in glob.glob(d + "/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    rows += c.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0 from kernels group by name").fetchall()
tot = sum(r[2] for r in rows)
rows.sort(key=lambda r: -r[2])
with open(out, "w") as fh:
    fh.write(f"# {title}\n\nCommand: `{cmd}`\n(microseconds; rocpd `kernels` view, all dispatches of the process)\n\n")
    fh.write("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
    for n, c_, t, a in rows[:28]:
        fh.write("| `%s` | %d | %.0f | %.3f | %.2f |\n" % (n.split("(")[0][:90], c_, t, a, 100 * t / tot))
    fh.write("\nTotal kernel time: %.1f ms over %d dispatches\n" % (tot / 1000, sum(r[1] for r in rows)))
print(open(out).read())

# usage: gpurun_prof.sh <script.py> [args]  -- rocprofv3 kernel stats (avg us per kernel)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats -d /tmp/prof -o p -- python $GRAFT_REPO_ROOT/"$1" "${@:2}" > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, glob
for f in glob.glob("/tmp/prof/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    for r in c.execute("select substr(name,1,40), count(*), avg(end-start)/1000.0, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 12"):
        print("%-42s calls %7d  avg %8.2f us  total %9.2f ms" % r)
PY

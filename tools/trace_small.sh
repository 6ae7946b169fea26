cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t64; rocprofv3 --kernel-trace --stats -d /tmp/t64 -- python $GRAFT_REPO_ROOT/tools/native_vs_python.py > /tmp/t64.log 2>&1
tail -4 /tmp/t64.log
python - <<'PY'
import sqlite3, glob
db=glob.glob('/tmp/t64/**/*.db', recursive=True)[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]
cols=[r[1] for r in c.execute("pragma table_info(%s)"%kt)]
rows=c.execute("select start, end from %s order by start"%kt).fetchall()
# last 2000 dispatches: native 64^3 run
rows=rows[-3000:]
busy=sum(e-s for s,e in rows); span=rows[-1][1]-rows[0][0]
print("dispatches",len(rows),"busy ms %.2f span ms %.2f busy frac %.2f"%(busy/1e6, span/1e6, busy/span))
PY

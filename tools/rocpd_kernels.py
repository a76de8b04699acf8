"""Kernel names, dispatch counts, mean and minimum durations out of a rocprofv3 rocpd (SQLite) output directory.
    python tools/rocpd_kernels.py DIR [rows = 8]
"""
import sqlite3,sys,glob
db=sqlite3.connect(glob.glob(sys.argv[1]+'/*.db')[0])
rows=db.execute("select name, count(*), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 8]: print(r[0][:80], r[1], round(r[2]), r[3])

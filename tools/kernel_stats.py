import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); c=db.cursor()
n=int(sys.argv[2]) if len(sys.argv)>2 else 1
rows=list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc"))
tot=sum(r[2] for r in rows)
print("total ms %.2f  per update (%d updates) %.2f"%(tot,n,tot/n))
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 14]:
    print("%-80s %6d %9.2f ms %9.1f us  per-update %6.2f ms"%(r[0][:80],r[1],r[2],r[3], r[2]/n))

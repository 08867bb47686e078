import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tables if 'copy' in t.lower() or 'memory' in t.lower()])
for t in tables:
    if 'memory_cop' in t.lower() or t == 'memory_copies':
        cols = [r[1] for r in db.execute(f"pragma table_info({t})")]
        print(t, cols)
        rows = list(db.execute(f"select * from {t} order by start desc limit 40"))
        for r in rows[:40]:
            print(r)
        break

"""every launch of the kernels whose name contains <pattern>, in start order: duration in us.  usage: db_kernel_list.py <results.db> <pattern>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, s, e in db.execute("select name, start, end from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)):
    print(f"{(e - s) / 1e3:9.2f}")

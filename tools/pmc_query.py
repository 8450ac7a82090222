"""Prints avg counter value and duration per glhip kernel from a rocprofv3 results database: python tools/pmc_query.py <db>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
     "group by kernel_name, counter_name order by avg(duration) desc")
for r in c.execute(q):
    if "glhip" in r[0]:
        print(r[0][:90], r[1], r[2], "%.4e" % r[3], "%.3f ms" % (r[4] / 1e6))

import sqlite3,sys
for d in sys.argv[1:]:
    c=sqlite3.connect(f"gpurun_out/{d}/p_results.db")
    try:
        for r in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%k_interp%' group by kernel_name, counter_name"):
            print(r[0][:40], r[1], r[2]/r[3], r[3])
    except Exception as e: print(e)
    try:
        for r in c.execute("select name,total_calls,average from top_kernels where name like '%k_interp%'"): print(r[0][:40], r[1], r[2])
    except Exception as e: print(e)

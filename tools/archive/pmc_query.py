import sqlite3, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    c = sqlite3.connect(f"gpurun_out/{d}/p_results.db")
    try:
        for r in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (f"%{pat}%",)):
            print(r[0][:40], r[1], r[2] / r[3], r[3])
    except Exception as e:
        print(e)
    try:
        for r in c.execute("select name,total_calls,average from top_kernels where name like ?", (f"%{pat}%",)):
            print(r[0][:40], r[1], r[2])
    except Exception as e:
        print(e)

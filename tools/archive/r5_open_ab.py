#!/usr/bin/env python3
"""open_table (dfdb_table_load of one 1e9-row Int64 column from /dev/shm) with and without the progressive decode, alternating"""
import sys,os,time,json,tempfile,shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dataframedbs.jl_amd")); sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
import dfdb
ctx=dfdb.default_context(0)
t=dfdb.DFTable.new(); t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, 1000000000)
d=tempfile.mkdtemp(dir="/dev/shm")
try:
    st=t.save(os.path.join(d,"tb")); t.close()
    for prog in (1,0,1,0,1,0):
        ctx.set_option("load_progressive", prog)
        t2=dfdb.open_table(os.path.join(d,"tb"), load=False)
        torch.cuda.synchronize(); t0=time.perf_counter(); t2.load(["x"]); torch.cuda.synchronize(); dt=time.perf_counter()-t0
        q=t2[("x", lambda x: x > 899999), dfdb.ALL]._query()
        print(json.dumps({"progressive":prog,"seconds":round(dt,4),"file_GBps":round(st["compressed"]/dt/1e9,2),"count":q.count()}), flush=True)
        t2.close()
finally:
    shutil.rmtree(d, ignore_errors=True)

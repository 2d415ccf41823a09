import os, sys, tempfile, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/dataframedbs.jl_amd')
import torch, dfdb
n=int(4e7)
d=tempfile.mkdtemp(dir="/dev/shm")
t=dfdb.DFTable.new(); t.add_generated("x", dfdb.GEN_I64_MOD1M, 1, n); t.save(d+"/tb"); t.close()
tb=dfdb.open_table(d+"/tb", load=False)
v=tb[("x", lambda x: x>899999), dfdb.ALL]
t0=time.perf_counter(); print(dfdb.nrow_streamed(v,64)); print("total", time.perf_counter()-t0)
import shutil; shutil.rmtree(d)

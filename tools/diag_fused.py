import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/dataframedbs.jl_amd')
import torch, dfdb
ctx = dfdb.default_context(0)
n=10**9
t = dfdb.DFTable.new(); t.add_generated("x", dfdb.GEN_I64_MOD1M, 0x9E3779B97F4A7C15, n)
q = t[("x", lambda x: x > 899999), dfdb.ALL]._query()
ns = q.count(); out = torch.empty(ns, dtype=torch.int64, device="cuda:0")
ctx.profile(True)
for name,fused,diag in [("unfused",0,0),("fused",1,0),("fused-noidx",1,1),("fused-nolook",1,2),("fused-neither",1,3)]*3:
    ctx.set_option("fused",fused); ctx.set_option("fused_diag",diag)
    a={k:ctx.profile_get(k) for k in ("scan_compact","scan_cmp","compact_indices","scan_counts")}
    q.reset(); q.indices_device(out.data_ptr(), ns)
    b={k:ctx.profile_get(k) for k in a}
    print(name, {k: round(b[k][1]-a[k][1],4) for k in a if b[k][0]!=a[k][0]})

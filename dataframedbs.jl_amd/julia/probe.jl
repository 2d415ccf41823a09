# probe.jl — executes the shim against the reference's own path (run by tests/test_gpu_julia_probe.py when a `julia` with
# DataFrameDBs.jl and DataFrames.jl installed exists on the GPU box; the build image has no Julia).
#
#   julia dataframedbs.jl_amd/julia/probe.jl [scratch_dir]
#
# Writes a small table with the REFERENCE's writer, evaluates a set of views with the stock methods, calls enable!(), evaluates the same
# views again (now on the MI355X) and compares: DataFrames bit for bit, nrow, one-column consumers, the fallback for untraceable closures.
using DataFrames, DataFrameDBs, Statistics
include(joinpath(@__DIR__, "DataFrameDBsAMD.jl"))
using .DataFrameDBsAMD

dir = length(ARGS) >= 1 ? ARGS[1] : mktempdir()
path = joinpath(dir, "probe_table")
n = 200_003
df = DataFrame(a = collect(Int64, 1:n), b = [Float64(i % 1000) / 7 for i in 1:n], s = [("apple", "sony", "dell", "asus")[i % 4 + 1] for i in 1:n],
               f = [i % 3 == 0 for i in 1:n])
t = create_table(path; from = df, block_size = 4096, show_progress = false)

views() = Dict(
    "pred_closure" => t[:a => x -> x > 150_000, :],
    "broadcast_and" => t[(t.a .> 1000) .& (t.s .== "sony"), [:a, :b]],
    "range_after_pred" => t[t.b .< 50.0, :][11:3:9000, :],
    "chained_comparison_falls_back" => t[:a => a -> 65 > a > 34, :],          # `&&` on a traced value: Unsupported -> stock path
    "bool_column" => t[t.f, [:a, :f]],
)
stock = Dict(k => (materialize(v), DataFrameDBs.nrow(v)) for (k, v) in views())
col = t[t.a .> 100_000, :].b
stock_col = (materialize(col), sum(col), mean(col), minimum(col), maximum(col), length(unique(t.s)), collect(col)[1:10])

DataFrameDBsAMD.enable!()
bad = String[]
for (k, v) in views()
    got = (materialize(v), DataFrameDBs.nrow(v))
    (isequal(got[1], stock[k][1]) && got[2] == stock[k][2]) || push!(bad, k)
    eltype.(eachcol(got[1])) == eltype.(eachcol(stock[k][1])) || push!(bad, k * " (eltypes)")
    typeof.(eachcol(got[1])) == typeof.(eachcol(stock[k][1])) || push!(bad, k * " (containers)")
end
col = t[t.a .> 100_000, :].b
got_col = (materialize(col), sum(col), mean(col), minimum(col), maximum(col), length(unique(t.s)), collect(col)[1:10])
got_col[1] == stock_col[1] || push!(bad, "materialize(::DFColumn)")
abs(got_col[2] - stock_col[2]) <= length(got_col[1]) * eps(Float64) * sum(abs, stock_col[1]) || push!(bad, "sum tolerance")
abs(got_col[3] - stock_col[3]) <= eps(Float64) * sum(abs, stock_col[1]) || push!(bad, "mean tolerance")
got_col[4:7] == stock_col[4:7] || push!(bad, "minimum / maximum / unique / collect")
dest = zeros(Float64, length(got_col[1])); dest .= col
dest == stock_col[1] || push!(bad, "broadcast copyto!")
if isempty(bad)
    println("PROBE OK: ", length(stock), " views and the one-column consumers equal the stock path")
else
    println("PROBE FAILED: ", join(bad, "; "))
    exit(1)
end

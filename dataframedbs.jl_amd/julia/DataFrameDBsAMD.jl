# DataFrameDBsAMD.jl — the reference-side binding of libdfdb_hip.so: a thin `ccall` shim that keeps
# DataFrameDBs.jl's DFTable / DFView API (selection(), projection(), materialize(), nrow/size) and sends the
# block-streamed decode + selection + projection + materialize hot path to the MI355X engine.
#
# NOT EXECUTED IN THE BUILD IMAGE (julia is not installed there: SURVEY.md "Facts").  It is the binding a
# DataFrameDBs.jl maintainer would add; INTEGRATION.md walks through it.  Every ccall below targets a symbol
# declared in include/dfdb.h, whose comments name the Julia method each one stands in for; julia/STATIC_REVIEW.md walks every
# ccall signature against that header.
#
#   using DataFrameDBs, DataFrameDBsAMD
#   DataFrameDBsAMD.enable!()                 # route materialize/nrow of DFView through the GPU
#   t = open_table("ecommerce")
#   materialize(t[(t.price .> 100) .& (t.brand .== "apple"), [:user_id, :price]])
#
# Lowering: a `BlockBroadcasting{RT,F,Args}` tree (src/tables/broadcast.jl:6-17) becomes the postfix IR of
# include/dfdb_ir.h.  Named Base functions map 1:1; closures (`:a => x -> x > c`) are traced by calling
# them on a symbolic `Tr` value.  Anything outside the IR op set makes the engine answer
# DFDB_ERR_UNSUPPORTED (or the tracer throw), and the call falls back to the stock Julia path.
#
# Closure grammar the tracer accepts: any composition of the functions in OPS / UNARY, `in(x, vector)`, `Float64(x)`, over the
# closure's arguments and literal numbers / strings / Dates / Chars — `x -> x > c`, `(a, b) -> (a % 10 == 0) & (b < 5.0)`,
# `x -> startswith(x, "so") | (sizeof(x) > 4)`.  A trace cannot pass anything that needs a real `Bool` out of a traced value — `&&`,
# `||`, `?:`, and therefore chained comparisons (`65 > a > 34` lowers to `(65 > a) && (a > 34)`; test/selection.jl:53): for those the
# closure's LOWERED CODE is walked symbolically (lower_closure below: `goto if not` diamonds whose arms are Bool-valued and cannot
# raise become `&` / `|` / `!`).  What is still outside — numeric `ifelse` / `?:`, loops, calls to functions outside the tables, arms
# with `÷` / `%` — raises `Unsupported`: the view is evaluated by the reference's own Julia path (same results, its speed) and the
# fallback is logged once per reason.  The DFColumn-broadcasting form of the same predicate, `(65 .> t.a) .& (t.a .> 34)`, never goes
# through a closure: its BlockBroadcasting tree names `>` and `&` directly and lowers 1:1.
module DataFrameDBsAMD

import Dates
import Statistics
using DataFrameDBs
using DataFrameDBs: DFTable, DFView, DFColumn, ColRef, BlockBroadcasting, SelectionQueue, Projection
import DataFrames

const LIB = joinpath(@__DIR__, "..", "libdfdb_hip.so")

# ---------------------------------------------------------------- status codes (include/dfdb.h)
const OK = Int32(0)
struct Unsupported <: Exception
    msg::String
end

function last_error()
    buf = Vector{UInt8}(undef, 1024)
    ccall((:dfdb_last_error, LIB), Int32, (Ptr{UInt8}, Csize_t), buf, 1024)
    GC.@preserve buf unsafe_string(pointer(buf))
end

function check(rc::Int32)
    rc == OK && return
    msg = last_error()
    # InexactError (Int8(300), UInt64(-1) inside a predicate) travels as DFDB_ERR_ARGUMENT with its name in the text (include/dfdb.h)
    rc == 1 && startswith(msg, "InexactError") && throw(InexactError(:convert, Integer, msg))
    rc == 1 && throw(ArgumentError(msg))
    rc == 4 && throw(KeyError(msg))
    rc == 5 && throw(BoundsError(msg))
    rc == 6 && throw(DivideError())
    rc == 7 && throw(Unsupported(msg))
    rc == 9 && throw(OutOfMemoryError())      # DFDB_ERR_NOMEM: with_query / with_gquery unload the table and answer block-streamed instead (below)
    error(msg)                       # IO / FORMAT / DEVICE -> ErrorException, like filesystem.jl:50-57
end

# ---------------------------------------------------------------- dtypes (include/dfdb_ir.h)
const DT = Dict{DataType,UInt8}(Int8 => 1, Int16 => 2, Int32 => 3, Int64 => 4, UInt8 => 5, UInt16 => 6, UInt32 => 7,
                                UInt64 => 8, Float32 => 9, Float64 => 10, Bool => 11, String => 12)
const JT = Dict(v => k for (k, v) in DT)
const NULLABLE = 0x80
dtype(::Type{Union{T,Missing}}) where {T} = DT[T] | NULLABLE
dtype(::Type{T}) where {T} = DT[T]
jltype(dt::Integer) = (dt & NULLABLE) != 0 ? Union{JT[UInt8(dt & 0x3f)],Missing} : JT[UInt8(dt & 0x3f)]
# the element type the REFERENCE declares for a projection column (Date, DateTime, Time, Char come back as Int64 / UInt32)
relabel(::Type{T}, v::Vector) where {T} = v
relabel(::Type{T}, v::Vector{Int64}) where {T<:Union{Dates.Date,Dates.DateTime,Dates.Time}} = T === Dates.Time ? Dates.Time.(Dates.Nanosecond.(v)) : reinterpret(T, v)
relabel(::Type{Char}, v::Vector{UInt32}) = reinterpret(Char, v)

# ---------------------------------------------------------------- IR emission
const OPS = Dict{Any,UInt8}(
    (+) => 0x10, (-) => 0x11, (*) => 0x12, (/) => 0x13, div => 0x14, (÷) => 0x14, rem => 0x15, (%) => 0x15, mod => 0x16,
    min => 0x19, max => 0x1a,
    (==) => 0x20, (!=) => 0x21, (<) => 0x22, (<=) => 0x23, (>) => 0x24, (>=) => 0x25,
    (&) => 0x30, (|) => 0x31, xor => 0x32,
    startswith => 0x41, endswith => 0x42, coalesce => 0x45)
const UNARY = Dict{Any,UInt8}((-) => 0x17, abs => 0x18, (!) => 0x33, ismissing => 0x43, sizeof => 0x44)

emit_col(io, ord::Integer) = (write(io, 0x01); write(io, UInt32(ord)))   # checked: a 0-based column ordinal, 0 <= ord < ncol
function emit_const(io, v::T) where {T<:Union{Int8,Int16,Int32,Int64,UInt8,UInt16,UInt32,UInt64,Float32,Float64,Bool}}
    write(io, 0x02); write(io, DT[T])
    # integers travel as their low 64 bits: `v % Int64` wraps (a UInt64 >= 2^63 keeps its bit pattern), `Int64(v)` would throw InexactError for it
    T == Float64 ? write(io, v) : T == Float32 ? (write(io, v); write(io, UInt32(0))) : write(io, v % Int64)
end
emit_const(io, s::AbstractString) = (write(io, 0x03); write(io, UInt32(sizeof(s))); write(io, String(s)))
# Date / DateTime / Time / Char columns are integer columns to the engine (dfdb_colinfo.logical): constants travel as the
# integers Julia itself stores (Dates.value: days / milliseconds / nanoseconds; reinterpret(UInt32, ::Char))
emit_const(io, d::Union{Dates.Date,Dates.DateTime,Dates.Time}) = emit_const(io, Int64(Dates.value(d)))
emit_const(io, c::Char) = emit_const(io, reinterpret(UInt32, c))
emit_const(io, r::Base.RefValue) = emit_const(io, r[])
function emit_const(io, v::AbstractVector{T}) where {T<:Union{Integer,AbstractFloat}}     # Ref([1,11,21]) for in.()
    E = T <: AbstractFloat ? Float64 : Int64
    # the IR's integer sets are Int64: a member above typemax(Int64) has no place in one (the stock path answers instead)
    T <: Unsigned && any(x -> x > typemax(Int64), v) && throw(Unsupported("in.() over unsigned values above typemax(Int64)"))
    write(io, 0x04); write(io, DT[E]); write(io, UInt32(length(v))); foreach(x -> write(io, convert(E, x)), v)
end
emit_const(io, v) = throw(Unsupported("constant of type $(typeof(v)) is outside the IR"))

# symbolic value used to trace closures: every operation appends postfix IR
struct Tr
    code::Vector{UInt8}
end
leaf(f) = (io = IOBuffer(); f(io); Tr(take!(io)))
tr(x::Tr) = x
tr(x) = leaf(io -> emit_const(io, x))
cat2(a, b, op::UInt8) = Tr(vcat(tr(a).code, tr(b).code, op))
for (f, op) in OPS
    fn = f
    @eval Base.$(nameof(fn))(a::Tr, b::Tr) = cat2(a, b, $op)
    @eval Base.$(nameof(fn))(a::Tr, b::Union{Number,AbstractString}) = cat2(a, b, $op)
    @eval Base.$(nameof(fn))(a::Union{Number,AbstractString}, b::Tr) = cat2(a, b, $op)
end
for (f, op) in UNARY
    @eval Base.$(nameof(f))(a::Tr) = Tr(vcat(a.code, $op))
end
Base.in(a::Tr, s::AbstractVector) = Tr(vcat(a.code, tr(s).code, 0x40))
# a set of strings: any(==(x), s) spelled out, (a == s1) | (a == s2) | ... — one bit-table lookup over a dictionary-coded column (K9)
Base.in(a::Tr, s::AbstractVector{<:AbstractString}) = isempty(s) ? throw(Unsupported("in.() over an empty set of strings")) : reduce((x, y) -> x | y, [a == v for v in s])
Base.Float64(a::Tr) = Tr(vcat(a.code, 0x50, DT[Float64]))
Base.convert(::Type{Float64}, a::Tr) = Float64(a)

# BlockBroadcasting / ColRef -> Tr.  `ord` maps a column Symbol to its 0-based table ordinal.
lower(c::ColRef, ord) = leaf(io -> emit_col(io, ord[c.name]))
lower(x, ord) = tr(x)
function lower(b::BlockBroadcasting, ord)
    args = map(a -> lower(a, ord), b.args)
    f = b.f
    if f === in && length(args) == 2
        return Tr(vcat(args[1].code, args[2].code, 0x40))
    elseif haskey(OPS, f) && length(args) == 2
        return cat2(args[1], args[2], OPS[f])
    elseif haskey(UNARY, f) && length(args) == 1
        return Tr(vcat(args[1].code, UNARY[f]))
    end
    try
        r = f(args...)                       # closure: trace it
        r isa Tr || throw(Unsupported("closure did not reduce to IR (returned $(typeof(r)))"))
        return r
    catch e
        e isa Unsupported && rethrow()
        # the trace died: most often on `&&` / `||` / a chained comparison / `?:`, which need a real Bool out of a traced value
        # (`65 > a > 34`, the reference's own test predicate: test/selection.jl:53).  Walk the closure's lowered code instead.
        try
            return lower_closure(f, collect(Any, args))
        catch e2
            e2 isa Unsupported && rethrow()
            throw(Unsupported("cannot trace $(f): $(e); cannot walk its lowered code either: $(e2)"))
        end
    end
end

# ---------------------------------------------------------------- closures with control flow: a symbolic walk over the lowered code
# Julia lowers `a && b`, `a || b`, `c ? x : y` and `65 > a > 34` (= `(65 > a) && (a > 34)`) to `goto #k if not %c` diamonds.  `Base.code_lowered(f)`
# gives that code before any type inference; the walk below evaluates it on the same symbolic `Tr` values the tracer uses:
#   * a statement whose operands hold no Tr is simply executed (constants, `getfield(#self#, :c)` for a captured variable, arithmetic on literals);
#   * a call with a Tr operand goes through the Tr methods above, exactly as in a trace;
#   * `goto #k if not %c` with a Tr condition FORKS the walk: both arms run to their `return` (a slot assigned in one arm and read after the join is
#     simply carried along: the continuation is walked once per arm), and the value is  (c & then) | (!c & else)  — `c & then` when the else arm is the
#     literal `false` (that is `&&`), `c | else` when the then arm is the literal `true` (`||`).
# Two conditions keep this exact: both arms must be Bool-valued (the IR has no select of numbers: `x > 0 ? x : -x` stays on the CPU path), and an arm that
# Julia evaluates only conditionally must not be able to raise — `x != 0 && 10 % x == 1` is fine in Julia for x == 0 and a DivideError once `&` evaluates
# both sides — so arms containing `÷`, `%`, `mod` or an integer conversion are refused.  Loops (a backward jump on a Tr condition) are refused too.
const RAISING_OPS = (0x14, 0x15, 0x16, 0x50)
may_raise(t::Tr) = any(b -> b in RAISING_OPS, t.code)          # (a constant's payload byte can look like one of them: a false alarm only sends the view to the CPU path)
may_raise(x) = false
const WALK_BUDGET = 4096                                       # statements walked in all arms together

mutable struct Walk
    ci::Core.CodeInfo
    steps::Int
end

function lower_closure(f, args::Vector{Any})
    cis = Base.code_lowered(f)
    length(cis) == 1 || throw(Unsupported("$(f) has $(length(cis)) methods: only a closure with one method is walked"))
    ci = cis[1]
    nslots = length(ci.slotnames)
    length(args) + 1 <= nslots || throw(Unsupported("argument count does not match the lowered code of $(f)"))
    slots = Vector{Any}(undef, nslots)
    slots[1] = f                                               # #self#: captured variables are its fields
    for (i, a) in enumerate(args); slots[i + 1] = a; end
    r = walk(Walk(ci, 0), 1, Dict{Int,Any}(), slots)
    r isa Tr || throw(Unsupported("the lowered code of $(f) did not reduce to IR (it returns $(typeof(r)))"))
    r
end

function wvalue(w::Walk, x, ssa, slots)
    x isa Core.SSAValue && return ssa[x.id]
    x isa Core.SlotNumber && return isassigned(slots, x.id) ? slots[x.id] : throw(Unsupported("a slot is read before it is assigned"))
    x isa Core.Argument && return slots[x.n]
    x isa GlobalRef && return getfield(x.mod, x.name)
    x isa QuoteNode && return x.value
    x isa Expr && return wexpr(w, x, ssa, slots)
    x
end

function wexpr(w::Walk, e::Expr, ssa, slots)
    if e.head === :call
        fn = wvalue(w, e.args[1], ssa, slots)
        av = Any[wvalue(w, a, ssa, slots) for a in e.args[2:end]]
        # with or without a Tr among the operands this is an ordinary call: the Tr methods do the emitting
        try
            return fn(av...)
        catch err
            err isa Unsupported && rethrow()
            throw(Unsupported("$(fn) is outside the IR operator set: $(err)"))
        end
    elseif e.head === :static_parameter || e.head === :the_exception || e.head === :foreigncall || e.head === :new
        throw(Unsupported("lowered code uses :$(e.head)"))
    elseif e.head === :meta || e.head === :loopinfo || e.head === :inbounds || e.head === :boundscheck
        return nothing
    end
    throw(Unsupported("lowered code uses :$(e.head)"))
end

select_bool(c::Tr, t::Bool, e::Bool) = t == e ? t : (t ? c : !c)
select_bool(c::Tr, t::Tr, e::Bool) = e ? (!c | t) : (c & t)                   # e == false: `c && t`
select_bool(c::Tr, t::Bool, e::Tr) = t ? (c | e) : (!c & e)                   # t == true:  `c || e`
select_bool(c::Tr, t::Tr, e::Tr) = (c & t) | (!c & e)
select_bool(c, t, e) = throw(Unsupported("a branch on a column value must choose between Bool values (got $(typeof(t)) and $(typeof(e)))"))

function walk(w::Walk, pc::Int, ssa::Dict{Int,Any}, slots::Vector{Any})
    code = w.ci.code
    while true
        (w.steps += 1) > WALK_BUDGET && throw(Unsupported("the lowered code is too long to walk"))
        pc <= length(code) || throw(Unsupported("the lowered code ends without a return"))
        st = code[pc]
        if st isa Core.ReturnNode
            isdefined(st, :val) || throw(Unsupported("unreachable code reached"))
            return wvalue(w, st.val, ssa, slots)
        elseif st isa Core.GotoNode
            st.label > pc || throw(Unsupported("the closure loops"))
            pc = st.label
        elseif st isa Core.GotoIfNot
            c = wvalue(w, st.cond, ssa, slots)
            if c isa Bool                                      # decided while walking (a literal, a captured flag)
                pc = c ? pc + 1 : st.dest
            elseif c isa Tr
                st.dest > pc || throw(Unsupported("the closure loops on a column value"))
                t = walk(w, pc + 1, copy(ssa), copy(slots))
                e = walk(w, st.dest, copy(ssa), copy(slots))
                (may_raise(t) || may_raise(e)) && throw(Unsupported("an arm of a short-circuit branch could raise: `&` would evaluate it for every row"))
                return select_bool(c, t, e)
            else
                throw(Unsupported("a branch on a $(typeof(c))"))
            end
        elseif st isa Core.NewvarNode || st === nothing || st isa LineNumberNode
            pc += 1
        elseif st isa Expr && st.head === :(=)
            lhs = st.args[1]
            lhs isa Core.SlotNumber || throw(Unsupported("assignment to $(typeof(lhs))"))
            slots[lhs.id] = wvalue(w, st.args[2], ssa, slots)
            ssa[pc] = slots[lhs.id]
            pc += 1
        else
            ssa[pc] = wvalue(w, st, ssa, slots)
            pc += 1
        end
    end
end

# ---------------------------------------------------------------- devices: one GPU, or a block-range sharded group of all of them
# With more than one GPU visible (and DFDB_GPUS != "1") every opened table is sharded by block range over ALL of them
# (dfdb_group_*: a host thread per GPU inside the library, RCCL all-reduce for nrow / sum / minimum / maximum, rank-order
# concatenation for materialize, per-shard device reduction + merge by key in rank order for unique / groupreduce; include/dfdb.h
# "multi-GPU groups").  The write side uses GPU 0 alone.
mutable struct Device
    ctx::Ptr{Cvoid}                       # single-GPU context (GPU 0)
    group::Ptr{Cvoid}                     # dfdb_group* or C_NULL
    tables::Dict{String,Ptr{Cvoid}}       # dfdb_table*  per opened DFTable path (GPU 0, whole table)
    gtables::Dict{String,Ptr{Cvoid}}      # dfdb_gtable* per opened DFTable path (sharded)
end
const DEV = Ref{Union{Nothing,Device}}(nothing)

function ngpus()
    n = Ref{Int32}(0)
    check(ccall((:dfdb_device_count, LIB), Int32, (Ptr{Int32},), n))
    want = tryparse(Int, get(ENV, "DFDB_GPUS", ""))
    want === nothing ? Int(n[]) : min(Int(n[]), want)   # checked: Int32 -> Int only widens
end

function device()
    if DEV[] === nothing
        ctx = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dfdb_ctx_create, LIB), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), 0, C_NULL, ctx))
        # Julia's exit hooks run before the C runtime's: the engine's background compiler (hipRTC) must be idle by then (include/dfdb.h: dfdb_shutdown)
        atexit(() -> check(ccall((:dfdb_shutdown, LIB), Int32, ())))
        # String columns with at most DFDB_STRING_DICTIONARY (default 4096) distinct values get 16-bit codes beside their flat form when they are
        # loaded: `t.brand .== "sony"` then scans 2 bytes per row (include/dfdb.h: dfdb_table_build_dictionary); 0 turns it off
        dictn = something(tryparse(Int, get(ENV, "DFDB_STRING_DICTIONARY", "")), 4096)
        check(ccall((:dfdb_ctx_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx[], "string_dictionary", dictn))
        # DFDB_KEEP_COMPRESSED=1: plain fixed-width columns keep their LZ4 blocks in HBM beside the decoded array (gpu_redecode! below; include/dfdb.h: keep_compressed);
        # DFDB_KEEP_COMPRESSED=2: COMPRESSED-ONLY — the blocks and nothing decoded (5.15 GB instead of 8 per 1e9 Int64 rows): predicates run inside the block decoder,
        # projections decode the blocks that kept a row; every result is the same, scans cost ~10 x more (INTEGRATION.md section 6)
        keepc = something(tryparse(Int, get(ENV, "DFDB_KEEP_COMPRESSED", "")), 0)
        check(ccall((:dfdb_ctx_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx[], "keep_compressed", keepc))
        # DFDB_HBM_BUDGET_MB: what ONE table may hold in HBM (per GPU when sharded); 0 / unset = whatever fits 80 % of the free HBM.  Only the columns a view
        # needs are ever loaded (dfdb_query_prepare: required_columns, view.jl:183-190); a view whose columns do not fit is answered BLOCK-STREAMED inside the
        # library, O(chunk) of HBM like the reference's O(block) (blocksiterator.jl:98-121; include/dfdb.h "out of core behind the ordinary entry points");
        # DFDB_OOC_CHUNK_BLOCKS = blocks per chunk of those streams (default 512)
        budget = something(tryparse(Int, get(ENV, "DFDB_HBM_BUDGET_MB", "")), 0)
        chunkb = something(tryparse(Int, get(ENV, "DFDB_OOC_CHUNK_BLOCKS", "")), 512)
        check(ccall((:dfdb_ctx_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx[], "hbm_budget_mb", budget))
        check(ccall((:dfdb_ctx_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx[], "ooc_chunk_blocks", chunkb))
        grp = Ref{Ptr{Cvoid}}(C_NULL)
        n = ngpus()
        if n > 1
            ids = Int32[i for i in 0:n-1]
            # exchange = 0 (DFDB_EXCHANGE_AUTO): RCCL over xGMI for distinct devices
            GC.@preserve ids check(ccall((:dfdb_group_create, LIB), Int32, (Ptr{Int32}, Int32, Int32, Ptr{Ptr{Cvoid}}), ids, n, 0, grp))
            check(ccall((:dfdb_group_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), grp[], "string_dictionary", dictn))
            check(ccall((:dfdb_group_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), grp[], "keep_compressed", keepc))
            check(ccall((:dfdb_group_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), grp[], "hbm_budget_mb", budget))
            check(ccall((:dfdb_group_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), grp[], "ooc_chunk_blocks", chunkb))
        end
        DEV[] = Device(ctx[], grp[], Dict{String,Ptr{Cvoid}}(), Dict{String,Ptr{Cvoid}}())
    end
    DEV[]
end
sharded() = device().group != C_NULL

# open_table: the meta and the column headers only (creators.jl:7-16) — NO column is read here.  Which columns become resident is decided per view, by
# dfdb_query_prepare in with_query below: exactly required_columns(v) (view.jl:183-190, blocksiterator.jl:20-33), and only when they fit.
function device_table(t::DFTable)
    d = device()
    get!(d.tables, t.path) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dfdb_table_open, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Ptr{Cvoid}}), d.ctx, t.path, h))
        h[]
    end
end
# the same, sharded: the block ranges are laid out over the GPUs at open, nothing is read (dfdb_group_query_prepare loads per view)
function device_gtable(t::DFTable)
    d = device()
    get!(d.gtables, t.path) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dfdb_group_table_open, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Ptr{Cvoid}}), d.group, t.path, h))
        h[]
    end
end

ordinals(t::DFTable) = Dict(m.name => i - 1 for (i, m) in enumerate(t.meta.columns))
struct PredCode                            # a lowered predicate stage (told apart from an index-vector stage by type)
    code::Vector{UInt8}
end

# ---------------------------------------------------------------- DFView -> dfdb_query / dfdb_gquery
# One builder, generated twice: the group entry points take the same arguments as the single-GPU ones (include/dfdb.h).
for (fname, tabfn, NEW, FREE, RANGE, INTEGER, INDICES, PRED, PROJ, PREPARE, UNLOAD, RESET) in (
        (:with_query, :device_table, :dfdb_query_new, :dfdb_query_free, :dfdb_query_add_range, :dfdb_query_add_integer,
         :dfdb_query_add_indices, :dfdb_query_add_predicate, :dfdb_query_set_projection, :dfdb_query_prepare, :dfdb_table_unload, :dfdb_query_reset),
        (:with_gquery, :device_gtable, :dfdb_group_query_new, :dfdb_group_query_free, :dfdb_group_query_add_range, :dfdb_group_query_add_integer,
         :dfdb_group_query_add_indices, :dfdb_group_query_add_predicate, :dfdb_group_query_set_projection, :dfdb_group_query_prepare,
         :dfdb_group_table_unload, :dfdb_group_query_reset))
    @eval function $fname(f, v::DFView)
        ord = ordinals(v.table)
        # lower everything BEFORE touching the device: an untraceable closure must fall back without side effects
        stages = Any[el isa BlockBroadcasting ? PredCode(lower(el, ord).code) : el for el in v.selection.queue]
        names = [string(k) for k in keys(v.projection)]
        codes = [lower(c, ord).code for c in values(v.projection.cols)]
        th = $tabfn(v.table)
        q = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall(($(QuoteNode(NEW)), LIB), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), th, q))
        try
            for el in stages                      # SelectionQueue stages, already composed by DataFrameDBs.add
                if el isa PredCode
                    code = el.code
                    GC.@preserve code check(ccall(($(QuoteNode(PRED)), LIB), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Csize_t), q[], code, length(code)))
                elseif el isa AbstractRange
                    check(ccall(($(QuoteNode(RANGE)), LIB), Int32, (Ptr{Cvoid}, Int64, Int64, Int64), q[], first(el), step(el), last(el)))
                elseif el isa Integer
                    check(ccall(($(QuoteNode(INTEGER)), LIB), Int32, (Ptr{Cvoid}, Int64), q[], el))
                else
                    idx = collect(Int64, el)
                    GC.@preserve idx check(ccall(($(QuoteNode(INDICES)), LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int64), q[], idx, length(idx)))
                end
            end
            lens = Csize_t[length(c) for c in codes]
            GC.@preserve names codes begin
                np = [pointer(n) for n in names]; cp = [pointer(c) for c in codes]   # rooted by names / codes (the GC.@preserve around this block)
                check(ccall(($(QuoteNode(PROJ)), LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Ptr{UInt8}}, Ptr{Ptr{UInt8}}, Ptr{Csize_t}),
                            q[], length(names), np, cp, lens))
            end
            # required_columns(v) — and no others — into HBM if they fit the budget (decoded, else compressed-only); if not, nothing is loaded and every
            # call f makes is answered block-streamed inside the library.  how: 0 resident already, 1 loaded, 2 compressed-only, 3 streamed.
            how = Ref{Int32}(0)
            check(ccall(($(QuoteNode(PREPARE)), LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}), q[], how))
            try
                return f(q[])
            catch e
                # HBM ran out while answering (result buffers, a transient decode): give the table's columns back and answer once more, block-streamed —
                # the reference's memory behaviour is O(block), an out-of-memory error is not among its outcomes (docs/src/index.md:182,192)
                e isa OutOfMemoryError || rethrow()
                check(ccall(($(QuoteNode(UNLOAD)), LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Int32), th, C_NULL, 0))
                check(ccall(($(QuoteNode(RESET)), LIB), Int32, (Ptr{Cvoid},), q[]))
                return f(q[])
            end
        finally
            ccall(($(QuoteNode(FREE)), LIB), Int32, (Ptr{Cvoid},), q[])
        end
    end
end

# struct dfdb_outcol (include/dfdb.h)
struct OutCol
    data::Ptr{Cvoid}; bytes::Ptr{UInt8}; missing::Ptr{UInt8}; bytes_cap::Int64
    memkind::Int32; dtype::Int32; count::Int64; nbytes::Int64
end

"nrow(v) on the device(s) (view.jl:192-206): one predicate scan per GPU, one RCCL all-reduce of the count when sharded."
function gpu_nrow(v::DFView)
    length(v.projection) == 0 && return 0            # isempty(it.streams): nothing to read (blocksiterator.jl:101)
    n = Ref{Int64}(0)
    if sharded()
        with_gquery(v) do q
            check(ccall((:dfdb_group_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n))
        end
    else
        with_query(v) do q
            check(ccall((:dfdb_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n))
        end
    end
    n[]
end

# shard 0's ordinary query handle: column types are the same on every shard
function gquery_coltype(gq, i)
    q0 = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dfdb_group_query_shard, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Ptr{Cvoid}}), gq, 0, q0))
    dt = Ref{Int32}(0)
    check(ccall((:dfdb_query_coltype, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}), q0[], i, dt))
    dt[]
end

# caller-owned Julia vectors for the n selected rows of every projection column; `strbytes(i)` = string bytes column i needs
function alloc_outputs(v::DFView, n::Integer, coltype, strbytes)
    ncols = length(v.projection)
    outs = Vector{OutCol}(undef, ncols)
    bufs = Any[]
    for i in 1:ncols
        dt = coltype(i - 1)
        T = jltype(dt)
        if (dt & 0x3f) == 12                                     # String: sizes + byte arena
            nb = strbytes(i - 1)
            sizes = Vector{Int32}(undef, n); arena = Vector{UInt8}(undef, nb)
            push!(bufs, (T, sizes, arena, nothing))
            outs[i] = OutCol(pointer(sizes), pointer(arena), C_NULL, nb, 0, 0, 0, 0)   # rooted by bufs (every caller: GC.@preserve bufs outs around its ccall)
        else
            B = Base.nonmissingtype(T)
            vals = Vector{B}(undef, n); miss = T === B ? nothing : Vector{UInt8}(undef, n)
            push!(bufs, (T, vals, nothing, miss))
            outs[i] = OutCol(pointer(vals), C_NULL, miss === nothing ? C_NULL : pointer(miss), 0, 0, 0, 0, 0)   # rooted by bufs (as above)
        end
    end
    outs, bufs
end

# engine buffers -> the containers the reference materialises into (make_materialization: Vector{T}, BitVector for Bool;
# FlatStrings -> Vector{String} like projection.jl:99-100), relabelled to the column's declared element type
function finish_columns(v::DFView, bufs)
    decl = [DataFrameDBs.coltype(v.projection, i) for i in 1:length(v.projection)]
    map(enumerate(bufs)) do (i, (T, a, arena, miss))
        if arena !== nothing
            res = Vector{T}(undef, length(a)); o = 0
            for (k, s) in enumerate(a)
                res[k] = s < 0 ? missing : (str = GC.@preserve arena unsafe_string(pointer(arena) + o, s); o += s; str)
            end
            res
        elseif miss !== nothing
            T[m != 0 ? missing : x for (x, m) in zip(a, miss)]
        elseif T === Bool
            BitVector(a)                                          # make_materialization(::Type{Bool}) = BitVector (materialization.jl:10)
        else
            relabel(Base.nonmissingtype(decl[i]), a)
        end
    end
end

"materialize(v) on the device(s) (materialization.jl:27-40): the selection is evaluated once, outputs are caller-owned Julia vectors."
function gpu_materialize_columns(v::DFView; first_occurrences_of::Int = -1)
    if sharded() && first_occurrences_of < 0
        return with_gquery(v) do q
            check(ccall((:dfdb_group_query_hint_materialize, LIB), Int32, (Ptr{Cvoid}, Int32), q, 1))
            n = Ref{Int64}(0)
            check(ccall((:dfdb_group_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n))     # one process drives every shard: the local rows are all rows
            outs, bufs = alloc_outputs(v, n[], i -> gquery_coltype(q, i), i -> begin
                nb = Ref{Int64}(0)
                check(ccall((:dfdb_group_result_string_bytes, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}), q, i, nb)); nb[]
            end)
            GC.@preserve bufs outs check(ccall((:dfdb_group_materialize, LIB), Int32, (Ptr{Cvoid}, Ptr{OutCol}, Int32), q, outs, length(outs)))
            finish_columns(v, bufs)
        end
    end
    with_query(v) do q
        check(ccall((:dfdb_query_hint_materialize, LIB), Int32, (Ptr{Cvoid}, Int32), q, 1))   # the count below is the scan: let it keep projected predicate columns
        # unique: narrow the selection to the first occurrence of every value of that projection column (Julia's order)
        first_occurrences_of >= 0 && check(ccall((:dfdb_query_unique, LIB), Int32, (Ptr{Cvoid}, Int32), q, first_occurrences_of))
        n = Ref{Int64}(0)
        check(ccall((:dfdb_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n))
        outs, bufs = alloc_outputs(v, n[], i -> begin
            dt = Ref{Int32}(0)
            check(ccall((:dfdb_query_coltype, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}), q, i, dt)); dt[]
        end, i -> begin
            nb = Ref{Int64}(0)
            check(ccall((:dfdb_result_string_bytes, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}), q, i, nb)); nb[]
        end)
        GC.@preserve bufs outs check(ccall((:dfdb_materialize, LIB), Int32, (Ptr{Cvoid}, Ptr{OutCol}, Int32), q, outs, length(outs)))
        finish_columns(v, bufs)
    end
end

function gpu_materialize(v::DFView; first_occurrences_of::Int = -1)
    cols = gpu_materialize_columns(v; first_occurrences_of = first_occurrences_of)
    DataFrames.DataFrame(collect(cols), collect(keys(v.projection)), copycols = false)
end

"materialize(c::DFColumn) on the device(s) (materialization.jl:46-52): the one projection column as a Vector{T} / BitVector."
gpu_materialize(c::DFColumn) = gpu_materialize_columns(c.view)[1]

"""
    gpu_redecode!(t::DFTable, col::Symbol) -> Int

Run the block decoder again over the LZ4 blocks column `col` kept in HBM (DFDB_KEEP_COMPRESSED=1 when the table was first touched) — the device-side
equivalent of re-reading the column through `BlockStream` (`read_block`, src/io/BlockStreams.jl:101-119) — and return how many blocks did NOT decode to
their stored size: 0 is the reference's `@assert size == sizes.origin "decompression error"` (:112) holding for every block.  Single-GPU tables only.
"""
function gpu_redecode!(t::DFTable, col::Symbol)
    sharded() && throw(Unsupported("gpu_redecode! over a sharded table"))
    h = device_table(t)
    ord = ordinals(t)[col] % Int32        # checked: a 0-based column ordinal, 0 <= ord < ncol
    check(ccall((:dfdb_table_decode_resident, LIB), Int32, (Ptr{Cvoid}, Int32), h, ord))
    bad = Ref{Int64}(0)
    check(ccall((:dfdb_table_decode_status, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}), h, ord, bad))
    Int(bad[])                            # checked: Int64 -> Int is the identity on a 64-bit host
end

"unique(col::DFColumn) on the device(s) (docs/src/index.md:171-182, 479-487): distinct values in order of first appearance over the WHOLE table —
sharded: every GPU reduces its block range, the per-shard distinct sets are merged by key in rank order inside the library (dfdb_group_query_unique)."
function gpu_unique(c::DFColumn)
    sharded() || return gpu_materialize_columns(c.view; first_occurrences_of = 0)[1]
    with_gquery(c.view) do q
        n = Ref{Int64}(0); kb = Ref{Int64}(0)
        check(ccall((:dfdb_group_query_unique, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}, Ptr{Int64}), q, 0, n, kb))
        outs, bufs = alloc_outputs(c.view, n[], i -> gquery_coltype(q, i), i -> kb[])
        GC.@preserve bufs outs check(ccall((:dfdb_group_query_unique_fetch, LIB), Int32, (Ptr{Cvoid}, Ptr{OutCol}), q, outs))
        finish_columns(c.view, bufs)[1]
    end
end

"""
sum / minimum / maximum of a DFColumn on the device(s) (the reference iterates the column: column.jl:102-126, docs/src/index.md:503-509).
The hint lets the scan that evaluates the selection reduce the selected values while it holds them (dfdb_query_hint_aggregate);
the count comes out of the same execution.  op: 1 = sum, 2 = minimum, 3 = maximum (DFDB_AGG_*).  Returns (value, count).
Integer results are exact (wrapping 64-bit sums like Julia's); a Float64 sum is within n*eps*sum|x| of the left-to-right sum.
"""
function gpu_aggregate(c::DFColumn{T}, op::Integer) where {T}
    T <: Union{Int8,Int16,Int32,Int64,UInt8,UInt16,UInt32,UInt64,Bool,Float64} || throw(Unsupported("aggregate over $(T)"))
    vi = Ref{Int64}(0); vf = Ref{Float64}(0.0); n = Ref{Int64}(0)
    if sharded()
        with_gquery(c.view) do q
            check(ccall((:dfdb_group_query_hint_aggregate, LIB), Int32, (Ptr{Cvoid}, Int32, Int32), q, op, 0))
            check(ccall((:dfdb_group_aggregate, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}, Ptr{Float64}), q, op, 0, vi, vf))
            check(ccall((:dfdb_group_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n))
        end
    else
        with_query(c.view) do q
            check(ccall((:dfdb_query_hint_aggregate, LIB), Int32, (Ptr{Cvoid}, Int32, Int32), q, op, 0))
            check(ccall((:dfdb_aggregate, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}, Ptr{Float64}), q, op, 0, vi, vf))
            check(ccall((:dfdb_count, LIB), Int32, (Ptr{Cvoid}, Ptr{Int64}), q, n))
        end
    end
    if T === Float64
        return (vf[], n[])
    elseif op == 1                                   # Base.sum widens: signed and Bool -> Int64, unsigned -> UInt64 (Base.add_sum)
        return (T <: Unsigned ? reinterpret(UInt64, vi[]) : vi[], n[])
    else                                             # minimum / maximum keep the element type
        return (T <: Unsigned ? (reinterpret(UInt64, vi[]) % T) : (T === Bool ? vi[] != 0 : vi[] % T), n[])
    end
end
gpu_sum(c::DFColumn) = gpu_aggregate(c, 1)[1]
gpu_minimum(c::DFColumn) = gpu_aggregate(c, 2)[1]
gpu_maximum(c::DFColumn) = gpu_aggregate(c, 3)[1]
gpu_mean(c::DFColumn) = ((s, n) = gpu_aggregate(c, 1); s / n)
gpu_sum_count(c::DFColumn) = gpu_aggregate(c, 1)

"""
groupreduce(view, (:by,); out = :col => Stat()) on the device (src/tables/aggregate.jl:1-36: exported by the reference, unfinished there — it numbers the
groups in order of first appearance and stops).  Returns a DataFrame with one row per distinct value of `by` in order of first appearance, the group's
row count and `stat(col)`, stat in (:count, :sum, :minimum, :maximum, :mean).  Sharded like everything else when a group is active.
"""
function gpu_groupreduce(v::DFView, by::Symbol, col::Union{Symbol,Nothing} = nothing, stat::Symbol = :count)
    code = Dict(:count => 0, :sum => 1, :minimum => 2, :maximum => 3, :mean => 1)[stat]
    two = col !== nothing && stat != :count
    sub = two ? v[:, [by, col]] : v[:, [by]]
    keyview = sub[:, [by]]
    finish = (keys, counts, vi, vf) -> begin
        res = DataFrames.DataFrame(by => keys, :count => counts)
        if two
            T = Base.nonmissingtype(DataFrameDBs.coltype(sub.projection, 2))
            vals = stat == :mean ? (T <: AbstractFloat ? vf : (T <: Unsigned ? Float64.(reinterpret(UInt64, vi)) : Float64.(vi))) ./ max.(counts, 1) :
                   T <: AbstractFloat ? vf : (T <: Unsigned ? reinterpret(UInt64, vi) : vi)
            res[!, stat] = vals
        end
        res
    end
    if sharded()          # every GPU reduces its block range; the groups are merged by key in rank order inside the library
        return with_gquery(sub) do q
            ng = Ref{Int64}(0); kb = Ref{Int64}(0)
            check(ccall((:dfdb_group_query_groupreduce, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int64}, Ptr{Int64}), q, 0, two ? 1 : -1, code, ng, kb))
            n = ng[]
            outs, bufs = alloc_outputs(keyview, n, i -> gquery_coltype(q, 0), i -> kb[])
            counts = Vector{Int64}(undef, n); vi = Vector{Int64}(undef, n); vf = Vector{Float64}(undef, n)
            GC.@preserve bufs outs counts vi vf check(ccall((:dfdb_group_query_groupreduce_fetch, LIB), Int32, (Ptr{Cvoid}, Ptr{OutCol}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), q, outs, counts, vi, vf))
            finish(finish_columns(keyview, bufs)[1], counts, vi, vf)
        end
    end
    with_query(sub) do q
        ng = Ref{Int64}(0); kb = Ref{Int64}(0)
        check(ccall((:dfdb_query_groupreduce, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Int64}, Ptr{Int64}), q, 0, two ? 1 : -1, code, ng, kb))
        n = ng[]
        outs, bufs = alloc_outputs(keyview, n, i -> begin
            dt = Ref{Int32}(0)
            check(ccall((:dfdb_query_coltype, LIB), Int32, (Ptr{Cvoid}, Int32, Ptr{Int32}), q, 0, dt)); dt[]
        end, i -> kb[])
        counts = Vector{Int64}(undef, n); vi = Vector{Int64}(undef, n); vf = Vector{Float64}(undef, n)
        GC.@preserve bufs outs counts vi vf check(ccall((:dfdb_query_groupreduce_fetch, LIB), Int32, (Ptr{Cvoid}, Ptr{OutCol}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}), q, outs, counts, vi, vf))
        finish(finish_columns(keyview, bufs)[1], counts, vi, vf)
    end
end

# ---------------------------------------------------------------- write side (create_table / add_column!)
struct SizeStatsC; rows::Int64; compressed::Int64; uncompressed::Int64; end

"create_table(path; from = v) on the device (creators.jl:18-60): the view is materialised column by column into a
device-resident table (dfdb_table_add_from_query: no host round trip), block bodies are packed and LZ4-compressed
on the GPU and written in the reference's format; the result opens with the stock `open_table`."
function gpu_create_table(path::AbstractString, v::DFView; block_size::Integer = DataFrameDBs.DEFAULT_BLOCK_SIZE)
    d = device()
    dst = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dfdb_table_new, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), d.ctx, block_size, dst))
    try
        for name in keys(v.projection)
            with_query(v[:, [name]]) do q                              # a one-column view: projection column 0
                check(ccall((:dfdb_table_add_from_query, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Cvoid}, Int32), dst[], String(name), q, 0))
            end
        end
        st = Ref(SizeStatsC(0, 0, 0))
        check(ccall((:dfdb_table_save, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{SizeStatsC}), dst[], path, st))
    finally
        ccall((:dfdb_table_close, LIB), Int32, (Ptr{Cvoid},), dst[])
    end
    DataFrameDBs.open_table(path)
end

"add_column!(t, name, col::DFColumn) (table.jl:96-124): the lazy column is evaluated on the device and the new
`<id>.bin` is written next to the table's other column files; the caller then re-reads meta like the stock method."
function gpu_write_column(file::AbstractString, col::DFColumn)
    with_query(col.view) do q
        d = device()
        tmp = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:dfdb_table_new, LIB), Int32, (Ptr{Cvoid}, Int64, Ptr{Ptr{Cvoid}}), d.ctx, DataFrameDBs.blocksize(col.view.table), tmp))
        try
            check(ccall((:dfdb_table_add_from_query, LIB), Int32, (Ptr{Cvoid}, Cstring, Ptr{Cvoid}, Int32), tmp[], "col", q, 0))
            st = Ref(SizeStatsC(0, 0, 0))
            check(ccall((:dfdb_table_save_column, LIB), Int32, (Ptr{Cvoid}, Int32, Cstring, Ptr{SizeStatsC}), tmp[], 0, file, st))
            return st[]
        finally
            ccall((:dfdb_table_close, LIB), Int32, (Ptr{Cvoid},), tmp[])
        end
    end
end

# ---------------------------------------------------------------- drop-in switch
# The overrides REPLACE the reference's methods (same signatures), so the stock implementation cannot be reached by `invoke`
# afterwards.  It stays reachable in the WORLD AGE recorded before the first override was defined: `Base.invoke_in_world(WORLD0[], f,
# args...)` dispatches among the methods that existed then — the reference's own `materialize(::DFView)`, whose inner `nrow(v)` call is
# resolved in that same old world too.  No recursion is possible: the old world does not contain the overrides.
const WORLD0 = Ref{UInt}(0)
const ENABLED = Ref(false)

"run `gpu()`; on `Unsupported` (an expression outside the IR, an untraceable closure) evaluate `stock` with the reference's own methods"
function with_fallback(gpu, f, args...)
    try
        return gpu()
    catch e
        e isa Unsupported || rethrow()
        # said once per distinct reason: the answer is the same, the speed is the stock package's (VERDICT r3: a silent fallback hides a 1000 x slower query)
        @warn "DataFrameDBsAMD: $(f) falls back to the stock CPU path: $(e.msg)" maxlog = 1 _id = Symbol("dfdb_fallback_", hash(e.msg))
        return Base.invoke_in_world(WORLD0[], f, args...)
    end
end

"""
Route the hot-path consumers of DataFrameDBs through the MI355X engine, falling back to the stock path when an expression is outside the IR:
`materialize(::DFView)`, `nrow(::DFView)` (and with it `size` / `length(::DFColumn)`), `materialize(::DFColumn)`, `copyto!(dest, ::DFColumn)`,
`copyto!(dest, ::Broadcasted{DFColumnStyle})` (`dest .= col_expr`), and `sum` / `minimum` / `maximum` / `Statistics.mean` / `unique` of a DFColumn
(which otherwise pull one element at a time through `Base.iterate(::DFColumn)`, column.jl:102-126).
"""
function enable!()
    ENABLED[] && return nothing
    WORLD0[] = Base.get_world_counter()               # only the reference's methods exist in this world
    @eval begin
        DataFrameDBs.materialize(v::DFView) = with_fallback(() -> gpu_materialize(v), DataFrameDBs.materialize, v)
        DataFrameDBs.nrow(v::DFView) = with_fallback(() -> gpu_nrow(v), DataFrameDBs.nrow, v)
        DataFrameDBs.materialize(c::DFColumn) = with_fallback(() -> gpu_materialize(c), DataFrameDBs.materialize, c)
        function Base.copyto!(dest::AbstractVector, src::DFColumn)
            with_fallback(Base.copyto!, dest, src) do
                vals = gpu_materialize(src)
                length(dest) >= length(vals) || throw(BoundsError(dest, length(vals)))
                copyto!(dest, 1, vals, 1, length(vals))
                dest
            end
        end
        function Base.copyto!(dest::AbstractArray, bc::Base.Broadcast.Broadcasted{DataFrameDBs.DFColumnStyle})
            with_fallback(Base.copyto!, dest, bc) do
                col = Base.Broadcast.materialize(bc)      # the lazy DFColumn of the whole expression (columnbroadcast.jl:35-62)
                vals = gpu_materialize(col)
                length(dest) >= length(vals) || throw(BoundsError(dest, length(vals)))
                copyto!(dest, 1, vals, 1, length(vals))
                dest
            end
        end
        Base.sum(c::DFColumn) = with_fallback(() -> gpu_sum(c), Base.sum, c)
        Base.minimum(c::DFColumn) = with_fallback(() -> gpu_minimum(c), Base.minimum, c)
        Base.maximum(c::DFColumn) = with_fallback(() -> gpu_maximum(c), Base.maximum, c)
        Statistics.mean(c::DFColumn) = with_fallback(() -> gpu_mean(c), Statistics.mean, c)
        Base.unique(c::DFColumn) = with_fallback(() -> gpu_unique(c), Base.unique, c)
    end
    ENABLED[] = true
    nothing
end

"forget the device-resident copies (e.g. after the table's files changed on disk)"
function reset!()
    d = DEV[]
    d === nothing && return nothing
    for h in values(d.gtables); ccall((:dfdb_group_table_close, LIB), Int32, (Ptr{Cvoid},), h); end
    for h in values(d.tables); ccall((:dfdb_table_close, LIB), Int32, (Ptr{Cvoid},), h); end
    empty!(d.gtables); empty!(d.tables)
    nothing
end

end # module

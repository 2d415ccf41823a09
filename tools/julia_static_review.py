#!/usr/bin/env python3
"""Static review of the Julia shim's FFI: every `ccall` in dataframedbs.jl_amd/julia/DataFrameDBsAMD.jl against the prototype of the
same symbol in include/dfdb.h — symbol exists, arity, and per-argument class and width (Int32 vs int32_t, Int64 vs int64_t, Csize_t vs
size_t, Cstring vs const char*, Ptr{…} vs any pointer), return type Int32 vs int32_t.  Also the field layout of the two structs the
shim mirrors (OutCol = dfdb_outcol, SizeStatsC = dfdb_sizestats).  Julia is not installed in the build image, so this is how the
binding is kept honest; tests/test_host_cpu.py runs check() on every CPU test run and `python tools/julia_static_review.py` rewrites
dataframedbs.jl_amd/julia/STATIC_REVIEW.md.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "DataFrameDBsAMD.jl")
HDR = os.path.join(ROOT, "include", "dfdb.h")


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_prototypes():
    txt = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint32_t\s+(dfdb_\w+)\s*\(([^;{]*?)\)\s*;", txt, re.S):
        args = " ".join(m.group(2).split())
        protos[m.group(1)] = [] if args in ("void", "") else split_top(args)
    return protos


def c_class(arg):
    a = re.sub(r"\b\w+\s*\[[^\]]*\]", "*", arg)         # uint8_t id[128] -> pointer
    a = a.replace("const", " ")
    if "*" in a:
        if re.search(r"\bchar\s*\*\s*\w*$", a.strip()) and a.count("*") == 1:
            return "cstring"
        return "ptr"
    t = a.split()[0]
    return {"int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "size_t": "size", "double": "f64"}.get(t, t)


def jl_class(t):
    t = t.strip()
    if t == "Cstring":
        return "cstring"
    if t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    return {"Int32": "i32", "Int64": "i64", "UInt64": "u64", "Csize_t": "size", "Float64": "f64"}.get(t, t)


def jl_ccalls():
    src = open(JL).read()
    calls = []
    # the generated builders name their symbols through $(QuoteNode(X)): expand both instantiations
    gen = re.search(r"for \(([\w, ]+)\) in \((.*?)\)\)\n", src, re.S)
    maps = []
    if gen:
        names = [n.strip() for n in gen.group(1).split(",")]
        for tup in re.findall(r"\(((?::\w+,?\s*)+)\)", gen.group(2) + ")"):
            syms = re.findall(r":(\w+)", tup)
            assert len(syms) == len(names), (names, syms)
            maps.append(dict(zip(names, syms)))
    assert maps, "the generated query builders were not found"
    for m in re.finditer(r"ccall\(\(", src):
        i = m.end()
        depth, j = 2, i
        while depth and j < len(src):
            depth += src[j] in "({["
            depth -= src[j] in ")}]"
            j += 1
        body = src[m.start() + len("ccall("):j - 1]
        parts = split_top(body)
        symexpr, ret, argt = parts[0], parts[1], parts[2]
        line = src.count("\n", 0, m.start()) + 1
        argtypes = split_top(argt.strip()[1:-1].rstrip(","))
        sm = re.match(r"\(:(\w+), LIB\)", symexpr)
        if sm:
            calls.append((sm.group(1), ret, argtypes, len(parts) - 3, line))
        else:
            var = re.match(r"\(\$\(QuoteNode\((\w+)\)\), LIB\)", symexpr).group(1)
            for mp in maps:
                calls.append((mp[var], ret, argtypes, len(parts) - 3, line))
    return calls


def struct_fields_c(name):
    txt = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + ";", txt, re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        ty, names = decl.rsplit(" ", 1)[0], decl
        base = decl.split()[0]
        for nm in decl[len(base):].split(","):
            nm = nm.strip()
            fields.append(("ptr" if "*" in nm or "*" in base else c_class(base + " x"), nm.replace("*", "").strip()))
    return fields


def struct_fields_jl(name):
    src = open(JL).read()
    m = re.search(r"struct " + name + r"\s*[;\n](.*?)\bend", src, re.S)
    fields = []
    for f in re.split(r"[;\n]", m.group(1)):
        f = f.strip()
        if f:
            nm, ty = f.split("::")
            fields.append((jl_class(ty), nm.strip()))
    return fields


def check():
    """returns (rows, errors): one row per ccall site"""
    protos = c_prototypes()
    rows, errors = [], []
    for sym, ret, argtypes, nargs, line in jl_ccalls():
        if sym not in protos:
            errors.append(f"line {line}: {sym} is not declared in include/dfdb.h"); continue
        cargs = protos[sym]
        ok = ret.strip() == "Int32" and len(cargs) == len(argtypes) == nargs
        detail = []
        for k, (ca, ja) in enumerate(zip(cargs, argtypes)):
            cc, jc = c_class(ca), jl_class(ja)
            same = cc == jc or (cc == "cstring" and jc == "ptr") or (cc == "ptr" and jc == "cstring" and "char" in ca)
            ok = ok and same
            detail.append(f"{ja} ↔ `{ca}`" + ("" if same else "  **MISMATCH**"))
        if not ok:
            errors.append(f"line {line}: {sym}: Julia ({', '.join(argtypes)}) -> {ret} with {nargs} values vs C ({', '.join(cargs)})")
        rows.append((line, sym, detail, ok))
    for jn, cn in (("OutCol", "dfdb_outcol"), ("SizeStatsC", "dfdb_sizestats")):
        cf, jf = struct_fields_c(cn), struct_fields_jl(jn)
        if [c for c, _ in cf] != [c for c, _ in jf] or [n for _, n in cf] != [n for _, n in jf]:
            errors.append(f"struct {jn} {jf} does not mirror {cn} {cf}")
    return rows, errors


# ---------------------------------------------------------------- semantic lint (round 3: a signature walk does not see these)
CHECKED_CONV = re.compile(r"(?<![\w.{])(U?Int(?:8|16|32|64)?)\(\s*([A-Za-z_]\w*(?:\[\])?)\s*\)")
UNCHECKED_OK = ("dfdb_last_error", "dfdb_query_free", "dfdb_group_query_free", "dfdb_table_close", "dfdb_group_table_close")


def _functions(src):
    """(first line, last line, text) of every top-level-ish function / do-block free span: good enough to scope a GC.@preserve search"""
    lines = src.split("\n")
    spans, start = [], None
    for i, ln in enumerate(lines):
        if re.match(r"\s*(function |[\w.!]+\(.*\) = |for \(fname)", ln) and start is None and not ln.startswith(" " * 8):
            start = i
        if start is not None and (ln.startswith("end") or (re.match(r"[\w.!]+\(.*\) = ", ln) and i == start and not ln.rstrip().endswith("begin"))):
            spans.append((start, i)); start = None
    return lines, spans


def lint():
    """findings the signature walk cannot make:
    (a) a CHECKED integer conversion `IntN(x)` / `UIntN(x)` of a variable — it throws InexactError for a value outside the target (round 2 shipped
        `Int64(v) % Int64` for constants, which throws for a UInt64 >= 2^63 before the `%` is reached) — must say why the value fits: `# checked: …`
        on its line; wrapping conversions are spelled `x % T`;
    (b) `pointer(x)` must sit under a `GC.@preserve` that names x (same line, or an enclosing `GC.@preserve … begin` of the same function), or name
        who roots the buffer for the call that uses the pointer: `# rooted by <name>`;
    (c) every ccall's status goes through `check(…)` before anything reads the Ref / buffer outputs (the cleanup calls in `finally` blocks and
        dfdb_last_error are the exceptions)."""
    src = open(JL).read()
    lines = src.split("\n")
    findings = []
    in_doc = False
    for i, ln in enumerate(lines):
        code = ln.split("#", 1)[0] if not ln.lstrip().startswith("#") else ""
        comment = ln[len(code):]
        if ln.count('"""') % 2 == 1:
            in_doc = not in_doc
        if in_doc or ln.lstrip().startswith('"'):
            continue
        # (a)
        for m in CHECKED_CONV.finditer(code):
            ty, arg = m.group(1), m.group(2)
            if arg in ("undef",) or "checked:" in comment:
                continue
            findings.append(f"line {i + 1}: checked conversion `{ty}({arg})` of a variable without a `# checked:` justification (use `{arg} % {ty}` to wrap)")
        # (b)
        for m in re.finditer(r"\bpointer\(\s*([A-Za-z_]\w*)\s*\)", code):
            var = m.group(1)
            if re.search(r"GC\.@preserve[^\n]*\b" + re.escape(var) + r"\b", code[:m.start()]) or "rooted by" in comment:
                continue
            # an enclosing `GC.@preserve a b begin` above, inside the same function (stop at a line that starts a function)
            ok = False
            for j in range(i - 1, max(i - 40, -1), -1):
                up = lines[j]
                if re.match(r"\s*function ", up) or re.match(r"^\S.*\) = ", up):
                    break
                if re.search(r"GC\.@preserve[^\n]*\b" + re.escape(var) + r"\b[^\n]*\bbegin\b", up):
                    ok = True; break
            if not ok:
                findings.append(f"line {i + 1}: `pointer({var})` outside any `GC.@preserve` naming {var} (or a `# rooted by …` note)")
        # (c)
        for m in re.finditer(r"ccall\(\(", code):
            sym = re.match(r"(?::(\w+)|\$\(QuoteNode\((\w+)\)\)), LIB\)", code[m.end():])
            name = sym.group(1) or sym.group(2) if sym else "?"
            if name in UNCHECKED_OK or name == "FREE":
                continue
            if not re.search(r"check\(\s*$", code[:m.start()]):
                findings.append(f"line {i + 1}: the status of `{name}` does not go through check(…) before its outputs are read")
    return findings


WALKER_RULES = [
    ("entry", "return lower_closure(f, collect(Any, args))", "a trace that dies (a `Tr` where Julia needs a real Bool) goes on to the lowered-code walk before anything is declared unsupported"),
    ("source of the code", "cis = Base.code_lowered(f)", "the walk reads the closure's LOWERED code (before inference): `&&`, `||`, `?:` and chained comparisons are `goto #k if not %c` there"),
    ("one method", "length(cis) == 1 || throw(Unsupported(", "only a closure with exactly one method is walked"),
    ("captured variables", "slots[1] = f", "slot 1 is the closure itself: `getfield(#self#, :c)` runs as an ordinary call and yields the captured value"),
    ("fork", "st isa Core.GotoIfNot", "a branch on a traced value forks the walk; a branch on a real Bool is decided while walking"),
    ("no loops", "st.dest > pc || throw(Unsupported(\"the closure loops on a column value\"))", "a backward jump on a traced condition is refused (and every `goto` must go forward)"),
    ("and", "select_bool(c::Tr, t::Tr, e::Bool) = e ? (!c | t) : (c & t)", "`c && t` (else arm is the literal false) becomes `c & t`"),
    ("or", "select_bool(c::Tr, t::Bool, e::Tr) = t ? (c | e) : (!c & e)", "`c || e` (then arm is the literal true) becomes `c | e`"),
    ("select", "select_bool(c::Tr, t::Tr, e::Tr) = (c & t) | (!c & e)", "two traced Bool arms: `(c & t) | (!c & e)`"),
    ("Bool arms only", "select_bool(c, t, e) = throw(Unsupported(", "an arm that is not Bool-valued (`x > 0 ? x : -x`) stays on the CPU path: the IR has no select of numbers"),
    ("no raising arms", "(may_raise(t) || may_raise(e)) && throw(Unsupported(", "an arm Julia evaluates conditionally must not be able to raise once `&` evaluates it for every row"),
    ("what raises", "const RAISING_OPS = (0x14, 0x15, 0x16, 0x50)", "DFIR_IDIV, DFIR_REM, DFIR_MOD (DivideError) and DFIR_CAST (InexactError), by opcode"),
    ("budget", "(w.steps += 1) > WALK_BUDGET && throw(Unsupported(", "the walk is bounded (4096 statements over all arms)"),
    ("fallback is said", "@warn \"DataFrameDBsAMD: $(f) falls back to the stock CPU path: $(e.msg)\" maxlog = 1", "a view that falls back is logged once per reason"),
]


def walker_review():
    """the closure-lowering rules of round 4, each pinned to the line of the shim that implements it"""
    src = open(JL).read()
    lines = src.split("\n")
    rows, missing = [], []
    for name, needle, what in WALKER_RULES:
        at = next((i + 1 for i, ln in enumerate(lines) if needle in ln), None)
        rows.append((name, at, what))
        if at is None:
            missing.append(f"closure lowering: the rule '{name}' is not in the shim (`{needle}`)")
    return rows, missing


def main():
    rows, errors = check()
    findings = lint()
    wrows, wmissing = walker_review()
    errors = errors + findings + wmissing
    out = ["# Static review of the Julia shim's FFI (generated by tools/julia_static_review.py)", "",
           "Julia is not installed in the build image, so `dataframedbs.jl_amd/julia/DataFrameDBsAMD.jl` cannot be executed here.  This file walks every",
           "`ccall` of the shim against the prototype of the same symbol in `include/dfdb.h`: the symbol is declared, the arity matches, every argument",
           "has the same class and width (Int32 ↔ int32_t, Int64 ↔ int64_t, Csize_t ↔ size_t, Cstring ↔ const char*, Ptr{…} ↔ pointer), and the",
           "return type is Int32 ↔ int32_t.  The same check runs in the CPU test suite (`tests/test_host_cpu.py::test_julia_shim_ccalls_match_the_header`),",
           "so the table cannot go stale silently.  Struct mirrors: `OutCol` ↔ `dfdb_outcol`, `SizeStatsC` ↔ `dfdb_sizestats` (field order, classes and names).", "",
           f"Result: **{len(rows)} ccall sites, {len(errors)} mismatches**.", "",
           "| shim line | symbol | arguments (Julia ↔ C) | ok |", "|---|---|---|---|"]
    for line, sym, detail, ok in sorted(rows):
        out.append(f"| {line} | `{sym}` | " + "; ".join(detail).replace("|", "\\|") + f" | {'yes' if ok else '**NO**'} |")
    out += ["", "## Semantic lint (tools/julia_static_review.py: lint)", "",
            "Three rules a signature walk cannot apply, added after round 2's reviewer found `Int64(v) % Int64` (throws for a `UInt64` constant ≥ 2^63 before the `%`):",
            "", "* a checked integer conversion `IntN(x)` / `UIntN(x)` of a variable must justify itself (`# checked: …`); wrapping is spelled `x % T`;",
            "* `pointer(x)` must sit under a `GC.@preserve` naming `x`, or name who roots the buffer (`# rooted by …`);",
            "* every `ccall` status goes through `check(…)` before its outputs are read (cleanup calls and `dfdb_last_error` excepted).",
            "", f"Result: **{len(findings)} findings**." + ("" if not findings else "  " + "; ".join(findings)),
            "", "The constant emitter itself is pinned dynamically: `tests/test_ir_golden.py::test_shim_constant_emitter_transcription` runs a Python transcription of",
            "`emit_const` (same branches, same wrapping) over UInt64 constants ≥ 2^63, negative Int8 … Int64, Bool and Float32 / Float64 and compares the bytes with `dfdb/ir.py`'s",
            "and with the hand-assembled golden bytes."]
    out += ["", "## Closures with control flow: the walk over lowered code (round 4)", "",
            "The tracer calls a closure on symbolic `Tr` values; `&&`, `||`, `?:` and chained comparisons need a real `Bool` and kill the trace — among them the",
            "reference's own test closure `(a)->65>a>34` (`test/selection.jl:53`).  `lower_closure` walks `Base.code_lowered(f)` instead.  The rules, each at the",
            "line of the shim that implements it (a rule that disappears from the source fails the CPU test suite):", "",
            "| rule | shim line | what it guarantees |", "|---|---|---|"]
    for name, at, what in wrows:
        out.append(f"| {name} | {at if at else '**MISSING**'} | {what} |")
    out += ["", "Pinned dynamically without a Julia: `tests/test_ir_golden.py::test_shim_lowered_code_walk_transcription` runs a Python transcription of `walk` /",
            "`select_bool` over the lowered code of `a -> 3 > a > -2`, `a -> a < -2 || a > 5`, `a -> (a > -2 && a < 5) | (a == 10)` and `(a) -> 65 > a > 34` and compares",
            "the bytes with the hand-assembled golden cases `closure_*` of `tests/golden/ir_golden.json`, which the oracle (CPU) and the engine (GPU) evaluate to the",
            "answers Julia gives.  What the walk cannot know without running Julia: the exact shape `code_lowered` gives a closure in a given Julia version (the",
            "statement kinds handled are `Core.GotoIfNot`, `Core.GotoNode`, `Core.ReturnNode`, `Core.NewvarNode`, slot assignment, `:call`; anything else is `Unsupported`)."]
    out += ["", "## Things a signature check cannot see (reviewed by reading)", "",
            "* **Fallback without recursion.** `enable!()` records `WORLD0 = Base.get_world_counter()` *before* it defines the overriding methods and every",
            "  fallback goes through `Base.invoke_in_world(WORLD0[], f, args...)`: in that world only the reference's own methods exist, including the `nrow(v)`",
            "  the stock `materialize(::DFView)` calls inside (`materialization.jl:29`), so an `Unsupported` can never re-enter an override (round 1's",
            "  `invoke(_cpu_materialize, …)` re-entered the replaced method).",
            "* **Untraceable closures.** `65 > a > 34` lowers to `(65 > a) && (a > 34)`; `&&` on a `Tr` raises `TypeError` inside `lower`'s `try`: the lowered-code walk above",
            "  takes over, and what it refuses is `Unsupported` → stock path, logged once.  Lowering happens before any device call (`with_query` lowers all stages first), so a",
            "  fallback leaves no device state behind.",
            "* **GC safety.** Every buffer whose pointer crosses the ABI is rooted: IR byte vectors, index vectors, name / code pointer arrays and the output",
            "  vectors are under `GC.@preserve` for the duration of the call; the engine copies what it keeps (`parse_ir`, `Stage::idx`), so nothing outlives the call.",
            "* **Ownership.** `with_query` frees its query in `finally`; tables and the group live in `DEV[]` until `reset!()`; `dfdb_group_query_shard` returns a borrowed handle.",
            "* **Element types.** Bool columns come back as `BitVector` (the reference's `make_materialization(::Type{Bool})`), Date / DateTime / Time / Char are",
            "  relabelled from the engine's Int64 / UInt32, `sum` widens like `Base.add_sum` (signed / Bool → Int64, unsigned → UInt64), `minimum` / `maximum` keep `T`.",
            "* **Runtime probe.** `dataframedbs.jl_amd/julia/probe.jl` executes the shim against the stock path on a freshly written table; `tests/test_gpu_julia_probe.py`",
            "  runs it when (and only when) a `julia` binary with DataFrameDBs.jl installed exists on the GPU box."]
    with open(os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "STATIC_REVIEW.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    for e in errors:
        print("MISMATCH:", e)
    print(f"{len(rows)} ccall sites, {len(errors)} mismatches")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())

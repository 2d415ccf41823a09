#!/usr/bin/env python3
"""Writes tests/golden/ir_golden.json: hand-assembled IR byte strings for EVERY operator of include/dfdb_ir.h with the answer Julia
gives for them, independent of the product's own encoder (dfdb/ir.py) and of the Julia shim's OPS / UNARY tables.

Why: the parity tests build the engine's and the oracle's predicates with the same encoder, so a mis-encoded operator would be
invisible to engine-vs-oracle comparison.  Here the bytes are assembled from mnemonics by the ten-line assembler below, which reads
the opcode VALUES out of the header text itself (never from dfdb/ir.py); the expected answers are literals typed from Julia's
semantics (rem / mod / div signs, `/` -> Float64, non-short-circuit `&`, exact Int-vs-Float comparison, InexactError of T(x)).
The tests then check, without a GPU, that (1) the oracle evaluates each byte string to the typed answer, (2) dfdb/ir.py emits
exactly these bytes for the same expression, (3) the Julia shim's tables carry the header's numbers; and on the GPU that (4) the
engine evaluates each byte string to the same answer.

Table the cases run over (5 rows, block_size 2):
  a  :: Int64            = [-7, -1, 0, 3, 10]
  x  :: Float64          = [0.5, -2.0, 3.0, 1e10, NaN]
  s  :: String           = ["apple", "sony", "", "sonic", "xs"]
  m  :: Union{Int64,Missing} = [1, missing, 3, missing, 0]
  u  :: UInt8            = [0, 1, 127, 128, 255]
round 3 (the corners engine and oracle could get wrong TOGETHER — they share an author):
  i8 :: Int8             = [-128, -1, 0, 1, 127]
  w  :: UInt64           = [0, 1, 2^63, 2^63 + 5, 2^64 - 1]
  z  :: Float64          = [-0.0, 0.0, NaN, 1.0, -1.0]
  f  :: Float32          = [0.1f0, -2.5f0, 16777216f0, NaN32, 3f0]
  b  :: Bool             = [true, false, true, false, true]
  mb :: Union{Bool,Missing} = [true, missing, false, missing, true]
  big:: Int64            = [typemin(Int64), -1, 0, 1, typemax(Int64)]
  us :: String           = ["é", "ß", "日本", "a", ""]
  ns :: Union{String,Missing} = ["a", missing, "", missing, "b"]

    python tests/golden/make_ir_golden.py        (rewrites ir_golden.json next to this file)
"""
import json
import os
import re
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "..", "include", "dfdb_ir.h")


def header_symbols():
    txt = open(HEADER).read()
    sym = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define\s+(DFIR_\w+)\s+(0x[0-9a-fA-F]+)", txt)}
    enum = re.search(r"enum\s*\{(.*?)\};", txt, re.S).group(1)
    for m in re.finditer(r"(DFDB_\w+)\s*=\s*(0x[0-9a-fA-F]+|\d+)", enum):
        sym[m.group(1)] = int(m.group(2), 0)
    return sym


S = header_symbols()
A, X, STR, M, U = 0, 1, 2, 3, 4      # column ordinals
I8C, W, Z, FF, B, MB, BIG = 5, 6, 7, 8, 9, 10, 11


def col(k): return bytes([S["DFIR_COL"]]) + struct.pack("<I", k)
def ci(v, dt="DFDB_I64"): return bytes([S["DFIR_CONST"], S[dt]]) + struct.pack("<q" if v < 0 else "<Q", v)
def cf(v): return bytes([S["DFIR_CONST"], S["DFDB_F64"]]) + struct.pack("<d", v)
def cb(v): return bytes([S["DFIR_CONST"], S["DFDB_BOOL"]]) + struct.pack("<q", 1 if v else 0)
def cs(t): return bytes([S["DFIR_CONST_STR"]]) + struct.pack("<I", len(t.encode())) + t.encode()
def cset(vals, dt="DFDB_I64"): return bytes([S["DFIR_CONST_SET"], S[dt]]) + struct.pack("<I", len(vals)) + b"".join(struct.pack("<q", v) for v in vals)
def csetf(vals): return bytes([S["DFIR_CONST_SET"], S["DFDB_F64"]]) + struct.pack("<I", len(vals)) + b"".join(struct.pack("<d", v) for v in vals)
def op(name): return bytes([S["DFIR_" + name]])
def cast(dt): return bytes([S["DFIR_CAST"], S[dt]])


T, F = True, False
NAN = float("nan")
MISS = "missing"
# (name, Julia expression, bytes, expected result type, expected values | "DivideError" | "InexactError", dfdb/ir.py expression)
CASES = [
    ("add", "a .+ 2", col(A) + ci(2) + op("ADD"), "Int64", [-5, 1, 2, 5, 12], "A + 2"),
    ("sub", "a .- 2", col(A) + ci(2) + op("SUB"), "Int64", [-9, -3, -2, 1, 8], "A - 2"),
    ("rsub", "2 .- a", ci(2) + col(A) + op("SUB"), "Int64", [9, 3, 2, -1, -8], "2 - A"),
    ("mul", "a .* 3", col(A) + ci(3) + op("MUL"), "Int64", [-21, -3, 0, 9, 30], "A * 3"),
    ("div", "a ./ 2", col(A) + ci(2) + op("DIV"), "Float64", [-3.5, -0.5, 0.0, 1.5, 5.0], "A / 2"),
    ("idiv", "a .÷ 2", col(A) + ci(2) + op("IDIV"), "Int64", [-3, 0, 0, 1, 5], "ir.div(A, 2)"),
    ("rem", "a .% 3", col(A) + ci(3) + op("REM"), "Int64", [-1, -1, 0, 0, 1], "A % 3"),
    ("mod", "mod.(a, 3)", col(A) + ci(3) + op("MOD"), "Int64", [2, 2, 0, 0, 1], "ir.mod(A, 3)"),
    ("mod_negative_divisor", "mod.(a, -3)", col(A) + ci(-3) + op("MOD"), "Int64", [-1, -1, 0, 0, -2], "ir.mod(A, -3)"),
    ("neg", "-a", col(A) + op("NEG"), "Int64", [7, 1, 0, -3, -10], "-A"),
    ("abs", "abs.(a)", col(A) + op("ABS"), "Int64", [7, 1, 0, 3, 10], "abs(A)"),
    ("min", "min.(a, 2)", col(A) + ci(2) + op("MIN"), "Int64", [-7, -1, 0, 2, 2], "ir.minimum(A, 2)"),
    ("max", "max.(a, 2)", col(A) + ci(2) + op("MAX"), "Int64", [2, 2, 2, 3, 10], "ir.maximum(A, 2)"),
    ("eq", "a .== 3", col(A) + ci(3) + op("EQ"), "Bool", [F, F, F, T, F], "A == 3"),
    ("ne", "a .!= 3", col(A) + ci(3) + op("NE"), "Bool", [T, T, T, F, T], "A != 3"),
    ("lt", "a .< 0", col(A) + ci(0) + op("LT"), "Bool", [T, T, F, F, F], "A < 0"),
    ("le", "a .<= 0", col(A) + ci(0) + op("LE"), "Bool", [T, T, T, F, F], "A <= 0"),
    ("gt", "a .> 0", col(A) + ci(0) + op("GT"), "Bool", [F, F, F, T, T], "A > 0"),
    ("ge", "a .>= 0", col(A) + ci(0) + op("GE"), "Bool", [F, F, T, T, T], "A >= 0"),
    ("and", "(a .> -2) .& (a .< 5)", col(A) + ci(-2) + op("GT") + col(A) + ci(5) + op("LT") + op("AND"), "Bool", [F, T, T, T, F], "(A > -2) & (A < 5)"),
    ("or", "(a .< -2) .| (a .> 5)", col(A) + ci(-2) + op("LT") + col(A) + ci(5) + op("GT") + op("OR"), "Bool", [T, F, F, F, T], "(A < -2) | (A > 5)"),
    ("xor", "xor.(a .> -2, a .> 1)", col(A) + ci(-2) + op("GT") + col(A) + ci(1) + op("GT") + op("XOR"), "Bool", [F, T, T, F, F], "(A > -2) ^ (A > 1)"),
    ("not", ".!(a .> 0)", col(A) + ci(0) + op("GT") + op("NOT"), "Bool", [T, T, T, F, F], "~(A > 0)"),
    ("bitand", "a .& 6", col(A) + ci(6) + op("AND"), "Int64", [0, 6, 0, 2, 2], "A & 6"),
    ("bitor", "a .| 1", col(A) + ci(1) + op("OR"), "Int64", [-7, -1, 1, 3, 11], "A | 1"),
    ("in_set", "in.(a, Ref([3, 10, 99]))", col(A) + cset([3, 10, 99]) + op("IN_SET"), "Bool", [F, F, F, T, T], "ir.isin(A, [3, 10, 99])"),
    ("str_eq", 's .== "sony"', col(STR) + cs("sony") + op("EQ"), "Bool", [F, T, F, F, F], 'St == "sony"'),
    ("str_ne", 's .!= "sony"', col(STR) + cs("sony") + op("NE"), "Bool", [T, F, T, T, T], 'St != "sony"'),
    ("str_lt", 's .< "sonic"', col(STR) + cs("sonic") + op("LT"), "Bool", [T, F, T, F, F], 'St < "sonic"'),
    ("startswith", 'startswith.(s, "so")', col(STR) + cs("so") + op("STARTSWITH"), "Bool", [F, T, F, T, F], 'ir.startswith(St, "so")'),
    ("endswith", 'endswith.(s, "s")', col(STR) + cs("s") + op("ENDSWITH"), "Bool", [F, F, F, F, T], 'ir.endswith(St, "s")'),
    ("sizeof", "sizeof.(s)", col(STR) + op("SIZEOF"), "Int64", [5, 4, 0, 5, 2], "ir.sizeof(St)"),
    ("ismissing", "ismissing.(m)", col(M) + op("ISMISSING"), "Bool", [F, T, F, T, F], "ir.ismissing(Mi)"),
    ("coalesce", "coalesce.(m, 42)", col(M) + ci(42) + op("COALESCE"), "Int64", [1, 42, 3, 42, 0], "ir.coalesce(Mi, 42)"),
    ("missing_propagates", "m .+ 1", col(M) + ci(1) + op("ADD"), "Missing(Int64)", [2, MISS, 4, MISS, 1], "Mi + 1"),
    ("float_lt_nan", "x .< 1.0", col(X) + cf(1.0) + op("LT"), "Bool", [T, T, F, F, F], "Xf < 1.0"),
    ("float_ne_nan", "x .!= 3.0", col(X) + cf(3.0) + op("NE"), "Bool", [T, T, F, T, T], "Xf != 3.0"),
    ("int_vs_float_exact", "a .< x", col(A) + col(X) + op("LT"), "Bool", [T, F, T, T, F], "A < Xf"),
    ("float_mul", "x .* 2.0", col(X) + cf(2.0) + op("MUL"), "Float64", [1.0, -4.0, 6.0, 2e10, NAN], "Xf * 2.0"),
    ("promote_int_float", "a .+ 0.5", col(A) + cf(0.5) + op("ADD"), "Float64", [-6.5, -0.5, 0.5, 3.5, 10.5], "A + 0.5"),
    ("uint8_widens", "u .+ 1", col(U) + ci(1) + op("ADD"), "Int64", [1, 2, 128, 129, 256], "Uc + 1"),
    ("uint8_wraps", "u .+ 0x01", col(U) + ci(1, "DFDB_U8") + op("ADD"), "UInt8", [1, 2, 128, 129, 0], "Uc + ir.const(1, ir.U8)"),
    ("bool_const", "(a .> 0) .& true", col(A) + ci(0) + op("GT") + cb(True) + op("AND"), "Bool", [F, F, F, T, T], "(A > 0) & True"),
    ("cast_float64", "Float64.(a)", col(A) + cast("DFDB_F64"), "Float64", [-7.0, -1.0, 0.0, 3.0, 10.0], "ir.cast(A, ir.F64)"),
    ("cast_int8", "Int8.(a)", col(A) + cast("DFDB_I8"), "Int8", [-7, -1, 0, 3, 10], "ir.cast(A, ir.I8)"),
    ("cast_uint8_to_int8_inexact", "Int8.(u)", col(U) + cast("DFDB_I8"), "Int8", "InexactError", "ir.cast(Uc, ir.I8)"),
    ("cast_negative_to_unsigned_inexact", "UInt64.(a)", col(A) + cast("DFDB_U64"), "UInt64", "InexactError", "ir.cast(A, ir.U64)"),
    ("cast_const_300_int8_inexact", "Int8.(a .* 0 .+ 300)", col(A) + ci(0) + op("MUL") + ci(300) + op("ADD") + cast("DFDB_I8"), "Int8", "InexactError", "ir.cast(A * 0 + 300, ir.I8)"),
    ("cast_300f_int8_inexact", "Int8.(a .* 0 .+ 300.0)", col(A) + ci(0) + op("MUL") + cf(300.0) + op("ADD") + cast("DFDB_I8"), "Int8", "InexactError", "ir.cast(A * 0 + 300.0, ir.I8)"),
    ("cast_minus1f_uint64_inexact", "UInt64.(a .* 0 .- 1.0)", col(A) + ci(0) + op("MUL") + cf(1.0) + op("SUB") + cast("DFDB_U64"), "UInt64", "InexactError", "ir.cast(A * 0 - 1.0, ir.U64)"),
    ("cast_2p63f_uint64", "UInt64.(a .* 0 .+ 2.0^63)", col(A) + ci(0) + op("MUL") + cf(2.0 ** 63) + op("ADD") + cast("DFDB_U64"), "UInt64", [2 ** 63] * 5, "ir.cast(A * 0 + 2.0 ** 63, ir.U64)"),
    ("cast_2p63f_int64_inexact", "Int64.(a .* 0 .+ 2.0^63)", col(A) + ci(0) + op("MUL") + cf(2.0 ** 63) + op("ADD") + cast("DFDB_I64"), "Int64", "InexactError", "ir.cast(A * 0 + 2.0 ** 63, ir.I64)"),
    ("cast_fraction_inexact", "Int64.(x)", col(X) + cast("DFDB_I64"), "Int64", "InexactError", "ir.cast(Xf, ir.I64)"),
    ("cast_float32_rounds", "Float32.(a ./ 3)", col(A) + ci(3) + op("DIV") + cast("DFDB_F32"), "Float32",
     [struct.unpack("<f", struct.pack("<f", v / 3))[0] for v in (-7, -1, 0, 3, 10)], "ir.cast(A / 3, ir.F32)"),
    ("rem_by_zero", "a .% 0", col(A) + ci(0) + op("REM"), "Int64", "DivideError", "A % 0"),
    ("idiv_by_zero", "a .÷ 0", col(A) + ci(0) + op("IDIV"), "Int64", "DivideError", "ir.div(A, 0)"),
    # round 4: what the Julia shim's walk over a closure's LOWERED code must emit (DataFrameDBsAMD.lower_closure: `goto if not` diamonds -> & | !).
    # `(a) -> 3 > a > -2` is the shape of the reference's own test closure `(a)->65>a>34` (test/selection.jl:53): Julia lowers it to
    #   %1 = 3 > a ; goto #3 if not %1 ; %3 = a > -2 ; return %3 ; #3: return false      ->  (3 > a) & (a > -2)
    ("closure_chained_comparison", "a -> 3 > a > -2", ci(3) + col(A) + op("GT") + col(A) + ci(-2) + op("GT") + op("AND"), "Bool", [F, T, T, F, F], "(ir.const(3) > A) & (A > -2)"),
    #   %1 = a < -2 ; goto #3 if not %1 ; return true ; #3: %4 = a > 5 ; return %4          ->  (a < -2) | (a > 5)
    ("closure_short_circuit_or", "a -> a < -2 || a > 5", col(A) + ci(-2) + op("LT") + col(A) + ci(5) + op("GT") + op("OR"), "Bool", [T, F, F, F, T], "(A < -2) | (A > 5)"),
    #   a slot assigned in both arms and read after the join: the continuation is walked once per arm,  (c & T) | (!c & E)  with c = a > -2,
    #   T = (a < 5) | (a == 10), E = false | (a == 10)
    ("closure_and_then_or_through_a_join", "a -> (a > -2 && a < 5) | (a == 10)",
     col(A) + ci(-2) + op("GT") + col(A) + ci(5) + op("LT") + col(A) + ci(10) + op("EQ") + op("OR") + op("AND") +
     col(A) + ci(-2) + op("GT") + op("NOT") + cb(False) + col(A) + ci(10) + op("EQ") + op("OR") + op("AND") + op("OR"), "Bool", [F, T, T, T, T],
     "((A > -2) & ((A < 5) | (A == 10))) | (~(A > -2) & (ir.const(False) | (A == 10)))"),
]

# ---- round 3: answers typed from Julia's semantics, never from the oracle's output ------------------------------------------------------
TMIN, TMAX = -(2 ** 63), 2 ** 63 - 1
INF = "Inf"


def f32(v): return struct.unpack("<f", struct.pack("<f", v))[0]          # round to Float32 (IEEE, independent of numpy and of the oracle)


F32COL = [f32(0.1), -2.5, 16777216.0, NAN, 3.0]
CASES += [
    # typemin: checked division, wrapping negation / abs (Base: div(typemin(T), -1) throws DivideError, rem(typemin, -1) == 0, abs(typemin) == typemin)
    ("typemin_idiv_minus1_divide_error", "big .÷ -1", col(BIG) + ci(-1) + op("IDIV"), "Int64", "DivideError", "ir.div(Big, -1)"),
    ("typemin_rem_minus1_is_zero", "big .% -1", col(BIG) + ci(-1) + op("REM"), "Int64", [0, 0, 0, 0, 0], "Big % -1"),
    ("typemin_mod_minus1_is_zero", "mod.(big, -1)", col(BIG) + ci(-1) + op("MOD"), "Int64", [0, 0, 0, 0, 0], "ir.mod(Big, -1)"),
    ("abs_typemin_wraps", "abs.(big)", col(BIG) + op("ABS"), "Int64", [TMIN, 1, 0, 1, TMAX], "abs(Big)"),
    ("neg_typemin_wraps", "-big", col(BIG) + op("NEG"), "Int64", [TMIN, 1, 0, -1, TMIN + 1], "-Big"),
    ("add_wraps_at_typemax", "big .+ 1", col(BIG) + ci(1) + op("ADD"), "Int64", [TMIN + 1, 0, 1, 2, TMIN], "Big + 1"),
    ("idiv_truncates_toward_zero", "big .÷ 2", col(BIG) + ci(2) + op("IDIV"), "Int64", [-(2 ** 62), 0, 0, 0, 2 ** 62 - 1], "ir.div(Big, 2)"),
    ("rem_keeps_the_dividends_sign", "big .% 3", col(BIG) + ci(3) + op("REM"), "Int64", [-2, -1, 0, 1, 1], "Big % 3"),
    ("mod_keeps_the_divisors_sign", "mod.(big, 3)", col(BIG) + ci(3) + op("MOD"), "Int64", [1, 2, 0, 1, 1], "ir.mod(Big, 3)"),
    ("int_over_zero_is_inf", "a ./ 0", col(A) + ci(0) + op("DIV"), "Float64", ["-Inf", "-Inf", NAN, INF, INF], "A / 0"),
    # Int8: arithmetic between Int8s stays Int8 and wraps; with an Int64 literal it widens
    ("abs_int8_typemin_wraps", "abs.(i8)", col(I8C) + op("ABS"), "Int8", [-128, 1, 0, 1, 127], "abs(I8c)"),
    ("neg_int8_typemin_wraps", "-i8", col(I8C) + op("NEG"), "Int8", [-128, 1, 0, -1, -127], "-I8c"),
    ("int8_add_wraps", "i8 .+ Int8(1)", col(I8C) + ci(1, "DFDB_I8") + op("ADD"), "Int8", [-127, 0, 1, 2, -128], "I8c + ir.const(1, ir.I8)"),
    ("int8_sub_wraps", "i8 .- Int8(1)", col(I8C) + ci(1, "DFDB_I8") + op("SUB"), "Int8", [127, -2, -1, 0, 126], "I8c - ir.const(1, ir.I8)"),
    ("int8_mul_wraps", "i8 .* Int8(2)", col(I8C) + ci(2, "DFDB_I8") + op("MUL"), "Int8", [0, -2, 0, 2, -2], "I8c * ir.const(2, ir.I8)"),
    ("int8_widens_with_int64", "i8 .+ 1", col(I8C) + ci(1) + op("ADD"), "Int64", [-127, 0, 1, 2, 128], "I8c + 1"),
    ("int8_typemin_idiv_minus1_divide_error", "i8 .÷ Int8(-1)", col(I8C) + ci(-1, "DFDB_I8") + op("IDIV"), "Int8", "DivideError", "ir.div(I8c, ir.const(-1, ir.I8))"),
    # signed zeros and NaN: == ignores the sign of zero, min / max do not, NaN propagates through both
    ("minus_zero_equals_zero", "z .== 0.0", col(Z) + cf(0.0) + op("EQ"), "Bool", [T, T, F, F, F], "Zf == 0.0"),
    ("nothing_is_below_minus_zero_but_minus_one", "z .< -0.0", col(Z) + cf(-0.0) + op("LT"), "Bool", [F, F, F, F, T], "Zf < -0.0"),
    ("min_orders_minus_zero_first", "min.(z, 0.0)", col(Z) + cf(0.0) + op("MIN"), "Float64", [-0.0, 0.0, NAN, 0.0, -1.0], "ir.minimum(Zf, 0.0)"),
    ("max_orders_plus_zero_last", "max.(z, -0.0)", col(Z) + cf(-0.0) + op("MAX"), "Float64", [-0.0, 0.0, NAN, 1.0, -0.0], "ir.maximum(Zf, -0.0)"),
    ("min_propagates_nan", "min.(x, 1.0)", col(X) + cf(1.0) + op("MIN"), "Float64", [0.5, -2.0, 1.0, 1.0, NAN], "ir.minimum(Xf, 1.0)"),
    ("max_propagates_nan", "max.(x, 1.0)", col(X) + cf(1.0) + op("MAX"), "Float64", [1.0, 1.0, 3.0, 1e10, NAN], "ir.maximum(Xf, 1.0)"),
    ("in_uses_double_equals_nan_is_never_in", "in.(z, Ref([NaN, 1.0]))", col(Z) + csetf([NAN, 1.0]) + op("IN_SET"), "Bool", [F, F, F, T, F], "ir.isin(Zf, [float('nan'), 1.0])"),
    ("in_minus_zero_is_in_zero", "in.(z, Ref([0.0]))", col(Z) + csetf([0.0]) + op("IN_SET"), "Bool", [T, T, F, F, F], "ir.isin(Zf, [0.0])"),
    # UInt64: compared with signed integers as the mathematical values; arithmetic with an Int64 literal is UInt64 (promote_type) and wraps
    ("uint64_above_minus_one", "w .> -1", col(W) + ci(-1) + op("GT"), "Bool", [T, T, T, T, T], "Wc > -1"),
    ("int64_below_uint64", "a .< w", col(A) + col(W) + op("LT"), "Bool", [T, T, T, T, T], "A < Wc"),
    ("uint64_le_int64", "w .<= big", col(W) + col(BIG) + op("LE"), "Bool", [F, F, F, F, F], "Wc <= Big"),
    ("uint64_never_equals_negative", "w .== a", col(W) + col(A) + op("EQ"), "Bool", [F, F, F, F, F], "Wc == A"),
    ("uint64_above_typemax_int64", "w .> typemax(Int64)", col(W) + ci(TMAX) + op("GT"), "Bool", [F, F, T, T, T], "Wc > %d" % TMAX),
    ("uint64_const_2p63", "w .== 0x8000000000000000", col(W) + ci(2 ** 63, "DFDB_U64") + op("EQ"), "Bool", [F, F, T, F, F], "Wc == ir.const(2 ** 63, ir.U64)"),
    ("uint64_plus_int64_is_uint64_and_wraps", "w .+ 1", col(W) + ci(1) + op("ADD"), "UInt64", [1, 2, 2 ** 63 + 1, 2 ** 63 + 6, 0], "Wc + 1"),
    ("uint64_minus_int64_wraps_below_zero", "w .- 1", col(W) + ci(1) + op("SUB"), "UInt64", [2 ** 64 - 1, 0, 2 ** 63 - 1, 2 ** 63 + 4, 2 ** 64 - 2], "Wc - 1"),
    ("uint64_idiv", "w .÷ 2", col(W) + ci(2) + op("IDIV"), "UInt64", [0, 0, 2 ** 62, 2 ** 62 + 2, 2 ** 63 - 1], "ir.div(Wc, 2)"),
    ("uint64_rem", "w .% 10", col(W) + ci(10) + op("REM"), "UInt64", [0, 1, 8, 3, 5], "Wc % 10"),
    ("uint64_to_float64_rounds_to_nearest_even", "w ./ 2", col(W) + ci(2) + op("DIV"), "Float64", [0.0, 0.5, 2.0 ** 62, 2.0 ** 62, 2.0 ** 63], "Wc / 2"),
    # Float32: stays Float32 against integers, becomes Float64 against a Float64; comparisons with a Float64 literal are exact (0.1f0 != 0.1)
    ("float32_plus_int64_is_float32", "f .+ a", col(FF) + col(A) + op("ADD"), "Float32", [f32(f32(0.1) + -7.0), -3.5, 16777216.0, NAN, 13.0], "Ff + A"),
    ("float32_times_int64_is_float32", "f .* 2", col(FF) + ci(2) + op("MUL"), "Float32", [f32(f32(0.1) * 2.0), -5.0, 33554432.0, NAN, 6.0], "Ff * 2"),
    ("float32_plus_float64_is_float64", "f .+ 0.5", col(FF) + cf(0.5) + op("ADD"), "Float64", [f32(0.1) + 0.5, -2.0, 16777216.5, NAN, 3.5], "Ff + 0.5"),
    ("float32_is_not_the_float64_literal", "f .== 0.1", col(FF) + cf(0.1) + op("EQ"), "Bool", [F, F, F, F, F], "Ff == 0.1"),
    ("float32_vs_float64_exact_order", "f .< 0.1", col(FF) + cf(0.1) + op("LT"), "Bool", [F, T, F, F, F], "Ff < 0.1"),
    # Bool is a number: true + true == 2 (Int64); -true == -1
    ("bool_plus_bool_is_int64", "b .+ b", col(B) + col(B) + op("ADD"), "Int64", [2, 0, 2, 0, 2], "Bc + Bc"),
    ("bool_plus_int64", "b .+ 1", col(B) + ci(1) + op("ADD"), "Int64", [2, 1, 2, 1, 2], "Bc + 1"),
    ("bool_times_float64", "b .* 2.5", col(B) + cf(2.5) + op("MUL"), "Float64", [2.5, 0.0, 2.5, 0.0, 2.5], "Bc * 2.5"),
    ("neg_bool_is_int64", "-b", col(B) + op("NEG"), "Int64", [-1, 0, -1, 0, -1], "-Bc"),
    # three-valued logic (Base: missing & false == false, missing | true == true, xor(missing, x) and !missing are missing)
    ("xor_missing_is_missing", "xor.(mb, true)", col(MB) + cb(True) + op("XOR"), "Missing(Bool)", [F, MISS, T, MISS, F], "Mb ^ True"),
    ("not_missing_is_missing", ".!mb", col(MB) + op("NOT"), "Missing(Bool)", [F, MISS, T, MISS, F], "~Mb"),
    ("missing_and_false_is_false", "mb .& false", col(MB) + cb(False) + op("AND"), "Missing(Bool)", [F, F, F, F, F], "Mb & False"),
    ("missing_or_true_is_true", "mb .| true", col(MB) + cb(True) + op("OR"), "Missing(Bool)", [T, T, T, T, T], "Mb | True"),
    ("missing_or_false_is_missing", "mb .| false", col(MB) + cb(False) + op("OR"), "Missing(Bool)", [T, MISS, F, MISS, T], "Mb | False"),
    ("coalesce_bool", "coalesce.(mb, false)", col(MB) + cb(False) + op("COALESCE"), "Bool", [T, F, F, F, T], "ir.coalesce(Mb, False)"),
    ("coalesce_chain", "coalesce.(m, m .+ 1, 9)", col(M) + col(M) + ci(1) + op("ADD") + op("COALESCE") + ci(9) + op("COALESCE"), "Int64", [1, 9, 3, 9, 0],
     "ir.coalesce(ir.coalesce(Mi, Mi + 1), 9)"),
    ("ismissing_of_an_expression", "ismissing.(m .* 2)", col(M) + ci(2) + op("MUL") + op("ISMISSING"), "Bool", [F, T, F, T, F], "ir.ismissing(Mi * 2)"),
    ("comparison_with_missing_is_missing", "m .== 1", col(M) + ci(1) + op("EQ"), "Missing(Bool)", [T, MISS, F, MISS, F], "Mi == 1"),
    ("three_valued_and", "(m .== 1) .& (a .> 0)", col(M) + ci(1) + op("EQ") + col(A) + ci(0) + op("GT") + op("AND"), "Missing(Bool)", [F, F, F, MISS, F], "(Mi == 1) & (A > 0)"),
    ("three_valued_or", "(m .== 1) .| (a .> 0)", col(M) + ci(1) + op("EQ") + col(A) + ci(0) + op("GT") + op("OR"), "Missing(Bool)", [T, MISS, F, T, T], "(Mi == 1) | (A > 0)"),
]

# ---- round 3, second batch: negative divisors, float rem / mod / div, exact Int-vs-Float comparison at the edges of the ranges, UTF-8 strings ----------
US, NS = 12, 13
CASES += [
    ("rem_negative_divisor", "a .% -3", col(A) + ci(-3) + op("REM"), "Int64", [-1, -1, 0, 0, 1], "A % -3"),
    ("idiv_negative_divisor", "a .÷ -3", col(A) + ci(-3) + op("IDIV"), "Int64", [2, 0, 0, -1, -3], "ir.div(A, -3)"),
    ("float_rem_keeps_the_dividends_sign", "x .% 2.0", col(X) + cf(2.0) + op("REM"), "Float64", [0.5, -0.0, 1.0, 0.0, NAN], "Xf % 2.0"),
    ("float_mod_takes_the_divisors_sign", "mod.(x, 2.0)", col(X) + cf(2.0) + op("MOD"), "Float64", [0.5, 0.0, 1.0, 0.0, NAN], "ir.mod(Xf, 2.0)"),
    ("float_mod_negative_divisor", "mod.(x, -2.0)", col(X) + cf(-2.0) + op("MOD"), "Float64", [-1.5, -0.0, -1.0, -0.0, NAN], "ir.mod(Xf, -2.0)"),
    ("float_idiv_truncates", "x .÷ 2.0", col(X) + cf(2.0) + op("IDIV"), "Float64", [0.0, -1.0, 1.0, 5e9, NAN], "ir.div(Xf, 2.0)"),
    ("neg_float_flips_the_sign_of_zero", "-z", col(Z) + op("NEG"), "Float64", [0.0, -0.0, NAN, -1.0, 1.0], "-Zf"),
    ("abs_float", "abs.(z)", col(Z) + op("ABS"), "Float64", [0.0, 0.0, NAN, 1.0, 1.0], "abs(Zf)"),
    ("nan_is_not_equal_to_itself", "x .== x", col(X) + col(X) + op("EQ"), "Bool", [T, T, T, T, F], "Xf == Xf"),
    ("nan_differs_from_itself", "x .!= x", col(X) + col(X) + op("NE"), "Bool", [F, F, F, F, T], "Xf != Xf"),
    # Int64 / UInt64 against Float64 are compared as the exact real values: Float64(typemax(Int64)) is 2^63, which no Int64 equals
    ("typemax_int64_is_not_2p63", "big .== 2.0^63", col(BIG) + cf(2.0 ** 63) + op("EQ"), "Bool", [F, F, F, F, F], "Big == 2.0 ** 63"),
    ("every_int64_is_below_2p63", "big .< 2.0^63", col(BIG) + cf(2.0 ** 63) + op("LT"), "Bool", [T, T, T, T, T], "Big < 2.0 ** 63"),
    ("typemin_int64_equals_minus_2p63", "big .== -2.0^63", col(BIG) + cf(-(2.0 ** 63)) + op("EQ"), "Bool", [T, F, F, F, F], "Big == -(2.0 ** 63)"),
    ("int64_vs_fraction", "big .> -0.5", col(BIG) + cf(-0.5) + op("GT"), "Bool", [F, F, T, T, T], "Big > -0.5"),
    ("uint64_equals_2p63_float", "w .== 2.0^63", col(W) + cf(2.0 ** 63) + op("EQ"), "Bool", [F, F, T, F, F], "Wc == 2.0 ** 63"),
    ("every_uint64_is_below_2p64", "w .< 2.0^64", col(W) + cf(2.0 ** 64) + op("LT"), "Bool", [T, T, T, T, T], "Wc < 2.0 ** 64"),
    ("uint64_max_is_not_2p64", "w .>= 2.0^64", col(W) + cf(2.0 ** 64) + op("GE"), "Bool", [F, F, F, F, F], "Wc >= 2.0 ** 64"),
    ("uint64_vs_negative_float", "w .> -1.5", col(W) + cf(-1.5) + op("GT"), "Bool", [T, T, T, T, T], "Wc > -1.5"),
    # Bool
    ("not_bool", ".!b", col(B) + op("NOT"), "Bool", [F, T, F, T, F], "~Bc"),
    ("bool_less_than_true", "b .< true", col(B) + cb(True) + op("LT"), "Bool", [F, T, F, T, F], "Bc < True"),
    ("bool_equals_int", "b .== 1", col(B) + ci(1) + op("EQ"), "Bool", [T, F, T, F, T], "Bc == 1"),
    # strings are compared byte-wise (= by code point for valid UTF-8); sizeof counts bytes
    ("utf8_sizeof", "sizeof.(us)", col(US) + op("SIZEOF"), "Int64", [2, 2, 6, 1, 0], "ir.sizeof(Us)"),
    ("utf8_equality", 'us .== "日本"', col(US) + cs("日本") + op("EQ"), "Bool", [F, F, T, F, F], 'Us == "日本"'),
    ("utf8_order", 'us .< "z"', col(US) + cs("z") + op("LT"), "Bool", [F, F, F, T, T], 'Us < "z"'),
    ("utf8_startswith", 'startswith.(us, "日")', col(US) + cs("日") + op("STARTSWITH"), "Bool", [F, F, T, F, F], 'ir.startswith(Us, "日")'),
    ("string_ge", 's .>= "sony"', col(STR) + cs("sony") + op("GE"), "Bool", [F, T, F, F, T], 'St >= "sony"'),
    ("empty_prefix_matches_everything", 'startswith.(s, "")', col(STR) + cs("") + op("STARTSWITH"), "Bool", [T, T, T, T, T], 'ir.startswith(St, "")'),
    ("empty_suffix_matches_everything", 'endswith.(s, "")', col(STR) + cs("") + op("ENDSWITH"), "Bool", [T, T, T, T, T], 'ir.endswith(St, "")'),
    ("nullable_string_equality_is_three_valued", 'ns .== "a"', col(NS) + cs("a") + op("EQ"), "Missing(Bool)", [T, MISS, F, MISS, F], 'Ns == "a"'),
    ("nullable_string_ismissing", "ismissing.(ns)", col(NS) + op("ISMISSING"), "Bool", [F, T, F, T, F], "ir.ismissing(Ns)"),
    ("nullable_string_coalesced_predicate", 'coalesce.(ns .== "a", false)', col(NS) + cs("a") + op("EQ") + cb(False) + op("COALESCE"), "Bool", [T, F, F, F, F], 'ir.coalesce(Ns == "a", False)'),
]


# ---- round 4: the routes this round added, pinned to Julia-typed answers rather than to the oracle's: narrow columns through k_scan_cmp_narrow (Int8 /
# UInt8 / Bool against Int64 / Float64 constants: Julia compares the exact values, so an out-of-range constant decides the comparison by itself), Float32
# against a Float64 constant (0.1f0 is 0.100000001490116...: not below 0.1), and `coalesce(<string comparison>, false)` over a Union{String,Missing} column,
# which K5 now answers itself (`==` / `!=` with missing are missing in Julia: Base.:(==)(::Missing, ::Any); startswith / endswith have no Missing method
# in Base, so they are not claimed here)
CASES += [
    ("nullable_string_ne_coalesced", 'coalesce.(ns .!= "a", false)', col(NS) + cs("a") + op("NE") + cb(False) + op("COALESCE"), "Bool", [F, F, T, F, T], 'ir.coalesce(Ns != "a", False)'),
    ("nullable_string_eq_empty_coalesced", 'coalesce.(ns .== "", false)', col(NS) + cs("") + op("EQ") + cb(False) + op("COALESCE"), "Bool", [F, F, T, F, F], 'ir.coalesce(Ns == "", False)'),
    ("nullable_string_ne_empty_coalesced", 'coalesce.(ns .!= "", false)', col(NS) + cs("") + op("NE") + cb(False) + op("COALESCE"), "Bool", [T, F, F, F, T], 'ir.coalesce(Ns != "", False)'),
    ("int8_gt_minus1", "i8 .> -1", col(I8C) + ci(-1) + op("GT"), "Bool", [F, F, T, T, T], "I8c > -1"),
    ("int8_le_typemin", "i8 .<= -128", col(I8C) + ci(-128) + op("LE"), "Bool", [T, F, F, F, F], "I8c <= -128"),
    ("int8_lt_constant_above_its_range", "i8 .< 1000", col(I8C) + ci(1000) + op("LT"), "Bool", [T, T, T, T, T], "I8c < 1000"),
    ("int8_eq_constant_above_its_range", "i8 .== 128", col(I8C) + ci(128) + op("EQ"), "Bool", [F, F, F, F, F], "I8c == 128"),
    ("int8_ne_constant_below_its_range", "i8 .!= -129", col(I8C) + ci(-129) + op("NE"), "Bool", [T, T, T, T, T], "I8c != -129"),
    ("int8_ge_fraction", "i8 .>= 0.5", col(I8C) + cf(0.5) + op("GE"), "Bool", [F, F, F, T, T], "I8c >= 0.5"),
    ("uint8_gt_negative_constant", "u .> -1", col(U) + ci(-1) + op("GT"), "Bool", [T, T, T, T, T], "Uc > -1"),
    ("uint8_eq_255", "u .== 255", col(U) + ci(255) + op("EQ"), "Bool", [F, F, F, F, T], "Uc == 255"),
    ("uint8_ge_128", "u .>= 128", col(U) + ci(128) + op("GE"), "Bool", [F, F, F, T, T], "Uc >= 128"),
    ("uint8_lt_fraction", "u .< 127.5", col(U) + cf(127.5) + op("LT"), "Bool", [T, T, T, F, F], "Uc < 127.5"),
    ("uint8_le_constant_above_its_range", "u .<= 256", col(U) + ci(256) + op("LE"), "Bool", [T, T, T, T, T], "Uc <= 256"),
    ("bool_eq_true", "b .== true", col(B) + cb(True) + op("EQ"), "Bool", [T, F, T, F, T], "Bc == True"),
    ("bool_ne_true", "b .!= true", col(B) + cb(True) + op("NE"), "Bool", [F, T, F, T, F], "Bc != True"),
    ("float32_tenth_is_not_below_the_float64_tenth", "f .< 0.1", col(FF) + cf(0.1) + op("LT"), "Bool", [F, T, F, F, F], "Ff < 0.1"),
    ("float32_ge_integer_constant", "f .>= 3", col(FF) + ci(3) + op("GE"), "Bool", [F, F, T, F, T], "Ff >= 3"),
    ("float32_ne_nan_row", "f .!= 3", col(FF) + ci(3) + op("NE"), "Bool", [T, T, T, T, F], "Ff != 3"),
]

def main():
    out = {"comment": "hand-assembled from include/dfdb_ir.h by tests/golden/make_ir_golden.py; expected = Julia semantics",
           "table": {"a": [-7, -1, 0, 3, 10], "x": [0.5, -2.0, 3.0, 1e10, "NaN"], "s": ["apple", "sony", "", "sonic", "xs"],
                     "m": [1, None, 3, None, 0], "u": [0, 1, 127, 128, 255],
                     "i8": [-128, -1, 0, 1, 127], "w": [0, 1, 2 ** 63, 2 ** 63 + 5, 2 ** 64 - 1], "z": [-0.0, 0.0, "NaN", 1.0, -1.0],
                     "f": ["NaN" if v != v else v for v in F32COL], "b": [True, False, True, False, True], "mb": [True, None, False, None, True],
                     "big": [TMIN, -1, 0, 1, TMAX], "us": ["é", "ß", "日本", "a", ""], "ns": ["a", None, "", None, "b"], "block_size": 2},
           "opcodes": {k: v for k, v in sorted(S.items()) if k.startswith("DFIR_")},
           "dtypes": {k: v for k, v in sorted(S.items()) if k.startswith("DFDB_")},
           "cases": []}
    for name, julia, b, rtype, want, py in CASES:
        if isinstance(want, list):
            want = ["NaN" if isinstance(v, float) and v != v else v for v in want]
        out["cases"].append({"name": name, "julia": julia, "hex": b.hex(), "type": rtype, "expect": want, "ir_py": py})
    with open(os.path.join(HERE, "ir_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(out["cases"]), "cases")


if __name__ == "__main__":
    main()

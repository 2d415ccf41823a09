/* dfdb_ir.h — predicate / computed-column expression IR shared by every front-end.
 *
 * The reference expresses predicates and computed columns as a tree of
 * `BlockBroadcasting{RT,F,Args}` nodes whose leaves are `ColRef{T}` or 0-dim
 * scalars (reference: src/tables/broadcast.jl:2-17) and JIT-fuses any Julia
 * function over them (broadcast.jl:60-68).  A C ABI cannot carry closures, so
 * the tree is serialised as a little-endian POSTFIX byte stream over the closed
 * operator set the reference's tests and docs exercise.
 *
 *   stream := token*            (evaluation leaves exactly one value on the stack)
 *   token  := opcode:u8 payload
 *
 * Result types are NOT carried: every consumer infers them with Julia's
 * promotion rules (see DESIGN.md "IR typing"), the same way the reference gets
 * RT from Base._return_type (broadcast.jl:13).
 */
#ifndef DFDB_IR_H
#define DFDB_IR_H

/* ---- column / scalar dtypes (ColumnTypes names: src/columntypes/base.jl:108-126) ---- */
enum {
  DFDB_I8 = 1, DFDB_I16 = 2, DFDB_I32 = 3, DFDB_I64 = 4,
  DFDB_U8 = 5, DFDB_U16 = 6, DFDB_U32 = 7, DFDB_U64 = 8,
  DFDB_F32 = 9, DFDB_F64 = 10, DFDB_BOOL = 11, DFDB_STRING = 12,
  DFDB_DTYPE_MASK = 0x3f,
  DFDB_NULLABLE = 0x80 /* Union{T,Missing}: "Missing(T)" on disk */
};

/* ---- leaves ---- */
#define DFIR_COL        0x01 /* payload: u32 column ordinal (0-based position in the table) */
#define DFIR_CONST      0x02 /* payload: u8 dtype, 8 bytes (value bit pattern, zero/sign extended) */
#define DFIR_CONST_STR  0x03 /* payload: u32 nbytes, bytes (Julia: "x" / Ref("x")) */
#define DFIR_CONST_SET  0x04 /* payload: u8 dtype, u32 n, n*8 bytes (Julia: Ref([..]) for in.()) */

/* ---- arithmetic (binary unless noted) ---- */
#define DFIR_ADD   0x10
#define DFIR_SUB   0x11
#define DFIR_MUL   0x12
#define DFIR_DIV   0x13 /* Julia `/`  : integers -> Float64 */
#define DFIR_IDIV  0x14 /* Julia `÷`  : truncating, DivideError on 0 */
#define DFIR_REM   0x15 /* Julia `%`  : sign of dividend, DivideError on 0 */
#define DFIR_MOD   0x16 /* Julia mod(): sign of divisor */
#define DFIR_NEG   0x17 /* unary */
#define DFIR_ABS   0x18 /* unary */
#define DFIR_MIN   0x19
#define DFIR_MAX   0x1a

/* ---- comparisons -> Bool (Int vs Float compared exactly, like Julia) ---- */
#define DFIR_EQ    0x20
#define DFIR_NE    0x21
#define DFIR_LT    0x22
#define DFIR_LE    0x23
#define DFIR_GT    0x24
#define DFIR_GE    0x25

/* ---- logic: Bool (non-short-circuit, like `&` in selection.jl:46) or bitwise on ints ---- */
#define DFIR_AND   0x30
#define DFIR_OR    0x31
#define DFIR_XOR   0x32
#define DFIR_NOT   0x33 /* unary `!` */

/* ---- set / string / missing ---- */
#define DFIR_IN_SET      0x40 /* stack: value, set        -> Bool   (in.(a, Ref(v)), numeric sets; a set of strings is lowered by the front ends to (a == v1) | (a == v2) | ...) */
#define DFIR_STARTSWITH  0x41 /* stack: string col, const -> Bool */
#define DFIR_ENDSWITH    0x42
#define DFIR_ISMISSING   0x43 /* unary on a nullable column -> Bool */
#define DFIR_SIZEOF      0x44 /* unary: sizeof(string) -> Int64 */
#define DFIR_COALESCE    0x45 /* binary: coalesce(a, b) = a unless it is missing, else b (same base type; result nullable iff b is) */

/* ---- conversion ---- */
#define DFIR_CAST  0x50 /* payload: u8 dtype ; Julia T(x) / convert */

#endif

/* orc_internal.h — private structures of the CPU oracle (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* the three liblz4 entry points CodecLz4 binds (src/io/BlockStreams.jl:39,42-48,110-111) */
int LZ4_compressBound(int inputSize);
int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration);
int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);

int orc_fail(int code, const char* fmt, ...);

typedef struct { uint8_t* p; size_t n, cap; } bytes_t;
int bytes_reserve(bytes_t* b, size_t cap);
int bytes_append(bytes_t* b, const void* src, size_t n);

static inline int dt_base(int32_t dt) { return dt & DFDB_DTYPE_MASK; }
static inline int dt_nullable(int32_t dt) { return (dt & DFDB_NULLABLE) != 0; }
static inline int dt_isint(int32_t dt) { int b = dt_base(dt); return b >= DFDB_I8 && b <= DFDB_U64; }
static inline int dt_issigned(int32_t dt) { int b = dt_base(dt); return b >= DFDB_I8 && b <= DFDB_I64; }
static inline int dt_isfloat(int32_t dt) { int b = dt_base(dt); return b == DFDB_F32 || b == DFDB_F64; }
static inline int dt_isnum(int32_t dt) { return dt_isint(dt) || dt_isfloat(dt) || dt_base(dt) == DFDB_BOOL; }
int dt_width(int32_t dt);               /* bytes per row on disk (String: 0) */
const char* dt_name(int32_t dt);        /* ColumnTypes.typestring */
int dt_parse(const char* s, size_t n, int32_t* out);
int dt_parse_ex(const char* s, size_t n, int32_t* out, char* logical /*[32]*/);   /* + the bits types carried as integers */
const char* dt_type_string(int32_t dt, const char* logical);

/* ---- table ---- */
typedef struct {
  int64_t id;
  char name[128];
  int32_t dtype;
  char logical[32]; /* bits types stored as an integer: "Date", "DateTime", "Time" (Int64), "Char" (UInt32); "" otherwise */
  bytes_t image;   /* <id>.bin : header + blocks */
  size_t  data_off; /* first block */
  int64_t nrows;
} col_t;

struct orc_table {
  int64_t block_size, format_version;
  int ncols;
  col_t* cols;
};

/* ---- decoded block buffer: make_buffer (materialization.jl:1-8) ---- */
typedef struct {
  int32_t dtype;
  int64_t rows;
  uint8_t* data; size_t data_cap;       /* Vector{T} */
  uint8_t* missing; size_t miss_cap;    /* Union{T,Missing}: 1 = missing */
  /* FlatStringsVector (FlatStringsVectors.jl:5-9) */
  int32_t* sizes; int64_t* offsets; size_t str_cap;
  uint8_t* sdata; size_t sdata_cap; int64_t datasize;
  int external;                          /* data pointers borrowed (selexec_apply) */
} colbuf_t;
void colbuf_free(colbuf_t* b);

/* ---- BlockStream (BlockStreams.jl:9-15) over an in-memory file image ---- */
typedef struct {
  const uint8_t* img; size_t n, pos;
  bytes_t uncomp, comp;
} stream_t;
void stream_init(stream_t* s, const uint8_t* img, size_t n, size_t pos);
void stream_free(stream_t* s);
static inline int stream_eof(const stream_t* s) { return s->pos >= s->n; }
int stream_skip_block(stream_t* s, orc_sizestats* st);
int stream_read_block(stream_t* s, colbuf_t* buf, orc_sizestats* st);

/* ---- expressions ---- */
typedef struct node {
  int op;               /* DFIR_* */
  int32_t dtype;        /* inferred result type */
  int col;              /* DFIR_COL */
  int32_t cdtype;       /* const dtype */
  int64_t ci; double cf;/* const value */
  uint8_t* str; int32_t slen;
  int64_t* set_i; double* set_f; int32_t nset; int32_t set_dtype;
  int cast_to;
  struct node *a, *b;
} node_t;
int expr_parse(const orc_table* t, const uint8_t* ir, size_t len, node_t** out);
node_t* expr_clone(const node_t* n);
void expr_free(node_t* n);
/* appends ordinals in first-appearance order (columns_buffers merge order: broadcast.jl:70-80) */
void expr_required(const node_t* n, int32_t* ords, int* count, int cap);
node_t* expr_and(node_t* a, node_t* b); /* BlockBroadcasting(&, (old, new)): selection.jl:44-47 */

typedef struct { uint8_t* base; size_t used, cap; void* overflow; } arena_t;
void* arena_alloc(arena_t* a, size_t n);
void arena_reset(arena_t* a);
void arena_free(arena_t* a);

/* result of evaluating a node over `n` gathered rows */
typedef struct {
  int32_t dtype; int is_const; int64_t n;
  int64_t* i; double* f; uint8_t* b;   /* by class: int (I8..U64 widened to 64 bit) / float / bool */
  uint8_t* miss;                        /* non-NULL if nullable */
  const colbuf_t* scol;                 /* string column reference (evaluated through idx) */
  const uint8_t* cstr; int32_t cstr_len;
} vec_t;

/* BroadcastExecutor.eval_on_range (broadcast.jl:121-133): gather the inputs at idx[0..n) (NULL = all
 * rows) then evaluate; result in arena */
int expr_eval(const node_t* nd, const colbuf_t* bufs, const int32_t* idx, int64_t n, arena_t* ar, vec_t* out);

#endif

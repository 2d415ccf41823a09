/* orc_codec.c — file format, block codec and block bodies of the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Restates:
 *   src/io/BlockStreams.jl:31-119   (framing + LZ4)
 *   src/io/blocks.jl:2-71           (block bodies)
 *   src/FlatStringsVectors.jl:61-70 (offset rebuild)
 *   src/io/filesystem.jl:8-54, src/io/table_io.jl:1-33, src/io/common_io.jl:1-8 (files)
 */
#include "orc_internal.h"
#include <stdarg.h>
#include <errno.h>
#include <sys/stat.h>

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
int orc_fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return code;
}

int bytes_reserve(bytes_t* b, size_t cap) {
  if (cap <= b->cap) return 0;
  size_t nc = b->cap ? b->cap : 64;
  while (nc < cap) nc *= 2;
  uint8_t* p = (uint8_t*)realloc(b->p, nc);
  if (!p) return orc_fail(ORC_ERR_NOMEM, "out of memory");
  b->p = p; b->cap = nc; return 0;
}
int bytes_append(bytes_t* b, const void* src, size_t n) {
  int rc = bytes_reserve(b, b->n + n); if (rc) return rc;
  memcpy(b->p + b->n, src, n); b->n += n; return 0;
}

/* ---------------------------------------------------------------- dtypes */
static const char* k_names[] = {0, "Int8", "Int16", "Int32", "Int64", "UInt8", "UInt16", "UInt32",
                                "UInt64", "Float32", "Float64", "Bool", "String"};
static const int k_width[] = {0, 1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 1, 0};
int dt_width(int32_t dt) { return k_width[dt_base(dt)]; }
const char* dt_name(int32_t dt) { /* columntypes/base.jl:108-126, complex.jl:1-8 */
  static __thread char buf[64];
  int b = dt_base(dt);
  if (b < 1 || b > 12) return "?";
  if (dt_nullable(dt)) { snprintf(buf, sizeof buf, "Missing(%s)", k_names[b]); return buf; }
  return k_names[b];
}
int dt_parse(const char* s, size_t n, int32_t* out) {
  int32_t flag = 0;
  if (n > 9 && memcmp(s, "Missing(", 8) == 0 && s[n - 1] == ')') { flag = DFDB_NULLABLE; s += 8; n -= 9; }
  for (int b = 1; b <= 12; b++)
    if (strlen(k_names[b]) == n && memcmp(k_names[b], s, n) == 0) { *out = b | flag; return 0; }
  return orc_fail(ORC_ERR_UNSUPPORTED, "undefined column type '%.*s'", (int)n, s);
}

/* Julia bits types whose blocks are plain integers (read_block_body! is a memcpy for every isbits T: blocks.jl:37-44;
 * type strings: columntypes/base.jl:108-126, Dates in columntypes/complex.jl).  The hot path treats them as their
 * integer representation (Date: days, DateTime: milliseconds, Time: nanoseconds since the Rata Die epoch / midnight,
 * Char: the UInt32 the UTF-8 bytes are left-aligned in); the type string travels beside the storage dtype. */
static const struct { const char* name; int dtype; } k_alias[] = {{"Date", DFDB_I64}, {"DateTime", DFDB_I64}, {"Time", DFDB_I64}, {"Char", DFDB_U32}};
int dt_parse_ex(const char* s, size_t n, int32_t* out, char* logical) {
  int32_t flag = 0;
  if (logical) logical[0] = 0;
  const char* b = s; size_t bn = n;
  if (n > 9 && memcmp(s, "Missing(", 8) == 0 && s[n - 1] == ')') { flag = DFDB_NULLABLE; b += 8; bn -= 9; }
  for (size_t k = 0; k < sizeof k_alias / sizeof k_alias[0]; k++)
    if (strlen(k_alias[k].name) == bn && memcmp(k_alias[k].name, b, bn) == 0) {
      *out = k_alias[k].dtype | flag;
      if (logical) snprintf(logical, 32, "%s", k_alias[k].name);
      return 0;
    }
  return dt_parse(s, n, out);
}
const char* dt_type_string(int32_t dt, const char* logical) {
  static __thread char buf[64];
  if (!logical || !logical[0]) return dt_name(dt);
  if (dt_nullable(dt)) { snprintf(buf, sizeof buf, "Missing(%s)", logical); return buf; }
  return logical;
}

/* ---------------------------------------------------------------- little-endian IO helpers */
static int put_i32(bytes_t* b, int32_t v) { return bytes_append(b, &v, 4); }
static int put_i64(bytes_t* b, int64_t v) { return bytes_append(b, &v, 8); }
static int put_string(bytes_t* b, const char* s) { /* write_string: common_io.jl:1-4 */
  int32_t n = (int32_t)strlen(s);
  int rc = put_i32(b, n); if (rc) return rc;
  return bytes_append(b, s, (size_t)n);
}
typedef struct { const uint8_t* p; size_t n, pos; } rd_t;
static int get_i32(rd_t* r, int32_t* v) { if (r->pos + 4 > r->n) return -1; memcpy(v, r->p + r->pos, 4); r->pos += 4; return 0; }
static int get_i64(rd_t* r, int64_t* v) { if (r->pos + 8 > r->n) return -1; memcpy(v, r->p + r->pos, 8); r->pos += 8; return 0; }
static int get_string(rd_t* r, char* out, size_t cap, size_t* len) { /* read_string: common_io.jl:5-8 */
  int32_t n; if (get_i32(r, &n) || n < 0 || r->pos + (size_t)n > r->n || (size_t)n + 1 > cap) return -1;
  memcpy(out, r->p + r->pos, (size_t)n); out[n] = 0; r->pos += (size_t)n; if (len) *len = (size_t)n; return 0;
}

/* ---------------------------------------------------------------- tables */
int orc_table_create(int64_t block_size, orc_table** out) {
  orc_table* t = (orc_table*)calloc(1, sizeof *t);
  if (!t) return orc_fail(ORC_ERR_NOMEM, "out of memory");
  t->block_size = block_size > 0 ? block_size : ORC_DEFAULT_BLOCK_SIZE;
  t->format_version = ORC_FORMAT_VERSION;
  *out = t; return 0;
}
void orc_table_free(orc_table* t) {
  if (!t) return;
  for (int i = 0; i < t->ncols; i++) free(t->cols[i].image.p);
  free(t->cols); free(t);
}
int orc_table_ncols(orc_table* t) { return t->ncols; }
int64_t orc_table_block_size(orc_table* t) { return t->block_size; }
int orc_table_colinfo(orc_table* t, int i, int64_t* id, char* name, size_t cap, int32_t* dtype) {
  if (i < 0 || i >= t->ncols) return orc_fail(ORC_ERR_BOUNDS, "column %d out of range", i);
  if (id) *id = t->cols[i].id;
  if (name) snprintf(name, cap, "%s", t->cols[i].name);
  if (dtype) *dtype = t->cols[i].dtype;
  return 0;
}
int orc_table_find(orc_table* t, const char* name) {
  for (int i = 0; i < t->ncols; i++) if (strcmp(t->cols[i].name, name) == 0) return i;
  return -1;
}
const uint8_t* orc_table_image(orc_table* t, int i, size_t* nbytes) {
  if (i < 0 || i >= t->ncols) return NULL;
  if (nbytes) *nbytes = t->cols[i].image.n;
  return t->cols[i].image.p;
}

/* make_column_file header: Int64 block_size + type string (filesystem.jl:14-23) */
static int write_col_header(col_t* c, int64_t block_size) {
  int rc = put_i64(&c->image, block_size); if (rc) return rc;
  rc = put_string(&c->image, dt_type_string(c->dtype, c->logical)); if (rc) return rc;
  c->data_off = c->image.n; return 0;
}

/* commit_block_write! (BlockStreams.jl:36-60): LZ4_compress_fast(acceleration 2), header
 * Int32 rows | Int64 origin | Int64 compressed, then the compressed bytes. */
static int append_block(bytes_t* img, const uint8_t* body, int64_t body_bytes, int32_t rows) {
  if (body_bytes == 0) return 0; /* size_to_compress == 0 && return (0,0): BlockStreams.jl:38 */
  if (body_bytes > 0x7E000000) return orc_fail(ORC_ERR_ARGUMENT, "block body too large for LZ4");
  int bound = LZ4_compressBound((int)body_bytes);
  int rc = bytes_reserve(img, img->n + 20 + (size_t)bound); if (rc) return rc;
  int csz = LZ4_compress_fast((const char*)body, (char*)img->p + img->n + 20, (int)body_bytes, bound, ORC_COMPRESSION_LEVEL);
  if (csz <= 0) return orc_fail(ORC_ERR_FORMAT, "LZ4_compress_fast failed");
  int64_t origin = body_bytes, comp = csz;
  memcpy(img->p + img->n, &rows, 4);
  memcpy(img->p + img->n + 4, &origin, 8);
  memcpy(img->p + img->n + 12, &comp, 8);
  img->n += 20 + (size_t)csz;
  return 0;
}
int orc_block_encode(const uint8_t* body, int64_t body_bytes, int32_t rows, uint8_t* out, size_t cap, size_t* written) {
  bytes_t b = {0, 0, 0};
  int rc = append_block(&b, body, body_bytes, rows);
  if (!rc) {
    if (b.n > cap) rc = orc_fail(ORC_ERR_ARGUMENT, "output too small");
    else { memcpy(out, b.p, b.n); *written = b.n; }
  }
  free(b.p); return rc;
}

/* write_block_body (blocks.jl:2-33) for rows [r0, r0+rows) of a caller column */
static int build_body(bytes_t* body, int32_t dtype, int64_t r0, int64_t rows, const void* data,
                      const uint8_t* sbytes, const int64_t* soffsets, const uint8_t* missing) {
  body->n = 0;
  int w = dt_width(dtype);
  if (dt_base(dtype) == DFDB_STRING) { /* Int32 datasize, sizes, bytes: blocks.jl:21-33 */
    const int32_t* sz = (const int32_t*)data + r0;
    int64_t total = 0;
    for (int64_t i = 0; i < rows; i++) if (sz[i] > 0) total += sz[i];
    if (total > 0x7fffffff) return orc_fail(ORC_ERR_ARGUMENT, "string block exceeds Int32 bytes");
    int rc = put_i32(body, (int32_t)total); if (rc) return rc;
    rc = bytes_append(body, sz, (size_t)rows * 4); if (rc) return rc;
    return bytes_append(body, sbytes + soffsets[r0], (size_t)total);
  }
  if (dt_nullable(dtype)) { /* BitArray chunks then values: blocks.jl:9-18 */
    int64_t nchunks = (rows + 63) / 64;
    int rc = bytes_reserve(body, (size_t)nchunks * 8 + (size_t)rows * w); if (rc) return rc;
    uint64_t* ch = (uint64_t*)body->p; memset(ch, 0, (size_t)nchunks * 8);
    for (int64_t i = 0; i < rows; i++) if (missing && missing[r0 + i]) ch[i >> 6] |= 1ull << (i & 63);
    body->n = (size_t)nchunks * 8;
    return bytes_append(body, (const uint8_t*)data + r0 * w, (size_t)rows * w);
  }
  return bytes_append(body, (const uint8_t*)data + r0 * w, (size_t)rows * w);
}

int orc_table_add_column(orc_table* t, const char* name, int32_t dtype, int64_t nrows,
                         const void* data, const uint8_t* bytes, const uint8_t* missing) {
  return orc_table_add_column_as(t, name, dtype, NULL, nrows, data, bytes, missing);
}
int orc_table_col_logical(orc_table* t, int i, char* buf, size_t cap) {
  if (i < 0 || i >= t->ncols) return orc_fail(ORC_ERR_BOUNDS, "column %d out of range", i);
  snprintf(buf, cap, "%s", t->cols[i].logical); return 0;
}
/* logical: NULL, or "Date" / "DateTime" / "Time" (dtype Int64) / "Char" (dtype UInt32): the type string written to the files */
int orc_table_add_column_as(orc_table* t, const char* name, int32_t dtype, const char* logical, int64_t nrows,
                            const void* data, const uint8_t* bytes, const uint8_t* missing) {
  if (logical && logical[0]) {
    int32_t want; char lg[32];
    if (dt_parse_ex(logical, strlen(logical), &want, lg) || !lg[0] || dt_base(want) != dt_base(dtype))
      return orc_fail(ORC_ERR_ARGUMENT, "logical type %s does not go with dtype %d", logical, dtype);
  }
  if (orc_table_find(t, name) >= 0) return orc_fail(ORC_ERR_ARGUMENT, "Duplicated column %s", name);
  if (dt_base(dtype) < 1 || dt_base(dtype) > 12) return orc_fail(ORC_ERR_UNSUPPORTED, "unsupported dtype %d", dtype);
  col_t* nc = (col_t*)realloc(t->cols, sizeof(col_t) * (size_t)(t->ncols + 1));
  if (!nc) return orc_fail(ORC_ERR_NOMEM, "out of memory");
  t->cols = nc;
  col_t* c = &t->cols[t->ncols];
  memset(c, 0, sizeof *c);
  c->id = t->ncols + 1; /* DFTableMeta ctor numbers ids 1..n: meta.jl:26-29 */
  snprintf(c->name, sizeof c->name, "%s", name);
  c->dtype = dtype; c->nrows = nrows;
  if (logical) snprintf(c->logical, sizeof c->logical, "%s", logical);
  int rc = write_col_header(c, t->block_size); if (rc) return rc;
  int64_t* soff = NULL;
  if (dt_base(dtype) == DFDB_STRING) {
    soff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nrows + 1));
    const int32_t* sz = (const int32_t*)data; int64_t o = 0;
    for (int64_t i = 0; i < nrows; i++) { soff[i] = o; if (sz[i] > 0) o += sz[i]; }
    soff[nrows] = o;
  }
  bytes_t body = {0, 0, 0};
  for (int64_t r0 = 0; r0 < nrows && !rc; r0 += t->block_size) {
    int64_t rows = nrows - r0 < t->block_size ? nrows - r0 : t->block_size;
    rc = build_body(&body, dtype, r0, rows, data, bytes, soff, missing);
    if (!rc) rc = append_block(&c->image, body.p, (int64_t)body.n, (int32_t)rows);
  }
  free(body.p); free(soff);
  if (rc) { free(c->image.p); return rc; }
  t->ncols++;
  return 0;
}

int orc_table_save(orc_table* t, const char* path) {
  if (mkdir(path, 0777) != 0 && errno != EEXIST) return orc_fail(ORC_ERR_IO, "cannot create %s", path);
  bytes_t m = {0, 0, 0}; /* write_table_meta: table_io.jl:9-19 */
  put_i64(&m, t->format_version); put_i64(&m, t->block_size); put_i64(&m, t->ncols);
  for (int i = 0; i < t->ncols; i++) { put_i64(&m, t->cols[i].id); put_string(&m, t->cols[i].name); put_string(&m, dt_type_string(t->cols[i].dtype, t->cols[i].logical)); }
  char fn[1024];
  snprintf(fn, sizeof fn, "%s/meta.bin", path);
  FILE* f = fopen(fn, "wb"); if (!f) { free(m.p); return orc_fail(ORC_ERR_IO, "cannot write %s", fn); }
  fwrite(m.p, 1, m.n, f); fclose(f); free(m.p);
  for (int i = 0; i < t->ncols; i++) {
    snprintf(fn, sizeof fn, "%s/%lld.bin", path, (long long)t->cols[i].id); /* columnpath: filesystem.jl:11 */
    f = fopen(fn, "wb"); if (!f) return orc_fail(ORC_ERR_IO, "cannot write %s", fn);
    fwrite(t->cols[i].image.p, 1, t->cols[i].image.n, f); fclose(f);
  }
  return 0;
}

static int slurp(const char* fn, bytes_t* out) {
  FILE* f = fopen(fn, "rb"); if (!f) return -1;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  if (bytes_reserve(out, (size_t)n + 1)) { fclose(f); return -1; }
  out->n = fread(out->p, 1, (size_t)n, f); fclose(f);
  return out->n == (size_t)n ? 0 : -1;
}

int orc_table_open(const char* path, orc_table** out) {
  char fn[1024]; snprintf(fn, sizeof fn, "%s/meta.bin", path);
  bytes_t m = {0, 0, 0};
  if (slurp(fn, &m)) { free(m.p); return orc_fail(ORC_ERR_IO, "table %s don't exists", path); }
  rd_t r = {m.p, m.n, 0};
  orc_table* t = (orc_table*)calloc(1, sizeof *t);
  int64_t ncols = 0; int rc = 0;
  if (get_i64(&r, &t->format_version) || get_i64(&r, &t->block_size) || get_i64(&r, &ncols) || ncols < 0 || ncols > 100000)
    rc = orc_fail(ORC_ERR_FORMAT, "bad meta.bin in %s", path);
  if (!rc) t->cols = (col_t*)calloc((size_t)ncols + 1, sizeof(col_t));
  for (int64_t i = 0; i < ncols && !rc; i++) {
    col_t* c = &t->cols[i]; char ty[128]; size_t tl;
    if (get_i64(&r, &c->id) || get_string(&r, c->name, sizeof c->name, NULL) || get_string(&r, ty, sizeof ty, &tl))
      rc = orc_fail(ORC_ERR_FORMAT, "bad meta.bin in %s", path);
    else rc = dt_parse_ex(ty, tl, &c->dtype, c->logical);
    if (!rc) t->ncols++;
  }
  free(m.p);
  for (int i = 0; i < t->ncols && !rc; i++) { /* check_column_file: filesystem.jl:56-61 */
    col_t* c = &t->cols[i];
    snprintf(fn, sizeof fn, "%s/%lld.bin", path, (long long)c->id);
    if (slurp(fn, &c->image)) { rc = orc_fail(ORC_ERR_IO, "column file '%s' for column %s don't exists", fn, c->name); break; }
    rd_t h = {c->image.p, c->image.n, 0}; int64_t bs; char ty[128]; size_t tl; int32_t dt;
    if (get_i64(&h, &bs) || get_string(&h, ty, sizeof ty, &tl)) { rc = orc_fail(ORC_ERR_FORMAT, "bad header in %s", fn); break; }
    if (bs != t->block_size) { rc = orc_fail(ORC_ERR_FORMAT, "column %s has blocksize %lld, but table has blocksize %lld", c->name, (long long)bs, (long long)t->block_size); break; }
    char lg[32];
    rc = dt_parse_ex(ty, tl, &dt, lg); if (rc) break;
    if (dt != c->dtype || strcmp(lg, c->logical) != 0) { rc = orc_fail(ORC_ERR_FORMAT, "column %s stored type is %s, but another expected", c->name, ty); break; }
    c->data_off = h.pos;
  }
  if (rc) { orc_table_free(t); return rc; }
  *out = t; return 0;
}

/* ---------------------------------------------------------------- BlockStream */
void stream_init(stream_t* s, const uint8_t* img, size_t n, size_t pos) { memset(s, 0, sizeof *s); s->img = img; s->n = n; s->pos = pos; }
void stream_free(stream_t* s) { free(s->uncomp.p); free(s->comp.p); s->uncomp.p = s->comp.p = NULL; }

int orc_block_sizes(const uint8_t* p, size_t avail, int32_t* rows, int64_t* origin, int64_t* compressed) {
  if (avail < 20) return orc_fail(ORC_ERR_FORMAT, "truncated block header");
  memcpy(rows, p, 4); memcpy(origin, p + 4, 8); memcpy(compressed, p + 12, 8);
  if (*rows < 0 || *origin < 0 || *compressed < 0 || (uint64_t)*compressed > avail - 20)
    return orc_fail(ORC_ERR_FORMAT, "corrupt block header");
  return 0;
}

/* stats_from_block adds sizeof((Int32,Int64,Int64)) = 24 (quirk Q10: BlockStreams.jl:7,23) */
static void fill_stats(orc_sizestats* st, int32_t rows, int64_t comp, int64_t origin) {
  if (st) { st->rows = rows; st->compressed = comp + 24; st->uncompressed = origin; }
}

int stream_skip_block(stream_t* s, orc_sizestats* st) { /* skip_block: BlockStreams.jl:74-78 */
  int32_t rows; int64_t origin, comp;
  int rc = orc_block_sizes(s->img + s->pos, s->n - s->pos, &rows, &origin, &comp); if (rc) return rc;
  s->pos += 20 + (size_t)comp;
  fill_stats(st, rows, comp, origin);
  return 0;
}

void colbuf_free(colbuf_t* b) {
  if (!b->external) { free(b->data); free(b->missing); free(b->sizes); free(b->sdata); }
  free(b->offsets);
  memset(b, 0, sizeof *b);
}
static int ensure(uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap) return 0;
  size_t nc = *cap ? *cap : 1024; while (nc < need) nc *= 2;
  uint8_t* q = (uint8_t*)realloc(*p, nc); if (!q) return orc_fail(ORC_ERR_NOMEM, "out of memory");
  *p = q; *cap = nc; return 0;
}

/* unsafe_remake_offsets! (FlatStringsVectors.jl:61-70): serial prefix sum, datasize = sum of positive sizes */
static void fsv_remake_offsets(colbuf_t* v) {
  int64_t n = v->rows, total = 0;
  if (n > 0) {
    v->offsets[0] = 0;
    for (int64_t i = 1; i < n; i++) v->offsets[i] = v->offsets[i - 1] + (v->sizes[i - 1] >= 0 ? v->sizes[i - 1] : 0);
  }
  for (int64_t i = 0; i < n; i++) if (v->sizes[i] > 0) total += v->sizes[i];
  v->datasize = total;
}

/* read_block_body! (blocks.jl:37-71) from the decompressed buffer */
static int read_body(const uint8_t* u, int64_t origin, int32_t rows, colbuf_t* v) {
  int w = dt_width(v->dtype); int rc;
  v->rows = rows;
  if (dt_base(v->dtype) == DFDB_STRING) {
    if (origin < 4 + (int64_t)rows * 4) return orc_fail(ORC_ERR_FORMAT, "string block too short");
    int32_t datasize; memcpy(&datasize, u, 4);
    if (datasize < 0 || 4 + (int64_t)rows * 4 + datasize > origin) return orc_fail(ORC_ERR_FORMAT, "string block datasize out of range");
    if ((size_t)rows > v->str_cap) {
      size_t nc = v->str_cap ? v->str_cap : 1024; while (nc < (size_t)rows) nc *= 2;
      v->sizes = (int32_t*)realloc(v->sizes, nc * 4); v->offsets = (int64_t*)realloc(v->offsets, nc * 8); v->str_cap = nc;
    }
    memcpy(v->sizes, u + 4, (size_t)rows * 4);
    /* resize_data!: doubling growth that copies the old bytes (FlatStringsVectors.jl:93-104) */
    if ((size_t)datasize > v->sdata_cap) {
      size_t nc = v->sdata_cap ? v->sdata_cap * 2 : 1024; while (nc < (size_t)datasize) nc *= 2;
      uint8_t* nd = (uint8_t*)malloc(nc); if (!nd) return orc_fail(ORC_ERR_NOMEM, "out of memory");
      if (v->sdata) memcpy(nd, v->sdata, (size_t)v->datasize);
      free(v->sdata); v->sdata = nd; v->sdata_cap = nc;
    }
    memcpy(v->sdata, u + 4 + (size_t)rows * 4, (size_t)datasize);
    fsv_remake_offsets(v);
    return 0;
  }
  if (dt_nullable(v->dtype)) { /* bits then values: blocks.jl:46-60 */
    int64_t nchunks = ((int64_t)rows + 63) / 64;
    if (nchunks * 8 + (int64_t)rows * w > origin) return orc_fail(ORC_ERR_FORMAT, "nullable block too short");
    rc = ensure(&v->missing, &v->miss_cap, (size_t)rows); if (rc) return rc;
    rc = ensure(&v->data, &v->data_cap, (size_t)rows * w); if (rc) return rc;
    const uint64_t* ch = (const uint64_t*)u;
    for (int64_t i = 0; i < rows; i++) v->missing[i] = (uint8_t)((ch[i >> 6] >> (i & 63)) & 1);
    memcpy(v->data, u + nchunks * 8, (size_t)rows * w);
    return 0;
  }
  if ((int64_t)rows * w > origin) return orc_fail(ORC_ERR_FORMAT, "block too short");
  rc = ensure(&v->data, &v->data_cap, (size_t)rows * w); if (rc) return rc;
  memcpy(v->data, u, (size_t)rows * w); /* read!(io, v): blocks.jl:43 */
  return 0;
}

/* read_block (BlockStreams.jl:101-119) */
int stream_read_block(stream_t* s, colbuf_t* buf, orc_sizestats* st) {
  int32_t rows; int64_t origin, comp;
  int rc = orc_block_sizes(s->img + s->pos, s->n - s->pos, &rows, &origin, &comp); if (rc) return rc;
  if ((rc = bytes_reserve(&s->comp, (size_t)comp))) return rc;     /* ensureroom :105 */
  if ((rc = bytes_reserve(&s->uncomp, (size_t)origin + 8))) return rc; /* ensureroom :106 */
  memcpy(s->comp.p, s->img + s->pos + 20, (size_t)comp);            /* unsafe_read :108 */
  int got = LZ4_decompress_safe((const char*)s->comp.p, (char*)s->uncomp.p, (int)comp, (int)origin);
  if (got != origin) return orc_fail(ORC_ERR_FORMAT, "decompression error"); /* @assert :112 */
  s->pos += 20 + (size_t)comp;
  rc = read_body(s->uncomp.p, origin, rows, buf); if (rc) return rc;
  fill_stats(st, rows, comp, origin);
  return 0;
}

int orc_block_decode(const uint8_t* p, size_t avail, uint8_t* out, size_t cap, int32_t* rows, int64_t* origin, size_t* consumed) {
  int64_t comp;
  int rc = orc_block_sizes(p, avail, rows, origin, &comp); if (rc) return rc;
  if ((size_t)*origin > cap) return orc_fail(ORC_ERR_ARGUMENT, "output too small");
  int got = LZ4_decompress_safe((const char*)p + 20, (char*)out, (int)comp, (int)*origin);
  if (got != *origin) return orc_fail(ORC_ERR_FORMAT, "decompression error");
  if (consumed) *consumed = 20 + (size_t)comp;
  return 0;
}

int orc_table_column_stats(orc_table* t, int i, orc_sizestats* out, int64_t* nblocks) {
  if (i < 0 || i >= t->ncols) return orc_fail(ORC_ERR_BOUNDS, "column %d out of range", i);
  stream_t s; stream_init(&s, t->cols[i].image.p, t->cols[i].image.n, t->cols[i].data_off);
  orc_sizestats tot = {0, 0, 0}; int64_t nb = 0; int rc = 0;
  while (!stream_eof(&s)) {
    orc_sizestats st; rc = stream_skip_block(&s, &st); if (rc) break;
    tot.rows += st.rows; tot.compressed += st.compressed; tot.uncompressed += st.uncompressed; nb++;
  }
  if (out) *out = tot;
  if (nblocks) *nblocks = nb;
  return rc;
}

/* ---------------------------------------------------------------- synthetic data */
uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
void orc_gen_i64_mod1m(uint64_t seed, int64_t row_first, int64_t n, int64_t* out) {
  for (int64_t i = 0; i < n; i++) out[i] = (int64_t)(orc_splitmix64(seed + (uint64_t)(row_first + i)) % 1000000ull);
}
void orc_gen_f64_u2000(uint64_t seed, int64_t row_first, int64_t n, double* out) {
  for (int64_t i = 0; i < n; i++) out[i] = (double)(orc_splitmix64(seed + (uint64_t)(row_first + i)) >> 11) * (1.0 / 9007199254740992.0) * 2000.0;
}
static const char* k_brands[10] = {"apple", "samsung", "huawei", "microsoft", "dell", "xbox", "sony", "intel", "lenovo", "asus"};
int64_t orc_gen_str_brands10(uint64_t seed, int64_t row_first, int64_t n, int32_t* sizes, uint8_t* bytes) {
  int64_t o = 0;
  for (int64_t i = 0; i < n; i++) {
    const char* b = k_brands[orc_splitmix64(seed + (uint64_t)(row_first + i)) % 10ull];
    int32_t l = (int32_t)strlen(b);
    sizes[i] = l; memcpy(bytes + o, b, (size_t)l); o += l;
  }
  return o;
}

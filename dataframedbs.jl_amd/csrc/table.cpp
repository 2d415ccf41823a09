// table.cpp — DFTable on the device: metadata parsing, column residency, block loading.
//
// Replaces (paths under /root/reference): open_table src/tables/creators.jl:7-16, read_table_meta
// src/io/table_io.jl:21-33, check_column_head src/io/filesystem.jl:47-54, and — for the data — the
// per-block BlockStream.read_block + read_block_body! loop of src/io/BlockStreams.jl:101-119 and
// src/io/blocks.jl:37-71, which here becomes: stage the compressed file bytes in HBM once, LZ4-decode all
// blocks of a column in ONE launch (K7, a wave per block), and leave the decoded column contiguous in HBM.
#include "engine.hpp"
#include <atomic>
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <fcntl.h>
#include <sys/stat.h>
#include <thread>
#include <unordered_map>
#include <unistd.h>

namespace dfdb {

// ---------------------------------------------------------------- little-endian readers
struct Rd {
  const uint8_t* p; size_t n, pos;
  bool i32(int32_t& v) { if (pos + 4 > n) return false; memcpy(&v, p + pos, 4); pos += 4; return true; }
  bool i64(int64_t& v) { if (pos + 8 > n) return false; memcpy(&v, p + pos, 8); pos += 8; return true; }
  bool str(std::string& s) {  // read_string: common_io.jl:5-8 (Int32 nbytes + bytes)
    int32_t l; if (!i32(l) || l < 0 || pos + (size_t)l > n) return false;
    s.assign((const char*)p + pos, (size_t)l); pos += (size_t)l; return true;
  }
};

static std::vector<uint8_t> slurp(const std::string& fn, bool& ok, size_t max_bytes = SIZE_MAX) {
  std::vector<uint8_t> v; ok = false;
  FILE* f = fopen(fn.c_str(), "rb"); if (!f) return v;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  size_t want = (size_t)n < max_bytes ? (size_t)n : max_bytes;
  v.resize(want);
  ok = fread(v.data(), 1, want, f) == want;
  fclose(f); return v;
}

// file bytes [lo, hi) -> dst with a few concurrent preads (page cache -> pinned memory is a memcpy: one core moves ~5 GB/s)
bool read_file_range_fd(int fd, uint8_t* dst, int64_t lo, int64_t hi);
bool read_file_range(const std::string& file, uint8_t* dst, int64_t lo, int64_t hi) {
  const int fd = open(file.c_str(), O_RDONLY);
  if (fd < 0) return false;
  const bool ok = read_file_range_fd(fd, dst, lo, hi);
  close(fd);
  return ok;
}
// how many concurrent preads one range is split into (ctx option "io_threads", read when a table is loaded / a stream is opened; process-wide)
static std::atomic<int> g_io_threads{8};
void set_io_threads(int64_t n) { g_io_threads.store((int)std::min<int64_t>(64, std::max<int64_t>(1, n)), std::memory_order_relaxed); }
bool read_file_range_fd(int fd, uint8_t* dst, int64_t lo, int64_t hi) {
  const int64_t n = hi - lo;
  const int parts = (int)std::min<int64_t>(g_io_threads.load(std::memory_order_relaxed), std::max<int64_t>(1, n / (2 << 20)));
  std::vector<std::thread> th;
  std::vector<char> ok((size_t)parts, 1);
  static const bool dbg_cpus = getenv("DFDB_STREAM_DEBUG_CPUS") != nullptr;
  int dbg_buf[64]; int* dbg = dbg_cpus ? dbg_buf : nullptr;
  for (int k = 0; k < parts; k++) {
    const int64_t a = lo + n * k / parts, b = lo + n * (k + 1) / parts;
    auto work = [fd, dst, lo, a, b, k, &ok, dbg] {
      if (dbg) dbg[k] = sched_getcpu();
      int64_t got = a;
      while (got < b) { const ssize_t r = pread(fd, dst + (got - lo), (size_t)(b - got), (off_t)got); if (r <= 0) { ok[(size_t)k] = 0; return; } got += r; }
    };
    if (k + 1 < parts) th.emplace_back(work); else work();
  }
  for (auto& t : th) t.join();
  if (dbg) { std::string l; for (int k = 0; k < parts; k++) l += " " + std::to_string(dbg[k]); fprintf(stderr, "[pread] cpus:%s\n", l.c_str()); }
  for (char c : ok) if (!c) return false;
  return true;
}

void table_open(dfdb_ctx* ctx, const char* path, dfdb_table** out) {
  const std::string dir(path);
  bool ok;
  std::vector<uint8_t> m = slurp(dir + "/meta.bin", ok);   // metapath: filesystem.jl:8
  if (!ok) fail(DFDB_ERR_IO, "table %s don't exists", path);
  auto t = std::make_unique<dfdb_table>();
  t->ctx = ctx; t->path = dir;
  Rd r{m.data(), m.size(), 0};
  int64_t ncols = 0;
  if (!r.i64(t->format_version) || !r.i64(t->block_size) || !r.i64(ncols) || ncols < 0 || ncols > 1000000 || t->block_size <= 0)
    fail(DFDB_ERR_FORMAT, "bad meta.bin in %s", path);
  for (int64_t i = 0; i < ncols; i++) {
    Column c; std::string ty;
    if (!r.i64(c.id) || !r.str(c.name) || !r.str(ty)) fail(DFDB_ERR_FORMAT, "bad meta.bin in %s", path);
    c.dtype = dt_parse_ex(ty, &c.logical);
    c.file = dir + "/" + std::to_string(c.id) + ".bin";   // columnpath: filesystem.jl:11
    t->cols.push_back(std::move(c));
  }
  for (auto& c : t->cols) {   // check_column_file / check_column_head: filesystem.jl:47-61
    std::vector<uint8_t> h = slurp(c.file, ok, 4096);
    if (!ok) fail(DFDB_ERR_IO, "column file '%s' for column %s don't exists", c.file.c_str(), c.name.c_str());
    Rd hr{h.data(), h.size(), 0};
    int64_t bs; std::string ty;
    if (!hr.i64(bs) || !hr.str(ty)) fail(DFDB_ERR_FORMAT, "bad column header in %s", c.file.c_str());
    if (bs != t->block_size) fail(DFDB_ERR_FORMAT, "column %s has blocksize %lld, but table has blocksize %lld", c.name.c_str(), (long long)bs, (long long)t->block_size);
    std::string lg;
    if (dt_parse_ex(ty, &lg) != c.dtype || lg != c.logical) fail(DFDB_ERR_FORMAT, "column %s stored type is %s, but %s expected", c.name.c_str(), ty.c_str(), dt_type_string(c.dtype, c.logical).c_str());
    c.data_off = hr.pos;
  }
  *out = t.release();
}

static void set_table_rows(dfdb_table* t, int64_t nrows) {
  if (t->nrows >= 0 && t->nrows != nrows)
    fail(DFDB_ERR_ARGUMENT, "column has %lld rows but the table holds %lld resident rows", (long long)nrows, (long long)t->nrows);
  t->nrows = nrows;
}

static Column& new_column(dfdb_table* t, const char* name, int32_t dtype) {
  for (auto& c : t->cols) if (c.name == name) fail(DFDB_ERR_ARGUMENT, "ArgumentError: Duplicated column %s", name);
  if (dt_base(dtype) < 1 || dt_base(dtype) > 12) fail(DFDB_ERR_UNSUPPORTED, "unsupported dtype %d", dtype);
  Column c; c.name = name; c.dtype = dtype; c.id = (int64_t)t->cols.size() + 1;
  t->cols.push_back(std::move(c));
  return t->cols.back();
}

// bitmap words for n rows, padded so that K2's 64-word (4096-row) reads stay in bounds
static size_t padded_words(int64_t nrows) { return (size_t)(round_up(nrows > 0 ? nrows : 1, kCTileRows) / 64 + 64); }

// K4: byte offset of every 1024-row string tile = exclusive scan of the per-tile sums of max(size,0)
// (the parallel form of unsafe_remake_offsets!: FlatStringsVectors.jl:61-70)
void set_string_tile_offsets(dfdb_ctx* ctx, Column& c) {
  const int64_t ntiles = ceil_div(c.nrows, kStrTileRows);
  DevBuf tb, scratch;
  tb.ensure((size_t)(ntiles + 1) * 4);
  scratch.ensure(scan_counts_scratch_bytes(ntiles));
  c.tile_off.ensure((size_t)(ntiles + 2) * 8);
  uint32_t* dmax = tb.as<uint32_t>() + ntiles;    // (the spare word behind the per-tile sums)
  HIP_CHECK(hipMemsetAsync(dmax, 0, 4, ctx->stream));
  launch_str_tile_bytes(ctx->stream, c.data.as<int32_t>(), tb.as<uint32_t>(), c.nrows, dmax);
  launch_scan_counts(ctx->stream, tb.as<uint32_t>(), c.tile_off.as<uint64_t>(), ntiles, scratch.as<uint64_t>());
  c.max_tile_bytes = 0;
  HIP_CHECK(hipMemcpyAsync(&c.max_tile_bytes, dmax, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_CHECK(hipStreamSynchronize(ctx->stream));   // tb/scratch die here
}

void table_add_column(dfdb_table* t, const char* name, int32_t dtype, int64_t nrows, const void* data, const uint8_t* bytes,
                      int64_t nbytes, const uint8_t* missing) {
  if (nrows < 0) fail(DFDB_ERR_ARGUMENT, "negative row count");
  if (t->nrows >= 0 && t->nrows != nrows)
    fail(DFDB_ERR_ARGUMENT, "column has %lld rows but the table holds %lld resident rows", (long long)nrows, (long long)t->nrows);
  // the column is built aside and only joins the table (and sets its row count) once its upload has succeeded
  struct Rollback { dfdb_table* t; size_t n; int64_t rows; bool armed = true; ~Rollback() { if (armed) { t->cols.resize(n); t->nrows = rows; } } } rb{t, t->cols.size(), t->nrows};
  Column& c = new_column(t, name, dtype);
  set_table_rows(t, nrows);
  hipStream_t s = t->ctx->stream;
  c.nrows = nrows;
  if (dt_base(dtype) == DFDB_STRING) {
    c.data.ensure((size_t)nrows * 4 + 256);
    if (nrows) HIP_CHECK(hipMemcpyAsync(c.data.p, data, (size_t)nrows * 4, hipMemcpyHostToDevice, s));
    c.nbytes = nbytes;
    c.bytes.ensure((size_t)nbytes + 64);
    if (nbytes) HIP_CHECK(hipMemcpyAsync(c.bytes.p, bytes, (size_t)nbytes, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync((char*)c.bytes.p + nbytes, 0, 64, s));
    set_string_tile_offsets(t->ctx, c);
  } else {
    const int w = dt_width(dtype);
    c.data.ensure((size_t)nrows * w + 256);
    if (nrows) HIP_CHECK(hipMemcpyAsync(c.data.p, data, (size_t)nrows * w, hipMemcpyHostToDevice, s));
    if (dt_nullable(dtype)) {   // pack the caller's byte flags into the 1-bit/row device layout
      const size_t nw = padded_words(nrows);
      std::vector<uint64_t> bits(nw, 0);
      if (missing) for (int64_t i = 0; i < nrows; i++) if (missing[i]) bits[(size_t)i >> 6] |= 1ull << (i & 63);
      c.missing.ensure(nw * 8);
      HIP_CHECK(hipMemcpyAsync(c.missing.p, bits.data(), nw * 8, hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
    }
  }
  HIP_CHECK(hipStreamSynchronize(s));
  c.resident = true;
  rb.armed = false;
  if (dt_base(dtype) == DFDB_STRING && !dt_nullable(dtype))
    if (const int64_t dn = ctx_option(t->ctx, "string_dictionary", 0)) table_build_dictionary(t, (int32_t)(t->cols.size() - 1), dn);
}

void table_add_generated(dfdb_table* t, const char* name, int32_t gen, uint64_t seed, int64_t row_first, int64_t nrows) {
  if (nrows < 0) fail(DFDB_ERR_ARGUMENT, "negative row count");
  if (t->nrows >= 0 && t->nrows != nrows)
    fail(DFDB_ERR_ARGUMENT, "column has %lld rows but the table holds %lld resident rows", (long long)nrows, (long long)t->nrows);
  if (gen < DFDB_GEN_I64_MOD1M || gen > DFDB_GEN_STR_BRANDS10_MISSING) fail(DFDB_ERR_ARGUMENT, "unknown generator %d", gen);
  for (auto& c : t->cols) if (c.name == name) fail(DFDB_ERR_ARGUMENT, "ArgumentError: Duplicated column %s", name);
  struct Rollback { dfdb_table* t; size_t n; int64_t rows; bool armed = true; ~Rollback() { if (armed) { t->cols.resize(n); t->nrows = rows; } } } rb{t, t->cols.size(), t->nrows};
  set_table_rows(t, nrows);
  hipStream_t s = t->ctx->stream;
  switch (gen) {
    case DFDB_GEN_I64_MOD1M: case DFDB_GEN_I64_IOTA: {
      Column& c = new_column(t, name, DFDB_I64); c.nrows = nrows;
      c.data.ensure((size_t)nrows * 8 + 256);
      if (gen == DFDB_GEN_I64_MOD1M) launch_gen_i64_mod1m(s, c.data.as<int64_t>(), seed, row_first, nrows);
      else launch_gen_i64_iota(s, c.data.as<int64_t>(), row_first, nrows);
      c.resident = true; break;
    }
    case DFDB_GEN_F64_U2000: {
      Column& c = new_column(t, name, DFDB_F64); c.nrows = nrows;
      c.data.ensure((size_t)nrows * 8 + 256);
      launch_gen_f64_u2000(s, c.data.as<double>(), seed, row_first, nrows);
      c.resident = true; break;
    }
    case DFDB_GEN_STR_BRANDS10: case DFDB_GEN_STR_BRANDS10_MISSING: {
      const bool wm = gen == DFDB_GEN_STR_BRANDS10_MISSING;
      Column& c = new_column(t, name, wm ? (DFDB_STRING | DFDB_NULLABLE) : DFDB_STRING); c.nrows = nrows;
      c.data.ensure((size_t)nrows * 4 + 256);
      launch_gen_brand_sizes(s, c.data.as<int32_t>(), seed, row_first, nrows, wm);
      set_string_tile_offsets(t->ctx, c);
      const int64_t ntiles = ceil_div(nrows, kStrTileRows);
      uint64_t total = 0;
      HIP_CHECK(hipMemcpy(&total, c.tile_off.as<uint64_t>() + ntiles, 8, hipMemcpyDeviceToHost));
      c.nbytes = (int64_t)total;
      c.bytes.ensure((size_t)total + 64);
      HIP_CHECK(hipMemsetAsync((char*)c.bytes.p + total, 0, 64, s));
      launch_gen_brand_bytes(s, c.data.as<int32_t>(), (const int64_t*)c.tile_off.p, c.bytes.as<uint8_t>(), seed, row_first, nrows);
      c.resident = true;
      if (!wm) if (const int64_t dn = ctx_option(t->ctx, "string_dictionary", 0)) table_build_dictionary(t, (int32_t)(t->cols.size() - 1), dn);
      break;
    }
    default: fail(DFDB_ERR_ARGUMENT, "unknown generator %d", gen);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  rb.armed = false;
}

// ---------------------------------------------------------------- block loading (D1-D6)
struct BlockHdr { int32_t rows; int64_t origin, compressed; size_t body_off; };

// read_sizes over the whole image (BlockStreams.jl:68-72): host walks the 20-byte headers only
static std::vector<BlockHdr> walk_blocks(const uint8_t* img, size_t n, size_t pos) {
  std::vector<BlockHdr> v;
  while (pos < n) {
    if (pos + 20 > n) fail(DFDB_ERR_FORMAT, "truncated block header");
    BlockHdr h; memcpy(&h.rows, img + pos, 4); memcpy(&h.origin, img + pos + 4, 8); memcpy(&h.compressed, img + pos + 12, 8);
    if (h.rows < 0 || h.origin < 0 || h.compressed < 0 || (uint64_t)h.compressed > n - pos - 20) fail(DFDB_ERR_FORMAT, "corrupt block header");
    h.body_off = pos + 20;
    pos += 20 + (size_t)h.compressed;
    v.push_back(h);
  }
  return v;
}

// K8 + block bodies on the device (k_decode.hip)
void launch_unpack_nullable(hipStream_t s, const uint8_t* bodies, const int64_t* body_off, const int64_t* row_off, const int64_t* rows_of, int32_t nblocks, int width,
                            uint8_t* values, uint64_t* missing_bits);
void launch_unpack_strings(hipStream_t s, const uint8_t* bodies, const int64_t* body_off, const int64_t* row_off, const int64_t* rows_of, const int64_t* byte_off,
                           int32_t nblocks, int32_t* sizes, uint8_t* bytes);

// the blocks hs[0..nb) of column c — block ordinals block_first.. — whose compressed bodies already sit in t->ld_staged (the body of
// hs[i] at staged + hs[i].body_off - comp_lo): K7 + K8 / string unpack into the column
//
// row_pos != nullptr (stream.cpp's late materialization): hs[] is a SUBSET of the blocks of a range that holds total_rows rows, block i starting at
// row row_pos[i] of the column.  The rows of the blocks that were left out keep whatever the buffers held (no selected row points at them);
// a String column gives them size 0, so that the tile byte offsets of the rows that ARE there come out of the same prefix scan.
// predecoded > 0 (load_from_file's progressive decode of a plain fixed-width column): the first `predecoded` blocks were decoded into c.data while the rest of
// the file was still on its way, their statuses sit in t->ld_status: only the others are decoded here, all are validated
static void decode_staged(dfdb_table* t, Column& c, const BlockHdr* hs, int64_t nb, int64_t block_first, int64_t comp_lo, dfdb_sizestats* stats,
                          const int64_t* row_pos = nullptr, int64_t total_rows = -1, int64_t predecoded = 0) {
  dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  int64_t nrows = 0;
  for (int64_t i = 0; i < nb; i++) nrows += hs[i].rows;
  if (row_pos) nrows = total_rows;
  if (t->nrows >= 0 && t->nrows != nrows)
    fail(DFDB_ERR_ARGUMENT, "column %s would load %lld rows but the table holds %lld resident rows", c.name.c_str(), (long long)nrows, (long long)t->nrows);
  if (t->nrows >= 0 && t->block_first != block_first) fail(DFDB_ERR_ARGUMENT, "all columns of a table must load the same block range");
  const int w = dt_width(c.dtype);
  const bool is_str = dt_base(c.dtype) == DFDB_STRING, is_null = dt_nullable(c.dtype) && !is_str;
  DevBuf &staged = t->ld_staged, &bodies = t->ld_bodies, &dblocks = t->ld_blocks, &dstatus = t->ld_status, &d_aux = t->ld_aux;
  std::vector<Lz4Block> blocks((size_t)nb);
  std::vector<int64_t> body_off((size_t)nb + 1), row_off((size_t)nb + 1), byte_off((size_t)nb + 1), rows_of((size_t)nb + 1, 0);
  int64_t bo = 0, ro = 0, so = 0;
  for (int64_t i = 0; i < nb; i++) {
    const BlockHdr& h = hs[i];
    if (h.origin > 0x7fffffffLL || h.compressed > 0x7fffffffLL) fail(DFDB_ERR_FORMAT, "block larger than the LZ4 block limit");
    int64_t expect_min = is_str ? 4 + 4ll * h.rows : (is_null ? 8 * ceil_div(h.rows, 64) + (int64_t)w * h.rows : (int64_t)w * h.rows);
    if (is_str ? h.origin < expect_min : h.origin != expect_min)
      fail(DFDB_ERR_FORMAT, "block %lld of column %s has body of %lld bytes, expected %lld", (long long)(block_first + i), c.name.c_str(), (long long)h.origin, (long long)expect_min);
    blocks[i].src_off = (int64_t)h.body_off - comp_lo; blocks[i].src_len = (int32_t)h.compressed; blocks[i].dst_len = (int32_t)h.origin;
    blocks[i].dst_off = bo;
    if (row_pos) ro = row_pos[i];
    body_off[i] = bo; row_off[i] = ro; byte_off[i] = so; rows_of[i] = h.rows;
    bo += round_up(h.origin, 16); ro += h.rows; if (is_str) so += h.origin - 4 - 4ll * h.rows;
  }
  body_off[nb] = bo; row_off[nb] = row_pos ? nrows : ro; byte_off[nb] = so;

  c.nrows = nrows;
  uint8_t* decode_dst;
  // ctx option keep_compressed = 2: a plain fixed-width column stays COMPRESSED-ONLY — its blocks are validated (and their sequence starts recorded) by a
  // decode whose output goes to the per-wave history rings and nowhere else; there is no decoded array, now or later (k_decode.hip HIST)
  const bool comp_only = !is_str && !is_null && nb && !row_pos && !t->keep_load_scratch && ctx_option(ctx, "keep_compressed", 0) == 2;
  if (!is_str && !is_null) {   // plain fixed width: decode straight into the column (no read!(io, v) copy: blocks.jl:43)
    if (comp_only) { c.data.release(); decode_dst = nullptr; }
    else { c.data.ensure((size_t)nrows * w + 256); decode_dst = c.data.as<uint8_t>(); }
    for (int64_t i = 0; i < nb; i++) blocks[i].dst_off = (int64_t)w * row_off[i];
  } else {
    bodies.ensure((size_t)bo + 64);
    decode_dst = bodies.as<uint8_t>();
  }
  if (nb) {
    dblocks.ensure(sizeof(Lz4Block) * (size_t)nb);
    dstatus.ensure(4 * (size_t)nb);
    HIP_CHECK(hipMemcpyAsync(dblocks.p, blocks.data(), sizeof(Lz4Block) * (size_t)nb, hipMemcpyHostToDevice, s));
    if (predecoded > nb || comp_only || is_str || is_null) predecoded = 0;
    HIP_CHECK(hipMemsetAsync(dstatus.as<int32_t>() + predecoded, 0, 4 * (size_t)(nb - predecoded), s));
    if (predecoded == nb) {}                                          // everything was decoded on the way
    else if (comp_only) {
      // (the index this launch records is the column's: comp / comp_index are set up here, ahead of the move below)
      if (c.comp.p || c.comp_index.p) { HIP_CHECK(hipStreamSynchronize(s)); c.comp_index.release(); }
      c.comp_index_state = 0;
      uint32_t* idx = nullptr;
      if (ctx_option(ctx, "lz4_index", 1) != 0) {
        try { c.comp_index.ensure((staged.bytes + 7) / 8 + 1024); idx = c.comp_index.as<uint32_t>(); HIP_CHECK(hipMemsetAsync(c.comp_index.p, 0, c.comp_index.bytes, s)); }
        catch (const Error&) { (void)hipGetLastError(); idx = nullptr; }
      }
      int waves = 0; uint8_t* scratch = ctx_hist_scratch(ctx, &waves);
      LaunchTimer lt(ctx, "lz4_decode_hist");
      prof_note(ctx, idx ? "lz4_decode_hist.recording" : "lz4_decode_hist.plain");
      launch_lz4_decode_hist(s, staged.as<uint8_t>(), scratch, waves, dblocks.as<Lz4Block>(), (int32_t)nb, dstatus.as<int32_t>(), nullptr, idx, idx ? 1 : 0);
    } else
    { LaunchTimer lt(ctx, "lz4_decode"); launch_lz4_decode(s, staged.as<uint8_t>(), decode_dst, dblocks.as<Lz4Block>() + predecoded, (int32_t)(nb - predecoded), dstatus.as<int32_t>() + predecoded, (int)ctx_option(ctx, "lz4_pipeline", -1)); }
    std::vector<int32_t> st((size_t)nb);
    HIP_CHECK(hipMemcpyAsync(st.data(), dstatus.p, 4 * (size_t)nb, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < nb; i++) if (st[i] != 0) fail(DFDB_ERR_FORMAT, "decompression error in block %lld of column %s", (long long)(block_first + i), c.name.c_str());
  }
  if (is_str || is_null) {
    d_aux.ensure(8 * 4 * ((size_t)nb + 1));
    int64_t* d_body = d_aux.as<int64_t>(); int64_t* d_row = d_body + nb + 1; int64_t* d_byte = d_row + nb + 1; int64_t* d_rows = d_byte + nb + 1;
    HIP_CHECK(hipMemcpyAsync(d_rows, rows_of.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_body, body_off.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_row, row_off.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_byte, byte_off.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, s));
    if (is_str) {
      c.data.ensure((size_t)nrows * 4 + 256);
      c.nbytes = so; c.bytes.ensure((size_t)so + 64);
      HIP_CHECK(hipMemsetAsync((char*)c.bytes.p + so, 0, 64, s));
      if (row_pos && nrows) HIP_CHECK(hipMemsetAsync(c.data.p, 0, (size_t)nrows * 4, s));      // the blocks left out: empty strings
      if (nb) launch_unpack_strings(s, bodies.as<uint8_t>(), d_body, d_row, d_rows, d_byte, (int32_t)nb, c.data.as<int32_t>(), c.bytes.as<uint8_t>());
      set_string_tile_offsets(ctx, c);
      // datasize must equal the sum of the positive sizes (unsafe_remake_offsets!: FlatStringsVectors.jl:69)
      uint64_t total = 0;
      HIP_CHECK(hipMemcpy(&total, c.tile_off.as<uint64_t>() + ceil_div(nrows, kStrTileRows), 8, hipMemcpyDeviceToHost));
      if ((int64_t)total != so) fail(DFDB_ERR_FORMAT, "string column %s: sizes sum to %llu bytes but blocks hold %lld", c.name.c_str(), (unsigned long long)total, (long long)so);
    } else {
      c.data.ensure((size_t)nrows * w + 256);
      const size_t nw = padded_words(nrows);
      c.missing.ensure(nw * 8);
      HIP_CHECK(hipMemsetAsync(c.missing.p, 0, nw * 8, s));
      if (nb) launch_unpack_nullable(s, bodies.as<uint8_t>(), d_body, d_row, d_rows, (int32_t)nb, w, c.data.as<uint8_t>(), c.missing.as<uint64_t>());
    }
    HIP_CHECK(hipStreamSynchronize(s));
  }
  if (t->nrows < 0) { t->nrows = nrows; t->block_first = block_first; t->row_base = block_first * t->block_size; }
  c.resident = true;
  c.dict_n = 0; c.dict_host.clear();
  if (is_str && !is_null && !t->keep_load_scratch)                 // (keep_load_scratch: a stream slot, whose columns live for one chunk)
    if (const int64_t dn = ctx_option(ctx, "string_dictionary", 0)) table_build_dictionary(t, (int32_t)(&c - t->cols.data()), dn);
  // what an earlier load of this column kept is stale now, whatever this load keeps (a reload with keep_compressed = 0 left the old
  // descriptors behind and dfdb_table_decode_resident / decode_on_scan would have decoded them into the new array)
  DevBuf fresh_index;
  if (comp_only) fresh_index = std::move(c.comp_index);                        // (recorded by this very load: it stays)
  if (c.comp.p || c.comp_blocks.p || c.comp_status.p || c.comp_index.p) {      // (a stream slot, which never keeps anything, pays no drain here)
    HIP_CHECK(hipStreamSynchronize(s));
    c.comp.release(); c.comp_blocks.release(); c.comp_status.release(); c.comp_index.release();
  }
  c.comp_nblocks = 0; c.comp_index_state = 0; c.comp_only = false; c.transient = false; c.comp_blocks_host.clear();
  if (!is_str && !is_null && nb && ctx_option(ctx, "keep_compressed", 0) != 0) {   // the compressed blocks stay: dfdb_table_decode_resident
    c.comp = std::move(staged); c.comp_blocks = std::move(dblocks); c.comp_status = std::move(dstatus); c.comp_nblocks = nb;
    c.comp_blocks_host = blocks;
    if (comp_only) { c.comp_only = true; c.comp_index = std::move(fresh_index); c.comp_index_state = c.comp_index.p ? 1 : 0; }
  }
  if (!t->keep_load_scratch) { HIP_CHECK(hipStreamSynchronize(s)); staged.release(); bodies.release(); dblocks.release(); dstatus.release(); d_aux.release(); }
  if (stats) {   // SizeStats incl. the 24-byte header quirk (BlockStreams.jl:7,23)
    stats->rows = nrows;
    for (int64_t i = 0; i < nb; i++) { stats->compressed += hs[i].compressed + 24; stats->uncompressed += hs[i].origin; }
  }
}

static void load_from_image(dfdb_table* t, Column& c, const uint8_t* img, size_t nbytes, size_t data_off, int64_t block_first,
                            int64_t block_last, dfdb_sizestats* stats) {
  hipStream_t s = t->ctx->stream;
  std::vector<BlockHdr> all = walk_blocks(img, nbytes, data_off);
  const int64_t nb_total = (int64_t)all.size();
  if (block_last < 0 || block_last > nb_total) block_last = nb_total;
  if (block_first < 0) block_first = 0;
  if (block_first > block_last) block_first = block_last;
  const int64_t nb = block_last - block_first;
  for (int64_t b = 0; b + 1 < nb_total; b++)
    if (all[b].rows != t->block_size) fail(DFDB_ERR_FORMAT, "block %lld of column %s holds %d rows, expected block_size %lld", (long long)b, c.name.c_str(), all[b].rows, (long long)t->block_size);
  int64_t comp_lo = 0, comp_hi = 0;
  if (nb) { comp_lo = (int64_t)all[block_first].body_off; comp_hi = (int64_t)(all[block_last - 1].body_off + all[block_last - 1].compressed); }
  // stage the compressed byte range in HBM
  DevBuf& staged = t->ld_staged;
  staged.ensure((size_t)(comp_hi - comp_lo) + 64);
  if (comp_hi > comp_lo && !t->ld_prestaged) HIP_CHECK(hipMemcpyAsync(staged.p, img + comp_lo, (size_t)(comp_hi - comp_lo), hipMemcpyHostToDevice, s));
  decode_staged(t, c, all.data() + block_first, nb, block_first, comp_lo, stats);
}

// dfdb_table_load: the column file -> HBM without ever holding it in host memory.  Blocks before block_first are skipped header by
// header (skip_block, BlockStreams.jl:74-78); from there the file is read in 64-MB pieces with concurrent preads into one of two pinned
// bounce buffers while the previous piece is in flight to the device, the 20-byte headers are walked as the bytes arrive, and the
// blocks are decoded by ONE K7 launch once the last piece has been queued.
static void load_from_file(dfdb_table* t, Column& c, int64_t block_first, int64_t block_last, dfdb_sizestats* stats) {
  dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  NodeBind bind(ctx);                                                          // the preads and the bounce buffers on the GPU's NUMA node
  set_io_threads(ctx_option(ctx, "io_threads", 8));                            // (process-wide: the last table loaded / stream opened decides)
  const int fd = open(c.file.c_str(), O_RDONLY);
  if (fd < 0) fail(DFDB_ERR_IO, "cannot read %s", c.file.c_str());
  struct FdClose { int fd; ~FdClose() { close(fd); } } fdg{fd};
  const int64_t fsz = (int64_t)lseek(fd, 0, SEEK_END);
  if (block_first < 0) block_first = 0;
  auto parse = [&](const uint8_t* h, int64_t pos, BlockHdr& o) {
    memcpy(&o.rows, h, 4); memcpy(&o.origin, h + 4, 8); memcpy(&o.compressed, h + 12, 8);
    if (o.rows < 0 || o.origin < 0 || o.compressed < 0 || o.compressed > fsz - pos - 20) fail(DFDB_ERR_FORMAT, "corrupt block header");
    o.body_off = (size_t)pos + 20;
  };
  auto check_rows = [&](const BlockHdr& h, int64_t b, int64_t next_pos) {      // every block but the last holds block_size rows
    if (next_pos < fsz && h.rows != t->block_size)
      fail(DFDB_ERR_FORMAT, "block %lld of column %s holds %d rows, expected block_size %lld", (long long)b, c.name.c_str(), h.rows, (long long)t->block_size);
  };
  int64_t pos = (int64_t)c.data_off, b = 0;
  uint8_t hb[20];
  while (b < block_first && pos < fsz) {
    if (pos + 20 > fsz || pread(fd, hb, 20, (off_t)pos) != 20) fail(DFDB_ERR_FORMAT, "truncated block header");
    BlockHdr h; parse(hb, pos, h);
    pos += 20 + h.compressed; check_rows(h, b, pos); b++;
  }
  if (b < block_first) block_first = b;                                        // fewer blocks than asked for: an empty range at the end
  const int64_t lo = pos;
  std::vector<BlockHdr> hs;
  int64_t hi = fsz;
  if (block_last >= 0) {                                                       // a block range (sharded load): find where it ends
    while (b < block_last && pos < fsz) {
      if (pos + 20 > fsz || pread(fd, hb, 20, (off_t)pos) != 20) fail(DFDB_ERR_FORMAT, "truncated block header");
      BlockHdr h; parse(hb, pos, h);
      pos += 20 + h.compressed; check_rows(h, b, pos); b++;
      hs.push_back(h);
    }
    hi = pos;
  }
  const bool walk = block_last < 0;
  const int64_t kPiece = std::max<int64_t>(4096, ctx_option(ctx, "load_piece_kb", 64 << 10) << 10);      // (64 MB; smaller only in tests)
  ensure_pin_ring(ctx, (size_t)kPiece);
  DevBuf& staged = t->ld_staged;
  staged.ensure((size_t)(hi - lo) + 64);
  // PROGRESSIVE DECODE (round 5): a plain fixed-width column is decoded batch by batch while the rest of its file is still being read and copied — K7 over
  // the blocks that have arrived, on the same stream behind their copies — instead of in one launch after the last byte (17 ms of a 100-ms load of 1e9 rows
  // during which PCIe idled).  The column array must exist before the first batch, i.e. the number of rows must be known: a block range knows its headers
  // already; a load to the end of the file walks the 20-byte headers on a side thread (page-cache preads, ~10 ms per 15 000 blocks) while the first pieces
  // are read, and decoding starts once that walk is done.  ctx option "load_progressive" = 0: one launch at the end.
  const int w_ = dt_width(c.dtype);
  const bool plain = dt_base(c.dtype) != DFDB_STRING && !dt_nullable(c.dtype);
  const bool progressive = plain && walk && ctx_option(ctx, "load_progressive", 1) != 0 && ctx_option(ctx, "keep_compressed", 0) != 2 && hi - lo >= 4 * kPiece;
  const int64_t prog_batch = std::max<int64_t>(1, ctx_option(ctx, "load_progressive_blocks", 768));
  struct PreWalk { std::thread th; std::atomic<int> state{0}; int64_t rows = 0, blocks = 0; } pre;      // state 1: done, -1: failed (the main walk will say why)
  if (progressive) {
    pre.th = std::thread([&, lo, fsz] {
      int64_t p = lo, r = 0, n = 0; uint8_t h20[20];
      while (p < fsz) {
        if (p + 20 > fsz || pread(fd, h20, 20, (off_t)p) != 20) { pre.state = -1; return; }
        int32_t rows; int64_t comp; memcpy(&rows, h20, 4); memcpy(&comp, h20 + 12, 8);
        if (rows < 0 || comp < 0 || comp > fsz - p - 20) { pre.state = -1; return; }
        r += rows; n++; p += 20 + comp;
      }
      pre.rows = r; pre.blocks = n; pre.state = 1;
    });
  }
  struct JoinPre { PreWalk& p; ~JoinPre() { if (p.th.joinable()) p.th.join(); } } join_pre{pre};
  int64_t predecoded = 0;                                                      // blocks already handed to K7
  bool prog_ready = false;
  // the batches decode on a stream of their own, each behind an event that marks its last copy on the engine stream: on ONE stream a batch's K7 (a few
  // milliseconds) would hold up the copies queued behind it and PCIe would idle exactly as before (measured: 127 ms against 104)
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> prog_ev;
  // (declared BEFORE the guard, so destroyed after it: these are the pageable sources of the side stream's asynchronous copies, and an exception that unwinds
  // this frame must find them alive until SideGuard has synchronised that stream — ADVICE r5)
  std::vector<std::vector<Lz4Block>> prog_desc;
  struct SideGuard { hipStream_t& st; std::vector<hipEvent_t>& ev; ~SideGuard() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } for (hipEvent_t e : ev) (void)hipEventDestroy(e); } } side_guard{side, prog_ev};
  auto progress = [&](bool last) {                                             // decode what has arrived since the last batch (blocks hs[predecoded ..) whose bodies end <= staged bytes)
    if (!progressive || pre.state.load() != 1) return;
    if (!prog_ready) {
      if (t->nrows >= 0 && t->nrows != pre.rows) return;                       // (decode_staged will raise the mismatch)
      c.data.ensure((size_t)pre.rows * (size_t)w_ + 256);
      t->ld_blocks.ensure(sizeof(Lz4Block) * (size_t)std::max<int64_t>(pre.blocks, 1));
      t->ld_status.ensure(4 * (size_t)std::max<int64_t>(pre.blocks, 1));
      HIP_CHECK(hipMemsetAsync(t->ld_status.p, 0, 4 * (size_t)std::max<int64_t>(pre.blocks, 1), s));
      HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
      prog_ready = true;
    }
    const int64_t have = (int64_t)hs.size();
    if (have > pre.blocks) { if (last && side) { (void)hipStreamSynchronize(side); } return; }   // the two walks disagree: leave everything to decode_staged
    if (!last && have - predecoded < prog_batch) return;
    auto join_side = [&] {                                                     // the engine stream goes on behind the batches
      hipEvent_t evd; HIP_CHECK(hipEventCreateWithFlags(&evd, hipEventDisableTiming)); prog_ev.push_back(evd);
      HIP_CHECK(hipEventRecord(evd, side));
      HIP_CHECK(hipStreamWaitEvent(s, evd, 0));
    };
    if (have == predecoded) { if (last) join_side(); return; }
    std::vector<Lz4Block> d((size_t)(have - predecoded));
    int64_t row = 0;
    for (int64_t i = 0; i < predecoded; i++) row += hs[(size_t)i].rows;
    for (int64_t i = predecoded; i < have; i++) {
      const BlockHdr& h = hs[(size_t)i];
      if (h.origin != (int64_t)w_ * h.rows || h.origin > 0x7fffffffLL || h.compressed > 0x7fffffffLL || row + h.rows > pre.rows) return;   // decode_staged reports it
      Lz4Block& x = d[(size_t)(i - predecoded)];
      x.src_off = (int64_t)h.body_off - lo; x.src_len = (int32_t)h.compressed; x.dst_len = (int32_t)h.origin; x.dst_off = (int64_t)w_ * row;
      row += h.rows;
    }
    prog_desc.push_back(std::move(d));
    const std::vector<Lz4Block>& dd = prog_desc.back();
    hipEvent_t ev; HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); prog_ev.push_back(ev);
    HIP_CHECK(hipEventRecord(ev, s));                                          // everything this batch reads has been queued on the engine stream by now
    HIP_CHECK(hipStreamWaitEvent(side, ev, 0));
    HIP_CHECK(hipMemcpyAsync(t->ld_blocks.as<Lz4Block>() + predecoded, dd.data(), sizeof(Lz4Block) * dd.size(), hipMemcpyHostToDevice, side));
    { LaunchTimer lt(ctx, "lz4_decode", side); prof_note(ctx, "lz4_decode.progressive");
      launch_lz4_decode(side, staged.as<uint8_t>(), c.data.as<uint8_t>(), t->ld_blocks.as<Lz4Block>() + predecoded, (int32_t)dd.size(), t->ld_status.as<int32_t>() + predecoded,
                        (int)ctx_option(ctx, "lz4_pipeline", -1)); }
    predecoded = have;
    if (last) join_side();
  };
  uint8_t tail[20]; int64_t tail_end = -1;                                     // the last 20 bytes of the previous piece (a header may straddle)
  pos = lo;
  bool used[2] = {false, false};
  int k = 0;
  for (int64_t a = lo; a < hi; a += kPiece, k ^= 1) {
    const int64_t e = std::min(hi, a + kPiece);
    uint8_t* buf = ctx->pin_ring[k];
    const auto tw0 = std::chrono::steady_clock::now();
    if (used[k]) HIP_CHECK(hipEventSynchronize(ctx->pin_ev[k]));              // its previous copy has left the buffer
    const auto tw1 = std::chrono::steady_clock::now();
    if (!read_file_range_fd(fd, buf, a, e)) fail(DFDB_ERR_IO, "short read from %s", c.file.c_str());
    if (getenv("DFDB_STREAM_DEBUG")) { const auto tw2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[load] piece at %lld: waited %.2f ms for its buffer, read %.1f MB in %.2f ms\n", (long long)(a - lo), std::chrono::duration<double, std::milli>(tw1 - tw0).count(),
              (double)(e - a) / 1e6, std::chrono::duration<double, std::milli>(tw2 - tw1).count()); }
    HIP_CHECK(hipMemcpyAsync(staged.as<uint8_t>() + (a - lo), buf, (size_t)(e - a), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipEventRecord(ctx->pin_ev[k], s)); used[k] = true;
    if (walk) {
      while (pos + 20 <= e) {
        const uint8_t* hp;
        if (pos >= a) hp = buf + (pos - a);
        else { const int64_t n0 = a - pos; memcpy(hb, tail + (20 - (tail_end - pos)), (size_t)n0); memcpy(hb + n0, buf, (size_t)(20 - n0)); hp = hb; }
        BlockHdr h; parse(hp, pos, h);
        pos += 20 + h.compressed; check_rows(h, b, pos); b++;
        hs.push_back(h);
      }
      if (e - a >= 20) { memcpy(tail, buf + (e - a - 20), 20); tail_end = e; }
      else if (e < hi) fail(DFDB_ERR_FORMAT, "truncated block header");
      // (the last header parsed may describe a block whose body is not staged yet: it waits for the next batch)
      if (!hs.empty() && (int64_t)hs.back().body_off + hs.back().compressed > e) { const BlockHdr lastb = hs.back(); hs.pop_back(); progress(false); hs.push_back(lastb); }
      else progress(false);
    }
  }
  if (walk && pos != hi) fail(DFDB_ERR_FORMAT, "truncated block header");
  if (pre.th.joinable()) pre.th.join();
  progress(true);                                                              // (a block range — !walk — never decodes on the way: its headers are known up front, its bodies are not)
  decode_staged(t, c, hs.data(), (int64_t)hs.size(), block_first, lo, stats, nullptr, -1, prog_ready && pre.blocks == (int64_t)hs.size() ? predecoded : 0);
}

// K7 again over the compressed blocks a column kept at load time (option "keep_compressed"): every block of the column decoded into
// its resident array, asynchronously.  What a query over a compressed-resident column pays before its scan.
void table_decode_resident(dfdb_table* t, int32_t ordinal) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  if (!c.comp_nblocks) fail(DFDB_ERR_ARGUMENT, "column %s holds no compressed blocks (load it with option keep_compressed = 1)", c.name.c_str());
  if (c.comp_only) fail(DFDB_ERR_ARGUMENT, "column %s is compressed-only (keep_compressed = 2): it has no decoded array to decode into", c.name.c_str());
  dfdb_ctx* ctx = t->ctx;
  for (dfdb_query* q : t->queries) { q->executed_stages = -1; q->count = -1; q->prefix_valid = false; }
  const int pipe = (int)ctx_option(ctx, "lz4_pipeline", -1);
  const int mode = column_lz4_index(ctx, c, lz4_decode_takes_index((int32_t)c.comp_nblocks, pipe));
  LaunchTimer lt(ctx, "lz4_decode");
  prof_note(ctx, mode == 2 ? "lz4_decode.indexed" : mode == 1 ? "lz4_decode.recording" : "lz4_decode.plain");
  launch_lz4_decode(ctx->stream, c.comp.as<uint8_t>(), c.data.as<uint8_t>(), c.comp_blocks.as<Lz4Block>(), (int32_t)c.comp_nblocks, c.comp_status.as<int32_t>(), pipe,
                    c.comp_index.as<uint32_t>(), mode);
}

int64_t table_decode_status(dfdb_table* t, int32_t ordinal) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  if (!c.comp_nblocks) fail(DFDB_ERR_ARGUMENT, "column %s holds no compressed blocks (load it with option keep_compressed = 1)", c.name.c_str());
  std::vector<int32_t> st((size_t)c.comp_nblocks);
  HIP_CHECK(hipMemcpyAsync(st.data(), c.comp_status.p, st.size() * 4, hipMemcpyDeviceToHost, t->ctx->stream));
  HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
  int64_t bad = 0;
  for (int32_t v : st) bad += v != 0;
  // a decode that went wrong may have been reading (or writing) the sequence-start index: drop it, the next decode parses for itself and records a new one
  if (bad > 0) { c.comp_index.release(); c.comp_index_state = 0; }
  return bad;
}

// ---- compressed-only columns (keep_compressed = 2)
uint8_t* ctx_hist_scratch(dfdb_ctx* ctx, int* waves) {
  int w = (int)ctx_option(ctx, "lz4_hist_waves", 0);
  if (w <= 0) w = lz4_hist_default_waves(ctx->prop.multiProcessorCount);
  if (w > 65536) w = 65536;
  if (!ctx->hist.p || ctx->hist_waves != w) {
    if (ctx->hist.p) HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->hist.release();
    ctx->hist.ensure(lz4_hist_scratch_bytes(w));
    ctx->hist_waves = w;
  }
  if (waves) *waves = w;
  return ctx->hist.as<uint8_t>();
}
static void transient_decode_checked(dfdb_table* t, Column& c);
// the decoded array of a fixed-width column; a compressed-only one is decoded whole, now, for the ABI call in progress (the statuses of that decode are
// read right behind the launch: transient_decode_checked)
const void* column_data(dfdb_table* t, Column& c) {
  if (!c.comp_only || c.data.p) return c.data.p;
  c.data.ensure((size_t)c.nrows * (size_t)dt_width(c.dtype) + 256);
  c.transient = true;
  transient_decode_checked(t, c);
  return c.data.p;
}
void table_drop_transient(dfdb_table* t) {
  bool any = false;
  for (Column& c : t->cols) any = any || c.transient;
  if (!any) return;
  (void)hipStreamSynchronize(t->ctx->stream);                 // whatever the call launched over the transient arrays has to be done with them
  for (Column& c : t->cols) if (c.transient) { c.data.release(); c.transient = false; }
}
// A whole-column transient decode is only as good as its statuses (the reference's `@assert size == sizes.origin "decompression error"`, BlockStreams.jl:112):
// they are read right behind the launch — a consumer that fed on garbage must never hand it to the caller.  A bad status with the sequence-start
// index in use drops the index and decodes once more by parsing; bad again = the resident blocks are damaged.
static void transient_decode_checked(dfdb_table* t, Column& c) {
  dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int pipe = (int)ctx_option(ctx, "lz4_pipeline", -1);
  for (int attempt = 0; attempt < 2; attempt++) {
    const int mode = attempt == 0 ? column_lz4_index(ctx, c, lz4_decode_takes_index((int32_t)c.comp_nblocks, pipe)) : 0;
    { LaunchTimer lt(ctx, "lz4_decode");
      prof_note(ctx, "lz4_decode.transient");
      launch_lz4_decode(s, c.comp.as<uint8_t>(), c.data.as<uint8_t>(), c.comp_blocks.as<Lz4Block>(), (int32_t)c.comp_nblocks, c.comp_status.as<int32_t>(), pipe,
                        mode ? c.comp_index.as<uint32_t>() : nullptr, mode); }
    std::vector<int32_t> st((size_t)c.comp_nblocks);
    HIP_CHECK(hipMemcpyAsync(st.data(), c.comp_status.p, st.size() * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    int64_t bad = 0; for (int32_t v : st) bad += v != 0;
    if (!bad) return;
    if (mode != 0) { c.comp_index.release(); c.comp_index_state = 0; HIP_CHECK(hipMemsetAsync(c.comp_status.p, 0, st.size() * 4, s)); continue; }   // the index may be what is damaged
    c.data.release(); c.transient = false;
    fail(DFDB_ERR_FORMAT, "column %s: %lld of its resident LZ4 blocks do not decode", c.name.c_str(), (long long)bad);
  }
}
void table_resident_bytes(dfdb_table* t, int32_t ordinal, int64_t* decoded, int64_t* compressed) {
  int64_t d = 0, k = 0;
  auto one = [&](const Column& c) {
    if (!c.transient) d += (int64_t)c.data.bytes;
    d += (int64_t)c.bytes.bytes + (int64_t)c.tile_off.bytes + (int64_t)c.missing.bytes + (int64_t)c.dict_codes.bytes + (int64_t)c.dict_bytes.bytes;
    k += (int64_t)c.comp.bytes + (int64_t)c.comp_blocks.bytes + (int64_t)c.comp_status.bytes + (int64_t)c.comp_index.bytes;
  };
  if (ordinal < 0) for (const Column& c : t->cols) one(c);
  else { if ((size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal); one(t->cols[(size_t)ordinal]); }
  if (decoded) *decoded = d;
  if (compressed) *compressed = k;
}

// The sequence-start index of a column's resident LZ4 blocks (k_decode.hip INDEX): the first decode that can take it records it, the later ones decode with it.
// Returns launch_lz4_decode's index_mode (0: none — ctx option "lz4_index" = 0, or a launch form that takes none).
int column_lz4_index(dfdb_ctx* ctx, Column& c, bool form_takes_index) {
  if (!form_takes_index || ctx_option(ctx, "lz4_index", 1) == 0 || !c.comp.p) return 0;
  if (!c.comp_index.p) {
    const size_t bytes = (c.comp.bytes + 7) / 8 + 1024;                  // + the 64-dword register window's reach past the last bit
    try { c.comp_index.ensure(bytes); } catch (const Error&) { (void)hipGetLastError(); return 0; }   // no room for it: decode without
    HIP_CHECK(hipMemsetAsync(c.comp_index.p, 0, c.comp_index.bytes, ctx->stream));
    c.comp_index_state = 0;
  }
  if (c.comp_index_state == 0) { c.comp_index_state = 1; return 1; }     // this launch records (stream order makes it complete before the next one reads)
  return 2;
}

void table_load_image(dfdb_table* t, int32_t ordinal, const uint8_t* image, size_t nbytes, int64_t block_first, int64_t block_last,
                      dfdb_sizestats* stats) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  // validate the header like check_column_head does
  Rd hr{image, nbytes, 0}; int64_t bs; std::string ty;
  if (!hr.i64(bs) || !hr.str(ty)) fail(DFDB_ERR_FORMAT, "bad column header");
  if (bs != t->block_size) fail(DFDB_ERR_FORMAT, "column %s has blocksize %lld, but table has blocksize %lld", c.name.c_str(), (long long)bs, (long long)t->block_size);
  std::string lg;
  if (dt_parse_ex(ty, &lg) != c.dtype || lg != c.logical) fail(DFDB_ERR_FORMAT, "column %s stored type is %s, but %s expected", c.name.c_str(), ty.c_str(), dt_type_string(c.dtype, c.logical).c_str());
  dfdb_sizestats st{0, 0, 0};
  load_from_image(t, c, image, nbytes, hr.pos, block_first, block_last, &st);
  if (stats) *stats = st;
}

// stream.cpp, late materialization (blocksiterator.jl:111-113: a block without survivors never decompresses its projection-only columns): a SUBSET of
// the blocks of a chunk, their compressed bodies already queued into t->ld_staged at staged_off, decoded to their own rows of the chunk's column
void table_decode_staged_blocks(dfdb_table* t, int32_t ordinal, const StagedBlock* bl, int64_t n, int64_t total_rows, dfdb_sizestats* stats) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  std::vector<BlockHdr> hs((size_t)n);
  std::vector<int64_t> pos((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; i++) {
    hs[(size_t)i] = BlockHdr{bl[i].rows, bl[i].origin, bl[i].compressed, (size_t)bl[i].staged_off};
    pos[(size_t)i] = bl[i].row_pos;
    if (bl[i].row_pos < 0 || bl[i].row_pos + bl[i].rows > total_rows) fail(DFDB_ERR_FORMAT, "block %lld of column %s lies outside its chunk", (long long)i, c.name.c_str());
  }
  dfdb_sizestats st{0, 0, 0};
  decode_staged(t, c, hs.data(), n, t->nrows >= 0 ? t->block_first : 0, 0, &st, pos.data(), total_rows);
  if (stats) *stats = st;
}

// ---------------------------------------------------------------- K9: dictionary of a low-cardinality String column
// The dictionary starts from the distinct strings of the column's first rows; k_dict_encode then codes every row against an open-addressing
// table of it and reports the rows it could not code (their strings land in a staging arena); the host adds those and runs the pass again.
// Gives up (returns 0, nothing kept) past max_entries, for a nullable column, or for a string too long for the staging arena.
int64_t table_build_dictionary(dfdb_table* t, int32_t ordinal, int64_t max_entries) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  if (dt_base(c.dtype) != DFDB_STRING) fail(DFDB_ERR_ARGUMENT, "ArgumentError: column %s is not a String column", c.name.c_str());
  if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
  c.dict_n = 0; c.dict_host.clear();
  if (dt_nullable(c.dtype) || c.nrows <= 0) return 0;
  if (max_entries <= 0) return 0;
  if (max_entries > 65535) max_entries = 65535;                   // 0xffff is "not coded"
  dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  constexpr int64_t kMaxRecords = 8192; constexpr int32_t kMaxLen = 4096; constexpr int64_t kArena = 4 << 20;
  std::vector<std::string> dict;
  std::unordered_map<std::string, uint32_t> index;
  auto add = [&](std::string&& v) { if (index.emplace(v, (uint32_t)dict.size()).second) dict.push_back(std::move(v)); };
  // seed: the first tile's rows
  {
    const int64_t m = std::min<int64_t>(c.nrows, kStrTileRows);
    std::vector<int32_t> sz((size_t)m);
    HIP_CHECK(hipMemcpyAsync(sz.data(), c.data.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    int64_t nb = 0; for (int32_t v : sz) nb += v > 0 ? v : 0;
    std::vector<uint8_t> by((size_t)nb + 1);
    if (nb) HIP_CHECK(hipMemcpy(by.data(), c.bytes.p, (size_t)nb, hipMemcpyDeviceToHost));
    int64_t o = 0;
    for (int32_t v : sz) { const int64_t l = v > 0 ? v : 0; add(std::string((const char*)by.data() + o, (size_t)l)); o += l; if ((int64_t)dict.size() > max_entries) return 0; }
  }
  DevBuf codes, slots, dbytes, mcount, moff, mlen, marena;
  codes.ensure((size_t)round_up(c.nrows, kCTileRows) * 2 + 256);
  mcount.ensure(64); moff.ensure(kMaxRecords * 4); mlen.ensure(kMaxRecords * 4); marena.ensure(kArena + 64);
  std::vector<uint32_t> off_host;
  for (int round = 0; round < 64; round++) {
    // the table: 4 x entries slots, a power of two
    uint32_t nslots = 64; while (nslots < 4 * dict.size()) nslots <<= 1;
    std::vector<DictSlot> hs(nslots, DictSlot{0, 0xffffffffu, 0, 0, 0});
    std::vector<uint8_t> hb; off_host.assign(dict.size(), 0);
    for (size_t k = 0; k < dict.size(); k++) { off_host[k] = (uint32_t)hb.size(); hb.insert(hb.end(), dict[k].begin(), dict[k].end()); }
    hb.resize(hb.size() + 64, 0);                                    // (8-byte probes read past the last entry)
    for (size_t k = 0; k < dict.size(); k++) {
      const uint32_t len = (uint32_t)dict[k].size();
      uint64_t key8 = 0; memcpy(&key8, dict[k].data(), std::min<size_t>(8, len));
      uint32_t at = (uint32_t)dict_hash_host(key8, len) & (nslots - 1);
      while (hs[at].len != 0xffffffffu) at = (at + 1) & (nslots - 1);
      hs[at] = DictSlot{key8, len, (uint32_t)k, off_host[k], 0};
    }
    slots.ensure(hs.size() * sizeof(DictSlot)); dbytes.ensure(hb.size());
    HIP_CHECK(hipMemcpyAsync(slots.p, hs.data(), hs.size() * sizeof(DictSlot), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(dbytes.p, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(mcount.p, 0, 64, s));
    HIP_CHECK(hipStreamSynchronize(s));                               // hs / hb are pageable host memory
    DictMiss miss{(unsigned long long*)mcount.p, (unsigned long long*)mcount.p + 1, moff.as<uint32_t>(), mlen.as<int32_t>(), marena.as<uint8_t>(), kMaxRecords, kArena, kMaxLen};
    { LaunchTimer lt(ctx, "dict_encode");
      launch_dict_encode(s, c.data.as<int32_t>(), (const int64_t*)c.tile_off.p, c.bytes.as<uint8_t>(), slots.as<DictSlot>(), nslots, dbytes.as<uint8_t>(), codes.as<uint16_t>(), c.nrows, miss); }
    unsigned long long cnt[2] = {0, 0};
    HIP_CHECK(hipMemcpyAsync(cnt, mcount.p, 16, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (cnt[0] == 0) {                                                // every row has its code: keep
      c.dict_codes = std::move(codes); c.dict_bytes = std::move(dbytes);
      std::vector<int32_t> lens(dict.size()); for (size_t k = 0; k < dict.size(); k++) lens[k] = (int32_t)dict[k].size();
      c.dict_len.ensure(lens.size() * 4 + 64); c.dict_off.ensure(off_host.size() * 4 + 64);
      HIP_CHECK(hipMemcpy(c.dict_len.p, lens.data(), lens.size() * 4, hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(c.dict_off.p, off_host.data(), off_host.size() * 4, hipMemcpyHostToDevice));
      c.dict_host = std::move(dict); c.dict_n = (int32_t)c.dict_host.size();
      return c.dict_n;
    }
    const int64_t nrec = (int64_t)std::min<unsigned long long>(cnt[0], (unsigned long long)kMaxRecords);
    std::vector<uint32_t> ro((size_t)nrec); std::vector<int32_t> rl((size_t)nrec);
    const size_t used = (size_t)std::min<unsigned long long>(cnt[1], (unsigned long long)kArena);
    std::vector<uint8_t> ar(used + 1);
    HIP_CHECK(hipMemcpy(ro.data(), moff.p, (size_t)nrec * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(rl.data(), mlen.p, (size_t)nrec * 4, hipMemcpyDeviceToHost));
    if (used) HIP_CHECK(hipMemcpy(ar.data(), marena.p, used, hipMemcpyDeviceToHost));
    size_t before = dict.size();
    for (int64_t k = 0; k < nrec; k++) {
      if (rl[(size_t)k] == -2) return 0;                              // a string longer than the dictionary takes
      if (rl[(size_t)k] < 0) continue;                                // no room this round
      add(std::string((const char*)ar.data() + ro[(size_t)k], (size_t)rl[(size_t)k]));
      if ((int64_t)dict.size() > max_entries) return 0;
    }
    if (dict.size() == before) return 0;                              // (cannot happen: a reported row always adds its string)
  }
  return 0;
}

void table_load(dfdb_table* t, const int32_t* ordinals, int32_t ncols, int64_t block_first, int64_t block_last, dfdb_sizestats* stats) {
  dfdb_sizestats tot{0, 0, 0};
  std::vector<int32_t> all;
  if (!ordinals) { for (size_t i = 0; i < t->cols.size(); i++) all.push_back((int32_t)i); ordinals = all.data(); ncols = (int32_t)all.size(); }
  for (int32_t k = 0; k < ncols; k++) {
    const int32_t o = ordinals[k];
    if (o < 0 || (size_t)o >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", o);
    Column& c = t->cols[(size_t)o];
    if (c.resident) continue;
    if (c.file.empty()) fail(DFDB_ERR_IO, "column %s has no backing file", c.name.c_str());
    dfdb_sizestats st{0, 0, 0};
    load_from_file(t, c, block_first, block_last, &st);
    tot.rows = st.rows; tot.compressed += st.compressed; tot.uncompressed += st.uncompressed;
  }
  if (stats) *stats = tot;
}

}  // namespace dfdb

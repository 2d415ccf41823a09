// writer.cpp — resident columns -> table directory in the reference's on-disk format.
//
// Replaces (paths under /root/reference): write_table_meta src/io/table_io.jl:9-19, write_column_head
// src/io/filesystem.jl:14-23, the write_column block loop src/tables/columns.jl:39-53, write_block_body
// src/io/blocks.jl:2-33 and commit_block_write! src/io/BlockStreams.jl:36-60 (LZ4_compress + the 20-byte
// block header: Int32 rows, Int64 origin, Int64 compressed).  The column never leaves HBM decoded: block
// bodies are packed (nullable / String) and LZ4-compressed on the device, a wave per block (k_encode.hip);
// only compressed bytes cross PCIe.  Blocks are processed in batches so the staging arenas stay bounded.
#include "engine.hpp"
#include <cerrno>
#include <cstring>
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace dfdb {

void launch_lz4_compress(hipStream_t s, const uint8_t* src, uint8_t* dst, const Lz4Block* blocks, int32_t nblocks, int32_t* out_len);
void launch_pack_file_image(hipStream_t s, const uint8_t* comp, const Lz4Block* blocks, const int32_t* lens, const int64_t* pos, const int32_t* rows,
                            int32_t nblocks, uint8_t* image);
void set_lz4_enc_near(int64_t v);  // k_encode.hip: a far match is given up for a near one that ends as late (ctx option "lz4_enc_near": the near reach in bytes, 0 = off)
void set_lz4_enc_variant(int v);   // 0 = v1 (one sequence per step), 1 = v2 (every match of a 64-byte window per step, default)
void launch_pack_nullable(hipStream_t s, const uint8_t* values, const uint64_t* missing_bits, const int64_t* row_off, const int64_t* body_off,
                          int32_t nblocks, int width, uint8_t* bodies);
void launch_pack_strings(hipStream_t s, const int32_t* sizes, const uint8_t* bytes, const int64_t* row_off, const int64_t* byte_off,
                         const int64_t* body_off, int32_t nblocks, uint8_t* bodies);
void launch_row_byte_offsets(hipStream_t s, const int32_t* sizes, const int64_t* tile_off, const int64_t* rows, int32_t n, int64_t nrows, int64_t* out);

namespace {

struct Wr {   // little-endian writer (native Julia `write`)
  std::vector<uint8_t> b;
  void i32(int32_t v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + 4); }
  void i64(int64_t v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + 8); }
  void str(const std::string& s) { i32((int32_t)s.size()); b.insert(b.end(), s.begin(), s.end()); }   // write_string: common_io.jl:1-4
};

struct File {   // sequential writer over a file descriptor; large pieces go out as concurrent pwrites
  int fd = -1; std::string name; int64_t pos = 0;
  // O_EXCL: an existing file is an error, like make_column_file (filesystem.jl:14-17: "Column file ... already exists")
  File(const std::string& fn) : name(fn) {
    fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0666);
    if (fd < 0) fail(DFDB_ERR_IO, errno == EEXIST ? "file %s already exists" : "cannot create %s: %s", fn.c_str(), strerror(errno));
  }
  ~File() { if (fd >= 0) ::close(fd); }
  static bool pwrite_all(int fd, const uint8_t* p, int64_t n, int64_t at) {
    while (n > 0) { const ssize_t r = pwrite(fd, p, (size_t)n, (off_t)at); if (r <= 0) return false; p += r; n -= r; at += r; }
    return true;
  }
  void put(const void* p, size_t n) {
    if (!n) return;
    const uint8_t* b = (const uint8_t*)p;
    if (n < (size_t)(8 << 20)) {
      if (!pwrite_all(fd, b, (int64_t)n, pos)) fail(DFDB_ERR_IO, "short write to %s", name.c_str());
      pos += (int64_t)n; return;
    }
    // a large piece: write()s to one file serialise on the inode lock, page faults on a shared mapping do not — grow the file,
    // map the piece and fill it with concurrent memcpys
    // RESERVE the space first: a sparse ftruncate always succeeds and a full disk would then surface as SIGBUS inside the memcpy
    const int fa = posix_fallocate(fd, (off_t)pos, (off_t)n);
    if (fa == ENOSPC || fa == EDQUOT || fa == EFBIG) fail(DFDB_ERR_IO, "cannot reserve %zu bytes in %s: %s", n, name.c_str(), strerror(fa));
    const int64_t page = 4096, m0 = pos / page * page, mlen = pos + (int64_t)n - m0;
    void* mp = fa == 0 ? mmap(nullptr, (size_t)mlen, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)m0) : MAP_FAILED;
    if (mp == MAP_FAILED) {                                            // (a file system without fallocate or shared mappings: plain writes report errors themselves)
      if (!pwrite_all(fd, b, (int64_t)n, pos)) fail(DFDB_ERR_IO, "short write to %s", name.c_str());
      pos += (int64_t)n; return;
    }
    uint8_t* d = (uint8_t*)mp + (pos - m0);
    const int parts = 8;
    std::vector<std::thread> th;
    for (int k = 0; k < parts; k++) {
      const int64_t a = (int64_t)n * k / parts, e = (int64_t)n * (k + 1) / parts;
      auto work = [d, b, a, e] { memcpy(d + a, b + a, (size_t)(e - a)); };
      if (k + 1 < parts) th.emplace_back(work); else work();
    }
    for (auto& t : th) t.join();
    munmap(mp, (size_t)mlen);
    pos += (int64_t)n;
  }
  void close(bool sync = false) {
    if (fd >= 0 && sync && fdatasync(fd) != 0) { ::close(fd); fd = -1; fail(DFDB_ERR_IO, "cannot sync %s: %s", name.c_str(), strerror(errno)); }
    if (fd >= 0 && ::close(fd) != 0) { fd = -1; fail(DFDB_ERR_IO, "cannot close %s", name.c_str()); }
    fd = -1;
  }
};

inline int64_t lz4_bound(int64_t n) { return n + n / 255 + 16; }   // LZ4_COMPRESSBOUND

constexpr int64_t kBatchBodyBytes = 512ll << 20;   // uncompressed bytes encoded per batch

}  // namespace

// one column file: header, then every block of the RESIDENT rows
void table_save_column(dfdb_table* t, int32_t ordinal, const char* file, dfdb_sizestats* stats) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
  if (t->block_first != 0) fail(DFDB_ERR_UNSUPPORTED, "a block-range shard cannot be saved as a whole column");
  dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  if (c.comp_only) (void)column_data(t, c);             // a compressed-only column is decoded for the duration of the save (dropped by the caller's TransientScope)
  NodeBind bind(ctx);                                   // bounce buffers and the threads that fill the file mapping on the GPU's NUMA node
  const int64_t B = t->block_size, nrows = c.nrows, nb = ceil_div(nrows, B);
  const int w = dt_width(c.dtype);
  const bool is_str = dt_base(c.dtype) == DFDB_STRING, is_null = dt_nullable(c.dtype) && !is_str;

  File f(file);
  { Wr h; h.i64(B); h.str(dt_type_string(c.dtype, c.logical)); f.put(h.b.data(), h.b.size()); }   // write_column_head: filesystem.jl:14-23
  dfdb_sizestats st{0, 0, 0};
  st.rows = nrows;

  // String columns: arena offset of the first row of every block
  std::vector<int64_t> blk_byte((size_t)nb + 1, 0);
  if (is_str && nb) {
    std::vector<int64_t> rows((size_t)nb + 1);
    for (int64_t b = 0; b <= nb; b++) rows[(size_t)b] = std::min(b * B, nrows);
    DevBuf d; d.ensure(16 * ((size_t)nb + 1));
    HIP_CHECK(hipMemcpyAsync(d.p, rows.data(), 8 * ((size_t)nb + 1), hipMemcpyHostToDevice, s));
    launch_row_byte_offsets(s, c.data.as<int32_t>(), (const int64_t*)c.tile_off.p, d.as<int64_t>(), (int32_t)(nb + 1), nrows, d.as<int64_t>() + nb + 1);
    HIP_CHECK(hipMemcpyAsync(blk_byte.data(), d.as<int64_t>() + nb + 1, 8 * ((size_t)nb + 1), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }
  auto body_bytes = [&](int64_t b) -> int64_t {   // blocks.jl:2-33
    const int64_t rows = std::min(B, nrows - b * B);
    if (is_str) return 4 + 4 * rows + (blk_byte[(size_t)b + 1] - blk_byte[(size_t)b]);
    if (is_null) return 8 * ceil_div(rows, 64) + (int64_t)w * rows;
    return (int64_t)w * rows;
  };

  DevBuf bodies, comp, dblocks, dlens, daux, image, dpos;
  int64_t b0 = 0;
  while (b0 < nb) {
    // batch [b0, b1): bounded by kBatchBodyBytes of bodies
    int64_t b1 = b0, tot = 0;
    while (b1 < nb && (b1 == b0 || tot + body_bytes(b1) <= kBatchBodyBytes)) { tot += round_up(body_bytes(b1), 16); b1++; }
    const int64_t n = b1 - b0;
    std::vector<Lz4Block> blocks((size_t)n);
    std::vector<int64_t> body_off((size_t)n + 1), row_off((size_t)n + 1), byte_off((size_t)n + 1);
    int64_t bo = 0, co = 0;
    for (int64_t i = 0; i < n; i++) {
      const int64_t b = b0 + i, len = body_bytes(b);
      if (len > 0x7e000000LL) fail(DFDB_ERR_UNSUPPORTED, "block body larger than the LZ4 block limit");
      body_off[(size_t)i] = bo; row_off[(size_t)i] = b * B; byte_off[(size_t)i] = blk_byte[(size_t)b];
      blocks[(size_t)i].src_off = (is_str || is_null) ? bo : b * B * w;   // plain columns are compressed straight out of the column
      blocks[(size_t)i].src_len = (int32_t)len; blocks[(size_t)i].dst_len = (int32_t)lz4_bound(len);
      blocks[(size_t)i].dst_off = co;
      bo += round_up(len, 16); co += round_up(lz4_bound(len), 16);
    }
    body_off[(size_t)n] = bo; row_off[(size_t)n] = std::min(b1 * B, nrows); byte_off[(size_t)n] = blk_byte[(size_t)b1];

    const uint8_t* csrc = c.data.as<uint8_t>();
    if (is_str || is_null) {
      bodies.ensure((size_t)bo + 64);
      daux.ensure(8 * 3 * ((size_t)n + 1));
      int64_t* d_body = daux.as<int64_t>(); int64_t* d_row = d_body + n + 1; int64_t* d_byte = d_row + n + 1;
      HIP_CHECK(hipMemcpyAsync(d_body, body_off.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(d_row, row_off.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(d_byte, byte_off.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, s));
      if (is_str) launch_pack_strings(s, c.data.as<int32_t>(), c.bytes.as<uint8_t>(), d_row, d_byte, d_body, (int32_t)n, bodies.as<uint8_t>());
      else launch_pack_nullable(s, c.data.as<uint8_t>(), c.missing.as<uint64_t>(), d_row, d_body, (int32_t)n, w, bodies.as<uint8_t>());
      csrc = bodies.as<uint8_t>();
    }
    comp.ensure((size_t)co + 64);
    dblocks.ensure(sizeof(Lz4Block) * (size_t)n);
    dlens.ensure(4 * (size_t)n);
    HIP_CHECK(hipMemcpyAsync(dblocks.p, blocks.data(), sizeof(Lz4Block) * (size_t)n, hipMemcpyHostToDevice, s));
    set_lz4_enc_variant((int)ctx_option(ctx, "lz4_enc_variant", 1));
    set_lz4_enc_near(ctx_option(ctx, "lz4_enc_near", 0));
    { LaunchTimer lt(ctx, "lz4_compress"); launch_lz4_compress(s, csrc, comp.as<uint8_t>(), dblocks.as<Lz4Block>(), (int32_t)n, dlens.as<int32_t>()); }
    std::vector<int32_t> lens((size_t)n);
    HIP_CHECK(hipMemcpyAsync(lens.data(), dlens.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    // the batch's piece of the file — 20-byte header + compressed bytes of every block, back to back (commit_block_write!:
    // BlockStreams.jl:50-53) — is assembled on the device and leaves through the two pinned bounce buffers: the copy of one piece
    // overlaps the pwrite of the previous one
    std::vector<int64_t> fpos((size_t)n + 1); std::vector<int32_t> brow((size_t)n);
    int64_t fo = 0;
    for (int64_t i = 0; i < n; i++) {
      if (lens[(size_t)i] <= 0 || lens[(size_t)i] > blocks[(size_t)i].dst_len) fail(DFDB_ERR_DEVICE, "LZ4 compression failed in block %lld of column %s", (long long)(b0 + i), c.name.c_str());
      fpos[(size_t)i] = fo; fo += 20 + lens[(size_t)i];
      brow[(size_t)i] = (int32_t)std::min(B, nrows - (b0 + i) * B);
      st.compressed += lens[(size_t)i] + 24; st.uncompressed += blocks[(size_t)i].src_len;   // SizeStats incl. the 24-byte header quirk (:7,23)
    }
    fpos[(size_t)n] = fo;
    image.ensure((size_t)fo + 64);
    dpos.ensure(8 * ((size_t)n + 1) + 4 * (size_t)n);
    int64_t* d_pos = dpos.as<int64_t>(); int32_t* d_rows = (int32_t*)(d_pos + n + 1);
    HIP_CHECK(hipMemcpyAsync(d_pos, fpos.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_rows, brow.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s));
    launch_pack_file_image(s, comp.as<uint8_t>(), dblocks.as<Lz4Block>(), dlens.as<int32_t>(), d_pos, d_rows, (int32_t)n, image.as<uint8_t>());
    constexpr int64_t kPiece = 64ll << 20;
    ensure_pin_ring(ctx, (size_t)kPiece);
    const int64_t npieces = ceil_div(fo, kPiece);
    auto start_copy = [&](int64_t k) {
      const int64_t a = k * kPiece, e = std::min(fo, a + kPiece);
      HIP_CHECK(hipMemcpyAsync(ctx->pin_ring[k & 1], image.as<uint8_t>() + a, (size_t)(e - a), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipEventRecord(ctx->pin_ev[k & 1], s));
    };
    if (npieces) start_copy(0);
    for (int64_t k = 0; k < npieces; k++) {
      HIP_CHECK(hipEventSynchronize(ctx->pin_ev[k & 1]));
      if (k + 1 < npieces) start_copy(k + 1);                     // into the other buffer, which the previous pwrite has left
      const int64_t a = k * kPiece, e = std::min(fo, a + kPiece);
      f.put(ctx->pin_ring[k & 1], (size_t)(e - a));
    }
    b0 = b1;
  }
  f.close(ctx_option(ctx, "save_fsync", 0) != 0);      // option save_fsync: the column's bytes are on stable storage before meta.bin names the table
  if (stats) *stats = st;
}

// A resident plain fixed-width column -> its COMPRESSED-ONLY (mode 2) or compressed-resident (mode 1) form without a file in between: the blocks are
// encoded on the device exactly as table_save_column encodes them — the bytes a saved file would hold, 20-byte headers included — and stay in HBM
// with their descriptors; mode 2 then releases the decoded array (what dfdb_table_load leaves behind under ctx option keep_compressed).
void table_compress_column(dfdb_table* t, int32_t ordinal, int32_t mode, dfdb_sizestats* stats) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_KEY, "KeyError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
  if (mode != 1 && mode != 2) fail(DFDB_ERR_ARGUMENT, "ArgumentError: mode 1 (keep the decoded array) or 2 (compressed-only)");
  if (dt_base(c.dtype) == DFDB_STRING || dt_nullable(c.dtype)) fail(DFDB_ERR_UNSUPPORTED, "column %s: only plain fixed-width columns have a compressed-resident form", c.name.c_str());
  if (c.comp_only && c.comp_nblocks) { if (stats) { stats->rows = c.nrows; stats->uncompressed = (int64_t)c.nrows * dt_width(c.dtype); stats->compressed = (int64_t)c.comp.bytes; } return; }
  dfdb_ctx* ctx = t->ctx; hipStream_t s = ctx->stream;
  const int64_t B = t->block_size, nrows = c.nrows, nb = ceil_div(nrows, B);
  const int w = dt_width(c.dtype);
  if (nb == 0) fail(DFDB_ERR_ARGUMENT, "column %s has no rows", c.name.c_str());
  HIP_CHECK(hipStreamSynchronize(s));
  c.comp.release(); c.comp_blocks.release(); c.comp_status.release(); c.comp_index.release();
  c.comp_nblocks = 0; c.comp_index_state = 0; c.comp_blocks_host.clear();
  DevBuf scratch, dblocks, dlens, dpos, image;
  std::vector<Lz4Block> all((size_t)nb);
  int64_t used = 0, b0 = 0;
  set_lz4_enc_variant((int)ctx_option(ctx, "lz4_enc_variant", 1));
  set_lz4_enc_near(ctx_option(ctx, "lz4_enc_near", 0));
  while (b0 < nb) {
    int64_t b1 = b0, co = 0;
    std::vector<Lz4Block> blocks;
    while (b1 < nb && (b1 == b0 || (b1 - b0) * B * w < kBatchBodyBytes)) {
      const int64_t len = std::min(B, nrows - b1 * B) * w;
      if (len > 0x7e000000LL) fail(DFDB_ERR_UNSUPPORTED, "block body larger than the LZ4 block limit");
      Lz4Block x; x.src_off = b1 * B * w; x.src_len = (int32_t)len; x.dst_len = (int32_t)lz4_bound(len); x.dst_off = co;
      co += round_up(lz4_bound(len), 16); blocks.push_back(x); b1++;
    }
    const int64_t n = b1 - b0;
    scratch.ensure((size_t)co + 64); dblocks.ensure(sizeof(Lz4Block) * (size_t)n); dlens.ensure(4 * (size_t)n);
    HIP_CHECK(hipMemcpyAsync(dblocks.p, blocks.data(), sizeof(Lz4Block) * (size_t)n, hipMemcpyHostToDevice, s));
    { LaunchTimer lt(ctx, "lz4_compress"); launch_lz4_compress(s, c.data.as<uint8_t>(), scratch.as<uint8_t>(), dblocks.as<Lz4Block>(), (int32_t)n, dlens.as<int32_t>()); }
    std::vector<int32_t> lens((size_t)n);
    HIP_CHECK(hipMemcpyAsync(lens.data(), dlens.p, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int64_t> fpos((size_t)n + 1); std::vector<int32_t> brow((size_t)n);
    int64_t fo = 0;
    for (int64_t i = 0; i < n; i++) {
      if (lens[(size_t)i] <= 0 || lens[(size_t)i] > blocks[(size_t)i].dst_len) fail(DFDB_ERR_DEVICE, "LZ4 compression failed in block %lld of column %s", (long long)(b0 + i), c.name.c_str());
      fpos[(size_t)i] = fo; fo += 20 + lens[(size_t)i];
      brow[(size_t)i] = (int32_t)std::min(B, nrows - (b0 + i) * B);
      Lz4Block& a = all[(size_t)(b0 + i)];
      a.src_off = used + fpos[(size_t)i] + 20; a.src_len = lens[(size_t)i]; a.dst_len = blocks[(size_t)i].src_len; a.dst_off = (b0 + i) * B * w;
    }
    fpos[(size_t)n] = fo;
    // the column's image grows by this batch's piece (sized after the first batch's ratio, with room to spare; grown by copy if the guess was short)
    if ((int64_t)image.bytes < used + fo + 64) {
      const double done_frac = (double)b1 / (double)nb;
      const int64_t guess = (int64_t)((double)(used + fo) / done_frac * 1.03) + (64 << 10);
      DevBuf bigger; bigger.ensure((size_t)std::max<int64_t>(guess, used + fo + 64));
      if (used) HIP_CHECK(hipMemcpyAsync(bigger.p, image.p, (size_t)used, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
      image = std::move(bigger);
    }
    dpos.ensure(8 * ((size_t)n + 1) + 4 * (size_t)n);
    int64_t* d_pos = dpos.as<int64_t>(); int32_t* d_rows = (int32_t*)(d_pos + n + 1);
    HIP_CHECK(hipMemcpyAsync(d_pos, fpos.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_rows, brow.data(), 4 * (size_t)n, hipMemcpyHostToDevice, s));
    launch_pack_file_image(s, scratch.as<uint8_t>(), dblocks.as<Lz4Block>(), dlens.as<int32_t>(), d_pos, d_rows, (int32_t)n, image.as<uint8_t>() + used);
    HIP_CHECK(hipStreamSynchronize(s));                 // (fpos / brow are pageable host memory)
    used += fo; b0 = b1;
  }
  c.comp = std::move(image);
  c.comp_blocks.ensure(sizeof(Lz4Block) * (size_t)nb); c.comp_status.ensure(4 * (size_t)nb);
  HIP_CHECK(hipMemcpyAsync(c.comp_blocks.p, all.data(), sizeof(Lz4Block) * (size_t)nb, hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemsetAsync(c.comp_status.p, 0, 4 * (size_t)nb, s));
  HIP_CHECK(hipStreamSynchronize(s));
  c.comp_nblocks = nb; c.comp_blocks_host = std::move(all);
  if (mode == 2) {
    for (dfdb_query* q : t->queries) { q->executed_stages = -1; q->count = -1; q->prefix_valid = false; }
    c.data.release(); c.mask_pref.release(); c.mask_calibrated = false; c.comp_only = true; c.transient = false;
  }
  if (stats) { stats->rows = nrows; stats->uncompressed = nrows * w; stats->compressed = used; }
}

// make_table / create_table: meta.bin + one file per column (creators.jl:18-60, table_io.jl:9-19)
void table_save(dfdb_table* t, const char* path, dfdb_sizestats* stats) {
  const std::string dir(path);
  // table_exists(path) = isdir(path) (filesystem.jl:38): an existing directory IS an existing table for make_table_files (:31)
  if (mkdir(dir.c_str(), 0777) != 0) {
    if (errno == EEXIST) fail(DFDB_ERR_IO, "Table %s already exists", path);
    fail(DFDB_ERR_IO, "cannot create directory %s: %s", path, strerror(errno));
  }
  for (auto& c : t->cols) if (!c.resident) fail(DFDB_ERR_ARGUMENT, "column %s is not resident on the device (dfdb_table_load it first)", c.name.c_str());
  dfdb_sizestats tot{0, 0, 0};
  for (size_t i = 0; i < t->cols.size(); i++) {
    dfdb_sizestats st{0, 0, 0};
    table_save_column(t, (int32_t)i, (dir + "/" + std::to_string(t->cols[i].id) + ".bin").c_str(), &st);   // columnpath: filesystem.jl:11
    tot.rows = st.rows; tot.compressed += st.compressed; tot.uncompressed += st.uncompressed;
  }
  Wr m; m.i64(t->format_version); m.i64(t->block_size); m.i64((int64_t)t->cols.size());
  for (auto& c : t->cols) { m.i64(c.id); m.str(c.name); m.str(dt_type_string(c.dtype, c.logical)); }
  File f(dir + "/meta.bin");   // written last: a table without meta.bin "don't exists" (creators.jl:9)
  f.put(m.b.data(), m.b.size());
  f.close(ctx_option(t->ctx, "save_fsync", 0) != 0);
  if (stats) *stats = tot;
}

}  // namespace dfdb

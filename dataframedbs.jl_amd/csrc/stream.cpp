// stream.cpp — block-streamed execution of a query over a table that is NOT resident in HBM:
// the reference's Base.iterate(::BlocksIterator) (src/io/blocksiterator.jl:98-145) at the granularity of a CHUNK of
// blocks instead of one block.
//
// Each call of stream_next() hands out a query over the next chunk (block range [b0, b1) of every required column,
// decoded in HBM); the caller uses the ordinary dfdb_count / dfdb_select_indices / dfdb_materialize on it, exactly
// as the reference's consumers use the NamedTuple an iteration yields (valid until the next iterate).  While the
// caller works on chunk i, two loader threads read the compressed bytes of chunks i+1 and i+2 from the column files,
// copy them to the device and LZ4-decode them, each slot on ITS OWN HIP stream (three contexts, three streams, slots
// used round-robin), so file I/O of one chunk overlaps PCIe + K7 of the other and the caller's scan/gather kernels.
// HBM holds three chunks, never the table.
//
// Late materialization at BLOCK granularity (round 4; blocksiterator.jl:111-113 `if rows > 0 ... read_block!(projection-only columns)`, skip_block
// BlockStreams.jl:74-78; quirk Q6): a loader first reads, copies and decodes only the columns the selection reads, evaluates the chunk's selection
// on the slot's own stream — every stage up to the first range stage that numbers the survivors of earlier CHUNKS, whose base is not known yet:
// the survivors of that prefix are a superset of the final ones — and then reads the projection-only columns ONLY for the blocks that kept a row:
// their byte ranges are the only ones that leave the file, cross PCIe and go through K7.  A chunk without survivors never touches the projection
// files.  dfdb_stream_read_stats reports what was really read per column.  ctx option "stream_late_materialize" = 0: every required column whole.
//
// The reference's per-stage running state carries over between chunks the same way it carries over between blocks:
//   * a leading range stage numbers table rows            -> dfdb_table row_base = b0 * block_size
//   * a range stage after a predicate numbers survivors    -> stage_base += survivors of the chunk (RangeToProcess.offset,
//     selection.jl:68-75,107)
//   * skip_if_can / is_finished (selection.jl:177-196)     -> chunks before the first requested row are never read, and the
//     stream ends as soon as a range stage has passed its last element
#include "engine.hpp"
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <set>
#include <thread>
#include <unistd.h>
#include <chrono>

namespace dfdb {
static double dbg_ms() { static const auto e = std::chrono::steady_clock::now(); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - e).count(); }

void required_columns(const Node& n, std::vector<int>& out);

namespace {

// walk the headers of `bi`'s file until it holds at least `upto` blocks or the file ends (read_sizes / skip_block: BlockStreams.jl:68-78)
void extend_index(BlockIndex& bi, int64_t upto) {
  std::lock_guard<std::mutex> lk(bi.mu);
  if (bi.complete || (int64_t)bi.v.size() >= upto) return;
  const int fd = open(bi.file.c_str(), O_RDONLY);
  if (fd < 0) fail(DFDB_ERR_IO, "column file '%s' don't exists", bi.file.c_str());
  struct FdClose { int fd; ~FdClose() { close(fd); } } fdg{fd};
  const off_t end = lseek(fd, 0, SEEK_END);
  if (bi.next_pos < 0) bi.next_pos = (int64_t)bi.data_off;
  off_t pos = (off_t)bi.next_pos;
  uint8_t h[20];
  while (pos < end && (int64_t)bi.v.size() < upto) {
    if (pread(fd, h, 20, pos) != 20) fail(DFDB_ERR_FORMAT, "truncated block header in %s", bi.file.c_str());
    BlockLoc b; b.off = pos; memcpy(&b.rows, h, 4); memcpy(&b.origin, h + 4, 8); memcpy(&b.compressed, h + 12, 8);
    if (b.rows < 0 || b.origin < 0 || b.compressed < 0 || b.compressed > end - pos - 20) fail(DFDB_ERR_FORMAT, "corrupt block header in %s", bi.file.c_str());
    pos += 20 + b.compressed;
    bi.v.push_back(b);
  }
  bi.next_pos = (int64_t)pos;
  if (pos >= end) bi.complete = true;
}
// the index of a column file: the one its table column already holds if the file still looks the same, a fresh one otherwise
std::shared_ptr<BlockIndex> index_of(Column& c) {
  struct stat sb;
  if (c.file.empty() || stat(c.file.c_str(), &sb) != 0) fail(DFDB_ERR_IO, "column file '%s' for column %s don't exists", c.file.c_str(), c.name.c_str());
  const int64_t mt = (int64_t)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
  if (!c.bix || c.bix->file != c.file || c.bix->data_off != c.data_off || c.bix->file_size != (int64_t)sb.st_size || c.bix->file_mtime_ns != mt) {
    c.bix = std::make_shared<BlockIndex>();
    c.bix->file = c.file; c.bix->data_off = c.data_off; c.bix->file_size = (int64_t)sb.st_size; c.bix->file_mtime_ns = mt;
  }
  return c.bix;
}

struct Slot {
  dfdb_ctx* ctx = nullptr;
  dfdb_table* tbl = nullptr;
  dfdb_query* q = nullptr;
  int64_t b0 = 0, b1 = 0;
  bool loading = false;            // a load of this slot is queued or running on the loader thread
  bool has_chunk = false;          // tbl holds a decoded chunk the caller may be using
  int err_code = 0; std::string err_msg;
  uint8_t* pin = nullptr; size_t pin_cap = 0;   // pinned host staging for the file bytes (DMA-able: the H2D copy is truly async)
  // a whole-column chunk load goes through the first kRing pieces of `pin` in turn (round 5): a piece is read, queued for its copy, and reused once that copy
  // has left — 96 MB that stay in the socket's last-level cache instead of a chunk-sized buffer the preads write to memory and the DMA reads back from there
  static constexpr int kRing = 3;
  hipEvent_t ring_ev[kRing] = {nullptr, nullptr, nullptr};
  bool ring_used[kRing] = {false, false, false};
  std::vector<std::vector<BlockLoc>> locs;   // per required column: the chunk's blocks [b0, b1) (copied out of the shared index when the load is requested)
  bool pre_executed = false;       // the loader's evaluation of the selection IS the chunk's (no stage depends on earlier chunks, nothing raised): q keeps it
};

}  // namespace
}  // namespace dfdb

using namespace dfdb;

struct dfdb_stream {
  // everything the stream needs from the caller's query and table is COPIED at open (stages, block size, the required columns' files):
  // the stream stays valid after dfdb_query_free / dfdb_table_close of the handles it was opened from
  std::vector<dfdb::Stage> stages;
  int64_t block_size = 0;
  std::string path;
  struct ColSrc { std::string name, file; size_t data_off; };
  std::vector<ColSrc> colsrc;      // per required column
  int64_t chunk_blocks = 0, nblocks = 0, next_block = 0, chunks_issued = 0;
  int64_t win_first = 0, win_last = -1;   // the table's block window (a group shard streams its own block range)
  std::vector<int> required;       // table ordinals the query touches
  std::vector<std::shared_ptr<dfdb::BlockIndex>> index;   // per required column: walked lazily, a chunk ahead of the loaders (shared with the table's column)
  int64_t checked_blocks = 0;      // blocks whose row counts have been checked across the columns
  bool index_complete = false;     // every required column's file has been walked to its end: nblocks is final
  std::vector<int64_t> base;       // per stage: survivors of stages [0,k) in the chunks already consumed
  static constexpr int kSlots = 8, kLoaders = 7;   // capacity; a stream uses nslots slots and nslots - 1 loaders (ctx option "stream_slots", default 8)
  int nslots = 8;
  Slot slot[kSlots];
  int cur = -1;                    // slot handed to the caller (-1: none yet)
  bool done = false;
  int64_t compressed = 0, uncompressed = 0, rows = 0;
  // late materialization: which required columns the loaders' evaluation of the selection reads (loaded whole, first), how many stages it covers
  std::vector<char> sel_col;
  int kprefix = 0;
  bool late = true;
  std::vector<dfdb_sizestats> read_stats;   // per required column: rows / compressed (+24 per block, quirk Q10) / uncompressed bytes the loaders have read (under mu)
  // the loader threads live as long as the stream (a fresh host thread pays the HIP runtime's per-thread set-up, ~45 ms,
  // on its first call): requests are slot numbers served first in first out, completion is signalled per slot
  std::thread loader[kLoaders];
  std::mutex mu; std::condition_variable cv;
  std::deque<int> requests;        // slots waiting for a loader
  bool quit = false;
  bool slot_done[kSlots] = {};
  // At most `max_readers` loaders READ (page cache -> pinned, queueing the copies) at a time; the others are waiting for their copies and their decode.
  // Without the limit the loaders run in lockstep — all reading (the host's memcpy bandwidth split five ways, PCIe waiting for pieces), then all
  // waiting for PCIe and K7 at once while nobody reads — and the caller gets its chunks in bursts: 42 GB/s of file bytes on 8 slots, measured.
  int64_t piece_bytes = 64ll << 20;   // a whole-column chunk load reads and copies this much at a time (ctx option "stream_piece_mb")
  // the pinned rings of the whole-column loads belong to the READING TURNS, not to the slots: `max_readers` rings of three pieces are all the pinned memory
  // the reads ever write (a few hundred MB that stay in the socket's last-level cache) however many slots the stream has; a ring goes with the turn
  struct TurnRing { uint8_t* pin = nullptr; size_t cap = 0; hipEvent_t ev[3] = {nullptr, nullptr, nullptr}; bool used[3] = {false, false, false}; bool taken = false; int next = 0; };
  TurnRing turn_ring[kLoaders];
  int max_readers = 3, readers = 0;
  std::set<int64_t> waiting_readers;   // first blocks of the chunks whose loaders wait for a turn: the EARLIEST chunk reads first (the caller consumes in order)
  dfdb_ctx* parent = nullptr; std::weak_ptr<int> parent_alive;   // where a closed stream parks (if that context still exists)
};

namespace dfdb {


namespace {

struct ReadTurn {                   // RAII: one of the stream's `max_readers` reading turns, granted in chunk order
  dfdb_stream* s;
  int ring = 0;                     // which of the stream's turn rings this turn reads through
  ReadTurn(dfdb_stream* st, int64_t first_block);
  ~ReadTurn();
};
ReadTurn::ReadTurn(dfdb_stream* st, int64_t first_block) : s(st) {
  std::unique_lock<std::mutex> lk(s->mu);
  s->waiting_readers.insert(first_block);
  s->cv.wait(lk, [&] { return s->readers < s->max_readers && *s->waiting_readers.begin() == first_block; });
  s->waiting_readers.erase(first_block);
  s->readers++;
  for (int k = 0; k < s->max_readers; k++) if (!s->turn_ring[k].taken) { ring = k; break; }
  s->turn_ring[ring].taken = true;
  lk.unlock();
  s->cv.notify_all();                 // (the next chunk in line may have a turn left to take)
}
ReadTurn::~ReadTurn() {
  { std::lock_guard<std::mutex> lk(s->mu); s->readers--; s->turn_ring[ring].taken = false; }
  s->cv.notify_all();
}
void ring_drain(Slot* sl);
void ensure_pin(Slot* sl, size_t need) {
  if (need <= sl->pin_cap) return;
  ring_drain(sl);
  if (sl->pin) (void)hipHostFree(sl->pin);
  sl->pin = nullptr; sl->pin_cap = 0;
  HIP_CHECK(hipHostMalloc((void**)&sl->pin, need + need / 4, hipHostMallocDefault));
  sl->pin_cap = need + need / 4;
}
void note_read(dfdb_stream* s, size_t k, const dfdb_sizestats& st) {
  std::lock_guard<std::mutex> lk(s->mu);
  s->read_stats[k].rows += st.rows; s->read_stats[k].compressed += st.compressed; s->read_stats[k].uncompressed += st.uncompressed;
}

// every block of the chunk of required column k: file -> pinned -> HBM -> K7, the byte range read in 32-MB pieces (concurrent preads), each piece on
// its way to HBM while the next is read — PCIe and the page-cache copy overlap inside the chunk, not only across the loaders (which fall into
// lockstep: all reading, then all copying)
void ring_drain(Slot* sl) {            // every copy out of the slot's pinned ring has left it (the buffer is about to be used differently)
  for (int r = 0; r < Slot::kRing; r++) if (sl->ring_used[r]) { (void)hipEventSynchronize(sl->ring_ev[r]); sl->ring_used[r] = false; }
}
void load_column_whole(dfdb_stream* s, Slot* sl, size_t k) {
  dfdb_table* tb = sl->tbl;
  const dfdb_stream::ColSrc& c = s->colsrc[k];
  const std::vector<BlockLoc>& ix = sl->locs[k];              // the chunk's blocks, chunk-relative
  const int64_t lo = ix.front().off, hi = ix.back().off + 20 + ix.back().compressed;
  const int64_t kPiece = s->piece_bytes;
  auto t0 = std::chrono::steady_clock::now();
  DevBuf& staged = tb->ld_staged;
  staged.ensure((size_t)(hi - lo) + 64);
  const bool dbg_copy = getenv("DFDB_STREAM_DEBUG_COPIES") != nullptr;     // (timing events around every piece's copy: how long PCIe took, how long it sat idle before)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> dbg_ev; std::vector<double> dbg_mb;
  {
    ReadTurn turn(s, sl->b0);
    dfdb_stream::TurnRing& R = s->turn_ring[turn.ring];        // (mine for the duration of the turn; its events may belong to copies another slot queued)
    if (R.cap < (size_t)(3 * kPiece)) {
      for (int r = 0; r < 3; r++) if (R.used[r]) { (void)hipEventSynchronize(R.ev[r]); R.used[r] = false; }
      if (R.pin) (void)hipHostFree(R.pin);
      R.pin = nullptr; R.cap = 0;
      HIP_CHECK(hipHostMalloc((void**)&R.pin, (size_t)(3 * kPiece), hipHostMallocDefault));
      R.cap = (size_t)(3 * kPiece);
    }
    for (int r = 0; r < 3; r++) if (!R.ev[r]) HIP_CHECK(hipEventCreateWithFlags(&R.ev[r], hipEventDisableTiming));
    t0 = std::chrono::steady_clock::now();
    const int fd = open(c.file.c_str(), O_RDONLY);
    if (fd < 0) fail(DFDB_ERR_IO, "cannot read %s", c.file.c_str());
    struct FdClose { int fd; ~FdClose() { close(fd); } } fdg{fd};
    // (the ring goes on where the previous turn left it: starting every chunk at piece 0 made its first read wait for the previous chunk's second-to-last
    // copy — 0.6-1.6 ms per chunk — while the piece after it had been free for milliseconds)
    int& r = R.next;
    for (int64_t a = lo; a < hi; a += kPiece, r = (r + 1) % 3) {
      const int64_t e = std::min(hi, a + kPiece);
      uint8_t* buf = R.pin + (size_t)r * (size_t)kPiece;
      const auto tw0 = std::chrono::steady_clock::now();
      if (R.used[r]) HIP_CHECK(hipEventSynchronize(R.ev[r]));                 // its previous copy (this chunk's or an earlier turn's) has left the buffer
      const auto tw1 = std::chrono::steady_clock::now();
      if (!read_file_range_fd(fd, buf, a, e)) fail(DFDB_ERR_IO, "short read from %s", c.file.c_str());
      if (getenv("DFDB_STREAM_DEBUG")) { const auto tw2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[stream] slot %d piece at %lld: waited %.2f ms for its buffer, read %.1f MB in %.2f ms\n", (int)(sl - s->slot), (long long)(a - lo),
                std::chrono::duration<double, std::milli>(tw1 - tw0).count(), (double)(e - a) / 1e6, std::chrono::duration<double, std::milli>(tw2 - tw1).count()); }
      // (on the slot's own engine stream: one DMA queue per slot.  ONE queue for every slot's pieces, the slots waiting by event, was measured slower
      // — 45 GB/s against 50 with two or three reading turns: copies of different queues overlap their starts and ends)
      hipEvent_t d0 = nullptr, d1 = nullptr;
      if (dbg_copy) { HIP_CHECK(hipEventCreate(&d0)); HIP_CHECK(hipEventCreate(&d1)); HIP_CHECK(hipEventRecord(d0, sl->ctx->stream)); }
      HIP_CHECK(hipMemcpyAsync(staged.as<uint8_t>() + (a - lo), buf, (size_t)(e - a), hipMemcpyHostToDevice, sl->ctx->stream));
      if (dbg_copy) { HIP_CHECK(hipEventRecord(d1, sl->ctx->stream)); dbg_ev.push_back({d0, d1}); dbg_mb.push_back((double)(e - a) / 1e6); }
      HIP_CHECK(hipEventRecord(R.ev[r], sl->ctx->stream)); R.used[r] = true;
    }
  }
  if (dbg_copy) {
    for (size_t i = 0; i < dbg_ev.size(); i++) {
      (void)hipEventSynchronize(dbg_ev[i].second);
      float ms = 0, gap = 0; (void)hipEventElapsedTime(&ms, dbg_ev[i].first, dbg_ev[i].second);
      if (i) (void)hipEventElapsedTime(&gap, dbg_ev[i - 1].second, dbg_ev[i].first);
      fprintf(stderr, "[stream] slot %d copy %zu: %.1f MB in %.2f ms (%.1f GB/s), %.2f ms after the previous one ended\n", (int)(sl - s->slot), i, dbg_mb[i], ms, dbg_mb[i] / ms, gap);
    }
    for (auto& pr : dbg_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  }
  const auto t1 = std::chrono::steady_clock::now();
  // the blocks as the index describes them (their headers were walked and checked against the other columns when the chunk was planned)
  std::vector<StagedBlock> bl; bl.reserve(ix.size());
  int64_t row = 0;
  for (const BlockLoc& L : ix) { bl.push_back(StagedBlock{L.rows, L.origin, L.compressed, (L.off - lo) + 20, row}); row += L.rows; }
  dfdb_sizestats st{0, 0, 0};
  table_decode_staged_blocks(tb, s->required[k], bl.data(), (int64_t)bl.size(), row, &st);
  st.rows = row;
  note_read(s, k, st);
  if (getenv("DFDB_STREAM_DEBUG")) {
    const auto t2 = std::chrono::steady_clock::now();
    fprintf(stderr, "[stream] t=%.2f ms slot %d blocks %lld-%lld col %s: read %.2f ms (%.1f MB), load+decode %.2f ms\n",
            dbg_ms() - std::chrono::duration<double, std::milli>(t2 - t0).count(), (int)(sl - s->slot), (long long)sl->b0, (long long)sl->b1, c.name.c_str(),
            std::chrono::duration<double, std::milli>(t1 - t0).count(), (double)(hi - lo) / 1e6, std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
}

// only the blocks of the chunk with keep[b] != 0 (b relative to the chunk): every run of consecutive kept blocks is one byte range of the file
void load_column_blocks(dfdb_stream* s, Slot* sl, size_t k, const std::vector<char>& keep, int64_t chunk_rows) {
  dfdb_table* tb = sl->tbl;
  const dfdb_stream::ColSrc& c = s->colsrc[k];
  const std::vector<BlockLoc>& ix = sl->locs[k];              // the chunk's blocks, chunk-relative
  const int64_t nb = sl->b1 - sl->b0;
  size_t need = 0;
  for (int64_t b = 0; b < nb; b++) if (keep[(size_t)b]) need += 20 + (size_t)ix[(size_t)b].compressed;
  ring_drain(sl);                                            // (this path fills the pinned buffer from its start)
  ensure_pin(sl, need + 64);
  DevBuf& staged = tb->ld_staged;
  staged.ensure(need + 64);
  std::vector<StagedBlock> bl;
  const auto t0 = std::chrono::steady_clock::now();
  if (need) {
    ReadTurn turn(s, sl->b0);
    const int fd = open(c.file.c_str(), O_RDONLY);
    if (fd < 0) fail(DFDB_ERR_IO, "cannot read %s", c.file.c_str());
    struct FdClose { int fd; ~FdClose() { close(fd); } } fdg{fd};
    constexpr int64_t kPiece = 32ll << 20;
    size_t poff = 0;
    for (int64_t b = 0; b < nb;) {
      if (!keep[(size_t)b]) { b++; continue; }
      int64_t e = b;
      while (e < nb && keep[(size_t)e]) e++;
      const BlockLoc& first = ix[(size_t)b]; const BlockLoc& last = ix[(size_t)(e - 1)];
      const int64_t lo = first.off, hi = last.off + 20 + last.compressed;
      for (int64_t a = lo; a < hi; a += kPiece) {
        const int64_t z = std::min(hi, a + kPiece);
        if (!read_file_range_fd(fd, sl->pin + poff + (size_t)(a - lo), a, z)) fail(DFDB_ERR_IO, "short read from %s", c.file.c_str());
        HIP_CHECK(hipMemcpyAsync(staged.as<uint8_t>() + poff + (size_t)(a - lo), sl->pin + poff + (size_t)(a - lo), (size_t)(z - a), hipMemcpyHostToDevice, sl->ctx->stream));
      }
      for (int64_t j = b; j < e; j++) {
        const BlockLoc& L = ix[(size_t)j];
        bl.push_back(StagedBlock{L.rows, L.origin, L.compressed, (int64_t)poff + (L.off - lo) + 20, j * s->block_size});
      }
      poff += (size_t)(hi - lo);
      b = e;
    }
  }
  const auto t1 = std::chrono::steady_clock::now();
  dfdb_sizestats st{0, 0, 0};
  table_decode_staged_blocks(tb, s->required[k], bl.data(), (int64_t)bl.size(), chunk_rows, &st);
  st.rows = 0; for (const StagedBlock& x : bl) st.rows += x.rows;
  note_read(s, k, st);
  if (getenv("DFDB_STREAM_DEBUG")) {
    const auto t2 = std::chrono::steady_clock::now();
    fprintf(stderr, "[stream] t=%.2f ms slot %d blocks %lld-%lld col %s: %zu of %lld blocks kept, read %.2f ms (%.1f MB), load+decode %.2f ms\n",
            dbg_ms() - std::chrono::duration<double, std::milli>(t2 - t0).count(), (int)(sl - s->slot), (long long)sl->b0, (long long)sl->b1, c.name.c_str(), bl.size(), (long long)nb,
            std::chrono::duration<double, std::milli>(t1 - t0).count(), (double)need / 1e6, std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
}

// loader thread: column files -> HBM (decoded) for blocks [b0, b1), into the slot's persistent table (buffers are reused)
void load_chunk(dfdb_stream* s, Slot* sl) {
  try {
    HIP_CHECK(hipSetDevice(sl->ctx->device));
    dfdb_table* tb = sl->tbl;
    tb->nrows = -1; tb->block_first = 0;
    sl->pre_executed = false;
    for (Column& c : tb->cols) c.resident = false;
    const int64_t nb = sl->b1 - sl->b0;
    int64_t chunk_rows = 0;
    for (const BlockLoc& L : sl->locs[0]) chunk_rows += L.rows;
    bool any_late = false;
    for (size_t k = 0; k < s->required.size(); k++) {
      if (s->late && !s->sel_col[k]) { any_late = true; continue; }
      load_column_whole(s, sl, k);
    }
    tb->block_first = 0;
    tb->row_base = sl->b0 * s->block_size;                // (a leading range stage numbers table rows: known before the selection runs)
    if (any_late) {
      std::vector<char> keep((size_t)nb, 1);
      dfdb_query* q = sl->q;
      if (tb->nrows < 0) tb->nrows = chunk_rows;          // the selection reads no column at all (range stages only): the block headers say how many rows there are
      if (s->kprefix > 0 && chunk_rows > 0) {
        q->executed_stages = -1; q->count = -1; q->prefix_valid = false; q->bitmap_rows = -1;
        for (Stage& st : q->stages) st.stage_base = 0;    // (no stage of the prefix numbers survivors of earlier chunks)
        // nothing is raised here: whether a DivideError / InexactError of a predicate is reached is the caller's full execution's to say (query.cpp:
        // error_is_reached); the erroring rows count as not selected meanwhile, and that execution raises or drops them for good
        q->err_checking = true;
        try { query_execute(q, s->kprefix); } catch (...) { q->err_checking = false; throw; }
        q->err_checking = false;
        std::vector<int64_t> counts;
        query_block_counts(q, s->block_size, counts);
        for (int64_t b = 0; b < nb && b < (int64_t)counts.size(); b++) keep[(size_t)b] = counts[(size_t)b] > 0;
        sl->pre_executed = s->kprefix == (int)q->stages.size() && q->err_row[0] == ~0ull && q->err_row[1] == ~0ull;
        if (!sl->pre_executed) { q->executed_stages = -1; q->count = -1; q->prefix_valid = false; q->err_row[0] = q->err_row[1] = ~0ull; }
      }
      bool all = true;
      for (char kp : keep) all = all && kp;
      for (size_t k = 0; k < s->required.size(); k++) {
        if (s->sel_col[k]) continue;
        if (all) load_column_whole(s, sl, k); else load_column_blocks(s, sl, k, keep, chunk_rows);
      }
    }
    tb->block_first = 0;
    tb->row_base = sl->b0 * s->block_size;
    sl->has_chunk = true;
  } catch (const Error& e) { sl->err_code = e.code; sl->err_msg = e.what(); }
  catch (const std::exception& e) { sl->err_code = DFDB_ERR_DEVICE; sl->err_msg = e.what(); }
}

void wait_loaded(dfdb_stream* s, Slot& sl) {
  if (!sl.loading) return;
  std::unique_lock<std::mutex> lk(s->mu);
  const int idx = (int)(&sl - s->slot);
  s->cv.wait(lk, [&] { return s->slot_done[idx]; });
  sl.loading = false;
}
void release_slot(dfdb_stream* s, Slot& sl) {   // the caller is done with this slot's chunk (buffers stay for the next one)
  wait_loaded(s, sl);
  (void)hipStreamSynchronize(sl.ctx->stream);
  sl.has_chunk = false;
}
void loader_main(dfdb_stream* s) {
  NodeBind bind(s->slot[0].ctx);          // for the thread's life: its preads (and the pread threads it starts), its slot's pinned buffer
  for (;;) {
    int idx;
    {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [&] { return s->quit || !s->requests.empty(); });
      if (s->quit) return;
      idx = s->requests.front(); s->requests.pop_front();
    }
    load_chunk(s, &s->slot[idx]);
    { std::lock_guard<std::mutex> lk(s->mu); s->slot_done[idx] = true; }
    s->cv.notify_all();
  }
}

bool range_like(const Stage& st) { return st.kind != ST_PRED; }

// every required column's index walked to `upto` blocks (or its file's end), the new blocks checked against each other (all columns of a table share
// their block boundaries: check_column_head, filesystem.jl:47-54).  Returns how many blocks are known for ALL columns.
int64_t walk_index(dfdb_stream* s, int64_t upto) {
  if (s->index.empty()) return 0;
  if (s->index_complete) return s->nblocks;
  const size_t nc = s->index.size();
  std::vector<int64_t> have(nc);
  std::vector<char> ended(nc);
  std::vector<std::vector<int32_t>> rows(nc);              // row counts of the blocks not checked yet, copied out under each index's lock
  for (size_t k = 0; k < nc; k++) {
    extend_index(*s->index[k], upto);
    std::lock_guard<std::mutex> lk(s->index[k]->mu);
    const std::vector<BlockLoc>& v = s->index[k]->v;
    have[k] = (int64_t)v.size(); ended[k] = s->index[k]->complete;
    for (int64_t b = s->checked_blocks; b < have[k]; b++) rows[k].push_back(v[(size_t)b].rows);
  }
  const int64_t known = *std::min_element(have.begin(), have.end()), most = *std::max_element(have.begin(), have.end());
  bool complete = true;
  for (size_t k = 0; k < nc; k++) {
    complete = complete && ended[k];
    if (ended[k] && have[k] < most) fail(DFDB_ERR_FORMAT, "columns of %s have different block counts", s->path.c_str());
  }
  for (size_t k = 0; k < nc; k++)
    for (int64_t b = s->checked_blocks; b < known; b++) {
      const int32_t r = rows[k][(size_t)(b - s->checked_blocks)];
      if (r != rows[0][(size_t)(b - s->checked_blocks)]) fail(DFDB_ERR_FORMAT, "columns of %s have different block boundaries", s->path.c_str());
      const bool last_of_file = ended[k] && b + 1 == have[k];
      if (!last_of_file && r != s->block_size) fail(DFDB_ERR_FORMAT, "block %lld holds %d rows, expected block_size %lld", (long long)b, r, (long long)s->block_size);
    }
  s->checked_blocks = known;
  if (complete) { s->index_complete = true; s->nblocks = known; }
  return known;
}

// start loading the next chunk that can still contribute rows into `sl`; false when the stream is exhausted
bool prefetch(dfdb_stream* s, Slot* sl) {
  const int64_t B = s->block_size;
  // skip_if_can (selection.jl:177-190): a leading range stage whose first element lies beyond a chunk skips it unread
  if (!s->stages.empty() && range_like(s->stages[0])) {
    const Stage& st = s->stages[0];
    const bool empty = (st.kind == ST_RANGE && st.n == 0) || (st.kind != ST_RANGE && st.idx.empty());
    if (empty) return false;
    const int64_t first_block = (st.first() - 1) / B;
    if (first_block > s->next_block) s->next_block = std::max(s->win_first, (first_block / s->chunk_blocks) * s->chunk_blocks);   // keep chunk boundaries fixed
    if (st.last() <= s->next_block * B) return false;                                  // is_finished (:192-196)
  }
  if (s->win_last >= 0 && s->next_block >= s->win_last) return false;                  // the end of the table's block window
  // the headers of the next chunk (and, after a skip, of everything before it: skip_block), walked now; the loaders get their own copy of the slice
  // the first chunks of a scan with large chunks are shorter (a quarter, then half of chunk_blocks): the caller gets its first rows after a quarter of the
  // read + copy + decode latency of a full chunk, and the loaders start out staggered instead of in step
  int64_t want = s->chunk_blocks;
  if (s->chunk_blocks >= 256 && s->chunks_issued < 2) want = s->chunk_blocks >> (2 - s->chunks_issued);
  s->chunks_issued++;
  if (s->win_last >= 0) want = std::min(want, s->win_last - s->next_block);
  const int64_t known = walk_index(s, s->next_block + want);
  if (s->next_block >= known) return false;
  sl->b0 = s->next_block; sl->b1 = std::min(known, s->next_block + want);
  s->next_block = sl->b1;
  sl->locs.assign(s->index.size(), {});
  for (size_t k = 0; k < s->index.size(); k++) {
    std::lock_guard<std::mutex> lk(s->index[k]->mu);
    sl->locs[k].assign(s->index[k]->v.begin() + sl->b0, s->index[k]->v.begin() + sl->b1);
  }
  sl->err_code = 0; sl->err_msg.clear();
  sl->loading = true;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    const int idx = (int)(sl - s->slot);
    s->slot_done[idx] = false; s->requests.push_back(idx);
  }
  s->cv.notify_all();
  return true;
}

}  // namespace

static void stream_open_impl(dfdb_query* q, int64_t chunk_blocks, dfdb_stream* s);
static void stream_destroy(dfdb_stream* s);
// a parked stream becomes a fresh one: everything that described the previous query goes, what was expensive to make stays (slot contexts,
// pinned buffers, the slots' tables with their device buffers — stream_open_impl moves those into the new tables —, loader threads)
static void stream_rearm(dfdb_stream* s) {
  s->stages.clear(); s->colsrc.clear(); s->required.clear(); s->index.clear(); s->base.clear();
  s->chunk_blocks = s->nblocks = s->next_block = s->chunks_issued = 0; s->cur = -1; s->done = false;
  s->win_first = 0; s->win_last = -1;
  s->compressed = s->uncompressed = s->rows = 0;
  s->sel_col.clear(); s->read_stats.clear(); s->kprefix = 0; s->index_complete = false; s->checked_blocks = 0;
  s->requests.clear();
  for (int i = 0; i < dfdb_stream::kSlots; i++) {
    Slot& sl = s->slot[i];
    s->slot_done[i] = false; sl.loading = false; sl.has_chunk = false; sl.pre_executed = false; sl.locs.clear(); sl.err_code = 0; sl.err_msg.clear(); sl.b0 = sl.b1 = 0;
    delete sl.q; sl.q = nullptr;
  }
}
void stream_open(dfdb_query* q, int64_t chunk_blocks, dfdb_stream** out) {
  if (getenv("DFDB_STREAM_DEBUG")) fprintf(stderr, "[stream] open begins at %.2f ms\n", dbg_ms());
  if (q->t->path.empty()) fail(DFDB_ERR_ARGUMENT, "streaming needs a table opened from files (dfdb_table_open)");
  dfdb_ctx* pc = q->t->ctx;
  dfdb_stream* s = nullptr;
  if (pc->parked_stream) { s = (dfdb_stream*)pc->parked_stream; pc->parked_stream = nullptr; stream_rearm(s); }
  else s = new dfdb_stream();
  s->parent = pc; s->parent_alive = pc->alive;
  try { stream_open_impl(q, chunk_blocks, s); } catch (...) { stream_destroy(s); throw; }
  *out = s;
  if (getenv("DFDB_STREAM_DEBUG")) fprintf(stderr, "[stream] open ends at %.2f ms\n", dbg_ms());
}
static void stream_open_impl(dfdb_query* q, int64_t chunk_blocks, dfdb_stream* s) {
  dfdb_table* t = q->t;
  s->block_size = t->block_size; s->path = t->path;
  for (const Stage& st : q->stages) {
    Stage c; c.kind = st.kind; c.start = st.start; c.step = st.step; c.stop = st.stop; c.n = st.n; c.idx = st.idx; c.stage_base = st.stage_base;   // (predicates live in the slots' queries)
    s->stages.push_back(std::move(c));
  }
  s->win_first = std::max<int64_t>(0, t->win_first); s->win_last = t->win_last;
  s->next_block = s->win_first;
  // one block is decoded by one wave in ~5-8 ms (K7 is serial inside a block) and the chip holds ~5000 waves, so a chunk
  // should hold several hundred blocks: 512 blocks = 0.25 GB of Int64 per slot, four slots
  s->chunk_blocks = chunk_blocks > 0 ? chunk_blocks : 512;
  // required_columns(view) (view.jl:183-190): selection columns first, then projection-only columns
  std::vector<int> req;
  for (const Stage& st : q->stages) if (st.kind == ST_PRED) required_columns(*st.pred, req);
  for (const ProjCol& p : q->proj) required_columns(*p.expr, req);
  // no required column at all (range stages + a projection of constants): the reference iterates nothing (`isempty(it.streams)`, blocksiterator.jl:101);
  // an empty projection (count only) takes its block sizes from the first column
  const bool nothing_to_read = req.empty() && !q->proj.empty();
  if (req.empty() && !t->cols.empty()) req.push_back(0);
  std::sort(req.begin(), req.end()); req.erase(std::unique(req.begin(), req.end()), req.end());
  s->required = req;
  // late materialization: the loaders evaluate stages [0, kprefix) — everything before the first range-like stage that follows another stage (its base is
  // the survivors of earlier chunks) — and the columns those stages' predicates read are the ones loaded whole
  s->late = ctx_option(t->ctx, "stream_late_materialize", 1) != 0;
  s->kprefix = (int)q->stages.size();
  for (size_t k = 1; k < q->stages.size(); k++) if (q->stages[k].kind != ST_PRED) { s->kprefix = (int)k; break; }
  std::vector<int> selreq;
  for (int k = 0; k < s->kprefix; k++) if (q->stages[(size_t)k].kind == ST_PRED) required_columns(*q->stages[(size_t)k].pred, selreq);
  s->sel_col.assign(req.size(), 0);
  for (size_t k = 0; k < req.size(); k++) s->sel_col[k] = std::find(selreq.begin(), selreq.end(), req[k]) != selreq.end();
  s->read_stats.assign(req.size(), dfdb_sizestats{0, 0, 0});
  for (int o : req) {
    Column& c = t->cols[(size_t)o];
    if (c.file.empty()) fail(DFDB_ERR_IO, "column %s has no backing file", c.name.c_str());
    s->index.push_back(index_of(c));                     // (nothing is walked yet: prefetch walks a chunk's headers at a time)
    s->colsrc.push_back(dfdb_stream::ColSrc{c.name, c.file, c.data_off});
  }
  s->nblocks = 0; s->index_complete = false; s->checked_blocks = 0;
  // (a stage's base starts at the survivors on the LOWER RANKS of a group — dfdb_query_set_stage_base, group.cpp plan_stage_bases — and grows by every chunk's)
  s->base.assign(q->stages.size(), 0);
  for (size_t k = 0; k < q->stages.size(); k++) s->base[k] = q->stages[k].stage_base;
  set_io_threads(ctx_option(t->ctx, "io_threads", 8));
  s->max_readers = (int)std::min<int64_t>(dfdb_stream::kLoaders, std::max<int64_t>(1, ctx_option(t->ctx, "stream_readers", 3)));
  s->readers = 0; s->waiting_readers.clear();
  s->piece_bytes = std::min<int64_t>(512, std::max<int64_t>(1, ctx_option(t->ctx, "stream_piece_mb", 64))) << 20;
  s->nslots = (int)std::min<int64_t>(dfdb_stream::kSlots, std::max<int64_t>(2, ctx_option(t->ctx, "stream_slots", 8)));
  for (int i = 0; i < s->nslots; i++) {
    Slot& sl = s->slot[i];
    if (!sl.ctx) { if (ctx_create_like(t->ctx, &sl.ctx) != 0) fail(DFDB_ERR_DEVICE, "cannot create a stream context"); }
    else sl.ctx->options = t->ctx->options;                // a re-armed slot: same device (same parent context), today's options
    sl.ctx->options["keep_compressed"] = 0;               // (a slot's staging buffer is reused by the next chunk)
    sl.ctx->options["placement_calibrate"] = 0;          // a slot's column lives for one chunk: nothing to calibrate for (and a 1024-block chunk is exactly 2^26 rows)
    auto tb = std::make_unique<dfdb_table>();            // the slot's chunk table: same columns, its own stream, reused buffers
    tb->ctx = sl.ctx; tb->path = t->path; tb->block_size = t->block_size; tb->format_version = t->format_version;
    tb->keep_load_scratch = true;
    for (const Column& c : t->cols) { Column n; n.name = c.name; n.id = c.id; n.dtype = c.dtype; n.logical = c.logical; n.file = c.file; n.data_off = c.data_off; tb->cols.push_back(std::move(n)); }
    auto cq = std::make_unique<dfdb_query>();             // the caller's query re-stated over the chunk table
    cq->t = tb.get();
    for (const Stage& st : q->stages) {
      Stage c; c.kind = st.kind; c.start = st.start; c.step = st.step; c.stop = st.stop; c.n = st.n; c.idx = st.idx;
      if (st.pred) c.pred = st.pred->clone();
      cq->stages.push_back(std::move(c));
    }
    for (const ProjCol& p : q->proj) cq->proj.push_back(ProjCol{p.name, p.expr->clone()});
    cq->stream_owned = true;                              // dfdb_query_free refuses it: it dies with the stream
    tb->queries.push_back(cq.get());
    if (sl.tbl) {                                         // re-armed: the previous tables' device buffers carry over (grown on demand by ensure())
      dfdb_table* old = sl.tbl;
      tb->ld_staged = std::move(old->ld_staged); tb->ld_bodies = std::move(old->ld_bodies); tb->ld_blocks = std::move(old->ld_blocks);
      tb->ld_status = std::move(old->ld_status); tb->ld_aux = std::move(old->ld_aux);
      // only the required columns ever held buffers: hand them to the new required columns in order
      std::vector<Column*> had;
      for (Column& oc : old->cols) if (oc.data.p || oc.bytes.p || oc.missing.p) had.push_back(&oc);
      size_t h = 0;
      for (int o : s->required) {
        if (h >= had.size()) break;
        Column& nc = tb->cols[(size_t)o]; Column& oc = *had[h++];
        nc.data = std::move(oc.data); nc.bytes = std::move(oc.bytes); nc.missing = std::move(oc.missing); nc.tile_off = std::move(oc.tile_off);
      }
      old->queries.clear();
      delete old;
    }
    sl.tbl = tb.release(); sl.q = cq.release();
  }
  for (int i = 0; i + 1 < s->nslots; i++) if (!s->loader[i].joinable()) s->loader[i] = std::thread(loader_main, s);
  if (nothing_to_read) { s->done = true; return; }
  if (!prefetch(s, &s->slot[0])) s->done = true;
  else for (int i = 1; i + 1 < s->nslots; i++) if (!prefetch(s, &s->slot[i])) break;
}

// the next chunk as a query (nullptr at the end).  The previous chunk's query dies here.
dfdb_query* stream_next(dfdb_stream* s, int64_t* chunk_rows, int64_t* first_row) {
  // 1. retire the chunk the caller just used: its survivors move the bases of the later range stages (RangeToProcess.offset)
  if (s->cur >= 0) {
    Slot& old = s->slot[s->cur];
    if (old.has_chunk) {
      for (size_t k = 1; k < s->stages.size(); k++)
        if (range_like(s->stages[k])) s->base[k] += query_count(old.q, (int)k);
      // is_finished: a range stage that has seen its last element ends the scan (selection.jl:192-196)
      for (size_t k = 1; k < s->stages.size(); k++) {
        const Stage& st = s->stages[k];
        if (!range_like(st)) continue;
        const bool empty = (st.kind == ST_RANGE && st.n == 0) || (st.kind != ST_RANGE && st.idx.empty());
        if (empty || st.last() <= s->base[k]) s->done = true;
      }
    }
  }
  const int K = s->nslots;
  const int nxt = s->cur < 0 ? 0 : (s->cur + 1) % K;
  Slot& sl = s->slot[nxt];
  if (s->cur >= 0) release_slot(s, s->slot[s->cur]);
  if (s->done || !sl.loading) {          // nothing was prefetched: end of stream
    for (int i = 0; i < s->nslots; i++) if (s->slot[i].loading) release_slot(s, s->slot[i]);
    s->done = true; s->cur = -1;
    return nullptr;
  }
  // 2. wait for the prefetched chunk, immediately start the one after it into the slot just retired
  const double tw0 = getenv("DFDB_STREAM_DEBUG") ? dbg_ms() : 0;
  wait_loaded(s, sl);
  if (tw0 != 0) fprintf(stderr, "[stream] t=%.2f ms next: slot %d blocks %lld-%lld handed out after waiting %.2f ms\n", dbg_ms(), nxt, (long long)sl.b0, (long long)sl.b1, dbg_ms() - tw0);
  if (sl.err_code) { const int c = sl.err_code; const std::string m = sl.err_msg; s->done = true; fail(c, "%s", m.c_str()); }
  s->cur = nxt;
  if (!prefetch(s, &s->slot[(nxt + K - 1) % K])) { /* no chunk left to start */ }
  // 3. the slot's query, re-based for this chunk
  for (size_t k = 0; k < sl.q->stages.size(); k++) sl.q->stages[k].stage_base = s->base[k];
  // (a chunk whose selection the loader has already evaluated in full keeps that evaluation: nothing in it depends on the chunks before)
  if (!sl.pre_executed) { sl.q->executed_stages = -1; sl.q->count = -1; sl.q->prefix_valid = false; sl.q->bitmap_rows = -1; }
  if (chunk_rows) *chunk_rows = sl.tbl->nrows < 0 ? 0 : sl.tbl->nrows;
  if (first_row) *first_row = sl.tbl->row_base;
  return sl.q;
}

static void stream_destroy(dfdb_stream* s) {
  for (Slot& sl : s->slot) if (sl.ctx) wait_loaded(s, sl);
  { std::lock_guard<std::mutex> lk(s->mu); s->quit = true; }
  s->cv.notify_all();
  for (auto& th : s->loader) if (th.joinable()) th.join();
  for (int i = 0; i < dfdb_stream::kSlots; i++) {
    Slot& sl = s->slot[i];
    if (sl.ctx) { release_slot(s, sl); }
    delete sl.q; sl.q = nullptr;
    delete sl.tbl; sl.tbl = nullptr;
    for (int r = 0; r < Slot::kRing; r++) if (sl.ring_ev[r]) { (void)hipEventDestroy(sl.ring_ev[r]); sl.ring_ev[r] = nullptr; sl.ring_used[r] = false; }
    if (sl.pin) (void)hipHostFree(sl.pin);
    if (sl.ctx) ctx_destroy(sl.ctx);
  }
  for (auto& R : s->turn_ring) {
    for (int r = 0; r < 3; r++) if (R.ev[r]) { if (R.used[r]) (void)hipEventSynchronize(R.ev[r]); (void)hipEventDestroy(R.ev[r]); }
    if (R.pin) (void)hipHostFree(R.pin);
  }
  delete s;
}
void stream_close(dfdb_stream* s) {
  if (!s) return;
  const bool dbg = getenv("DFDB_STREAM_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[stream] close begins at %.2f ms\n", dbg_ms());
  // park it on its context for the next dfdb_stream_open (ctx option "stream_cache" = 0: never), unless that context is gone or already holds one
  bool park = false;
  if (!s->parent_alive.expired() && s->parent && !s->parent->parked_stream && ctx_option(s->parent, "stream_cache", 1) != 0) {
    park = true;
    for (int i = 0; i < s->nslots; i++) if (!s->slot[i].ctx || !s->slot[i].tbl) park = false;      // (an open that failed half-way)
  }
  if (park) {
    for (int i = 0; i < s->nslots; i++) release_slot(s, s->slot[i]);         // loads in flight finish, the slots' streams drain
    s->parent->parked_stream = s;
  } else stream_destroy(s);
  if (dbg) fprintf(stderr, "[stream] close ends at %.2f ms (%s)\n", dbg_ms(), park ? "parked" : "destroyed");
}
void stream_drop_parked(dfdb_ctx* ctx) {
  if (!ctx->parked_stream) return;
  dfdb_stream* s = (dfdb_stream*)ctx->parked_stream; ctx->parked_stream = nullptr;
  stream_destroy(s);
}

// table_stats (src/tables/misc.jl:6-43): skip_block over one column file — block headers only, nothing is decoded
void table_column_stats(dfdb_table* t, int32_t ordinal, dfdb_sizestats* st) {
  if (ordinal < 0 || (size_t)ordinal >= t->cols.size()) fail(DFDB_ERR_BOUNDS, "BoundsError: column ordinal %d", ordinal);
  Column& c = t->cols[(size_t)ordinal];
  if (c.file.empty()) fail(DFDB_ERR_IO, "column %s has no backing file", c.name.c_str());
  st->rows = st->compressed = st->uncompressed = 0;
  std::shared_ptr<BlockIndex> bi = index_of(c);
  extend_index(*bi, INT64_MAX);
  std::lock_guard<std::mutex> lk(bi->mu);
  for (const BlockLoc& b : bi->v) { st->rows += b.rows; st->compressed += b.compressed + 24; st->uncompressed += b.origin; }   // +24: quirk Q10
}

// table_stats over the required columns: asks for every header of their files (the scan itself only walks a chunk ahead of its loaders)
void stream_stats(dfdb_stream* s, dfdb_sizestats* st) {
  walk_index(s, INT64_MAX);
  if (s->rows == 0 && s->compressed == 0) {
    for (size_t k = 0; k < s->index.size(); k++) {
      std::lock_guard<std::mutex> lk(s->index[k]->mu);
      const std::vector<BlockLoc>& v = s->index[k]->v;
      const int64_t b1 = s->win_last >= 0 ? std::min<int64_t>(s->win_last, (int64_t)v.size()) : (int64_t)v.size();
      for (int64_t bi = s->win_first; bi < b1; bi++) { const BlockLoc& b = v[(size_t)bi]; s->compressed += b.compressed + 24; s->uncompressed += b.origin; if (k == 0) s->rows += b.rows; }
    }
  }
  st->rows = s->rows; st->compressed = s->compressed; st->uncompressed = s->uncompressed;
}

// what the loaders have read so far of table column `ordinal` (-1: of every required column): the rows of the blocks read, their compressed bytes
// (+ 24 per block: quirk Q10, like dfdb_stream_stats) and their decoded bytes.  A column the query does not need reads nothing.
void stream_read_stats(dfdb_stream* s, int32_t ordinal, dfdb_sizestats* st) {
  std::lock_guard<std::mutex> lk(s->mu);
  *st = dfdb_sizestats{0, 0, 0};
  for (size_t k = 0; k < s->required.size() && k < s->read_stats.size(); k++) {
    if (ordinal >= 0 && s->required[k] != ordinal) continue;
    st->rows += s->read_stats[k].rows; st->compressed += s->read_stats[k].compressed; st->uncompressed += s->read_stats[k].uncompressed;
  }
}

}  // namespace dfdb

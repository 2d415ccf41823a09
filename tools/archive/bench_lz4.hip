// bench_lz4.hip — standalone harness for the K7 LZ4 block decoders: N blocks of the benchmark's Int64 column compressed by the
// system liblz4 (dlopen), decoded by each variant, verified byte for byte, timed with HIP events; the v5 kernel is compiled with
// cycle probes (DFDB_LZ4_PROF) and prints where block 0 spent its cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDFDB_LZ4_PROF -Idataframedbs.jl_amd/csrc tools/bench_lz4.hip -o tools/bench_lz4 -ldl
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../dataframedbs.jl_amd/csrc/k_decode.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace dfdb;

int main(int argc, char** argv) {
  const int nblocks = argc > 1 ? atoi(argv[1]) : 4096;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const int use_index = argc > 4 ? atoi(argv[4]) : 0;    // 1: the first launch records the sequence-start index, the following ones decode with it (k_decode.hip INDEX)
  const int pipe = argc > 3 ? atoi(argv[3]) : -1;        // -1 by block count (default), 0 one wave per block, 1 two-wave pipeline.  mode 0: h mod 1e6 (benchmark column), 1: 1..n, 2: h mod 1000, 3: random doubles, 4: zeros, 5: runs + noise, 6: string body
  void* h = dlopen("liblz4.so.1", RTLD_NOW);
  if (!h) { printf("no liblz4.so.1\n"); return 1; }
  auto compress = (int (*)(const char*, char*, int, int))dlsym(h, "LZ4_compress_default");
  const int rows = 65536, body = rows * 8;
  const int distinct = nblocks < 64 ? nblocks : 64;      // compress 64 different blocks, repeat them
  std::vector<std::vector<uint8_t>> comp((size_t)distinct);
  std::vector<int64_t> plain((size_t)distinct * rows);
  for (int b = 0; b < distinct; b++) {
    int64_t* v = plain.data() + (size_t)b * rows;
    for (int i = 0; i < rows; i++) {
      const uint64_t r = splitmix64(0x9E3779B97F4A7C15ull + (uint64_t)b * rows + (uint64_t)i);
      v[i] = mode == 0 ? (int64_t)(r % 1000000ull) : mode == 1 ? (int64_t)b * rows + i + 1 : mode == 2 ? (int64_t)(r % 1000ull) : 0;
      if (mode == 3) { const double x = (double)(r >> 11) * 0x1.0p-53 * 2000.0; memcpy(&v[i], &x, 8); }          // incompressible
      if (mode == 5) v[i] = (int64_t)((r % 10ull) < 3 ? r % 1000000ull : (uint64_t)i / 7);                           // runs of equal values between noisy ones
    }
    if (mode == 6) {   // a String block body: Int32 datasize, Int32 sizes, bytes of 10 brand names
      static const char* brands[10] = {"apple", "samsung", "huawei", "microsoft", "dell", "xbox", "sony", "intel", "lenovo", "asus"};
      uint8_t* p8 = (uint8_t*)v; int32_t* sz = (int32_t*)p8; const int nr = 52000; size_t o = 4 + 4 * (size_t)nr;
      for (int i = 0; i < nr; i++) { const char* w = brands[splitmix64((uint64_t)b * 65536 + i) % 10]; const int L = (int)strlen(w); sz[1 + i] = L; memcpy(p8 + o, w, L); o += L; }
      sz[0] = (int32_t)(o - 4 - 4 * (size_t)nr);
      memset(p8 + o, 0, body - o);
    }
    comp[(size_t)b].resize((size_t)body + body / 255 + 64);
    const int n = compress((const char*)v, (char*)comp[(size_t)b].data(), body, (int)comp[(size_t)b].size());
    comp[(size_t)b].resize((size_t)n);
    if (mode == 7) {     // a compressed block from a file (tools/dbg_index.py): int32 origin (must be 524288), compressed bytes
      FILE* f = fopen(getenv("LZ4_BLOCK_FILE"), "rb"); if (!f) { printf("no LZ4_BLOCK_FILE\n"); return 1; }
      int32_t origin = 0; fread(&origin, 4, 1, f);
      std::vector<uint8_t> cb((size_t)body + 4096); const size_t got = fread(cb.data(), 1, cb.size(), f); fclose(f); cb.resize(got);
      auto dec = (int (*)(const char*, char*, int, int))dlsym(h, "LZ4_decompress_safe");
      memset(v, 0, body);
      const int dn = dec((const char*)cb.data(), (char*)v, (int)got, body);
      if (b == 0) printf("block file: origin %d, compressed %zu, liblz4 decodes %d bytes\n", origin, got, dn);
      comp[(size_t)b] = cb;
    }
  }
  std::vector<Lz4Block> blk((size_t)nblocks);
  std::vector<uint8_t> img;
  for (int b = 0; b < nblocks; b++) {
    const auto& c = comp[(size_t)(b % distinct)];
    while (img.size() % 8) img.push_back(0);
    blk[(size_t)b] = Lz4Block{(int64_t)img.size(), (int32_t)c.size(), mode == 7 ? atoi(getenv("LZ4_BLOCK_ORIGIN")) : body, (int64_t)b * body};
    img.insert(img.end(), c.begin(), c.end());
  }
  img.resize(img.size() + 64);
  printf("blocks %d, compressed %.1f MB, ratio %.3f\n", nblocks, img.size() / 1e6, (double)nblocks * body / img.size());
  uint8_t *dsrc, *ddst; Lz4Block* dblk; int32_t* dstat;
  CK(hipMalloc(&dsrc, img.size())); CK(hipMalloc(&ddst, (size_t)nblocks * body + 64)); CK(hipMalloc(&dblk, blk.size() * sizeof(Lz4Block))); CK(hipMalloc(&dstat, nblocks * 4));
  CK(hipMemcpy(dsrc, img.data(), img.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dblk, blk.data(), blk.size() * sizeof(Lz4Block), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<uint8_t> back((size_t)distinct * body);
  std::vector<int32_t> stat((size_t)nblocks);
  uint32_t* dindex = nullptr;
  if (use_index) { CK(hipMalloc(&dindex, img.size() / 8 + 1024)); CK(hipMemset(dindex, 0, img.size() / 8 + 1024)); }
  int launch_no = 0;
  for (int variant : {4, 4, 4}) {
    CK(hipMemset(ddst, 0xAB, (size_t)nblocks * body));
    const int imode = use_index ? (launch_no == 0 ? 1 : 2) : 0; launch_no++;
    if (use_index) printf("index mode %d: ", imode);
    CK(hipEventRecord(e0, nullptr));
    launch_lz4_decode(nullptr, dsrc, ddst, dblk, nblocks, dstat, pipe, dindex, imode);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(stat.data(), dstat, nblocks * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(back.data(), ddst + (size_t)(nblocks - distinct) * body, back.size(), hipMemcpyDeviceToHost));
    int bad = 0; for (int b = 0; b < nblocks; b++) bad += stat[(size_t)b] != 0;
    int wrong = 0;
    for (int k = 0; k < distinct; k++) {
      const int b = nblocks - distinct + k;
      wrong += memcmp(back.data() + (size_t)k * body, plain.data() + (size_t)(b % distinct) * rows, (size_t)blk[(size_t)b].dst_len) != 0;
    }
    printf("variant %d: %.3f ms  %.1f GB/s out  status!=0: %d  wrong blocks: %d\n", variant, ms, (double)nblocks * body / ms / 1e6, bad, wrong);
#ifdef DFDB_LZ4_PROF
    if (variant >= 4) {
      unsigned long long pf[32];
      CK(hipMemcpyFromSymbol(pf, HIP_SYMBOL(g_lz4_prof), sizeof(pf)));
      printf("  block 0 cycles: total %llu | candidates %llu  walk+partial %llu  dense %llu  far %llu  resolve %llu  flush %llu  other %llu | superbatches %llu chunks %llu rounds %llu seqs %llu far %llu\n",
             pf[15], pf[0], pf[11], pf[12], pf[5], pf[2], pf[3], pf[4], pf[6], pf[7], pf[8], pf[9], pf[10]);
      printf("    walk: hops %llu  per-window rest %llu | production: ordinals+gather %llu  addresses %llu  pointer rounds %llu  bytes %llu\n", pf[13], pf[14], pf[16], pf[17], pf[18], pf[19]);
    }
#endif
  }
  if (use_index && getenv("LZ4_DUMP_INDEX")) {
    const auto& c = comp[0];
    std::vector<uint8_t> ix(img.size() / 8 + 1024);
    CK(hipMemcpy(ix.data(), dindex, ix.size(), hipMemcpyDeviceToHost));
    std::vector<uint8_t> truth(c.size() + 8, 0);
    size_t ip = 0;
    while (ip < c.size()) {
      truth[ip] = 1;
      const uint8_t tok = c[ip++]; size_t lit = tok >> 4;
      if (lit == 15) { uint8_t bb; do { bb = c[ip++]; lit += bb; } while (bb == 255); }
      ip += lit; if (ip >= c.size()) break;
      ip += 2; size_t ml = tok & 15; if (ml == 15) { uint8_t bb; do { bb = c[ip++]; ml += bb; } while (bb == 255); }
    }
    int diffs = 0;
    for (size_t k = 0; k < c.size(); k++) {
      const size_t bit = (size_t)blk[0].src_off + k;
      const int g = (ix[bit >> 3] >> (bit & 7)) & 1;
      if (g != truth[k] && diffs++ < 20) printf("index differs at input %zu: gpu %d truth %d\n", k, g, (int)truth[k]);
    }
    printf("index of block 0: %d differences over %zu input bytes\n", diffs, c.size());
  }
  return 0;
}

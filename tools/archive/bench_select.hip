// bench_select.hip — A/B of selection(x -> x > c) -> row indices on one MI355X: the three-launch form (K1 k_scan_cmp, count scan, K2
// k_compact_indices) against the single-pass K1S k_scan_select (decoupled look-back), interleaved rounds in one process, outputs compared
// word for word (bitmap, tile prefix, indices).  Links the shipped library's own launchers for the three-launch form; the single-pass kernel
// lives in tools/k_select_experiment.hip (it lost: profiles/r2_single_pass_select.txt).  -DDFDB_SELECT_STATS adds per-wave cycle totals and, with
// SELECT_DBG=1, publication / look-back timestamps per quad; -DDFDB_SELECT_NOLOOK + SELECT_NOLOOK=1 hands the kernel the finished prefix (upper bound).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idataframedbs.jl_amd/csrc tools/bench_select.hip -o tools/bench_select \
//        -Ldataframedbs.jl_amd -ldfdb_hip -Wl,-rpath,'$ORIGIN/../dataframedbs.jl_amd'
// Run:   tools/bench_select [rows=1000000000] [permille selected=100] [rounds=10]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "k_select_experiment.hip"      // the single-pass kernel under test (not in the library)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace dfdb;

__global__ void k_gen(int64_t* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (int64_t)(splitmix64(0x9E3779B97F4A7C15ull + (uint64_t)i) % 1000000ull);
}
__global__ void k_diff(const uint64_t* a, const uint64_t* b, int64_t n, unsigned long long* bad) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += a[i] != b[i];
  if (c) atomicAdd(bad, c);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000000ll;
  const int permille = argc > 2 ? atoi(argv[2]) : 100;
  const int rounds = argc > 3 ? atoi(argv[3]) : 10;
  const int64_t c = 1000000ll - 1 - (int64_t)permille * 1000;       // x > c keeps permille/1000 of the rows
  const int64_t ntiles = (n + 1023) / 1024, words = ((n + 4095) / 4096) * 64;
  hipStream_t s; CK(hipStreamCreate(&s));
  int64_t* col; CK(hipMalloc(&col, (size_t)n * 8));
  hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, s, col, n);
  uint64_t *bmA, *bmB, *pfA, *pfB, *scratch, *state; uint32_t *tcA, *tcB; int64_t *outA, *outB; unsigned long long* bad;
  CK(hipMalloc(&bmA, words * 8)); CK(hipMalloc(&bmB, words * 8));
  CK(hipMalloc(&pfA, (ntiles + 1) * 8)); CK(hipMalloc(&pfB, (ntiles + 1) * 8));
  CK(hipMalloc(&tcA, (ntiles + 8) * 4)); CK(hipMalloc(&tcB, (ntiles + 8) * 4));
  CK(hipMalloc(&scratch, scan_counts_scratch_bytes(ntiles))); CK(hipMalloc(&state, scan_select_state_bytes(n)));
  CK(hipMalloc(&bad, 8));
  CK(hipMemsetAsync(bmA, 0, words * 8, s)); CK(hipMemsetAsync(bmB, 0, words * 8, s));
  // count first (sizes the outputs)
  launch_scan_cmp(s, col, DFDB_I64, CMP_GT, (uint64_t)c, bmA, tcA, n, false, true, nullptr);
  launch_scan_counts(s, tcA, pfA, ntiles, scratch);
  uint64_t total = 0; CK(hipMemcpyAsync(&total, pfA + ntiles, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
  printf("rows %lld  selected %llu (%.4f)\n", (long long)n, (unsigned long long)total, (double)total / (double)n);
  const int64_t cap = (int64_t)total + 64;
  CK(hipMalloc(&outA, (size_t)cap * 8)); CK(hipMalloc(&outB, (size_t)cap * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto three = [&] {
    launch_scan_cmp(s, col, DFDB_I64, CMP_GT, (uint64_t)c, bmA, tcA, n, false, true, nullptr);
    launch_scan_counts(s, tcA, pfA, ntiles, scratch);
    launch_compact_indices(s, bmA, pfA, outA, n, 0, cap);
  };
  auto one = [&](int emit) {
    set_select_emit(emit);
    launch_scan_select(s, col, DFDB_I64, CMP_GT, (uint64_t)c, bmB, tcB, getenv("SELECT_NOLOOK") ? pfA : pfB, outB, cap, 0, n, state, 1);
  };
  auto timed = [&](auto&& f) { CK(hipEventRecord(e0, s)); f(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
  // correctness of both emit forms against the three-launch form
  three();
  for (int emit = 0; emit < 2; emit++) {
    CK(hipMemsetAsync(outB, 0xff, (size_t)cap * 8, s)); CK(hipMemsetAsync(pfB, 0xff, (ntiles + 1) * 8, s)); CK(hipMemsetAsync(bmB, 0, words * 8, s));
    one(emit);
    CK(hipMemsetAsync(bad, 0, 8, s));
    hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, s, (const uint64_t*)outA, (const uint64_t*)outB, (int64_t)total, bad);
    hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, s, (const uint64_t*)pfA, (const uint64_t*)pfB, ntiles + 1, bad);
    hipLaunchKernelGGL(k_diff, dim3(1024), dim3(256), 0, s, (const uint64_t*)bmA, (const uint64_t*)bmB, words, bad);
    unsigned long long nb = 0; CK(hipMemcpyAsync(&nb, bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    printf("emit %d: mismatching words (indices + prefix + bitmap): %llu\n", emit, nb);
  }
  std::vector<float> t3, t0, t1, tk1;
  for (int r = 0; r < rounds; r++) {
    t3.push_back(timed(three));
    t0.push_back(timed([&] { one(0); }));
    t1.push_back(timed([&] { one(1); }));
    tk1.push_back(timed([&] { launch_scan_cmp(s, col, DFDB_I64, CMP_GT, (uint64_t)c, bmA, tcA, n, false, true, nullptr); }));
  }
  for (int emit = 0; emit < 2; emit++) {     // DFDB_SELECT_STATS builds: per-wave cycle totals left in the state words
    one(emit); unsigned long long st[16]; CK(hipMemcpyAsync(st, state, sizeof st, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    if (st[8]) printf("emit %d stats: waves %llu  look-back steps %llu (spinning %llu)  cycles per wave: total %.0f  look-back %.0f  ticket+barrier %.0f  prefix+emit %.0f\n", emit, st[8], st[3], st[2],
                      (double)st[4] / st[8], (double)st[5] / st[8], (double)st[6] / st[8], (double)st[7] / st[8]);
  }
  if (getenv("SELECT_DBG")) {   // DFDB_SELECT_STATS builds: when did each quad publish, whom did its look-back wait for, for how long
    one(1); CK(hipStreamSynchronize(s));
    const int64_t ngroups = (ntiles + 3) / 4, nq = (ngroups + 3) / 4, nqp = nq + 4;
    std::vector<uint64_t> h((size_t)(16 + 5 * nqp)); CK(hipMemcpy(h.data(), state, h.size() * 8, hipMemcpyDeviceToHost));
    const uint64_t *pub = h.data() + 16 + nqp, *wait = pub + nqp, *spin0 = wait + nqp, *done = spin0 + nqp;
    uint64_t tmin = ~0ull; for (int64_t i = 0; i < nq; i++) if (pub[i] && pub[i] < tmin) tmin = pub[i];
    long long nwait = 0; double sum_wait = 0, sum_late = 0, sum_vis = 0, sum_dist = 0; long long late = 0;
    for (int64_t i = 0; i < nq; i++) if (spin0[i]) {
      nwait++; sum_wait += (double)(done[i] - spin0[i]); const uint64_t w = wait[i]; sum_dist += (double)(i - (int64_t)w);
      if (pub[w] > spin0[i]) { late++; sum_late += (double)(pub[w] - spin0[i]); } else sum_vis += (double)(spin0[i] - pub[w]);
    }
    printf("quads %lld, look-backs that spun %lld: mean wait %.1f us, mean distance to the awaited quad %.1f; awaited quad published AFTER the first look %lld times (mean %.1f us later), BEFORE it %lld times (mean %.1f us earlier: visibility lag)\n",
           (long long)nq, nwait, sum_wait / (nwait ? nwait : 1) / 100.0, sum_dist / (nwait ? nwait : 1), late, sum_late / (late ? late : 1) / 100.0, nwait - late, sum_vis / (nwait - late ? nwait - late : 1) / 100.0);
    for (int64_t i = 0; i < nq; i += nq / 24) printf("  quad %lld: published at %.1f us, look-back done at %.1f us%s\n", (long long)i, (pub[i] - tmin) / 100.0, (done[i] - tmin) / 100.0, spin0[i] ? " (spun)" : "");
  }
  auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mn = [](std::vector<float> v) { return *std::min_element(v.begin(), v.end()); };
  const double bytes = (double)n * (8 + 1.0 / 8 + 12.0 / 1024 + 8.0 / 16384) + 8.0 * (double)total;
  printf("three launches   median %.4f ms  min %.4f   (K1 alone median %.4f)\n", med(t3), mn(t3), med(tk1));
  printf("one pass, emit 0 median %.4f ms  min %.4f   %.0f GB/s algorithmic\n", med(t0), mn(t0), bytes / (med(t0) * 1e-3) / 1e9);
  printf("one pass, emit 1 median %.4f ms  min %.4f   %.0f GB/s algorithmic\n", med(t1), mn(t1), bytes / (med(t1) * 1e-3) / 1e9);
  return 0;
}

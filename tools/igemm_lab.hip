// Developer lab for the implicit-GEMM convolution kernel: times one geometry with HIP events and, built with
// -DMMSA_LAB_STAMPS, prints where a workgroup's cycles go (prologue / main loop / epilogue), the clock the chip
// held, and how the workgroups' start and end times spread.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMMSA_LAB_STAMPS tools/igemm_lab.hip -o tools/igemm_lab
//   tools/igemm_lab B H W Cin Cout ntaps [stats]
#include "../multimodal-sentiment-analysis_amd/csrc/conv.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                           \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                         \
    }                                                                   \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 7) return 1;
  const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), Cin = atoi(argv[4]), Cout = atoi(argv[5]);
  const int nt = atoi(argv[6]);
  const bool want_stats = argc > 7;
  int dh[9], dw[9];
  for (int i = 0; i < nt; ++i) {
    dh[i] = nt == 9 ? i / 3 - 1 : 0;
    dw[i] = nt == 9 ? i % 3 - 1 : (nt == 1 ? 0 : i);
  }
  const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)nt * Cin * Cout;
  std::vector<float> hx(nx), hw(nw);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.05f;
  float *X, *Wp, *Y, *stats;
  CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&Wp, nw * 4)); CK(hipMalloc(&Y, ny * 4));
  const int bm = mmsa_conv_tile_rows(B, H, W, Cout);
  const int tiles = (H * W + bm - 1) / bm;
  CK(hipMalloc(&stats, (size_t)B * tiles * 2 * Cout * 4));
  CK(hipMemcpy(X, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Wp, hw.data(), nw * 4, hipMemcpyHostToDevice));
  const long nwg = (long)B * tiles * (Cout / 64 > 0 ? Cout / 64 : 1) * 2;   // upper bound
  unsigned long long* stamps = nullptr;
#ifdef MMSA_LAB_STAMPS
  CK(hipMalloc(&stamps, nwg * 16 * sizeof(unsigned long long)));
  CK(hipMemset(stamps, 0, nwg * 16 * sizeof(unsigned long long)));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_stamps), &stamps, sizeof(stamps)));
#endif
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto run = [&]() { return mmsa_conv_fwd(X, Wp, Y, want_stats ? stats : nullptr, B, H, W, Cin, Cout, nt, dh, dw, 0, st); };
  for (int i = 0; i < 200; ++i)
    if (run() != 0) { printf("launch failed\n"); return 1; }
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) run();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * B * H * W * Cin * Cout * nt;
  printf("B=%d H=%d W=%d Cin=%d Cout=%d taps=%d tile=%d: %.1f us  %.1f TFLOP/s\n", B, H, W, Cin, Cout, nt, bm, us,
         fl / us / 1e6);
#ifdef MMSA_LAB_STAMPS
  std::vector<unsigned long long> h(nwg * 16);
  CK(hipMemcpy(h.data(), stamps, nwg * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> pro, loop, epi, tot, clk, l_plan, l_acc, l_issue, l_wait;
  unsigned long long r0 = ~0ull, r1 = 0;
  std::vector<unsigned long long> starts, ends;
  for (long i = 0; i < nwg; ++i) {
    const unsigned long long* p = &h[i * 16];
    if (!p[3]) continue;
    pro.push_back((double)(p[1] - p[0])); loop.push_back((double)(p[2] - p[1])); epi.push_back((double)(p[3] - p[2]));
    tot.push_back((double)(p[3] - p[0]));
    if (p[11]) {
      l_plan.push_back((double)(p[8] - p[0])); l_acc.push_back((double)(p[9] - p[8]));
      l_issue.push_back((double)(p[10] - p[9])); l_wait.push_back((double)(p[11] - p[10]));
    }
    const double rt = (double)(p[7] - p[4]);       // 100 MHz ticks
    if (rt > 0) clk.push_back((double)(p[3] - p[0]) / rt * 100.0);
    r0 = std::min(r0, p[4]); r1 = std::max(r1, p[7]);
    starts.push_back(p[4]); ends.push_back(p[7]);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  auto mx = [](std::vector<double> v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
  printf("workgroups %zu: cycles median prologue %.0f, loop %.0f, epilogue %.0f, total %.0f (max %.0f); clock %.0f MHz\n",
         tot.size(), med(pro), med(loop), med(epi), med(tot), mx(tot), med(clk));
  printf("loader wave: cycles median from workgroup start to plan done %.0f, to first request %.0f, issuing the prologue's requests %.0f, waiting for them %.0f\n",
         med(l_plan), med(l_acc), med(l_issue), med(l_wait));
  std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
  const size_t n = starts.size();
  if (n) {
    printf("kernel span %.2f us; starts: median +%.2f us, 90%% +%.2f us, last +%.2f us; ends: first +%.2f us, median +%.2f us\n",
           (r1 - r0) / 100.0, (starts[n / 2] - r0) / 100.0, (starts[n * 9 / 10] - r0) / 100.0, (starts[n - 1] - r0) / 100.0,
           (ends[0] - r0) / 100.0, (ends[n / 2] - r0) / 100.0);
  }
#endif
  return 0;
}

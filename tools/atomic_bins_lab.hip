// Lab: what does it cost a convolution's epilogue to add its per-channel sums into shared 64-bit integer bins
// (fixed point: integer adds commute, so the result does not depend on arrival order) instead of storing one row
// of partial sums per workgroup?   hipcc --offload-arch=gfx950 -O3 tools/atomic_bins_lab.hip -o tools/ab/atomic_bins_lab
// Every workgroup idles for a pseudo-random time within `spread` microseconds, then 128 threads store / add 64
// channels x {sum, sum of squares} (+ 64 per-image sums with mode 2).  Kernel time by events over 50 launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void epilogue(float* partial, unsigned long long* bins, unsigned long long* img, int mode,
                                                int replicas, int tiles_per_img, int spread_ticks) {
  const int t = threadIdx.x;
  // spread the arrivals
  if (spread_ticks > 0) {
    const unsigned h = (blockIdx.x * 2654435761u) >> 8;
    const long long until = wall_clock64() + (long long)(h % (unsigned)spread_ticks);
    while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
  }
  const float v = 1.0f + 0.001f * t;
  if (t < 128) {
    const int which = t >> 6, col = t & 63;
    if (mode == 0) {
      partial[((long)which * 64 + col) * gridDim.x + blockIdx.x] = v;
    } else {
      const long long q = __float2ll_rn(v * 4294967296.0f);
      const int r = blockIdx.x % replicas;
      atomicAdd(&bins[((long)r * 2 + which) * 64 + col], (unsigned long long)q);
      if (mode == 2 && which == 0) atomicAdd(&img[(long)(blockIdx.x / tiles_per_img) * 64 + col], (unsigned long long)q);
    }
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 1056;
  float* partial;
  unsigned long long *bins, *img;
  hipMalloc(&partial, sizeof(float) * 128 * G);
  hipMalloc(&bins, 8 * 128 * 64);
  hipMalloc(&img, 8 * 64 * 4096);
  hipMemset(bins, 0, 8 * 128 * 64);
  hipMemset(img, 0, 8 * 64 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int spread_us : {0, 5, 20}) {
    for (int mode = 0; mode < 3; ++mode)
      for (int rep : {1, 8}) {
        if (mode == 0 && rep != 1) continue;
        for (int i = 0; i < 5; ++i) epilogue<<<G, 256>>>(partial, bins, img, mode, rep, 33, spread_us * 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) epilogue<<<G, 256>>>(partial, bins, img, mode, rep, 33, spread_us * 100);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("G=%d spread=%2d us mode=%d (%s) replicas=%d: %.2f us per launch\n", G, spread_us, mode,
               mode == 0 ? "rows of partial sums" : mode == 1 ? "64-bit adds, 2 x 64 bins" : "adds + per-image bins",
               rep, ms * 1000.f / 50);
      }
  }
  unsigned long long h[2];
  hipMemcpy(h, bins, 16, hipMemcpyDeviceToHost);
  printf("bin[0] = %llu\n", h[0]);
  return 0;
}

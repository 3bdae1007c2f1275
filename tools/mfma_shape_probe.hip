// Developer probe: which fp32 MFMA shape delivers more FLOP/s by the wall clock when the chip sets its own clock
// under load?  Bare loops on random operands held in registers, the same 32 x 32 output tile per wave:
//   shape 0: v_mfma_f32_32x32x2_f32, one accumulator, two instructions per K = 4
//   shape 1: v_mfma_f32_16x16x4_f32, 2 x 2 accumulators, four instructions per K = 4
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shape_probe.hip -o tools/mfma_shape_probe
//   tools/mfma_shape_probe [waves per SIMD: 1 or 2]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ in, float* __restrict__ out, int iters,
                                             unsigned long long* clk) {
  const int t = threadIdx.x + blockIdx.x * blockDim.x;
  float a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = in[(t * 16 + k) & 0xfffff];
    b[k] = in[(t * 16 + 8 + k) & 0xfffff];
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  if (SHAPE == 0) {
    f32x16 acc = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[r];
  } else {
    f32x4 acc[2][2] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // K = 4 per instruction: 4 x (2 x 2) instructions = the 8 of the other shape x 2
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * k], b[2 * k], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * k], b[2 * k + 1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * k + 1], b[2 * k], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * k + 1], b[2 * k + 1], acc[1][1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[t] = sum;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = c1 - c0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const int threads = 256 * wps, blocks = 256;
  std::vector<float> h(1 << 20);
  unsigned s = 777;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float *in, *out;
  unsigned long long* clk;
  hipMalloc(&in, h.size() * 4);
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipMalloc(&clk, blocks * 16);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const int iters = 40000;     // x 16 K-steps of 4 x 2 x 32 x 32 flop
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int shape = 0; shape < 2; ++shape) {
    for (int rep = 0; rep < 3; ++rep) {
      const int n = 8;
      hipEventRecord(e0);
      for (int i = 0; i < n; ++i) {
        if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, clk);
        else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, clk);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> hc(blocks * 2);
      hipMemcpy(hc.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
      const double ghz = (double)hc[0] / (double)hc[1] * 0.1;
      const double flop = (double)n * blocks * (threads / 64) * iters * 8.0 * 2 * 32 * 32 * 2;
      printf("shape %s, %d wave(s) per SIMD: %.1f ms, %.1f TFLOP/s, in-kernel clock %.2f GHz, %.1f cycles per 4096 flop\n",
             shape == 0 ? "32x32x2" : "16x16x4", wps, ms, flop / ms / 1e9, ghz,
             (double)hc[0] / (iters * 8.0) / (wps));
    }
  }
  return 0;
}

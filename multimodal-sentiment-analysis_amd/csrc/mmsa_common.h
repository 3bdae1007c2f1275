// Shared device/host helpers for the MI355X (gfx950) kernels of this library.
// Wavefronts are 64 lanes wide; every block size used here is a multiple of 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMSA_OK 0
#define MMSA_ERR_ARG (-1)
#define MMSA_ERR_LAUNCH (-2)
#define MMSA_ERR_LIMIT (-3)

#define MMSA_EXPORT extern "C" __attribute__((visibility("default")))

#define MMSA_REQUIRE(cond)                 \
  do {                                     \
    if (!(cond)) return MMSA_ERR_ARG;      \
  } while (0)

static inline int mmsa_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMSA_OK : MMSA_ERR_LAUNCH;
}

static inline hipStream_t mmsa_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int mmsa_cdiv(long a, long b) { return static_cast<int>((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// counter-based uniform in [0,1): one 64-bit mix per element (dropout masks; not comparable with torch's RNG)
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// ---- wave / block reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over a whole block (blockDim.x multiple of 64, <= 1024). `red` is >= 16 floats of LDS.
// Every thread receives the result. Contains barriers: call from uniform control flow.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

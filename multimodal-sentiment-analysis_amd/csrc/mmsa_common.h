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

// A call that has to be made once per device (raising a kernel's dynamic-LDS limit belongs to the device that is
// current when it is made): pending() is true until done() has run with that device current.  A process that drives
// several GPUs gets the call on each of them.  Races are benign, the guarded calls are idempotent.
struct PerDeviceOnce {
  unsigned long long seen = 0;
  static int current() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : -1;
  }
  bool pending() const {
    const int d = current();
    return d < 0 || !((seen >> d) & 1ull);
  }
  void done() {
    const int d = current();
    if (d >= 0) seen |= 1ull << d;
  }
};

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
// Butterflies on the vector ALU's own cross-lane path (DPP row operations, then the two half-swaps of gfx950), not
// through the LDS crossbar: a ds_bpermute round trip is some 130 cycles and a 64-lane sum chains six of them (twelve
// for a double), which was half the run time of the one-workgroup-per-channel finishing kernels and of every
// LayerNorm stage of the fusion head.  Every lane ends with the same bits.  Call with all 64 lanes enabled.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int DPP_XOR1 = 0xB1;      // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;      // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;   // lane i <-> 7 - i of each 8: the other quad, once the quads are uniform
constexpr int DPP_MIRROR = 0x140;        // lane i <-> 15 - i of each 16: the other half row
// sum over each 16-lane row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<DPP_XOR1>(v);
  v += dpp_mov<DPP_XOR2>(v);
  v += dpp_mov<DPP_HALF_MIRROR>(v);
  v += dpp_mov<DPP_MIRROR>(v);
  return v;
}
// The half-swaps exchange the odd rows (upper half) of one register with the even rows (lower half) of another; of
// two copies of v they leave one holding the even-side and one the odd-side value in every lane.
__device__ __forceinline__ void swap16_pair(float v, float& a, float& b) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap32_pair(float v, float& a, float& b) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum(float v) {
  float a, b;
  v = row16_sum(v);
  swap16_pair(v, a, b);
  v = a + b;
  swap32_pair(v, a, b);
  return a + b;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  v += dpp_mov<DPP_XOR1>(v);
  v += dpp_mov<DPP_XOR2>(v);
  v += dpp_mov<DPP_HALF_MIRROR>(v);
  v += dpp_mov<DPP_MIRROR>(v);
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)bits, hi = (unsigned)(bits >> 32);
    const u32x2 rl = step ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false)
                          : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u32x2 rh = step ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false)
                          : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __longlong_as_double((long long)(((unsigned long long)rh[0] << 32) | rl[0])) +
        __longlong_as_double((long long)(((unsigned long long)rh[1] << 32) | rl[1]));
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  float a, b;
  v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
  swap16_pair(v, a, b);
  v = fmaxf(a, b);
  swap32_pair(v, a, b);
  return fmaxf(a, b);
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

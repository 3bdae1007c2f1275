// 512-point complex FFT for one 64-lane wavefront per transform: radix-8 Stockham, three passes through LDS with an
// 8-point DFT in registers.  A 1024-point real transform (forward or inverse) packs into one of these.
#pragma once

#include "mmsa_common.h"

// 8-point DFT in registers (outputs in natural order).
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
  auto add = [](float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); };
  auto sub = [](float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); };
  auto mul_mi = [](float2 a) { return make_float2(a.y, -a.x); };   // times -i
  const float2 a0 = add(v[0], v[4]), a4 = sub(v[0], v[4]), a2 = add(v[2], v[6]), a6 = mul_mi(sub(v[2], v[6]));
  const float2 a1 = add(v[1], v[5]), a5 = sub(v[1], v[5]), a3 = add(v[3], v[7]), a7 = mul_mi(sub(v[3], v[7]));
  const float2 b0 = add(a0, a2), b2 = sub(a0, a2), b4 = add(a4, a6), b6 = sub(a4, a6);
  const float2 b1 = add(a1, a3), b3 = mul_mi(sub(a1, a3)), b5 = add(a5, a7), b7 = sub(a5, a7);
  constexpr float h = 0.70710678118654752440f;
  const float2 w1b5 = make_float2(h * (b5.x + b5.y), h * (b5.y - b5.x));      // e^{-i pi/4} b5
  const float2 w3b7 = make_float2(h * (b7.y - b7.x), -h * (b7.x + b7.y));     // e^{-3i pi/4} b7
  v[0] = add(b0, b1); v[4] = sub(b0, b1);
  v[2] = add(b2, b3); v[6] = sub(b2, b3);
  v[1] = add(b4, w1b5); v[5] = sub(b4, w1b5);
  v[3] = add(b6, w3b7); v[7] = sub(b6, w3b7);
}

constexpr int FFT512_LD = 576;   // padded row length of the two LDS work buffers

// Forward transform of the 512 values held as v[r] = z[j + 64 r] by lane j of the wavefront that owns row `fa` / `fb`.
// On return fa[n] = Z[n], n = 0..511, in natural order.  `tws[m]` = e^{-2 pi i m / 512}.  Contains block barriers:
// every wavefront of the workgroup must call it together.
__device__ __forceinline__ void fft512(float2 (&v)[8], float2* __restrict__ fa, float2* __restrict__ fb,
                                       const float2* __restrict__ tws, int j) {
  dft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) fa[j * 9 + r] = v[r];            // element i lives at i + i / 8
  __syncthreads();
  {
    const int k = j & 7;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float2 x = fa[j + (j >> 3) + 72 * r];
      const float2 tw = tws[r * k * 8];
      v[r] = make_float2(x.x * tw.x - x.y * tw.y, x.x * tw.y + x.y * tw.x);
    }
    dft8(v);
    const int j0 = (j >> 3) * 72 + k;                           // element i lives at i + 8 * (i / 64)
#pragma unroll
    for (int r = 0; r < 8; ++r) fb[j0 + 8 * r] = v[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float2 x = fb[j + 72 * r];
    const float2 tw = tws[r * j];
    v[r] = make_float2(x.x * tw.x - x.y * tw.y, x.x * tw.y + x.y * tw.x);
  }
  dft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) fa[j + 64 * r] = v[r];
  __syncthreads();
}

// The same transform for a wavefront on its own: one LDS row, used in place, and no workgroup barrier.  A wavefront's
// LDS instructions execute in issue order, so every lane's reads of a pass are served before any lane's writes of
// that pass reach the row, and the next pass's reads see all of them; the fences only keep the compiler from
// moving LDS accesses across the pass boundaries.  `twh[m]` = e^{-2 pi i m / 512} for m < 256 (the second half of
// the circle is the negated first half).  On return row[n] = Z[n], n = 0..511, in natural order.
__device__ __forceinline__ float2 fft512_tw(const float2* __restrict__ twh, int m) {
  const float2 w = twh[m & 255];
  return (m & 256) ? make_float2(-w.x, -w.y) : w;
}

__device__ __forceinline__ void fft512_wave(float2 (&v)[8], float2* row, const float2* __restrict__ twh, int j) {
  dft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) row[j * 9 + r] = v[r];            // element i lives at i + i / 8
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  {
    const int k = j & 7;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float2 x = row[j + (j >> 3) + 72 * r];
      const float2 tw = fft512_tw(twh, r * k * 8);
      v[r] = make_float2(x.x * tw.x - x.y * tw.y, x.x * tw.y + x.y * tw.x);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // all reads of the row are issued before the writes
    dft8(v);
    const int j0 = (j >> 3) * 72 + k;                           // element i lives at i + 8 * (i / 64)
#pragma unroll
    for (int r = 0; r < 8; ++r) row[j0 + 8 * r] = v[r];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float2 x = row[j + 72 * r];
    const float2 tw = fft512_tw(twh, r * j);
    v[r] = make_float2(x.x * tw.x - x.y * tw.y, x.x * tw.y + x.y * tw.x);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  dft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) row[j + 64 * r] = v[r];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

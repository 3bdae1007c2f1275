// Bandwidth-bound kernels of the audio branch (everything that is not a dense contraction):
// the 1->32 channel stem, BatchNorm statistics/coefficients, squeeze-excite gates, the fused
// gate * BN + shortcut BN + ReLU + 2x2 max-pool, global average pooling, and all their backward passes.
//
// Reference semantics: `CNN implementation/multimodal_model.py:27-42` (SEBlock), `:45-70` (ResidualBlock),
// `:78-83,118-129` (stem, pools, global pool).  BatchNorm follows torch defaults: eps 1e-5, momentum 0.1,
// biased variance for normalisation, unbiased for the running estimate.
//
// Layout: activations channels-last [B][H][W][C]; BN coefficient tables are [4][C] = mean, 1/sqrt(var+eps),
// scale = gamma*invstd, shift = beta - mean*scale.  Every cross-workgroup reduction is two-stage (per-block
// partials, then a fixed-order sum), so results do not change from run to run.
#include "mmsa_common.h"

namespace {

constexpr int STEM_C = 32;

// ------------------------------------------------------------------------------------------------ stem
// conv value at column w of the row triple staged in `rows` (each row has a zero at index 0 and W+1)
__device__ __forceinline__ float stem_conv(const float* r0, const float* r1, const float* r2, int w, const float* k,
                                           float bias) {
  float a = bias;
  a = fmaf(k[0], r0[w], a); a = fmaf(k[1], r0[w + 1], a); a = fmaf(k[2], r0[w + 2], a);
  a = fmaf(k[3], r1[w], a); a = fmaf(k[4], r1[w + 1], a); a = fmaf(k[5], r1[w + 2], a);
  a = fmaf(k[6], r2[w], a); a = fmaf(k[7], r2[w + 1], a); a = fmaf(k[8], r2[w + 2], a);
  return a;
}

template <int NROWS>
__device__ __forceinline__ void stem_stage_rows(float* rows, int ldw, const float* xb, int h_first, int H, int W) {
  for (int i = threadIdx.x; i < NROWS * ldw; i += blockDim.x) {
    const int r = i / ldw, col = i - r * ldw;
    const int h = h_first + r, w = col - 1;
    float v = 0.f;
    if (h >= 0 && h < H && w >= 0 && w < W) v = xb[h * W + w];
    rows[i] = v;
  }
}

constexpr int STEM_ROWS = 8;   // image rows per workgroup of the statistics kernel (one per group of 32 lanes)

// 3x3 convolution value from a register window (same order of operations as stem_conv)
__device__ __forceinline__ float stem_conv_win(float a0, float a1, float a2, float b0, float b1, float b2, float c0,
                                               float c1, float c2, const float* k, float bias) {
  float a = bias;
  a = fmaf(k[0], a0, a); a = fmaf(k[1], a1, a); a = fmaf(k[2], a2, a);
  a = fmaf(k[3], b0, a); a = fmaf(k[4], b1, a); a = fmaf(k[5], b2, a);
  a = fmaf(k[6], c0, a); a = fmaf(k[7], c1, a); a = fmaf(k[8], c2, a);
  return a;
}

// one block per group of STEM_ROWS input rows of one image: per-channel sum / sum of squares of the conv output.
// Lane group pl (32 lanes = the 32 channels) walks along row h0 + pl with a 3x3 window in registers: three LDS reads
// (broadcast over the group) per nine multiply-adds.
__global__ __launch_bounds__(256) void stem_stats_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                         const float* __restrict__ bias, float* __restrict__ partial,
                                                         int H, int W, int groups_per_img) {
  extern __shared__ float sm[];
  const int ldw = W + 2;
  float* rows = sm;                               // [STEM_ROWS + 2][ldw]
  float* red = sm + (STEM_ROWS + 2) * ldw;        // [8][2][32]
  const int b = blockIdx.x / groups_per_img, h0 = (blockIdx.x - b * groups_per_img) * STEM_ROWS;
  stem_stage_rows<STEM_ROWS + 2>(rows, ldw, x + (long)b * H * W, h0 - 1, H, W);
  const int c = threadIdx.x & 31, pl = threadIdx.x >> 5;
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = wgt[c * 9 + i];
  const float bs = bias[c];
  __syncthreads();
  float s = 0.f, q = 0.f;
  if (h0 + pl < H) {
    const float* r0 = rows + pl * ldw;
    const float* r1 = r0 + ldw;
    const float* r2 = r1 + ldw;
    float a0 = r0[0], a1 = r0[1], b0 = r1[0], b1 = r1[1], c0 = r2[0], c1 = r2[1];
    for (int w = 0; w < W; ++w) {
      const float a2 = r0[w + 2], b2 = r1[w + 2], c2 = r2[w + 2];
      const float v = stem_conv_win(a0, a1, a2, b0, b1, b2, c0, c1, c2, k, bs);
      s += v;
      q = fmaf(v, v, q);
      a0 = a1; a1 = a2; b0 = b1; b1 = b2; c0 = c1; c1 = c2;
    }
  }
  red[(pl * 2 + 0) * 32 + c] = s;
  red[(pl * 2 + 1) * 32 + c] = q;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5;
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += red[(p * 2 + which) * 32 + c];
    partial[((long)which * STEM_C + c) * gridDim.x + blockIdx.x] = t;
  }
}

// one block per pooled row (b,h2): recompute conv, BN, ReLU, 2x2 max -> out[b][h2][w2][32].  Lane group pl takes a
// run of consecutive pooled columns and slides a 4x4 window along the four staged rows: eight LDS reads per pooled
// value (four convolution values).
__global__ __launch_bounds__(256) void stem_apply_pool_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ coeff, float* __restrict__ out,
                                                              int H, int W) {
  extern __shared__ float sm[];
  const int ldw = W + 2, H2 = H / 2, W2 = W / 2;
  const int b = blockIdx.x / H2, h2 = blockIdx.x - b * H2;
  stem_stage_rows<4>(sm, ldw, x + (long)b * H * W, 2 * h2 - 1, H, W);
  const int c = threadIdx.x & 31, pl = threadIdx.x >> 5;
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = wgt[c * 9 + i];
  const float bs = bias[c], sc = coeff[2 * STEM_C + c], sh = coeff[3 * STEM_C + c];
  __syncthreads();
  const int per = (W2 + 7) / 8;
  const int w_first = pl * per, w_last = min(W2, w_first + per);
  if (w_first >= w_last) return;
  float win[4][4];   // rows 0..3 of the staged block, columns 2*w2 .. 2*w2 + 3 (staged column index)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    win[r][0] = sm[r * ldw + 2 * w_first];
    win[r][1] = sm[r * ldw + 2 * w_first + 1];
  }
  float* o = out + (((long)b * H2 + h2) * W2) * STEM_C + c;
  for (int w2 = w_first; w2 < w_last; ++w2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][2] = sm[r * ldw + 2 * w2 + 2];
      win[r][3] = sm[r * ldw + 2 * w2 + 3];
    }
    float m = 0.f;  // ReLU folded into the max
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v = stem_conv_win(win[i][j], win[i][j + 1], win[i][j + 2], win[i + 1][j], win[i + 1][j + 1],
                                      win[i + 1][j + 2], win[i + 2][j], win[i + 2][j + 1], win[i + 2][j + 2], k, bs);
        m = fmaxf(m, fmaf(v, sc, sh));
      }
    o[(long)w2 * STEM_C] = m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][0] = win[r][2];
      win[r][1] = win[r][3];
    }
  }
}

// backward pass 1: route the pooled gradient to the window argmax (first maximum, as torch's max_pool2d),
// gate by ReLU, accumulate sum(dz) and sum(dz * xhat) per channel.  partial[2][32][B*H2].
__global__ __launch_bounds__(256) void stem_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ coeff,
                                                             const float* __restrict__ dout,
                                                             float* __restrict__ partial, int H, int W) {
  extern __shared__ float sm[];
  const int ldw = W + 2, H2 = H / 2, W2 = W / 2;
  float* red = sm + 4 * ldw;
  const int b = blockIdx.x / H2, h2 = blockIdx.x - b * H2;
  stem_stage_rows<4>(sm, ldw, x + (long)b * H * W, 2 * h2 - 1, H, W);
  const int c = threadIdx.x & 31, pl = threadIdx.x >> 5;
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = wgt[c * 9 + i];
  const float bs = bias[c];
  const float mean = coeff[c], istd = coeff[STEM_C + c], sc = coeff[2 * STEM_C + c], sh = coeff[3 * STEM_C + c];
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  // lane group pl takes a run of consecutive pooled columns and slides a 4x4 window along the staged rows
  const int per = (W2 + 7) / 8;
  const int w_first = pl * per, w_last = min(W2, w_first + per);
  float win[4][4];
  if (w_first < w_last) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][0] = sm[r * ldw + 2 * w_first];
      win[r][1] = sm[r * ldw + 2 * w_first + 1];
    }
  }
  for (int w2 = w_first; w2 < w_last; ++w2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][2] = sm[r * ldw + 2 * w2 + 2];
      win[r][3] = sm[r * ldw + 2 * w2 + 3];
    }
    float best = -INFINITY, bestconv = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float v = stem_conv_win(win[i][j], win[i][j + 1], win[i][j + 2], win[i + 1][j], win[i + 1][j + 1],
                                      win[i + 1][j + 2], win[i + 2][j], win[i + 2][j + 1], win[i + 2][j + 2], k, bs);
        const float y = fmaxf(fmaf(v, sc, sh), 0.f);
        if (y > best) { best = y; bestconv = v; }
      }
    if (best > 0.f) {
      const float d = dout[(((long)b * H2 + h2) * W2 + w2) * STEM_C + c];
      s1 += d;
      s2 = fmaf(d, (bestconv - mean) * istd, s2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][0] = win[r][2];
      win[r][1] = win[r][3];
    }
  }
  red[(pl * 2 + 0) * 32 + c] = s1;
  red[(pl * 2 + 1) * 32 + c] = s2;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int which = threadIdx.x >> 5;
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += red[(p * 2 + which) * 32 + c];
    partial[((long)which * STEM_C + c) * gridDim.x + blockIdx.x] = t;
  }
}

// backward pass 2: dconv = scale*(dz - S1/N - xhat*S2/N) at every full-resolution position;
// accumulate dW[c][tap] = sum dconv * x[pos+tap], dbias[c] = sum dconv.  partial[B*H2][32][10].
__global__ __launch_bounds__(256) void stem_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ coeff,
                                                         const float* __restrict__ dout, const float* __restrict__ s12,
                                                         float inv_n, float* __restrict__ partial, int H, int W) {
  extern __shared__ float sm[];
  const int ldw = W + 2, H2 = H / 2, W2 = W / 2;
  float* red = sm + 4 * ldw;  // [8][32][10]
  const int b = blockIdx.x / H2, h2 = blockIdx.x - b * H2;
  stem_stage_rows<4>(sm, ldw, x + (long)b * H * W, 2 * h2 - 1, H, W);
  const int c = threadIdx.x & 31, pl = threadIdx.x >> 5;
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = wgt[c * 9 + i];
  const float bs = bias[c];
  const float mean = coeff[c], istd = coeff[STEM_C + c], sc = coeff[2 * STEM_C + c], sh = coeff[3 * STEM_C + c];
  const float m1 = s12[c] * inv_n, m2 = s12[STEM_C + c] * inv_n;
  __syncthreads();
  float acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.f;
  // columns: pooled windows cover w < 2*W2; a trailing odd column still receives the dense BN terms
  const int ncol = (W + 1) / 2;
  // lane group pl takes a run of consecutive column pairs; the 4x4 window holds every input value the four
  // convolution values and their nine weight-gradient products need
  const int per = (ncol + 7) / 8;
  const int w_first = pl * per, w_last = min(ncol, w_first + per);
  float win[4][4];
  if (w_first < w_last) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][0] = sm[r * ldw + 2 * w_first];
      win[r][1] = sm[r * ldw + 2 * w_first + 1];
    }
  }
  for (int w2 = w_first; w2 < w_last; ++w2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][2] = sm[r * ldw + 2 * w2 + 2];
      win[r][3] = 2 * w2 + 3 < ldw ? sm[r * ldw + 2 * w2 + 3] : 0.f;   // past the staged row only for an odd width
    }
    float conv[2][2];
    float best = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int w = 2 * w2 + j;
        conv[i][j] = (w < W) ? stem_conv_win(win[i][j], win[i][j + 1], win[i][j + 2], win[i + 1][j], win[i + 1][j + 1],
                                             win[i + 1][j + 2], win[i + 2][j], win[i + 2][j + 1], win[i + 2][j + 2], k,
                                             bs)
                             : 0.f;
        const float y = fmaxf(fmaf(conv[i][j], sc, sh), 0.f);
        if (y > best) { best = y; bi = i * 2 + j; }
      }
    const bool pooled = w2 < W2;
    const float d = (pooled && best > 0.f) ? dout[(((long)b * H2 + h2) * W2 + w2) * STEM_C + c] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int w = 2 * w2 + j;
        if (w < W) {
          const float dz = (pooled && (i * 2 + j) == bi) ? d : 0.f;
          const float xh = (conv[i][j] - mean) * istd;
          const float dc = sc * (dz - m1 - xh * m2);
          acc[0] = fmaf(dc, win[i][j], acc[0]); acc[1] = fmaf(dc, win[i][j + 1], acc[1]);
          acc[2] = fmaf(dc, win[i][j + 2], acc[2]);
          acc[3] = fmaf(dc, win[i + 1][j], acc[3]); acc[4] = fmaf(dc, win[i + 1][j + 1], acc[4]);
          acc[5] = fmaf(dc, win[i + 1][j + 2], acc[5]);
          acc[6] = fmaf(dc, win[i + 2][j], acc[6]); acc[7] = fmaf(dc, win[i + 2][j + 1], acc[7]);
          acc[8] = fmaf(dc, win[i + 2][j + 2], acc[8]);
          acc[9] += dc;
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      win[r][0] = win[r][2];
      win[r][1] = win[r][3];
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) red[(pl * 32 + c) * 10 + i] = acc[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 320; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += red[p * 320 + i];
    partial[(long)blockIdx.x * 320 + i] = t;
  }
}

// sum partial[rows][320] (c*10+k) -> dW[32*9], dbias[32].  16 columns per workgroup of 256 threads: lane group
// t >> 4 takes every sixteenth row (sixteen loads in flight: the kernel is a chain of round trips on the step's
// tail), the sixteen groups are added in fixed order.
__global__ __launch_bounds__(256) void stem_bwd_w_finalize_kernel(const float* __restrict__ partial, int rows,
                                                                  float* __restrict__ dW, float* __restrict__ db) {
  __shared__ double red[16][16];
  const int cs = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cs;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (col < 320) {
    int r = rg;
    for (; r + 240 < rows; r += 256) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = partial[(long)(r + 16 * u) * 320 + col];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        s0 += (double)v[u]; s1 += (double)v[u + 1]; s2 += (double)v[u + 2]; s3 += (double)v[u + 3];
      }
    }
    for (; r + 48 < rows; r += 64) {
      s0 += (double)partial[(long)r * 320 + col];
      s1 += (double)partial[(long)(r + 16) * 320 + col];
      s2 += (double)partial[(long)(r + 32) * 320 + col];
      s3 += (double)partial[(long)(r + 48) * 320 + col];
    }
    for (; r < rows; r += 16) s0 += (double)partial[(long)r * 320 + col];
  }
  red[rg][cs] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < 320) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cs];
    const int c = col / 10, k = col - c * 10;
    if (k < 9) dW[c * 9 + k] = (float)t; else db[c] = (float)t;
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// partial: [2][C][rows] sums / sums of squares (rows = B * tiles, tiles of one image contiguous).
// One workgroup per channel: the 256 threads stride over that channel's contiguous rows (coalesced), double
// accumulation, fixed-order combination.
__device__ __forceinline__ void bn_finalize_body(const float* __restrict__ partial, int rows, int tiles_per_img,
                                                 double count, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ rmean,
                                                 float* __restrict__ rvar, long long* __restrict__ nbt, float momentum,
                                                 float eps, int training, float* __restrict__ coeff,
                                                 float* __restrict__ bsum, int C) {
  __shared__ double red[16];
  constexpr int STASH = 2048;          // rows of the channel's sums kept in LDS for the per-utterance pass below
  __shared__ float stash[STASH];
  const int c = blockIdx.x;
  if (training && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  const float* ps = partial + (long)c * rows;
  const float* pq = partial + ((long)C + c) * rows;
  const bool stashed = training && bsum && rows <= STASH;   // uniform
  // (what the last thread standing needs is requested ahead of the sums: behind their barriers it would be one more
  // round trip to memory on a kernel that is nothing but latency)
  const float gam = gamma[c], bet = beta[c];
  const float rm0 = rmean ? rmean[c] : 0.f, rv0 = rmean ? rvar[c] : 0.f;
  double s = 0.0, q = 0.0;
  if (training) {
    for (int r = threadIdx.x; r < rows; r += 256) {
      const float v = ps[r];
      if (stashed) stash[r] = v;
      s += (double)v;
      q += (double)pq[r];
    }
    s = block_sum_d(s, red);
    q = block_sum_d(q, red + 8);
  }
  if (threadIdx.x == 0) {
    float mean, var;
    if (training) {
      const double m = s / count;
      double v = q / count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m;
      var = (float)v;
      if (rmean) {
        const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
        rmean[c] = (1.f - momentum) * rm0 + momentum * mean;
        rvar[c] = (1.f - momentum) * rv0 + momentum * (float)unb;
      }
    } else {
      mean = rm0;
      var = rv0;
    }
    const float istd = 1.0f / sqrtf(var + eps);
    const float sc = gam * istd;
    coeff[c] = mean;
    coeff[C + c] = istd;
    coeff[2 * C + c] = sc;
    coeff[3 * C + c] = bet - mean * sc;
  }
  if (bsum) {
    // eight lanes per utterance: each takes every eighth tile, the eight sums meet through lane exchanges
    const int B = rows / tiles_per_img;
    const int sub = threadIdx.x & 7;
    // (the sums were read once already: from LDS this pass costs no further trips to memory -- each lane's chain of
    // up to five dependent loads was half of the kernel's time on the first block's 1 056 rows)
    for (int b = threadIdx.x >> 3; b < B; b += 32) {
      float t = 0.f;
      if (stashed)
        for (int i = sub; i < tiles_per_img; i += 8) t += stash[b * tiles_per_img + i];
      else
        for (int i = sub; i < tiles_per_img; i += 8) t += ps[(long)b * tiles_per_img + i];
      t += dpp_mov<DPP_XOR1>(t);          // (the eight lanes of an utterance are enabled together)
      t += dpp_mov<DPP_XOR2>(t);
      t += dpp_mov<DPP_HALF_MIRROR>(t);
      if (sub == 0) bsum[(long)b * C + c] = t;
    }
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int tiles_per_img,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, long long* __restrict__ nbt,
                                                          float momentum, float eps, int training,
                                                          float* __restrict__ coeff, float* __restrict__ bsum, int C) {
  bn_finalize_body(partial, rows, tiles_per_img, count, gamma, beta, rmean, rvar, nbt, momentum, eps, training, coeff,
                   bsum, C);
}

// Two BatchNorm layers over maps of the same geometry in one launch (blockIdx.y = layer): the first convolution of a
// residual block and its shortcut.
struct BnLayer {
  const float* partial;
  const float* gamma;
  const float* beta;
  float* rmean;
  float* rvar;
  long long* nbt;
  float* coeff;
};

__global__ __launch_bounds__(256) void bn_finalize_pair_kernel(const BnLayer l0, const BnLayer l1, int rows,
                                                               int tiles_per_img, double count, float momentum,
                                                               float eps, int training, int C) {
  const BnLayer& l = blockIdx.y ? l1 : l0;
  bn_finalize_body(l.partial, rows, tiles_per_img, count, l.gamma, l.beta, l.rmean, l.rvar, l.nbt, momentum, eps,
                   training, l.coeff, nullptr, C);
}

__global__ void bn_relu_apply_kernel(const f32x4* __restrict__ x, const float* __restrict__ coeff,
                                     f32x4* __restrict__ y, long n4, int C) {
  const int c4n = C >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const f32x4 v = x[i];
    const f32x4 sc = *reinterpret_cast<const f32x4*>(coeff + 2 * C + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(coeff + 3 * C + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
    y[i] = o;
  }
}

// ------------------------------------------------------------------------------------------------ squeeze-excite
// Squeeze-excite gate of utterance b by one 256-thread workgroup: gap = scale * mean_hw(raw) + shift; hid = relu(W1 gap);
// gate = sigmoid(W2 hid).  The gate is left in gs[C] (LDS; valid behind the function's last barrier); with `save` the
// three vectors also go to memory (the backward pass reads them).  g[256], hd[16], gs[256]: LDS of the caller.
__device__ __forceinline__ void se_gate_body(const float* __restrict__ bsum, const float* __restrict__ coeff,
                                             const float* __restrict__ W1, const float* __restrict__ W2, float inv_hw,
                                             float* __restrict__ gap, float* __restrict__ hid, float* __restrict__ gate,
                                             int b, int C, int Ch, bool save, float* g, float* hd, float* gs) {
  const int t = threadIdx.x;
  if (t < C) {
    const float v = fmaf(coeff[2 * C + t], bsum[(long)b * C + t] * inv_hw, coeff[3 * C + t]);
    g[t] = v;
    if (save) gap[(long)b * C + t] = v;
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
  // wave w takes hidden units w, w + 4, w + 8, w + 12 together (their weight rows are requested at once), then the
  // next sixteen; each unit's sum keeps its order
  for (int j0 = w; j0 < Ch; j0 += 16) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    // C <= 256: at most four 64-channel steps, whose sixteen weights are all requested before the first is used (one
    // trip to memory instead of four); the sums keep their order
    float wv[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * it;
        wv[it][u] = (c < C && j0 + 4 * u < Ch) ? W1[(long)(j0 + 4 * u) * C + c] : 0.f;
      }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane + 64 * it;
      if (c < C) {
        const float gc = g[c];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + 4 * u < Ch) a[u] = fmaf(wv[it][u], gc, a[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = fmaxf(wave_sum(a[u]), 0.f);
      if (lane == 0 && j0 + 4 * u < Ch) {
        hd[j0 + 4 * u] = v;
        if (save) hid[(long)b * Ch + j0 + 4 * u] = v;
      }
    }
  }
  __syncthreads();
  if (t < C) {
    float a = 0.f;
    const float* wr = W2 + (long)t * Ch;
    int j = 0;
    for (; j + 4 <= Ch && (Ch & 3) == 0; j += 4) {           // rows of W2 are 16-byte aligned when Ch % 4 == 0
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + j);
      a = fmaf(w4[0], hd[j], a); a = fmaf(w4[1], hd[j + 1], a);
      a = fmaf(w4[2], hd[j + 2], a); a = fmaf(w4[3], hd[j + 3], a);
    }
    for (; j < Ch; ++j) a = fmaf(wr[j], hd[j], a);
    const float s = 1.0f / (1.0f + expf(-a));
    gs[t] = s;
    if (save) gate[(long)b * C + t] = s;
  }
  __syncthreads();
}

// one block per utterance
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ bsum, const float* __restrict__ coeff,
                                                      const float* __restrict__ W1, const float* __restrict__ W2,
                                                      float inv_hw, float* __restrict__ gap, float* __restrict__ hid,
                                                      float* __restrict__ gate, int C, int Ch) {
  __shared__ float g[256], gs[256];
  __shared__ float hd[16];
  se_gate_body(bsum, coeff, W1, W2, inv_hw, gap, hid, gate, blockIdx.x, C, Ch, true, g, hd, gs);
}

// The squeeze-excite gate and the pooling that consumes it in one launch: grid = (chunks, B); every workgroup works the
// gate of its utterance out for itself (a few thousand multiply-adds on vectors that sit in L2; its map loads are in
// flight meanwhile), chunk 0 also stores gap / hid / gate for the backward pass; then
// out[b][h2][w2][c] = max over the 2x2 window of relu( bn2(c2)*gate[b][c] + bn_s(sc) ), QUADS channel quads per thread.
constexpr int SEPOOL_QUADS = 2;
__global__ __launch_bounds__(256) void res_segate_relu_pool_kernel(
    const float* __restrict__ c2, const float* __restrict__ sc, const float* __restrict__ coeff2,
    const float* __restrict__ coeffs, const float* __restrict__ bsum, const float* __restrict__ fc1,
    const float* __restrict__ fc2, float inv_hw, float* __restrict__ gap, float* __restrict__ hid,
    float* __restrict__ gate, float* __restrict__ out, int H, int W, int C, int Ch) {
  __shared__ float g[256], gs[256];
  __shared__ float hd[16];
  const int b = blockIdx.y;
  const int H2 = H / 2, W2 = W / 2, c4n = C >> 2;
  const int per_utt = H2 * W2 * c4n;                       // channel quads of one utterance's pooled map
  // this thread's quads and their eight map loads each, requested ahead of the gate
  f32x4 a[SEPOOL_QUADS][4], s[SEPOOL_QUADS][4];
  int idx[SEPOOL_QUADS];
#pragma unroll
  for (int q = 0; q < SEPOOL_QUADS; ++q) {
    const int i = ((int)blockIdx.x * SEPOOL_QUADS + q) * 256 + (int)threadIdx.x;
    idx[q] = i < per_utt ? i : -1;
    if (idx[q] >= 0) {
      const int c = (i % c4n) * 4;
      const int r = i / c4n;
      const int w2 = r % W2, h2 = r / W2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long off = (((long)b * H + 2 * h2 + (k >> 1)) * W + 2 * w2 + (k & 1)) * C + c;
        a[q][k] = *reinterpret_cast<const f32x4*>(c2 + off);
        s[q][k] = *reinterpret_cast<const f32x4*>(sc + off);
      }
    }
  }
  se_gate_body(bsum, coeff2, fc1, fc2, inv_hw, gap, hid, gate, b, C, Ch, blockIdx.x == 0, g, hd, gs);
#pragma unroll
  for (int q = 0; q < SEPOOL_QUADS; ++q) {
    if (idx[q] < 0) continue;
    const int c = (idx[q] % c4n) * 4;
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(coeff2 + 2 * C + c);
    const f32x4 h2c = *reinterpret_cast<const f32x4*>(coeff2 + 3 * C + c);
    const f32x4 ss = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + c);
    const f32x4 hs = *reinterpret_cast<const f32x4*>(coeffs + 3 * C + c);
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaf(fmaf(a[q][k][e], s2[e], h2c[e]), gs[c + e], fmaf(s[q][k][e], ss[e], hs[e]));
        m[e] = fmaxf(m[e], z);
      }
    *reinterpret_cast<f32x4*>(out + ((long)b * per_utt + idx[q]) * 4) = m;
  }
}

// out[b][h2][w2][c] = max over the 2x2 window of relu( bn2(c2)*gate[b][c] + bn_s(sc) )
__global__ void res_gate_relu_pool_kernel(const float* __restrict__ c2, const float* __restrict__ sc,
                                          const float* __restrict__ coeff2, const float* __restrict__ coeffs,
                                          const float* __restrict__ gate, float* __restrict__ out, int B, int H, int W,
                                          int C) {
  const int H2 = H / 2, W2 = W / 2, c4n = C >> 2;
  const long total = (long)B * H2 * W2 * c4n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    long r = i / c4n;
    const int w2 = (int)(r % W2); r /= W2;
    const int h2 = (int)(r % H2);
    const int b = (int)(r / H2);
    const f32x4 s2 = *reinterpret_cast<const f32x4*>(coeff2 + 2 * C + c);
    const f32x4 h2c = *reinterpret_cast<const f32x4*>(coeff2 + 3 * C + c);
    const f32x4 ss = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + c);
    const f32x4 hs = *reinterpret_cast<const f32x4*>(coeffs + 3 * C + c);
    const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + (long)b * C + c);
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const long off = (((long)b * H + 2 * h2 + dy) * W + 2 * w2 + dx) * C + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(c2 + off);
        const f32x4 s = *reinterpret_cast<const f32x4*>(sc + off);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float z = fmaf(fmaf(a[k], s2[k], h2c[k]), gt[k], fmaf(s[k], ss[k], hs[k]));
          m[k] = fmaxf(m[k], z);
        }
      }
    *reinterpret_cast<f32x4*>(out + i * 4) = m;
  }
}

// The last block of the audio branch: the same pooled values, but only their mean over the pooled positions is wanted
// (AdaptiveAvgPool2d(1) follows, multimodal_model.py:123-125).  grid = (C / 64, B): lane = channel, the four waves take
// every fourth pooled position, their sums are added in wave order.
__global__ __launch_bounds__(256) void res_gate_relu_pool_mean_kernel(const float* __restrict__ c2,
                                                                      const float* __restrict__ sc,
                                                                      const float* __restrict__ coeff2,
                                                                      const float* __restrict__ coeffs,
                                                                      const float* __restrict__ gate,
                                                                      float* __restrict__ feat, int H, int W, int C) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
  const int H2 = H / 2, W2 = W / 2, P2 = H2 * W2;
  const float s2 = coeff2[2 * C + c], h2c = coeff2[3 * C + c], ss = coeffs[2 * C + c], hs = coeffs[3 * C + c];
  const float gt = gate[(long)b * C + c];
  float acc = 0.f;
  for (int p2 = w; p2 < P2; p2 += 4) {
    const int h2 = p2 / W2, w2 = p2 - h2 * W2;
    float a[4], s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long off = (((long)b * H + 2 * h2 + (k >> 1)) * W + 2 * w2 + (k & 1)) * C + c;
      a[k] = c2[off];
      s[k] = sc[off];
    }
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaf(fmaf(a[k], s2, h2c), gt, fmaf(s[k], ss, hs)));
    acc += m;
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0) feat[(long)b * C + c] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)P2;
}

// res_gate_relu_pool_mean_kernel with the squeeze-excite gate worked out by every workgroup for its utterance (see
// res_segate_relu_pool_kernel); the workgroup of the first 64 channels stores gap / hid / gate.
__global__ __launch_bounds__(256) void res_segate_relu_pool_mean_kernel(
    const float* __restrict__ c2, const float* __restrict__ sc, const float* __restrict__ coeff2,
    const float* __restrict__ coeffs, const float* __restrict__ bsum, const float* __restrict__ fc1,
    const float* __restrict__ fc2, float inv_hw, float* __restrict__ gap, float* __restrict__ hid,
    float* __restrict__ gate, float* __restrict__ feat, int H, int W, int C, int Ch) {
  __shared__ float g[256], gs[256];
  __shared__ float hd[16];
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
  const int H2 = H / 2, W2 = W / 2, P2 = H2 * W2;
  const float s2 = coeff2[2 * C + c], h2c = coeff2[3 * C + c], ss = coeffs[2 * C + c], hs = coeffs[3 * C + c];
  se_gate_body(bsum, coeff2, fc1, fc2, inv_hw, gap, hid, gate, b, C, Ch, blockIdx.x == 0, g, hd, gs);
  const float gt = gs[c];
  float acc = 0.f;
  for (int p2 = w; p2 < P2; p2 += 4) {
    const int h2 = p2 / W2, w2 = p2 - h2 * W2;
    float a[4], s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long off = (((long)b * H + 2 * h2 + (k >> 1)) * W + 2 * w2 + (k & 1)) * C + c;
      a[k] = c2[off];
      s[k] = sc[off];
    }
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaf(fmaf(a[k], s2, h2c), gt, fmaf(s[k], ss, hs)));
    acc += m;
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w == 0) feat[(long)b * C + c] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)P2;
}

// The same arithmetic without the pooling, every part optional: out = [relu]( (a * s2 + h2) * gate + (s * ss + hs) ),
// a null coefficient table meaning scale 1 / shift 0, a null gate 1, a null second map 0.  Serves the stand-alone
// SEBlock (x * gate) and ResidualBlock (no pool; identity or projection shortcut) forwards.
__global__ void gate_affine_add_kernel(const f32x4* __restrict__ a, const float* __restrict__ coeff2,
                                       const float* __restrict__ gate, const f32x4* __restrict__ sc,
                                       const float* __restrict__ coeffs, f32x4* __restrict__ out, long n4, int P,
                                       int C, int relu) {
  const int c4n = C >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long b = i / ((long)c4n * P);
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 s2 = coeff2 ? *reinterpret_cast<const f32x4*>(coeff2 + 2 * C + c) : one;
    const f32x4 h2 = coeff2 ? *reinterpret_cast<const f32x4*>(coeff2 + 3 * C + c) : zero;
    const f32x4 gt = gate ? *reinterpret_cast<const f32x4*>(gate + b * C + c) : one;
    const f32x4 av = a[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf(av[k], s2[k], h2[k]) * gt[k];
    if (sc) {
      const f32x4 ss = coeffs ? *reinterpret_cast<const f32x4*>(coeffs + 2 * C + c) : one;
      const f32x4 hs = coeffs ? *reinterpret_cast<const f32x4*>(coeffs + 3 * C + c) : zero;
      const f32x4 sv = sc[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += fmaf(sv[k], ss[k], hs[k]);
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    out[i] = o;
  }
}

// grid = (ceil(C / 64), B): the four waves of a workgroup take every fourth position of 64 channels (four loads in
// flight per lane), and their sums are added in wave order.
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int P,
                                                      int C) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* xb = x + (long)b * P * C + c;
    int p = w;
    for (; p + 12 < P; p += 16) {
      s0 += xb[(long)p * C];
      s1 += xb[(long)(p + 4) * C];
      s2 += xb[(long)(p + 8) * C];
      s3 += xb[(long)(p + 12) * C];
    }
    for (; p < P; p += 4) s0 += xb[(long)p * C];
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && c < C) out[(long)b * C + c] = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) / (float)P;
}

__global__ void gap_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int B, int P, int C) {
  const long total = (long)B * P * C;
  const float inv = 1.0f / (float)P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int b = (int)(i / ((long)P * C));
    dx[i] = dout[(long)b * C + c] * inv;
  }
}

// ------------------------------------------------------------------------------------------------ block backward
// Pass A: dz (gradient w.r.t. the pre-ReLU sum z = bn2(c2)*gate + bn_s(sc)) routed through pool + ReLU, and
// per-(b, chunk) partial sums A1 = sum dz, A2 = sum dz*c2, A3 = sum dz*sc.   partial[b][chunk][3][C].
// Block: (C/4) channel quads x (1024/C) position lanes; each block covers `chunk` pooled positions of one image.
__global__ __launch_bounds__(256) void res_pool_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ c2,
                                                           const float* __restrict__ sc,
                                                           const float* __restrict__ coeff2,
                                                           const float* __restrict__ coeffs,
                                                           const float* __restrict__ gate, float* __restrict__ dz,
                                                           float* __restrict__ partial, int H, int W, int C, int chunk,
                                                           int nchunks, int dout_mean) {
  extern __shared__ float red[];  // [planes][3][C]
  const int c4n = C >> 2, planes = 256 / c4n;
  const int q = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int c = q * 4;
  const int b = blockIdx.x / nchunks, ch = blockIdx.x - b * nchunks;
  // dout_mean == 2: no pooling follows the block (a ResidualBlock used on its own): dout is [B][H][W][C] and every
  // position with z > 0 takes its own gradient; the 2x2 windows then only serve as the unit of work and cover an odd
  // last row / column too
  const bool unpooled = dout_mean == 2;
  const int H2 = unpooled ? (H + 1) / 2 : H / 2, W2 = unpooled ? (W + 1) / 2 : W / 2, P2 = H2 * W2;
  const f32x4 s2 = *reinterpret_cast<const f32x4*>(coeff2 + 2 * C + c);
  const f32x4 h2c = *reinterpret_cast<const f32x4*>(coeff2 + 3 * C + c);
  const f32x4 ss = *reinterpret_cast<const f32x4*>(coeffs + 2 * C + c);
  const f32x4 hs = *reinterpret_cast<const f32x4*>(coeffs + 3 * C + c);
  const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + (long)b * C + c);
  f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
  const int pend = min((ch + 1) * chunk, P2);
  for (int p2 = ch * chunk + pl; p2 < pend; p2 += planes) {
    const int h2 = p2 / W2, w2 = p2 - h2 * W2;
    // dout_mean: dout is [B][C], the gradient with respect to the mean over the pooled positions (global average pool
    // behind the last block): every position's gradient is that value / P2
    f32x4 av[4], sv[4];
    long offs[4];
    f32x4 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (unpooled) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int hh = 2 * h2 + (k >> 1), ww = 2 * w2 + (k & 1);
        if (hh >= H || ww >= W) continue;
        const long off = (((long)b * H + hh) * W + ww) * C + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(c2 + off);
        const f32x4 sv1 = *reinterpret_cast<const f32x4*>(sc + off);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + off);
        f32x4 ok = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = fmaf(fmaf(a[e], s2[e], h2c[e]), gt[e], fmaf(sv1[e], ss[e], hs[e]));
          if (z > 0.f) {
            ok[e] = g[e];
            a1[e] += g[e];
            a2[e] = fmaf(g[e], a[e], a2[e]);
            a3[e] = fmaf(g[e], sv1[e], a3[e]);
          }
        }
        *reinterpret_cast<f32x4*>(dz + off) = ok;
      }
      continue;
    }
    f32x4 d = *reinterpret_cast<const f32x4*>(dout + (dout_mean ? (long)b * C : ((long)b * P2 + p2) * C) + c);
    if (dout_mean) {
      const float inv = 1.0f / (float)P2;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] *= inv;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      offs[k] = (((long)b * H + 2 * h2 + (k >> 1)) * W + 2 * w2 + (k & 1)) * C + c;
      av[k] = *reinterpret_cast<const f32x4*>(c2 + offs[k]);
      sv[k] = *reinterpret_cast<const f32x4*>(sc + offs[k]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float best = -INFINITY;
      int bi = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float z = fmaf(fmaf(av[k][e], s2[e], h2c[e]), gt[e], fmaf(sv[k][e], ss[e], hs[e]));
        const float y = fmaxf(z, 0.f);
        if (y > best) { best = y; bi = k; }
      }
      if (best > 0.f) {
        const float g = d[e];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k == bi) {
            o[k][e] = g;
            a2[e] = fmaf(g, av[k][e], a2[e]);
            a3[e] = fmaf(g, sv[k][e], a3[e]);
          }
        a1[e] += g;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dz + offs[k]) = o[k];
    // positions outside every pooling window (last column of an odd width, last row of an odd height) receive no
    // routed gradient: the window next to them writes their zeros
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const bool edge_w = (W & 1) && w2 == W2 - 1, edge_h = (H & 1) && h2 == H2 - 1;
    if (edge_w) {
      *reinterpret_cast<f32x4*>(dz + (((long)b * H + 2 * h2) * W + W - 1) * C + c) = zero;
      *reinterpret_cast<f32x4*>(dz + (((long)b * H + 2 * h2 + 1) * W + W - 1) * C + c) = zero;
    }
    if (edge_h) {
      *reinterpret_cast<f32x4*>(dz + (((long)b * H + H - 1) * W + 2 * w2) * C + c) = zero;
      *reinterpret_cast<f32x4*>(dz + (((long)b * H + H - 1) * W + 2 * w2 + 1) * C + c) = zero;
      if (edge_w) *reinterpret_cast<f32x4*>(dz + (((long)b * H + H - 1) * W + W - 1) * C + c) = zero;
    }
  }
  float* mine = red + (long)pl * 3 * C;
  *reinterpret_cast<f32x4*>(mine + c) = a1;
  *reinterpret_cast<f32x4*>(mine + C + c) = a2;
  *reinterpret_cast<f32x4*>(mine + 2 * C + c) = a3;
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    float t = 0.f;
    for (int p = 0; p < planes; ++p) t += red[(long)p * 3 * C + i];
    partial[((long)blockIdx.x) * 3 * C + i] = t;
  }
}

// SE backward per utterance.  Outputs A[b][3][C] (chunk sums), du[b][C], dhp[b][Ch], dg[b][C].
__global__ __launch_bounds__(256) void se_bwd_kernel(const float* __restrict__ partial, int nchunks,
                                                     const float* __restrict__ coeff2, const float* __restrict__ gate,
                                                     const float* __restrict__ hid, const float* __restrict__ W1,
                                                     const float* __restrict__ W2, float* __restrict__ A,
                                                     float* __restrict__ du, float* __restrict__ dhp,
                                                     float* __restrict__ dg, int C, int Ch) {
  __shared__ float sdu[256];
  __shared__ float sdh[16];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < C) {
    float a[3] = {0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= nchunks; k += 4) {       // twelve loads in flight; the sums keep their order
      float v[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[u][j] = partial[(((long)b * nchunks + k + u) * 3 + j) * C + t];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] += v[u][j];
    }
    for (; k < nchunks; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) a[j] += partial[(((long)b * nchunks + k) * 3 + j) * C + t];
#pragma unroll
    for (int j = 0; j < 3; ++j) A[((long)b * 3 + j) * C + t] = a[j];
    const float ds = fmaf(coeff2[2 * C + t], a[1], coeff2[3 * C + t] * a[0]);   // sum dz * bn2(c2)
    const float s = gate[(long)b * C + t];
    const float u = ds * s * (1.f - s);
    sdu[t] = u;
    du[(long)b * C + t] = u;
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
  // wave w takes hidden units w, w + 4, w + 8, w + 12 together (four loads in flight per channel), then the next
  // sixteen; each unit's sum keeps its order
  for (int j0 = w; j0 < Ch; j0 += 16) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    float wv[4][4];       // C <= 256: all sixteen weights of the wave's four units requested at once
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = lane + 64 * it;
        wv[it][u] = (c < C && j0 + 4 * u < Ch) ? W2[(long)c * Ch + j0 + 4 * u] : 0.f;
      }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane + 64 * it;
      if (c < C) {
        const float uc = sdu[c];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + 4 * u < Ch) a[u] = fmaf(wv[it][u], uc, a[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v = wave_sum(a[u]);
      const int j = j0 + 4 * u;
      if (lane == 0 && j < Ch) {
        v = hid[(long)b * Ch + j] > 0.f ? v : 0.f;
        sdh[j] = v;
        dhp[(long)b * Ch + j] = v;
      }
    }
  }
  __syncthreads();
  if (t < C) {
    float a = 0.f;
    int j = 0;
    for (; j + 16 <= Ch; j += 16) {            // sixteen weight loads in flight; the sum keeps its order
      float wj[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) wj[i] = W1[(long)(j + i) * C + t];
#pragma unroll
      for (int i = 0; i < 16; ++i) a = fmaf(wj[i], sdh[j + i], a);
    }
    for (; j + 4 <= Ch; j += 4) {              // four weight loads in flight
      const float w0 = W1[(long)j * C + t], w1 = W1[(long)(j + 1) * C + t];
      const float w2 = W1[(long)(j + 2) * C + t], w3 = W1[(long)(j + 3) * C + t];
      a = fmaf(w0, sdh[j], a); a = fmaf(w1, sdh[j + 1], a); a = fmaf(w2, sdh[j + 2], a); a = fmaf(w3, sdh[j + 3], a);
    }
    for (; j < Ch; ++j) a = fmaf(W1[(long)j * C + t], sdh[j], a);
    dg[(long)b * C + t] = a;
  }
}

// Per-channel BN backward sums for bn2 and the shortcut BN, their affine gradients and the coefficient
// tables of pass B:   d_c2 = P[b][c]*dz + Q[c]*c2 + R[b][c],   d_sc = Ps[c]*dz + Qs[c]*sc + Rs[c].
// tab: P[B][C], R[B][C], then Q[C], Ps[C], Qs[C], Rs[C].
__global__ __launch_bounds__(256) void rb_bwd_coeff_kernel(const float* __restrict__ A, const float* __restrict__ gate,
                                                           const float* __restrict__ dg, const float* __restrict__ bsum,
                                                           const float* __restrict__ coeff2,
                                                           const float* __restrict__ coeffs, int B, int C, float hw,
                                                           float* __restrict__ dgamma2, float* __restrict__ dbeta2,
                                                           float* __restrict__ dgammas, float* __restrict__ dbetas,
                                                           float* __restrict__ tab, int training) {
  // 16 channels per block (16 consecutive lanes = 16 consecutive channels), sixteen lane groups split the utterances
  // (two each at batch 32, so every load of the kernel is in flight at once); fixed-order combine
  __shared__ double red[4][16][16];
  const int lc = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + lc;
  const bool cv = c < C;
  float mean2 = 0.f, istd2 = 0.f, sc2 = 0.f, means = 0.f, istds = 0.f, scs = 0.f;
  if (cv) {
    mean2 = coeff2[c]; istd2 = coeff2[C + c]; sc2 = coeff2[2 * C + c];
    means = coeffs[c]; istds = coeffs[C + c]; scs = coeffs[2 * C + c];
  }
  double S1 = 0.0, S2 = 0.0, S1s = 0.0, S2s = 0.0;
  if (cv)
#pragma unroll 2
    for (int b = grp; b < B; b += 16) {
      const float a1 = A[((long)b * 3 + 0) * C + c], a2 = A[((long)b * 3 + 1) * C + c];
      const float a3 = A[((long)b * 3 + 2) * C + c];
      const float s = gate[(long)b * C + c], g = dg[(long)b * C + c];
      S1 += (double)(s * a1 + g);
      S2 += (double)(s * (a2 - mean2 * a1) * istd2 + (g / hw) * (bsum[(long)b * C + c] - hw * mean2) * istd2);
      S1s += (double)a1;
      S2s += (double)((a3 - means * a1) * istds);
    }
  red[0][grp][lc] = S1; red[1][grp][lc] = S2; red[2][grp][lc] = S1s; red[3][grp][lc] = S2s;
  __syncthreads();
  if (!cv) return;
  S1 = S2 = S1s = S2s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    S1 += red[0][k][lc]; S2 += red[1][k][lc]; S1s += red[2][k][lc]; S2s += red[3][k][lc];
  }
  const float n = (float)B * hw;
  // in eval mode BatchNorm is a fixed affine map: only the scale survives
  // (training bit 1: the shortcut is a fixed affine map whatever bn2 is -- the identity shortcut of a block whose
  // channel count stays, which has no BatchNorm)
  const bool train2 = training & 1, train_s = train2 && !(training & 2);
  const float m1 = train2 ? (float)S1 / n : 0.f, m2 = train2 ? (float)S2 / n : 0.f;
  const float m1s = train_s ? (float)S1s / n : 0.f, m2s = train_s ? (float)S2s / n : 0.f;
  float* P = tab;
  float* R = tab + (long)B * C;
  float* Q = tab + 2L * B * C;
  if (grp == 0) {
    dgamma2[c] = (float)S2; dbeta2[c] = (float)S1;
    dgammas[c] = (float)S2s; dbetas[c] = (float)S1s;
    Q[c] = -sc2 * istd2 * m2;
    Q[C + c] = scs;                                          // Ps
    Q[2 * C + c] = -scs * istds * m2s;                       // Qs
    Q[3 * C + c] = -scs * m1s + scs * istds * m2s * means;   // Rs
  }
#pragma unroll 2
  for (int b = grp; b < B; b += 16) {
    const float s = gate[(long)b * C + c], g = dg[(long)b * C + c];
    P[(long)b * C + c] = sc2 * s;
    R[(long)b * C + c] = sc2 * (g / hw - m1) + sc2 * istd2 * m2 * mean2;
  }
}

// dW2[c][j] = sum_b du[b][c]*hid[b][j];  dW1[j][c] = sum_b dhp[b][j]*gap[b][c]
__global__ void se_wgrad_kernel(const float* __restrict__ du, const float* __restrict__ hid,
                                const float* __restrict__ dhp, const float* __restrict__ gap, float* __restrict__ dW1,
                                float* __restrict__ dW2, int B, int C, int Ch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = C * Ch;
  if (i < n) {  // dW2 [C][Ch]
    const int c = i / Ch, j = i - c * Ch;
    float s = 0.f;
#pragma unroll 8   // the operands of eight utterances are requested together; the sum keeps its order
    for (int b = 0; b < B; ++b) s = fmaf(du[(long)b * C + c], hid[(long)b * Ch + j], s);
    dW2[i] = s;
  } else if (i < 2 * n) {  // dW1 [Ch][C]
    const int k = i - n;
    const int j = k / C, c = k - j * C;
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) s = fmaf(dhp[(long)b * Ch + j], gap[(long)b * C + c], s);
    dW1[k] = s;
  }
}

// Pass B: d_c2 and d_sc from dz (d_sc may alias dz).
__global__ void rb_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ c2,
                                    const float* __restrict__ sc, const float* __restrict__ tab, float* __restrict__ dc2,
                                    float* dsc, int B, long HW, int C) {
  const int c4n = C >> 2;
  const long total = (long)B * HW * c4n;
  const float* P = tab;
  const float* R = tab + (long)B * C;
  const float* Q = tab + 2L * B * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const int b = (int)(i / (HW * c4n));
    const f32x4 d = *reinterpret_cast<const f32x4*>(dz + i * 4);
    const f32x4 a = *reinterpret_cast<const f32x4*>(c2 + i * 4);
    const f32x4 s = *reinterpret_cast<const f32x4*>(sc + i * 4);
    const f32x4 p = *reinterpret_cast<const f32x4*>(P + (long)b * C + c);
    const f32x4 r = *reinterpret_cast<const f32x4*>(R + (long)b * C + c);
    const f32x4 q = *reinterpret_cast<const f32x4*>(Q + c);
    const f32x4 ps = *reinterpret_cast<const f32x4*>(Q + C + c);
    const f32x4 qs = *reinterpret_cast<const f32x4*>(Q + 2 * C + c);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(Q + 3 * C + c);
    f32x4 o1, o2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o1[k] = fmaf(p[k], d[k], fmaf(q[k], a[k], r[k]));
      o2[k] = fmaf(ps[k], d[k], fmaf(qs[k], s[k], rs[k]));
    }
    *reinterpret_cast<f32x4*>(dc2 + i * 4) = o1;
    *reinterpret_cast<f32x4*>(dsc + i * 4) = o2;
  }
}

// ReLU + BN backward statistics for bn1: g = dh * [bn1(c1) > 0]; partial[2][C][blocks] = sum g, sum g*xhat
__global__ __launch_bounds__(256) void bn_relu_bwd_stats_kernel(const float* __restrict__ dh, const float* __restrict__ c1,
                                                                const float* __restrict__ coeff,
                                                                float* __restrict__ partial, long npos, int C,
                                                                int pos_per_block) {
  extern __shared__ float red[];  // [planes][2][C]
  const int c4n = C >> 2, planes = 256 / c4n;
  const int q = threadIdx.x % c4n, pl = threadIdx.x / c4n;
  const int c = q * 4;
  const f32x4 mean = *reinterpret_cast<const f32x4*>(coeff + c);
  const f32x4 istd = *reinterpret_cast<const f32x4*>(coeff + C + c);
  const f32x4 scl = *reinterpret_cast<const f32x4*>(coeff + 2 * C + c);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(coeff + 3 * C + c);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  const long pbeg = (long)blockIdx.x * pos_per_block;
  long pend = pbeg + pos_per_block;
  if (pend > npos) pend = npos;
  for (long p = pbeg + pl; p < pend; p += planes) {
    const f32x4 d = *reinterpret_cast<const f32x4*>(dh + p * C + c);
    const f32x4 x = *reinterpret_cast<const f32x4*>(c1 + p * C + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = fmaf(x[k], scl[k], sh[k]) > 0.f ? d[k] : 0.f;
      s1[k] += g;
      s2[k] = fmaf(g, (x[k] - mean[k]) * istd[k], s2[k]);
    }
  }
  float* mine = red + (long)pl * 2 * C;
  *reinterpret_cast<f32x4*>(mine + c) = s1;
  *reinterpret_cast<f32x4*>(mine + C + c) = s2;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int p = 0; p < planes; ++p) t += red[(long)p * 2 * C + i];
    partial[(long)i * gridDim.x + blockIdx.x] = t;
  }
}

// sums the partial rows ([2][C][rows]), writes dgamma/dbeta, s12 = [S1[C], S2[C]] and cb[3][C] = (scale, Q, R) with
// d_x = [relu mask]*scale*dy + Q*x + R.  One workgroup per channel, threads stride over the contiguous rows.
__global__ __launch_bounds__(256) void bn_bwd_coeff_kernel(const float* __restrict__ partial, int rows, double count,
                                                           const float* __restrict__ coeff, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ s12,
                                                           float* __restrict__ cb, int C, int training) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  const float* p1 = partial + (long)c * rows;
  const float* p2 = partial + ((long)C + c) * rows;
  // (what the last thread standing needs is requested ahead of the sums: behind their barriers it would be one more
  // round trip to memory on a kernel that is nothing but latency)
  float mean = 0.f, istd = 0.f, sc = 0.f;
  if (cb) { mean = coeff[c]; istd = coeff[C + c]; sc = coeff[2 * C + c]; }
  double s = 0.0, q = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) {
    s += (double)p1[r];
    q += (double)p2[r];
  }
  const double S1 = block_sum_d(s, red), S2 = block_sum_d(q, red + 8);
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[c] = (float)S2;
    if (dbeta) dbeta[c] = (float)S1;
    if (s12) { s12[c] = (float)S1; s12[C + c] = (float)S2; }
    if (cb) {
      const float m1 = training ? (float)(S1 / count) : 0.f, m2 = training ? (float)(S2 / count) : 0.f;
      cb[c] = sc;
      cb[C + c] = -sc * istd * m2;
      cb[2 * C + c] = -sc * m1 + sc * istd * m2 * mean;
    }
  }
}

// in place: dh <- [bn1(c1) > 0] * scale * dh + Q * c1 + R
__global__ void bn_relu_bwd_apply_kernel(float* __restrict__ dh, const float* __restrict__ c1,
                                         const float* __restrict__ coeff, const float* __restrict__ cb, long n4,
                                         int C) {
  const int c4n = C >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dh + i * 4);
    const f32x4 x = *reinterpret_cast<const f32x4*>(c1 + i * 4);
    const f32x4 scl = *reinterpret_cast<const f32x4*>(coeff + 2 * C + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(coeff + 3 * C + c);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(cb + c);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(cb + C + c);
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(cb + 2 * C + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = fmaf(x[k], scl[k], sh[k]) > 0.f ? d[k] * k0[k] : 0.f;
      o[k] = g + fmaf(k1[k], x[k], k2[k]);
    }
    *reinterpret_cast<f32x4*>(dh + i * 4) = o;
  }
}

inline int ew_blocks(long n, int per_block = 256) {
  long b = (n + per_block - 1) / per_block;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// ================================================================================================ C ABI
// Rows of the statistics table mmsa_stem_stats writes (partial[2][32][rows]; pass the same count to mmsa_bn_finalize).
MMSA_EXPORT int mmsa_stem_stats_rows(int B, int H) {
  if (B <= 0 || H <= 0) return MMSA_ERR_ARG;
  return B * ((H + STEM_ROWS - 1) / STEM_ROWS);
}

MMSA_EXPORT int mmsa_stem_stats(const float* x, const float* w, const float* bias, float* partial, int B, int H, int W,
                                void* stream) {
  MMSA_REQUIRE(x && w && bias && partial && B > 0 && H > 0 && W > 0 && W <= 1024);
  const int groups = (H + STEM_ROWS - 1) / STEM_ROWS;
  const size_t sh = ((STEM_ROWS + 2) * (W + 2) + 8 * 2 * 32) * sizeof(float);
  hipLaunchKernelGGL(stem_stats_kernel, dim3(B * groups), dim3(256), sh, mmsa_stream(stream), x, w, bias, partial, H, W,
                     groups);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_stem_apply_pool(const float* x, const float* w, const float* bias, const float* coeff, float* out,
                                     int B, int H, int W, void* stream) {
  MMSA_REQUIRE(x && w && bias && coeff && out && B > 0 && H >= 2 && H % 2 == 0 && W >= 2 && W <= 1024);
  const size_t sh = 4 * (W + 2) * sizeof(float);
  hipLaunchKernelGGL(stem_apply_pool_kernel, dim3(B * (H / 2)), dim3(256), sh, mmsa_stream(stream), x, w, bias, coeff,
                     out, H, W);
  return mmsa_launch_status();
}

// Stem backward.  scratch: at least B*(H/2)*(64 + 320) + 64 floats.  Writes dW[32*9], dbias[32], dgamma[32], dbeta[32].
MMSA_EXPORT int mmsa_stem_bwd(const float* x, const float* w, const float* bias, const float* coeff, const float* dout,
                              float* dW, float* dbias, float* dgamma, float* dbeta, float* scratch, int B, int H, int W,
                              int training, void* stream) {
  MMSA_REQUIRE(x && w && bias && coeff && dout && dW && dbias && dgamma && dbeta && scratch);
  MMSA_REQUIRE(B > 0 && H >= 2 && H % 2 == 0 && W >= 2 && W <= 1024);
  hipStream_t s = mmsa_stream(stream);
  const int rows = B * (H / 2);
  float* p1 = scratch;                    // [2][32][rows]
  float* s12 = p1 + (long)rows * 64;      // [2][32]
  float* p2 = s12 + 64;                   // [rows][320]
  const size_t sh1 = (4 * (W + 2) + 8 * 2 * 32) * sizeof(float);
  hipLaunchKernelGGL(stem_bwd_stats_kernel, dim3(rows), dim3(256), sh1, s, x, w, bias, coeff, dout, p1, H, W);
  hipLaunchKernelGGL(bn_bwd_coeff_kernel, dim3(32), dim3(256), 0, s, p1, rows, (double)B * H * W, coeff, dgamma, dbeta,
                     s12, (float*)nullptr, 32, training);
  const size_t sh2 = (4 * (W + 2) + 8 * 320) * sizeof(float);
  const float inv_n = training ? 1.0f / ((float)B * H * W) : 0.f;
  hipLaunchKernelGGL(stem_bwd_w_kernel, dim3(rows), dim3(256), sh2, s, x, w, bias, coeff, dout, s12, inv_n, p2, H, W);
  hipLaunchKernelGGL(stem_bwd_w_finalize_kernel, dim3(20), dim3(256), 0, s, p2, rows, dW, dbias);
  return mmsa_launch_status();
}

// BatchNorm coefficients from per-tile partial sums (training) or running statistics (eval).
// bsum (optional) receives per-utterance channel sums of the raw input, [B][C].
MMSA_EXPORT int mmsa_bn_finalize(const float* partial, int rows, int tiles_per_img, double count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, long long* nbt,
                                 float momentum, float eps, int training, float* coeff, float* bsum, int C,
                                 void* stream) {
  MMSA_REQUIRE(gamma && beta && coeff && C > 0 && rows >= 0 && tiles_per_img > 0);
  MMSA_REQUIRE(training ? (partial && count > 0) : (running_mean && running_var));
  MMSA_REQUIRE(!bsum || (partial && rows % tiles_per_img == 0));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, mmsa_stream(stream), partial, rows,
                     tiles_per_img, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, training, coeff,
                     bsum, C);
  return mmsa_launch_status();
}

// The same for two layers whose maps share a geometry (arrays of two pointers each, host memory).
MMSA_EXPORT int mmsa_bn_finalize_pair(const float* const* partial, int rows, int tiles_per_img, double count,
                                      const float* const* gamma, const float* const* beta, float* const* running_mean,
                                      float* const* running_var, long long* const* nbt, float momentum, float eps,
                                      int training, float* const* coeff, int C, void* stream) {
  MMSA_REQUIRE(partial && gamma && beta && running_mean && running_var && nbt && coeff && C > 0 && rows >= 0);
  MMSA_REQUIRE(tiles_per_img > 0);
  BnLayer l[2];
  for (int i = 0; i < 2; ++i) {
    MMSA_REQUIRE(gamma[i] && beta[i] && coeff[i]);
    MMSA_REQUIRE(training ? (partial[i] && count > 0) : (running_mean[i] && running_var[i]));
    l[i] = BnLayer{partial[i], gamma[i], beta[i], running_mean[i], running_var[i], nbt[i], coeff[i]};
  }
  hipLaunchKernelGGL(bn_finalize_pair_kernel, dim3(C, 2), dim3(256), 0, mmsa_stream(stream), l[0], l[1], rows,
                     tiles_per_img, count, momentum, eps, training, C);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_bn_relu_apply(const float* x, const float* coeff, float* y, long npos, int C, void* stream) {
  MMSA_REQUIRE(x && coeff && y && npos > 0 && C > 0 && C % 4 == 0);
  const long n4 = npos * (C / 4);
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, mmsa_stream(stream),
                     reinterpret_cast<const f32x4*>(x), coeff, reinterpret_cast<f32x4*>(y), n4, C);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_se_gate(const float* bsum, const float* coeff, const float* W1, const float* W2, int HW, float* gap,
                             float* hid, float* gate, int B, int C, int Ch, void* stream) {
  MMSA_REQUIRE(bsum && coeff && W1 && W2 && gap && hid && gate && B > 0 && C > 0 && C <= 256 && Ch > 0 && Ch <= 16);
  hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(256), 0, mmsa_stream(stream), bsum, coeff, W1, W2, 1.0f / (float)HW,
                     gap, hid, gate, C, Ch);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_res_gate_relu_pool(const float* c2, const float* sc, const float* coeff2, const float* coeffs,
                                        const float* gate, float* out, int B, int H, int W, int C, void* stream) {
  MMSA_REQUIRE(c2 && sc && coeff2 && coeffs && gate && out && B > 0 && H >= 2 && W >= 2 && C % 4 == 0);
  const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(res_gate_relu_pool_kernel, dim3(ew_blocks(total)), dim3(256), 0, mmsa_stream(stream), c2, sc,
                     coeff2, coeffs, gate, out, B, H, W, C);
  return mmsa_launch_status();
}

// out[B][P][C] = [relu]( (a * scale2 + shift2) * gate[b][c] + (sc * scale_s + shift_s) ); coeff2 / coeffs are
// [4][C] BatchNorm coefficient tables (rows 2, 3 = scale, shift) or null (identity), gate [B][C] or null (1), sc null (0).
// feat[B][C] = mean over the pooled positions of what mmsa_res_gate_relu_pool would store (C a multiple of 64)
MMSA_EXPORT int mmsa_res_gate_relu_pool_mean(const float* c2, const float* sc, const float* coeff2, const float* coeffs,
                                             const float* gate, float* feat, int B, int H, int W, int C, void* stream) {
  MMSA_REQUIRE(c2 && sc && coeff2 && coeffs && gate && feat && B > 0 && B <= 65535 && H >= 2 && W >= 2);
  MMSA_REQUIRE(C >= 64 && C % 64 == 0);
  hipLaunchKernelGGL(res_gate_relu_pool_mean_kernel, dim3(C / 64, B), dim3(256), 0, mmsa_stream(stream), c2, sc, coeff2,
                     coeffs, gate, feat, H, W, C);
  return mmsa_launch_status();
}

// mmsa_se_gate followed by mmsa_res_gate_relu_pool (feat == null: out [B][H/2][W/2][C]) or by
// mmsa_res_gate_relu_pool_mean (out == null: feat [B][C]) in one launch: the same values bit for bit, gap / hid / gate
// stored as mmsa_se_gate stores them.
MMSA_EXPORT int mmsa_res_se_gate_relu_pool(const float* c2, const float* sc, const float* coeff2, const float* coeffs,
                                           const float* bsum, const float* W1, const float* W2, float* gap, float* hid,
                                           float* gate, float* out, float* feat, int B, int H, int W, int C, int Ch,
                                           void* stream) {
  MMSA_REQUIRE(c2 && sc && coeff2 && coeffs && bsum && W1 && W2 && gap && hid && gate && (out != nullptr) != (feat != nullptr));
  MMSA_REQUIRE(B > 0 && B <= 65535 && H >= 2 && W >= 2 && C > 0 && C <= 256 && C % 4 == 0 && Ch > 0 && Ch <= 16);
  const float inv_hw = 1.0f / (float)(H * W);
  if (feat) {
    MMSA_REQUIRE(C % 64 == 0);
    hipLaunchKernelGGL(res_segate_relu_pool_mean_kernel, dim3(C / 64, B), dim3(256), 0, mmsa_stream(stream), c2, sc,
                       coeff2, coeffs, bsum, W1, W2, inv_hw, gap, hid, gate, feat, H, W, C, Ch);
  } else {
    const int per_utt = (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(res_segate_relu_pool_kernel, dim3(mmsa_cdiv(per_utt, 256 * SEPOOL_QUADS), B), dim3(256), 0,
                       mmsa_stream(stream), c2, sc, coeff2, coeffs, bsum, W1, W2, inv_hw, gap, hid, gate, out, H, W, C,
                       Ch);
  }
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_gate_affine_add(const float* a, const float* coeff2, const float* gate, const float* sc,
                                     const float* coeffs, float* out, int B, int P, int C, int relu, void* stream) {
  MMSA_REQUIRE(a && out && B > 0 && P > 0 && C > 0 && C % 4 == 0 && (sc || !coeffs));
  const long n4 = (long)B * P * C / 4;
  hipLaunchKernelGGL(gate_affine_add_kernel, dim3(ew_blocks(n4)), dim3(256), 0, mmsa_stream(stream),
                     reinterpret_cast<const f32x4*>(a), coeff2, gate, reinterpret_cast<const f32x4*>(sc), coeffs,
                     reinterpret_cast<f32x4*>(out), n4, P, C, relu);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_gap_fwd(const float* x, float* out, int B, int P, int C, void* stream) {
  MMSA_REQUIRE(x && out && B > 0 && P > 0 && C > 0);
  MMSA_REQUIRE(B <= 65535);
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(mmsa_cdiv(C, 64), B), dim3(256), 0, mmsa_stream(stream), x, out, B, P, C);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_gap_bwd(const float* dout, float* dx, int B, int P, int C, void* stream) {
  MMSA_REQUIRE(dout && dx && B > 0 && P > 0 && C > 0);
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_blocks((long)B * P * C)), dim3(256), 0, mmsa_stream(stream), dout, dx, B, P,
                     C);
  return mmsa_launch_status();
}

// Scratch floats needed by mmsa_resblock_bwd_ew for a given geometry.
// Pooled positions per workgroup of the element-wise backward's first pass: up to four per position lane, fewer while
// the launch would have under 512 workgroups (the small maps of the last blocks: a workgroup's passes are a chain of
// round trips, and 64 workgroups leave three quarters of the CUs idle).
static int rb_bwd_chunk(int B, int P2, int planes) {
  int passes = 4;
  while (passes > 1 && (long)B * ((P2 + planes * passes - 1) / (planes * passes)) < 512) passes >>= 1;
  return planes * passes;
}

// windows of the element-wise backward's first pass: 2x2 pooling windows, or (a block without pooling behind it) 2x2
// units of work that also cover an odd last row / column
static int rb_bwd_windows(int H, int W, bool unpooled) {
  return unpooled ? ((H + 1) / 2) * ((W + 1) / 2) : (H / 2) * (W / 2);
}

static long rb_bwd_scratch(int B, int H, int W, int C, bool unpooled) {
  if (B <= 0 || H < 2 || W < 2 || C < 64 || C > 256 || C % 64) return MMSA_ERR_ARG;
  const int P2 = rb_bwd_windows(H, W, unpooled);
  const int planes = 1024 / C, chunk = rb_bwd_chunk(B, P2, planes);
  const int nchunks = (P2 + chunk - 1) / chunk;
  const int Ch = C / 16;
  // partial[B][nchunks][3][C] + A[B][3][C] + du[B][C] + dhp[B][Ch] + dg[B][C] + tab[2*B*C + 4*C]
  return (long)B * nchunks * 3 * C + (long)B * 3 * C + (long)B * C + (long)B * Ch + (long)B * C + 2L * B * C + 4L * C;
}

MMSA_EXPORT long mmsa_resblock_bwd_unpooled_scratch_floats(int B, int H, int W, int C) {
  return rb_bwd_scratch(B, H, W, C, true);
}

MMSA_EXPORT long mmsa_resblock_bwd_scratch_floats(int B, int H, int W, int C) {
  if (B <= 0 || H < 2 || W < 2 || C < 64 || C > 256 || C % 64) return MMSA_ERR_ARG;
  const int planes = 1024 / C, chunk = rb_bwd_chunk(B, (H / 2) * (W / 2), planes);
  const int P2 = (H / 2) * (W / 2);
  const int nchunks = (P2 + chunk - 1) / chunk;
  const int Ch = C / 16;
  // partial[B][nchunks][3][C] + A[B][3][C] + du[B][C] + dhp[B][Ch] + dg[B][C] + tab[2*B*C + 4*C]
  return (long)B * nchunks * 3 * C + (long)B * 3 * C + (long)B * C + (long)B * Ch + (long)B * C + 2L * B * C + 4L * C;
}

// Elementwise part of a residual block's backward (everything between the pooled output gradient and the
// two convolution gradients):  dout[B,H/2,W/2,C] -> d_c2[B,H,W,C], d_sc[B,H,W,C] plus the gradients of
// bn2 / shortcut-BN affine parameters and of the two squeeze-excite matrices.
static int resblock_bwd_ew_impl(const float* dout, int dout_mean, const float* c2, const float* sc, const float* coeff2,
                                const float* coeffs, const float* gate, const float* hid, const float* gap,
                                const float* bsum, const float* W1, const float* W2, float* d_c2, float* d_sc,
                                float* dgamma2, float* dbeta2, float* dgammas, float* dbetas, float* dW1, float* dW2,
                                float* scratch, int B, int H, int W, int C, int training, void* stream) {
  MMSA_REQUIRE(dout && c2 && sc && coeff2 && coeffs && gate && hid && gap && bsum && W1 && W2 && d_c2 && d_sc);
  MMSA_REQUIRE(dgamma2 && dbeta2 && dgammas && dbetas && scratch && (dW1 != nullptr) == (dW2 != nullptr));
  MMSA_REQUIRE(B > 0 && H >= 2 && W >= 2 && C >= 64 && C <= 256 && C % 64 == 0);
  hipStream_t s = mmsa_stream(stream);
  const int P2 = rb_bwd_windows(H, W, dout_mean == 2);
  const int planes = 1024 / C, chunk = rb_bwd_chunk(B, P2, planes);
  const int nchunks = (P2 + chunk - 1) / chunk;
  const int Ch = C / 16;
  float* partial = scratch;
  float* A = partial + (long)B * nchunks * 3 * C;
  float* du = A + (long)B * 3 * C;
  float* dhp = du + (long)B * C;
  float* dg = dhp + (long)B * Ch;
  float* tab = dg + (long)B * C;
  float* dz = d_sc;  // pass A writes dz where d_sc will end up; pass B converts it in place
  hipLaunchKernelGGL(res_pool_bwd_kernel, dim3(B * nchunks), dim3(256), (size_t)planes * 3 * C * sizeof(float), s, dout,
                     c2, sc, coeff2, coeffs, gate, dz, partial, H, W, C, chunk, nchunks, dout_mean);
  hipLaunchKernelGGL(se_bwd_kernel, dim3(B), dim3(256), 0, s, partial, nchunks, coeff2, gate, hid, W1, W2, A, du, dhp,
                     dg, C, Ch);
  hipLaunchKernelGGL(rb_bwd_coeff_kernel, dim3(mmsa_cdiv(C, 16)), dim3(256), 0, s, A, gate, dg, bsum, coeff2, coeffs, B,
                     C, (float)(H * W), dgamma2, dbeta2, dgammas, dbetas, tab, training);
  // the squeeze-excite weight gradients feed only the optimiser: with null dW1 / dW2 they are left to
  // mmsa_resblock_bwd_se_wgrad, which the caller may put on another stream
  if (dW1)
    hipLaunchKernelGGL(se_wgrad_kernel, dim3(mmsa_cdiv(2L * C * Ch, 256)), dim3(256), 0, s, du, hid, dhp, gap, dW1, dW2,
                       B, C, Ch);
  const long total4 = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(rb_bwd_apply_kernel, dim3(ew_blocks(total4)), dim3(256), 0, s, dz, c2, sc, tab, d_c2, d_sc, B,
                     (long)H * W, C);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_resblock_bwd_ew(const float* dout, const float* c2, const float* sc, const float* coeff2,
                                     const float* coeffs, const float* gate, const float* hid, const float* gap,
                                     const float* bsum, const float* W1, const float* W2, float* d_c2, float* d_sc,
                                     float* dgamma2, float* dbeta2, float* dgammas, float* dbetas, float* dW1,
                                     float* dW2, float* scratch, int B, int H, int W, int C, int training,
                                     void* stream) {
  return resblock_bwd_ew_impl(dout, 0, c2, sc, coeff2, coeffs, gate, hid, gap, bsum, W1, W2, d_c2, d_sc, dgamma2, dbeta2,
                              dgammas, dbetas, dW1, dW2, scratch, B, H, W, C, training, stream);
}

// The same for the last block of the audio branch, whose pooled output only feeds the global average pool: dfeat is
// [B][C], the gradient with respect to that average (no [B][H/2][W/2][C] gradient map is formed).
MMSA_EXPORT int mmsa_resblock_bwd_ew_from_mean(const float* dfeat, const float* c2, const float* sc, const float* coeff2,
                                               const float* coeffs, const float* gate, const float* hid,
                                               const float* gap, const float* bsum, const float* W1, const float* W2,
                                               float* d_c2, float* d_sc, float* dgamma2, float* dbeta2, float* dgammas,
                                               float* dbetas, float* dW1, float* dW2, float* scratch, int B, int H,
                                               int W, int C, int training, void* stream) {
  return resblock_bwd_ew_impl(dfeat, 1, c2, sc, coeff2, coeffs, gate, hid, gap, bsum, W1, W2, d_c2, d_sc, dgamma2,
                              dbeta2, dgammas, dbetas, dW1, dW2, scratch, B, H, W, C, training, stream);
}

// The same for a residual block with no pooling behind it (the reference's ResidualBlock used on its own,
// multimodal_model.py:64-70): dout is [B][H][W][C], any H, W >= 2.  training: bit 0 batch statistics, bit 1 the
// shortcut is a fixed affine map whatever bit 0 says (coeffs = [0, 1, 1, 0] per channel for the identity shortcut
// of a block whose channel count stays; dgammas / dbetas are then scratch).  Scratch:
// mmsa_resblock_bwd_unpooled_scratch_floats.
MMSA_EXPORT int mmsa_resblock_bwd_ew_unpooled(const float* dout, const float* c2, const float* sc, const float* coeff2,
                                              const float* coeffs, const float* gate, const float* hid,
                                              const float* gap, const float* bsum, const float* W1, const float* W2,
                                              float* d_c2, float* d_sc, float* dgamma2, float* dbeta2, float* dgammas,
                                              float* dbetas, float* dW1, float* dW2, float* scratch, int B, int H,
                                              int W, int C, int training, void* stream) {
  return resblock_bwd_ew_impl(dout, 2, c2, sc, coeff2, coeffs, gate, hid, gap, bsum, W1, W2, d_c2, d_sc, dgamma2, dbeta2,
                              dgammas, dbetas, dW1, dW2, scratch, B, H, W, C, training, stream);
}

// Gradients of the two squeeze-excite matrices from the scratch mmsa_resblock_bwd_ew left behind (same geometry,
// same scratch pointer; valid until that scratch is written again).
MMSA_EXPORT int mmsa_resblock_bwd_se_wgrad(const float* scratch, const float* hid, const float* gap, float* dW1,
                                           float* dW2, int B, int H, int W, int C, void* stream) {
  MMSA_REQUIRE(scratch && hid && gap && dW1 && dW2);
  MMSA_REQUIRE(B > 0 && H >= 2 && W >= 2 && C >= 64 && C <= 256 && C % 64 == 0);
  const int planes = 1024 / C, chunk = rb_bwd_chunk(B, (H / 2) * (W / 2), planes);
  const int P2 = (H / 2) * (W / 2);
  const int nchunks = (P2 + chunk - 1) / chunk;
  const int Ch = C / 16;
  const float* du = scratch + (long)B * nchunks * 3 * C + (long)B * 3 * C;
  const float* dhp = du + (long)B * C;
  hipLaunchKernelGGL(se_wgrad_kernel, dim3(mmsa_cdiv(2L * C * Ch, 256)), dim3(256), 0, mmsa_stream(stream), du, hid, dhp,
                     gap, dW1, dW2, B, C, Ch);
  return mmsa_launch_status();
}

// positions per workgroup of the statistics pass: about 512 workgroups, whole passes of the position lanes
static int bn_relu_bwd_pos_per_block(long npos, int C) {
  const int planes = 1024 / C;
  long ppb = (npos + 511) / 512;
  ppb = (ppb + planes - 1) / planes * planes;
  if (ppb > 256) ppb = 256;
  return (int)ppb;
}

MMSA_EXPORT long mmsa_bn_relu_bwd_scratch_floats(long npos, int C) {
  if (npos <= 0 || C < 32 || C > 256 || C % 32) return MMSA_ERR_ARG;
  const int pos_per_block = bn_relu_bwd_pos_per_block(npos, C);
  const long blocks = (npos + pos_per_block - 1) / pos_per_block;
  return blocks * 2 * C + 3L * C;
}

// dh[npos][C] (gradient w.r.t. relu(bn(c1))) -> in place gradient w.r.t. c1; also dgamma/dbeta.
MMSA_EXPORT int mmsa_bn_relu_bwd(float* dh, const float* c1, const float* coeff, float* dgamma, float* dbeta,
                                 float* scratch, long npos, int C, int training, void* stream) {
  MMSA_REQUIRE(dh && c1 && coeff && dgamma && dbeta && scratch && npos > 0);
  MMSA_REQUIRE(C >= 32 && C <= 256 && C % 32 == 0);
  hipStream_t s = mmsa_stream(stream);
  const int pos_per_block = bn_relu_bwd_pos_per_block(npos, C);
  const int blocks = (int)((npos + pos_per_block - 1) / pos_per_block);
  float* partial = scratch;
  float* cb = scratch + (long)blocks * 2 * C;
  const int planes = 1024 / C;
  hipLaunchKernelGGL(bn_relu_bwd_stats_kernel, dim3(blocks), dim3(256), (size_t)planes * 2 * C * sizeof(float), s, dh,
                     c1, coeff, partial, npos, C, pos_per_block);
  hipLaunchKernelGGL(bn_bwd_coeff_kernel, dim3(C), dim3(256), 0, s, partial, blocks, (double)npos, coeff,
                     dgamma, dbeta, (float*)nullptr, cb, C, training);
  const long n4 = npos * (C / 4);
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dh, c1, coeff, cb, n4, C);
  return mmsa_launch_status();
}

// The same with the two reductions already done: partial [2][C][rows] holds per-tile sums of g = dh [bn(c1) > 0] and of
// g * xhat (left behind by mmsa_conv_fwd_bnrelu_bwd_stats, the convolution that produced dh).  cb: 3 C floats of scratch.
MMSA_EXPORT int mmsa_bn_relu_bwd_finish(float* dh, const float* c1, const float* coeff, const float* partial, int rows,
                                        float* dgamma, float* dbeta, float* cb, long npos, int C, int training,
                                        void* stream) {
  MMSA_REQUIRE(dh && c1 && coeff && partial && dgamma && dbeta && cb && npos > 0 && rows > 0);
  MMSA_REQUIRE(C >= 32 && C <= 256 && C % 32 == 0);
  hipStream_t s = mmsa_stream(stream);
  hipLaunchKernelGGL(bn_bwd_coeff_kernel, dim3(C), dim3(256), 0, s, partial, rows, (double)npos, coeff, dgamma, dbeta,
                     (float*)nullptr, cb, C, training);
  const long n4 = npos * (C / 4);
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dh, c1, coeff, cb, n4, C);
  return mmsa_launch_status();
}

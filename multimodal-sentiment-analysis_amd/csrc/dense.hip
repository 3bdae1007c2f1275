// Small dense layers of the fusion head and the classifiers: Linear (+ LayerNorm / ReLU / dropout epilogues),
// stand-alone LayerNorm with the learnable modality scale, BatchNorm1d + ReLU, the 2-way attention mix of
// the fusion layer, softmax cross-entropy, and their backward passes.
//
// Reference semantics: `CNN implementation/multimodal_model.py:202-225` (MultifactorFeatureNormalizer),
// `:228-285` (ModalityFusion), `:365-404` (the three classifier stacks), `:448-454` (softmax, CE mean).
// These layers are latency-bound (a few hundred KFLOP per utterance): one workgroup per utterance, weights
// stream from L2, every wave reduces along K with lanes on consecutive k (coalesced weight rows).
#include "mmsa_common.h"

#include <mutex>

namespace {

enum Epilogue {
  EPI_NONE = 0, EPI_LN_RELU = 1, EPI_RELU_LN = 2, EPI_RELU = 3,
  EPI_LN = 4, EPI_TANH = 5, EPI_SIGMOID = 6, EPI_LN_GELU = 7, EPI_LAST = 7
};
constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ bool epi_has_ln(int e) {
  return e == EPI_LN_RELU || e == EPI_RELU_LN || e == EPI_LN || e == EPI_LN_GELU;
}
// exact GELU (nn.GELU default): 0.5 x (1 + erf(x / sqrt 2)) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// out[b][n0 .. n0+63] = X[b] W^T + bias.  grid = (B, ceil(N/64)); every wave owns 16 outputs and computes four of
// them at a time (independent weight-row loads stay in flight); lanes run along K, so weight rows are read coalesced.
__global__ __launch_bounds__(256) void dense_matvec_kernel(const float* __restrict__ X, int ldx,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ out, int ldo, int K, int N) {
  extern __shared__ float xs[];  // [K]
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int k = t; k < K; k += 256) xs[k] = X[(long)b * ldx + k];
  __syncthreads();
  const int nbase = blockIdx.y * 64 + w * 16;
#pragma unroll 1
  for (int g4 = 0; g4 < 16; g4 += 4) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const float* wr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = W + (long)min(nbase + g4 + i, N - 1) * K;
    for (int k = lane; k < K; k += 64) {
      const float xv = xs[k];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = fmaf(wr[i][k], xv, a[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[i] = wave_sum(a[i]);
      const int n = nbase + g4 + i;
      if (lane == 0 && n < N) out[(long)b * ldo + n] = a[i] + (bias ? bias[n] : 0.f);
    }
  }
}

// Y[b][:] = drop( epi( pre[b][:] ) ), one block per utterance; saves mean / rstd for the LayerNorm epilogues.
__global__ __launch_bounds__(256) void dense_epilogue_kernel(const float* __restrict__ pre,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ Y,
                                                             int ldy, float* __restrict__ stats, int N, int epi,
                                                             float drop_p, unsigned long long seed,
                                                             const float* __restrict__ step) {
  if (step) seed += (unsigned long long)step[0] * 0x9E3779B97F4A7C15ull;   // replayed launches draw new masks
  __shared__ float red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* ps = pre + (long)b * N;
  float mean = 0.f, rstd = 1.f;
  if (epi_has_ln(epi)) {
    float s = 0.f;
    for (int n = t; n < N; n += 256) {
      float v = ps[n];
      if (epi == EPI_RELU_LN) v = fmaxf(v, 0.f);
      s += v;
    }
    mean = block_sum(s, red) / (float)N;
    float q = 0.f;
    for (int n = t; n < N; n += 256) {
      float v = ps[n];
      if (epi == EPI_RELU_LN) v = fmaxf(v, 0.f);
      q += (v - mean) * (v - mean);
    }
    rstd = 1.0f / sqrtf(block_sum(q, red) / (float)N + LN_EPS);
    if (t == 0 && stats) {
      stats[2 * b] = mean;
      stats[2 * b + 1] = rstd;
    }
  }
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  for (int n = t; n < N; n += 256) {
    float v = ps[n];
    if (epi == EPI_RELU_LN) v = fmaxf(v, 0.f);
    if (epi_has_ln(epi)) v = fmaf((v - mean) * rstd, gamma[n], beta[n]);
    if (epi == EPI_LN_RELU || epi == EPI_RELU) v = fmaxf(v, 0.f);
    else if (epi == EPI_LN_GELU) v = gelu_f(v);
    else if (epi == EPI_TANH) v = tanhf(v);
    else if (epi == EPI_SIGMOID) v = 1.0f / (1.0f + expf(-v));
    if (drop_p > 0.f) v = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? v * keep_scale : 0.f;
    Y[(long)b * ldy + n] = v;
  }
}

// dY[b][:] -> dPre[b][:] (gradient w.r.t. the linear output), dLn[b][:] (gradient w.r.t. the LayerNorm output,
// LN epilogues only), dX[b][:] (= or +=) dPre W.  grid = (B, ceil(K/64)): every block rebuilds the utterance's
// dPre row in LDS (cheap) and then produces 64 columns of dX, four thread groups splitting the sum over N.
__global__ __launch_bounds__(256) void dense_bwd_x_kernel(const float* __restrict__ dY, int ldy,
                                                          const float* __restrict__ pre, const float* __restrict__ stats,
                                                          const float* __restrict__ W, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ dPre,
                                                          float* __restrict__ dLn, float* __restrict__ dX, int ldx,
                                                          int K, int N, int epi, float drop_p, unsigned long long seed,
                                                          int accumulate_dx, const float* __restrict__ step) {
  if (step) seed += (unsigned long long)step[0] * 0x9E3779B97F4A7C15ull;   // replayed launches draw new masks
  extern __shared__ float sm[];
  float* g = sm;  // [N] gradient w.r.t. linear output
  __shared__ float red[16];
  __shared__ float part[4][64];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool first = blockIdx.y == 0;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  const bool ln = epi_has_ln(epi);
  float mean = 0.f, rstd = 1.f;
  if (ln) {
    mean = stats[2 * b];
    rstd = stats[2 * b + 1];
  }
  // pass 1: through dropout and the outer activation; LayerNorm needs two row sums
  float s1 = 0.f, s2 = 0.f;
  for (int n = t; n < N; n += 256) {
    float d = dY[(long)b * ldy + n];
    if (drop_p > 0.f) d = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? d * keep_scale : 0.f;
    const float p = epi == EPI_NONE ? 0.f : pre[(long)b * N + n];   // pre is null for plain linear layers
    if (ln) {
      const float xh = ((epi == EPI_RELU_LN ? fmaxf(p, 0.f) : p) - mean) * rstd;
      const float y = fmaf(xh, gamma[n], beta[n]);                  // LayerNorm output
      if (epi == EPI_LN_RELU) d = y > 0.f ? d : 0.f;
      else if (epi == EPI_LN_GELU) d *= gelu_df(y);
      if (dLn && first) dLn[(long)b * N + n] = d;
      const float dg = d * gamma[n];
      g[n] = dg;
      s1 += dg;
      s2 = fmaf(dg, xh, s2);
    } else if (epi == EPI_RELU) {
      g[n] = p > 0.f ? d : 0.f;
    } else if (epi == EPI_TANH) {
      const float th = tanhf(p);
      g[n] = d * (1.f - th * th);
    } else if (epi == EPI_SIGMOID) {
      const float sg = 1.0f / (1.0f + expf(-p));
      g[n] = d * sg * (1.f - sg);
    } else {
      g[n] = d;
    }
  }
  if (ln) {
    s1 = block_sum(s1, red) / (float)N;
    s2 = block_sum(s2, red) / (float)N;
    for (int n = t; n < N; n += 256) {
      const float p = pre[(long)b * N + n];
      const float v = epi == EPI_RELU_LN ? fmaxf(p, 0.f) : p;
      const float xh = (v - mean) * rstd;
      float d = rstd * (g[n] - s1 - xh * s2);
      if (epi == EPI_RELU_LN && p <= 0.f) d = 0.f;
      g[n] = d;
    }
  }
  __syncthreads();
  if (first)
    for (int n = t; n < N; n += 256) dPre[(long)b * N + n] = g[n];
  if (dX) {
    const int kl = t & 63, grp = t >> 6;
    const int k = blockIdx.y * 64 + kl;
    float a0 = 0.f, a1 = 0.f;
    if (k < K) {
      int n = grp;
      for (; n + 4 < N; n += 8) {
        a0 = fmaf(g[n], W[(long)n * K + k], a0);
        a1 = fmaf(g[n + 4], W[(long)(n + 4) * K + k], a1);
      }
      for (; n < N; n += 4) a0 = fmaf(g[n], W[(long)n * K + k], a0);
    }
    part[grp][kl] = a0 + a1;
    __syncthreads();
    if (grp == 0 && k < K) {
      const float a = (part[0][kl] + part[1][kl]) + (part[2][kl] + part[3][kl]);
      float* dst = dX + (long)b * ldx + k;
      *dst = accumulate_dx ? *dst + a : a;
    }
  }
}

// dW[n][k] = sum_b dPre[b][n] X[b][k];  db[n] = sum_b dPre[b][n];
// dgamma[n] = sum_b dLn[b][n] xhat[b][n];  dbeta[n] = sum_b dLn[b][n].
__global__ void dense_bwd_w_kernel(const float* __restrict__ dPre, const float* __restrict__ dLn,
                                   const float* __restrict__ pre, const float* __restrict__ stats,
                                   const float* __restrict__ X, int ldx, float* __restrict__ dW, float* __restrict__ db,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int K, int N, int epi) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long nk = (long)N * K;
  if (i < nk) {
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    float a = 0.f;
#pragma unroll 8
    for (int b = 0; b < B; ++b) a = fmaf(dPre[(long)b * N + n], X[(long)b * ldx + k], a);
    dW[i] = a;
  } else if (i < nk + N) {
    const int n = (int)(i - nk);
    float sb = 0.f, sg = 0.f, sbt = 0.f;
    for (int b = 0; b < B; ++b) {
      sb += dPre[(long)b * N + n];
      if (dgamma) {
        const float p = pre[(long)b * N + n];
        const float v = epi == EPI_RELU_LN ? fmaxf(p, 0.f) : p;
        const float xh = (v - stats[2 * b]) * stats[2 * b + 1];
        const float d = dLn[(long)b * N + n];
        sg = fmaf(d, xh, sg);
        sbt += d;
      }
    }
    if (db) db[n] = sb;
    if (dgamma) { dgamma[n] = sg; dbeta[n] = sbt; }
  }
}

// y[b] = (LN(x[b]) * gamma + beta) * scale[0] * keep      (keep: 0 or 1 per call, modality dropout)
__global__ __launch_bounds__(256) void ln_scale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ keep, float* __restrict__ y,
                                                           int ldy, float* __restrict__ stats, int N) {
  __shared__ float red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  float s = 0.f;
  for (int n = t; n < N; n += 256) s += x[(long)b * N + n];
  const float mean = block_sum(s, red) / (float)N;
  float q = 0.f;
  for (int n = t; n < N; n += 256) {
    const float d = x[(long)b * N + n] - mean;
    q = fmaf(d, d, q);
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)N + LN_EPS);
  if (t == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
  const float sc = scale[0] * (keep ? keep[0] : 1.f);
  for (int n = t; n < N; n += 256)
    y[(long)b * ldy + n] = fmaf((x[(long)b * N + n] - mean) * rstd, gamma[n], beta[n]) * sc;
}

// dx[b] and per-utterance partial of dscale;  dLn[b][n] = dy * scale * keep (gradient w.r.t. LN affine output)
__global__ __launch_bounds__(256) void ln_scale_bwd_kernel(const float* __restrict__ dy, int ldy,
                                                           const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ keep,
                                                           const float* __restrict__ stats, float* __restrict__ dx,
                                                           float* __restrict__ dLn, float* __restrict__ dscale_part,
                                                           int N) {
  __shared__ float red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const float mean = stats[2 * b], rstd = stats[2 * b + 1];
  const float kp = keep ? keep[0] : 1.f;
  const float sc = scale[0] * kp;
  float s1 = 0.f, s2 = 0.f, ds = 0.f;
  for (int n = t; n < N; n += 256) {
    const float d = dy[(long)b * ldy + n];
    const float xh = (x[(long)b * N + n] - mean) * rstd;
    const float dl = d * sc;
    dLn[(long)b * N + n] = dl;
    ds = fmaf(d * kp, fmaf(xh, gamma[n], beta[n]), ds);
    const float dg = dl * gamma[n];
    s1 += dg;
    s2 = fmaf(dg, xh, s2);
  }
  s1 = block_sum(s1, red) / (float)N;
  s2 = block_sum(s2, red) / (float)N;
  ds = block_sum(ds, red);
  if (t == 0) dscale_part[b] = ds;
  for (int n = t; n < N; n += 256) {
    const float xh = (x[(long)b * N + n] - mean) * rstd;
    dx[(long)b * N + n] = rstd * (dLn[(long)b * N + n] * gamma[n] - s1 - xh * s2);
  }
}

// dgamma[n] = sum_b dLn*xhat, dbeta[n] = sum_b dLn, dscale = sum_b dscale_part[b]
__global__ void ln_scale_bwd_params_kernel(const float* __restrict__ dLn, const float* __restrict__ x,
                                           const float* __restrict__ stats, const float* __restrict__ dscale_part,
                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                           float* __restrict__ dscale, int B, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) {
    float sg = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float d = dLn[(long)b * N + n];
      sg = fmaf(d, (x[(long)b * N + n] - stats[2 * b]) * stats[2 * b + 1], sg);
      sb += d;
    }
    dgamma[n] = sg;
    dbeta[n] = sb;
  }
  if (n == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dscale_part[b];
    dscale[0] = s;
  }
}

// fused[b][n] = pa*w0 + pt*w1 with (w0,w1) = softmax(att[b][0..1]); P = [pa | pt] rows of length 2H
__global__ void fusion_mix_fwd_kernel(const float* __restrict__ P, const float* __restrict__ att,
                                      float* __restrict__ wts, float* __restrict__ fused, int B, int Hd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Hd) return;
  const int b = i / Hd, n = i - b * Hd;
  const float a0 = att[2 * b], a1 = att[2 * b + 1];
  const float m = fmaxf(a0, a1);
  const float e0 = expf(a0 - m), e1 = expf(a1 - m);
  const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
  if (n == 0) { wts[2 * b] = w0; wts[2 * b + 1] = w1; }
  fused[i] = P[(long)b * 2 * Hd + n] * w0 + P[(long)b * 2 * Hd + Hd + n] * w1;
}

// one block per utterance: dP = [dfused*w0 | dfused*w1], datt = softmax backward of (sum dfused*pa, sum dfused*pt)
__global__ __launch_bounds__(256) void fusion_mix_bwd_kernel(const float* __restrict__ dfused,
                                                             const float* __restrict__ P, const float* __restrict__ wts,
                                                             float* __restrict__ dP, float* __restrict__ datt, int Hd) {
  __shared__ float red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const float w0 = wts[2 * b], w1 = wts[2 * b + 1];
  float d0 = 0.f, d1 = 0.f;
  for (int n = t; n < Hd; n += 256) {
    const float d = dfused[(long)b * Hd + n];
    d0 = fmaf(d, P[(long)b * 2 * Hd + n], d0);
    d1 = fmaf(d, P[(long)b * 2 * Hd + Hd + n], d1);
    dP[(long)b * 2 * Hd + n] = d * w0;
    dP[(long)b * 2 * Hd + Hd + n] = d * w1;
  }
  d0 = block_sum(d0, red);
  d1 = block_sum(d1, red);
  if (t == 0) {
    const float dot = w0 * d0 + w1 * d1;
    datt[2 * b] = w0 * (d0 - dot);
    datt[2 * b + 1] = w1 * (d1 - dot);
  }
}

// probs = softmax(logits); loss[0] = mean_b(-log probs[b][y_b]); dlogits = (probs - onehot)/B.  Single block.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits,
                                                         const long long* __restrict__ labels, float* __restrict__ probs,
                                                         float* __restrict__ loss, float* __restrict__ dlogits, int B,
                                                         int C) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* l = logits + (long)b * C;
    float m = l[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(l[c] - m);
    const float inv = 1.0f / s;
    const long long y = labels ? labels[b] : -1;
    for (int c = 0; c < C; ++c) {
      const float p = expf(l[c] - m) * inv;
      probs[(long)b * C + c] = p;
      if (dlogits) dlogits[(long)b * C + c] = (p - (c == y ? 1.f : 0.f)) / (float)B;
    }
    if (labels && y >= 0 && y < C) acc += (m + logf(s)) - l[y];
  }
  if (loss) {
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = acc / (float)B;
  }
}

// BatchNorm1d + ReLU over [B][N] (audio-only classifier).  One thread per feature.
__global__ void bn1d_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                     long long* __restrict__ nbt, float momentum, float eps, int training,
                                     float* __restrict__ y, float* __restrict__ coeff, int B, int N, float drop_p,
                                     unsigned long long seed, const float* __restrict__ step) {
  if (step) seed += (unsigned long long)step[0] * 0x9E3779B97F4A7C15ull;   // replayed launches draw new masks
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0 && training && nbt) *nbt += 1;
  if (n >= N) return;
  float mean, var;
  if (training) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += (double)x[(long)b * N + n];
    const double m = s / B;
    double q = 0.0;
    for (int b = 0; b < B; ++b) {
      const double d = (double)x[(long)b * N + n] - m;
      q += d * d;
    }
    mean = (float)m;
    var = (float)(q / B);
    const float unb = B > 1 ? (float)(q / (B - 1)) : var;
    rmean[n] = (1.f - momentum) * rmean[n] + momentum * mean;
    rvar[n] = (1.f - momentum) * rvar[n] + momentum * unb;
  } else {
    mean = rmean[n];
    var = rvar[n];
  }
  const float istd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[n] * istd, sh = beta[n] - mean * sc;
  coeff[n] = mean; coeff[N + n] = istd; coeff[2 * N + n] = sc; coeff[3 * N + n] = sh;
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  for (int b = 0; b < B; ++b) {
    float v = fmaxf(fmaf(x[(long)b * N + n], sc, sh), 0.f);
    if (drop_p > 0.f) v = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? v * keep_scale : 0.f;
    y[(long)b * N + n] = v;
  }
}

__global__ void bn1d_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                     const float* __restrict__ coeff, float* __restrict__ dx, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int B, int N, int training, float drop_p,
                                     unsigned long long seed, const float* __restrict__ step) {
  if (step) seed += (unsigned long long)step[0] * 0x9E3779B97F4A7C15ull;   // replayed launches draw new masks
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float mean = coeff[n], istd = coeff[N + n], sc = coeff[2 * N + n], sh = coeff[3 * N + n];
  const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const float xv = x[(long)b * N + n];
    float d = dy[(long)b * N + n];
    if (drop_p > 0.f) d = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? d * keep_scale : 0.f;
    if (fmaf(xv, sc, sh) <= 0.f) d = 0.f;
    s1 += d;
    s2 = fmaf(d, (xv - mean) * istd, s2);
  }
  dgamma[n] = s2;
  dbeta[n] = s1;
  const float m1 = training ? s1 / B : 0.f, m2 = training ? s2 / B : 0.f;
  for (int b = 0; b < B; ++b) {
    const float xv = x[(long)b * N + n];
    float d = dy[(long)b * N + n];
    if (drop_p > 0.f) d = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? d * keep_scale : 0.f;
    if (fmaf(xv, sc, sh) <= 0.f) d = 0.f;
    dx[(long)b * N + n] = sc * (d - m1 - (xv - mean) * istd * m2);
  }
}

// y = drop(x) elementwise (text classifier's input dropout); backward is the same map applied to dy
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float drop_p,
                               unsigned long long seed, const float* __restrict__ step) {
  if (step) seed += (unsigned long long)step[0] * 0x9E3779B97F4A7C15ull;   // replayed launches draw new masks
  const float keep_scale = 1.0f / (1.0f - drop_p);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = hash_uniform(seed, (unsigned long long)i) >= drop_p ? x[i] * keep_scale : 0.f;
}

// ---- attention fusion of two projected modalities (config "attention-fusion head") -------------------------
// M = [text_proj | audio_proj] rows of 2*Fd, scores [B][2]: (w0,w1) = softmax(scores[b]);
// Wm = [text_proj*w0 | audio_proj*w1] (the "combined" vector the gate reads).
__global__ void pair_attn_mix_fwd_kernel(const float* __restrict__ M, const float* __restrict__ scores,
                                         float* __restrict__ wts, float* __restrict__ Wm, int B, int Fd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Fd) return;
  const int b = i / Fd, n = i - b * Fd;
  const float a0 = scores[2 * b], a1 = scores[2 * b + 1];
  const float m = fmaxf(a0, a1);
  const float e0 = expf(a0 - m), e1 = expf(a1 - m);
  const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
  if (n == 0) { wts[2 * b] = w0; wts[2 * b + 1] = w1; }
  Wm[(long)b * 2 * Fd + n] = M[(long)b * 2 * Fd + n] * w0;
  Wm[(long)b * 2 * Fd + Fd + n] = M[(long)b * 2 * Fd + Fd + n] * w1;
}

// fused = wt*gate + wa*(1-gate)
__global__ void gate_mix_fwd_kernel(const float* __restrict__ Wm, const float* __restrict__ gate,
                                    float* __restrict__ fused, int B, int Fd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Fd) return;
  const int b = i / Fd, n = i - b * Fd;
  const float g = gate[i];
  fused[i] = Wm[(long)b * 2 * Fd + n] * g + Wm[(long)b * 2 * Fd + Fd + n] * (1.f - g);
}

// dfused -> dWm (= [dfused*gate | dfused*(1-gate)], accumulated on top of what the gate's Linear sent back when
// accumulate != 0) and dgate = dfused * (wt - wa)
__global__ void gate_mix_bwd_kernel(const float* __restrict__ dfused, const float* __restrict__ Wm,
                                    const float* __restrict__ gate, float* __restrict__ dWm, float* __restrict__ dgate,
                                    int B, int Fd, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Fd) return;
  const int b = i / Fd, n = i - b * Fd;
  const float d = dfused[i], g = gate[i];
  const long o = (long)b * 2 * Fd + n;
  if (dgate) dgate[i] = d * (Wm[o] - Wm[o + Fd]);
  if (dWm) {
    const float t0 = d * g, t1 = d * (1.f - g);
    dWm[o] = accumulate ? dWm[o] + t0 : t0;
    dWm[o + Fd] = accumulate ? dWm[o + Fd] + t1 : t1;
  }
}

// dWm -> dM (+= when accumulate) and dscores (softmax backward); one block per utterance
__global__ __launch_bounds__(256) void pair_attn_mix_bwd_kernel(const float* __restrict__ dWm, const float* __restrict__ M,
                                                                const float* __restrict__ wts, float* __restrict__ dM,
                                                                float* __restrict__ dscores, int Fd, int accumulate) {
  __shared__ float red[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const float w0 = wts[2 * b], w1 = wts[2 * b + 1];
  float d0 = 0.f, d1 = 0.f;
  for (int n = t; n < Fd; n += 256) {
    const long o = (long)b * 2 * Fd + n;
    const float g0 = dWm[o], g1 = dWm[o + Fd];
    d0 = fmaf(g0, M[o], d0);
    d1 = fmaf(g1, M[o + Fd], d1);
    dM[o] = accumulate ? dM[o] + g0 * w0 : g0 * w0;
    dM[o + Fd] = accumulate ? dM[o + Fd] + g1 * w1 : g1 * w1;
  }
  d0 = block_sum(d0, red);
  d1 = block_sum(d1, red);
  if (t == 0) {
    const float dot = w0 * d0 + w1 * d1;
    dscores[2 * b] = w0 * (d0 - dot);
    dscores[2 * b + 1] = w1 * (d1 - dot);
  }
}

}  // namespace

MMSA_EXPORT int mmsa_dense_fwd(const float* X, int ldx, const float* W, const float* bias, const float* gamma,
                               const float* beta, float* Y, int ldy, float* pre, float* stats, int B, int K, int N,
                               int epilogue, float drop_p, unsigned long long seed, const float* step,
                               void* stream) {
  MMSA_REQUIRE(X && W && Y && B > 0 && B <= 65535 && K > 0 && N > 0 && ldx >= K && ldy >= N && K <= 12 * 1024);
  MMSA_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_LAST && drop_p >= 0.f && drop_p < 1.f);
  const bool has_ln = epilogue == EPI_LN_RELU || epilogue == EPI_RELU_LN || epilogue == EPI_LN || epilogue == EPI_LN_GELU;
  if (has_ln) MMSA_REQUIRE(gamma && beta && stats);
  hipStream_t s = mmsa_stream(stream);
  const bool plain = epilogue == EPI_NONE && drop_p == 0.f;
  if (!plain) MMSA_REQUIRE(pre != nullptr);     // the linear output is kept for the backward pass
  dim3 grid(B, mmsa_cdiv(N, 64));
  hipLaunchKernelGGL(dense_matvec_kernel, grid, dim3(256), (size_t)K * sizeof(float), s, X, ldx, W, bias,
                     plain ? Y : pre, plain ? ldy : N, K, N);
  if (plain) {
    if (pre && hipMemcpy2DAsync(pre, (size_t)N * 4, Y, (size_t)ldy * 4, (size_t)N * 4, B, hipMemcpyDeviceToDevice, s) !=
                   hipSuccess)
      return MMSA_ERR_LAUNCH;
  } else {
    hipLaunchKernelGGL(dense_epilogue_kernel, dim3(B), dim3(256), 0, s, pre, gamma, beta, Y, ldy, stats, N, epilogue,
                       drop_p, seed, step);
  }
  return mmsa_launch_status();
}

// Backward of mmsa_dense_fwd.  dPre [B][N] is always written; dLn [B][N] for LN epilogues; dX optional
// (= or += by accumulate_dx); dW [N][K], db [N], dgamma/dbeta [N] are overwritten when non-null.
MMSA_EXPORT int mmsa_dense_bwd(const float* dY, int ldy, const float* X, int ldx, const float* pre, const float* stats,
                               const float* W, const float* gamma, const float* beta, float* dPre, float* dLn, float* dX,
                               int lddx, float* dW, float* db, float* dgamma, float* dbeta, int B, int K, int N,
                               int epilogue, float drop_p, unsigned long long seed, int accumulate_dx,
                               const float* step, void* stream) {
  MMSA_REQUIRE(dY && X && W && dPre && B > 0 && B <= 65535 && K > 0 && N > 0 && N <= 12 * 1024);
  MMSA_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_LAST);
  const bool ln = epilogue == EPI_LN_RELU || epilogue == EPI_RELU_LN || epilogue == EPI_LN || epilogue == EPI_LN_GELU;
  if (ln) MMSA_REQUIRE(pre && stats && gamma && beta && dLn);
  if (epilogue != EPI_NONE) MMSA_REQUIRE(pre != nullptr);
  hipStream_t s = mmsa_stream(stream);
  hipLaunchKernelGGL(dense_bwd_x_kernel, dim3(B, dX ? mmsa_cdiv(K, 64) : 1), dim3(256), (size_t)N * sizeof(float), s, dY,
                     ldy, pre, stats, W, gamma, beta, dPre, ln ? dLn : nullptr, dX, lddx, K, N, epilogue, drop_p, seed,
                     accumulate_dx, step);
  if (dW) {
    const long total = (long)N * K + N;
    hipLaunchKernelGGL(dense_bwd_w_kernel, dim3(mmsa_cdiv(total, 256)), dim3(256), 0, s, dPre, dLn, pre, stats, X, ldx,
                       dW, db, ln ? dgamma : nullptr, dbeta, B, K, N, epilogue);
  }
  return mmsa_launch_status();
}

// Parameter gradients of a dense layer alone, from the dPre / dLn rows a previous mmsa_dense_bwd (called with
// dW = null) left behind: lets a caller keep these off the critical path (they feed nothing but the optimiser).
MMSA_EXPORT int mmsa_dense_bwd_w(const float* dPre, const float* dLn, const float* pre, const float* stats,
                                 const float* X, int ldx, float* dW, float* db, float* dgamma, float* dbeta, int B,
                                 int K, int N, int epilogue, void* stream) {
  MMSA_REQUIRE(dPre && X && dW && B > 0 && K > 0 && N > 0 && ldx >= K);
  MMSA_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_LAST);
  const bool ln = epilogue == EPI_LN_RELU || epilogue == EPI_RELU_LN || epilogue == EPI_LN || epilogue == EPI_LN_GELU;
  if (ln) MMSA_REQUIRE(dLn && pre && stats && dgamma && dbeta);
  const long total = (long)N * K + N;
  hipLaunchKernelGGL(dense_bwd_w_kernel, dim3(mmsa_cdiv(total, 256)), dim3(256), 0, mmsa_stream(stream), dPre, dLn, pre,
                     stats, X, ldx, dW, db, ln ? dgamma : nullptr, dbeta, B, K, N, epilogue);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_ln_scale_fwd(const float* x, const float* gamma, const float* beta, const float* scale,
                                  const float* keep, float* y, int ldy, float* stats, int B, int N, void* stream) {
  MMSA_REQUIRE(x && gamma && beta && scale && y && stats && B > 0 && N > 0 && ldy >= N);
  hipLaunchKernelGGL(ln_scale_fwd_kernel, dim3(B), dim3(256), 0, mmsa_stream(stream), x, gamma, beta, scale, keep, y, ldy,
                     stats, N);
  return mmsa_launch_status();
}

// scratch: B*N + B floats
MMSA_EXPORT int mmsa_ln_scale_bwd(const float* dy, int ldy, const float* x, const float* gamma, const float* beta,
                                  const float* scale, const float* keep, const float* stats, float* dx, float* dgamma,
                                  float* dbeta, float* dscale, float* scratch, int B, int N, void* stream) {
  MMSA_REQUIRE(dy && x && gamma && beta && scale && stats && dx && dgamma && dbeta && dscale && scratch);
  MMSA_REQUIRE(B > 0 && N > 0 && ldy >= N);
  hipStream_t s = mmsa_stream(stream);
  float* dLn = scratch;
  float* part = scratch + (long)B * N;
  hipLaunchKernelGGL(ln_scale_bwd_kernel, dim3(B), dim3(256), 0, s, dy, ldy, x, gamma, beta, scale, keep, stats, dx, dLn,
                     part, N);
  hipLaunchKernelGGL(ln_scale_bwd_params_kernel, dim3(mmsa_cdiv(N, 256)), dim3(256), 0, s, dLn, x, stats, part, dgamma,
                     dbeta, dscale, B, N);
  return mmsa_launch_status();
}

// ---- the same backward for callers that cannot allocate or bring a workspace -------------------------------------
// The scratch of mmsa_ln_scale_bwd (dLn [B][N], then B partial sums) lives in a file-scope device array.  Each GPU
// holds its own copy of that array at its own address: the address is looked up once per device (with that device
// current) and kept; host code never uses the array's name as an address.  Every call on one GPU shares that GPU's
// array, so calls that overlap on one GPU race; nothing in it has to start at zero (every element read by the second
// kernel is written by the first one of the same call).
#define MMSA_LN_STATIC_FLOATS (1 << 20)
static __device__ __align__(256) float ln_static_scratch[MMSA_LN_STATIC_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_static_mu;
float* g_static_ptr[kMaxDevices] = {};

// device-side address of ln_static_scratch on the current device (nullptr on failure)
float* static_scratch_of_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (!g_static_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(ln_static_scratch)) != hipSuccess || !p) return nullptr;
    g_static_ptr[dev] = static_cast<float*>(p);
  }
  return g_static_ptr[dev];
}
}  // namespace

// Looks the current device's copy of the array up (once per device).  The first mmsa_ln_scale_bwd_static call on a
// device does the same by itself, but that look-up is a runtime call that a stream capture in progress does not
// allow: callers that capture call this first, outside the capture.
MMSA_EXPORT int mmsa_static_scratch_init(void) {
  return static_scratch_of_current_device() ? MMSA_OK : MMSA_ERR_LAUNCH;
}

MMSA_EXPORT int mmsa_ln_scale_bwd_static(const float* dy, int ldy, const float* x, const float* gamma,
                                         const float* beta, const float* scale, const float* keep,
                                         const float* stats, float* dx, float* dgamma, float* dbeta, float* dscale,
                                         int B, int N, void* stream) {
  MMSA_REQUIRE(dy && x && gamma && beta && scale && stats && dx && dgamma && dbeta && dscale);
  MMSA_REQUIRE(B > 0 && N > 0 && ldy >= N);
  if ((long)B * N + B > (long)MMSA_LN_STATIC_FLOATS) return MMSA_ERR_LIMIT;
  float* scratch = static_scratch_of_current_device();
  if (!scratch) return MMSA_ERR_LAUNCH;
  hipStream_t s = mmsa_stream(stream);
  float* dLn = scratch;
  float* part = scratch + (long)B * N;
  hipLaunchKernelGGL(ln_scale_bwd_kernel, dim3(B), dim3(256), 0, s, dy, ldy, x, gamma, beta, scale, keep, stats, dx, dLn,
                     part, N);
  hipLaunchKernelGGL(ln_scale_bwd_params_kernel, dim3(mmsa_cdiv(N, 256)), dim3(256), 0, s, dLn, x, stats, part, dgamma,
                     dbeta, dscale, B, N);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_fusion_mix_fwd(const float* P, const float* att, float* wts, float* fused, int B, int Hd,
                                    void* stream) {
  MMSA_REQUIRE(P && att && wts && fused && B > 0 && Hd > 0);
  hipLaunchKernelGGL(fusion_mix_fwd_kernel, dim3(mmsa_cdiv((long)B * Hd, 256)), dim3(256), 0, mmsa_stream(stream), P, att,
                     wts, fused, B, Hd);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_fusion_mix_bwd(const float* dfused, const float* P, const float* wts, float* dP, float* datt, int B,
                                    int Hd, void* stream) {
  MMSA_REQUIRE(dfused && P && wts && dP && datt && B > 0 && Hd > 0);
  hipLaunchKernelGGL(fusion_mix_bwd_kernel, dim3(B), dim3(256), 0, mmsa_stream(stream), dfused, P, wts, dP, datt, Hd);
  return mmsa_launch_status();
}

// Raises bit 1 of the sticky flag word `err` when a class label lies outside [0, C): the loss kernels treat such a row
// as having no target (no term in the loss, no -1 in the gradient), where torch's CrossEntropyLoss refuses it.  (Bit 0
// of the same word belongs to the embedding gather: a token id outside the table.)
__global__ __launch_bounds__(256) void check_labels_kernel(const long long* __restrict__ labels, int B, int C,
                                                           int* __restrict__ err) {
  bool bad = false;
  for (int b = threadIdx.x; b < B; b += 256) {
    const long long y = labels[b];
    bad |= y < 0 || y >= C;
  }
  if (bad) atomicOr(err, 2);
}

MMSA_EXPORT int mmsa_check_labels(const long long* labels, int B, int C, int* err, void* stream) {
  MMSA_REQUIRE(labels && err && B > 0 && C > 0);
  hipLaunchKernelGGL(check_labels_kernel, dim3(1), dim3(256), 0, mmsa_stream(stream), labels, B, C, err);
  return mmsa_launch_status();
}

// labels may be null (inference): probabilities only.  loss / dlogits optional.
MMSA_EXPORT int mmsa_softmax_ce(const float* logits, const long long* labels, float* probs, float* loss, float* dlogits,
                                int B, int C, void* stream) {
  MMSA_REQUIRE(logits && probs && B > 0 && C > 0);
  MMSA_REQUIRE(labels || (!loss && !dlogits));
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, mmsa_stream(stream), logits, labels, probs, loss, dlogits,
                     B, C);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_bn1d_relu_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, long long* nbt, float momentum, float eps, int training, float* y,
                                   float* coeff, int B, int N, float drop_p, unsigned long long seed,
                                   const float* step, void* stream) {
  MMSA_REQUIRE(x && gamma && beta && running_mean && running_var && y && coeff && B > 0 && N > 0);
  MMSA_REQUIRE(drop_p >= 0.f && drop_p < 1.f);
  hipLaunchKernelGGL(bn1d_relu_fwd_kernel, dim3(mmsa_cdiv(N, 64)), dim3(64), 0, mmsa_stream(stream), x, gamma, beta,
                     running_mean, running_var, nbt, momentum, eps, training, y, coeff, B, N, drop_p, seed, step);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_bn1d_relu_bwd(const float* dy, const float* x, const float* coeff, float* dx, float* dgamma,
                                   float* dbeta, int B, int N, int training, float drop_p, unsigned long long seed,
                                   const float* step, void* stream) {
  MMSA_REQUIRE(dy && x && coeff && dx && dgamma && dbeta && B > 0 && N > 0);
  hipLaunchKernelGGL(bn1d_relu_bwd_kernel, dim3(mmsa_cdiv(N, 64)), dim3(64), 0, mmsa_stream(stream), dy, x, coeff, dx,
                     dgamma, dbeta, B, N, training, drop_p, seed, step);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_dropout(const float* x, float* y, long n, float drop_p, unsigned long long seed, const float* step,
                             void* stream) {
  MMSA_REQUIRE(x && y && n > 0 && drop_p > 0.f && drop_p < 1.f);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, mmsa_stream(stream), x, y, n, drop_p, seed, step);
  return mmsa_launch_status();
}

// Attention fusion of two projected modalities (`Pretrained model implementations/early_fusion.py:136-156`).
MMSA_EXPORT int mmsa_pair_attn_mix_fwd(const float* M, const float* scores, float* wts, float* Wm, int B, int Fd,
                                       void* stream) {
  MMSA_REQUIRE(M && scores && wts && Wm && B > 0 && Fd > 0);
  hipLaunchKernelGGL(pair_attn_mix_fwd_kernel, dim3(mmsa_cdiv((long)B * Fd, 256)), dim3(256), 0, mmsa_stream(stream), M,
                     scores, wts, Wm, B, Fd);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_pair_attn_mix_bwd(const float* dWm, const float* M, const float* wts, float* dM, float* dscores,
                                       int B, int Fd, int accumulate, void* stream) {
  MMSA_REQUIRE(dWm && M && wts && dM && dscores && B > 0 && Fd > 0);
  hipLaunchKernelGGL(pair_attn_mix_bwd_kernel, dim3(B), dim3(256), 0, mmsa_stream(stream), dWm, M, wts, dM, dscores, Fd,
                     accumulate);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_gate_mix_fwd(const float* Wm, const float* gate, float* fused, int B, int Fd, void* stream) {
  MMSA_REQUIRE(Wm && gate && fused && B > 0 && Fd > 0);
  hipLaunchKernelGGL(gate_mix_fwd_kernel, dim3(mmsa_cdiv((long)B * Fd, 256)), dim3(256), 0, mmsa_stream(stream), Wm, gate,
                     fused, B, Fd);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_gate_mix_bwd(const float* dfused, const float* Wm, const float* gate, float* dWm, float* dgate,
                                  int B, int Fd, int accumulate, void* stream) {
  MMSA_REQUIRE(dfused && Wm && gate && (dWm || dgate) && B > 0 && Fd > 0);
  hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3(mmsa_cdiv((long)B * Fd, 256)), dim3(256), 0, mmsa_stream(stream), dfused,
                     Wm, gate, dWm, dgate, B, Fd, accumulate);
  return mmsa_launch_status();
}

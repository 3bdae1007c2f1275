// The fusion head of the 'both' model as two row-local kernels (forward, data-gradient backward) and one
// parameter-gradient kernel.
//
// Reference semantics: `CNN implementation/multimodal_model.py:216-225` (MultifactorFeatureNormalizer: LayerNorm x
// learnable scale per modality), `:263-285` (ModalityFusion: modality dropout, Linear+LayerNorm+ReLU+Dropout per
// modality, Linear(2H,2)+softmax, weighted sum), `:393-404` (fusion classifier: Linear, LN, ReLU, Dropout x2, Linear).
//
// Every layer of the head is row-local (LayerNorm, not BatchNorm), so one workgroup of 1024 threads carries one
// utterance through the whole stack with activations in LDS; the weights (0.8 MB) stream from L2.  The saved
// activations and the per-layer gradients are also written to HBM in the layout of the layer-by-layer kernels
// (dense.hip), so the parameter gradients, which sum over utterances, are formed by one further launch off the
// critical path.  Dropout masks come from the same counter-based generator as the layer-by-layer kernels with the
// same (seed, element) keys, so both paths draw identical masks.
#include "mmsa_common.h"

namespace {

constexpr int HT = 1024;     // threads per workgroup
constexpr int LPR = 16;      // lanes per output row in the forward mat-vec
constexpr float LN_EPS = 1e-5f;

// slots of the pointer table (device pointers; see include/mmsa.h)
enum Slot {
  S_AFEAT, S_TFEAT, S_AN_G, S_AN_B, S_A_SCALE, S_TN_G, S_TN_B, S_T_SCALE, S_KEEP_A, S_KEEP_T,
  S_WPA, S_BPA, S_GPA, S_BEPA, S_WPT, S_BPT, S_GPT, S_BEPT, S_WATT, S_BATT,
  S_W0, S_B0, S_G0, S_BE0, S_W1, S_B1, S_G1, S_BE1, S_W2, S_B2,
  S_AN, S_TN, S_AN_STATS, S_TN_STATS, S_PA_PRE, S_PA_STATS, S_PT_PRE, S_PT_STATS, S_P, S_ATT, S_WTS, S_FUSED,
  S_C0_PRE, S_C0_STATS, S_H1, S_C1_PRE, S_C1_STATS, S_H2, S_LOGITS,
  S_DLOGITS, S_C1_DPRE, S_C1_DLN, S_C0_DPRE, S_C0_DLN, S_ATT_DPRE, S_PA_DPRE, S_PA_DLN, S_PT_DPRE, S_PT_DLN,
  S_AN_DLN, S_TN_DLN, S_A_DSPART, S_T_DSPART, S_D_AFEAT, S_D_TFEAT,
  S_DWPA, S_DBPA, S_DGPA, S_DBEPA, S_DWPT, S_DBPT, S_DGPT, S_DBEPT, S_DWATT, S_DBATT,
  S_DW0, S_DB0, S_DG0, S_DBE0, S_DW1, S_DB1, S_DG1, S_DBE1, S_DW2, S_DB2,
  S_DAN_G, S_DAN_B, S_DA_SCALE, S_DTN_G, S_DTN_B, S_DT_SCALE,
  S_LABELS,   // optional (int64 [B]): the forward then also writes dlogits
  S_STEP,     // optional (float [1]): a device-resident step counter mixed into the dropout seed
  S_COUNT
};

struct HeadPtrs {
  void* p[S_COUNT];
};
struct HeadDims {
  int B, A, T, Hd, N0, N1, C;
  float drop_p;
  unsigned long long seed;
  float grad_scale;
};

__device__ __forceinline__ const float* cf(const HeadPtrs& h, int s) { return static_cast<const float*>(h.p[s]); }
// Dropout seed of this launch: the host's seed, advanced by the device-resident step counter when there is one, so
// that a replayed launch sequence (whose scalar arguments are frozen) still draws fresh masks every step.
__device__ __forceinline__ unsigned long long step_seed(const HeadPtrs& h, unsigned long long seed) {
  const float* step = static_cast<const float*>(h.p[S_STEP]);
  return step ? seed + (unsigned long long)step[0] * 0x9E3779B97F4A7C15ull : seed;
}
__device__ __forceinline__ float* mf(const HeadPtrs& h, int s) { return static_cast<float*>(h.p[s]); }

// out[n] = bias[n] + sum_k W[n][k] x[k] for n < N (x, out in LDS; K a multiple of 4, rows of W 16-byte aligned).
// Sixteen lanes share an output row and read it in 16-byte pieces (256 contiguous bytes per row and step), 64 rows
// per pass over the workgroup; two passes are in flight at a time.
// NIT: 64-column steps a row takes (K <= 64 NIT), so that only the pieces a layer has are requested: every request
// instruction costs the CU's address unit 16 cycles whether or not its lanes repeat a neighbour's address, and a
// 64 x 3 layer served with the 8 steps x 2 rows of the largest one spent 1.7 us on 255 redundant kilobytes.
template <int NIT>
__device__ __forceinline__ void matvec_steps(const float* __restrict__ W, const float* __restrict__ bias, const float* x,
                                             float* out, int K, int N) {
  const int t = threadIdx.x, sub = t % LPR, rloc = t / LPR;
  constexpr int RP = HT / LPR;
  const int wave_row0 = (t >> 6) * (64 / LPR);          // first row of this wave in a pass (uniform per wave)
  for (int n0 = 0; n0 < N; n0 += 2 * RP) {
    const int na = n0 + rloc, nb = n0 + RP + rloc;
    const bool wave_a = n0 + wave_row0 < N, wave_b = n0 + RP + wave_row0 < N;   // does this wave hold any row of the pass
    const float* wa = W + (long)min(na, N - 1) * K;
    const float* wb = W + (long)min(nb, N - 1) * K;
    float a = 0.f, b = 0.f;
    // all of a row pair's pieces are requested before any is used: one round trip to memory per pass instead of one
    // per step (within a wave that holds rows the loads are unconditional, from a clamped offset: a load under a
    // per-lane condition would be waited for on its own)
    f32x4 u[NIT], v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = min(4 * sub + it * 4 * LPR, K - 4);
      u[it] = v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (wave_a) u[it] = *reinterpret_cast<const f32x4*>(wa + k);
      if (wave_b) v[it] = *reinterpret_cast<const f32x4*>(wb + k);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = 4 * sub + it * 4 * LPR;
      const bool in = k < K;
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (in ? k : 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xe = in ? x4[e] : 0.f;
        a = fmaf(u[it][e], xe, a);
        b = fmaf(v[it][e], xe, b);
      }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      b += __shfl_xor(b, o, 64);
    }
    if (sub == 0) {
      if (na < N) out[na] = a + (bias ? bias[na] : 0.f);
      if (nb < N) out[nb] = b + (bias ? bias[nb] : 0.f);
    }
  }
}

__device__ __forceinline__ void matvec(const float* __restrict__ W, const float* __restrict__ bias, const float* x,
                                       float* out, int K, int N) {
  const int nit = (K + 4 * LPR - 1) / (4 * LPR);   // uniform
  if (nit <= 1) matvec_steps<1>(W, bias, x, out, K, N);
  else if (nit == 2) matvec_steps<2>(W, bias, x, out, K, N);
  else if (nit <= 4) matvec_steps<4>(W, bias, x, out, K, N);
  else if (nit <= 6) matvec_steps<6>(W, bias, x, out, K, N);
  else matvec_steps<8>(W, bias, x, out, K, N);
}

// dx[k] (= or +=) sum_n g[n] W[n][k] for k < K (g, dx in LDS; K a multiple of 4, rows of W 16-byte aligned).  A thread
// takes four consecutive k (16-byte pieces of the rows of W, coalesced), KQ = K / 4 rounded up to whole waves threads
// span a row, and the sum over n is split over G = HT / KQ thread groups (whole waves, so their row tests are uniform);
// sixteen row pieces are in flight per thread -- four times the bytes per round trip of a dword per thread, which is
// what the two projectors' 256 x 256 and 256 x 384 matrices need (one and two round trips instead of four and eight).
// The groups' partial sums meet in `part` ([4 * HT] floats of LDS) and are added in group order.
__device__ __forceinline__ void matvec_t(const float* __restrict__ W, const float* g, float* dx, float* part, int K,
                                         int N, bool accumulate) {
  const int t = threadIdx.x;
  const int KQ = ((K >> 2) + 63) & ~63;
  const int G = HT / KQ;
  const int kq = t % KQ, grp = t / KQ;
  const bool active = 4 * kq < K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = grp; n0 < N; n0 += 16 * G) {   // uniform per wave
    f32x4 wv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      wv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (n0 + i * G < N && active) wv[i] = *reinterpret_cast<const f32x4*>(W + (long)(n0 + i * G) * K + 4 * kq);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + i * G;
      const float gn = n < N ? g[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(gn, wv[i][e], acc[e]);
    }
  }
  __syncthreads();   // the previous user of `part` is done
  *reinterpret_cast<f32x4*>(part + 4 * t) = acc;
  __syncthreads();
  if (t < K) {
    float s = 0.f;
    for (int i = 0; i < G; ++i) s += part[4 * (i * KQ + (t >> 2)) + (t & 3)];
    dx[t] = accumulate ? dx[t] + s : s;
  }
  __syncthreads();
}

// Sum of v[0..N) (LDS) computed by every wave for itself (lanes stride over the row, then a wave reduction): all
// waves obtain the same value in the same order of operations and no workgroup barrier is needed.
__device__ __forceinline__ float wave_row_sum(const float* v, int N) {
  float s = 0.f;
  for (int n = threadIdx.x & 63; n < N; n += 64) s += v[n];
  return wave_sum(s);
}

// mean and 1/sqrt(var + eps) of v[0..N) (LDS), two passes as torch's LayerNorm, wave-redundant
__device__ __forceinline__ void row_stats(const float* v, int N, float& mean, float& rstd) {
  mean = wave_row_sum(v, N) / (float)N;
  float q = 0.f;
  for (int n = threadIdx.x & 63; n < N; n += 64) {
    const float d = v[n] - mean;
    q = fmaf(d, d, q);
  }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + LN_EPS);
}

// y = dropout(relu(LN(pre) * gamma + beta)); pre (LDS) is also stored to HBM with its statistics
__device__ __forceinline__ void ln_relu_drop(const float* pre, int N, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float* y_lds, float* y_g, float* pre_g,
                                             float* stats_g, int b, float drop_p, unsigned long long seed) {
  float mean, rstd;
  row_stats(pre, N, mean, rstd);
  if (threadIdx.x == 0) {
    stats_g[2 * b] = mean;
    stats_g[2 * b + 1] = rstd;
  }
  const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  for (int n = threadIdx.x; n < N; n += HT) {
    const float p = pre[n];
    pre_g[(long)b * N + n] = p;
    float v = fmaxf(fmaf((p - mean) * rstd, gamma[n], beta[n]), 0.f);
    if (drop_p > 0.f) v = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? v * ks : 0.f;
    y_lds[n] = v;
    y_g[n] = v;
  }
  __syncthreads();
}

// gradient through dropout, ReLU and LayerNorm of one Linear+LN+ReLU layer: d (LDS, gradient w.r.t. the layer output)
// -> g (LDS, gradient w.r.t. the linear output); dLn and dPre rows go to HBM for the parameter-gradient kernel.
// tmp: N floats of LDS scratch.
__device__ __forceinline__ void ln_relu_drop_bwd(const float* d_in, float* g, float* tmp, int N,
                                                 const float* __restrict__ pre_g, const float* __restrict__ stats_g,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* dln_g, float* dpre_g, int b, float drop_p,
                                                 unsigned long long seed) {
  const float mean = stats_g[2 * b], rstd = stats_g[2 * b + 1];
  const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  float xh_keep[1];   // N <= 512 < HT: one element per thread
  const int n = threadIdx.x;
  xh_keep[0] = 0.f;
  if (n < N) {
    float d = d_in[n];
    if (drop_p > 0.f) d = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? d * ks : 0.f;
    const float xh = (pre_g[(long)b * N + n] - mean) * rstd;
    const float y = fmaf(xh, gamma[n], beta[n]);
    d = y > 0.f ? d : 0.f;
    dln_g[(long)b * N + n] = d;
    const float dg = d * gamma[n];
    g[n] = dg;
    tmp[n] = dg * xh;
    xh_keep[0] = xh;
  }
  __syncthreads();
  const float s1 = wave_row_sum(g, N) / (float)N;
  const float s2 = wave_row_sum(tmp, N) / (float)N;
  __syncthreads();   // every wave has read g / tmp
  if (n < N) {
    const float v = rstd * (g[n] - s1 - xh_keep[0] * s2);
    g[n] = v;
    dpre_g[(long)b * N + n] = v;
  }
  __syncthreads();
}

__global__ __launch_bounds__(HT) void head_both_fwd_kernel(const HeadPtrs h, const HeadDims d) {
  __shared__ __attribute__((aligned(16))) float s_in[1024];    // normalised audio | text features
  __shared__ __attribute__((aligned(16))) float s_pre[1024];   // linear outputs of the current layer
  __shared__ __attribute__((aligned(16))) float s_P[1024];     // projected audio | text
  __shared__ __attribute__((aligned(16))) float s_x[512];      // fused / hidden vectors
  const int b = blockIdx.x, t = threadIdx.x;
  const int A = d.A, T = d.T, Hd = d.Hd;
  const unsigned long long seed = step_seed(h, d.seed);

  // feature normaliser: LayerNorm * scale * modality keep flag
  for (int n = t; n < A; n += HT) s_pre[n] = cf(h, S_AFEAT)[(long)b * A + n];
  for (int n = t; n < T; n += HT) s_pre[512 + n] = cf(h, S_TFEAT)[(long)b * T + n];
  __syncthreads();
  {
    float mean, rstd;
    row_stats(s_pre, A, mean, rstd);
    if (t == 0) { mf(h, S_AN_STATS)[2 * b] = mean; mf(h, S_AN_STATS)[2 * b + 1] = rstd; }
    const float sc = cf(h, S_A_SCALE)[0] * (h.p[S_KEEP_A] ? cf(h, S_KEEP_A)[0] : 1.f);
    for (int n = t; n < A; n += HT) {
      const float v = fmaf((s_pre[n] - mean) * rstd, cf(h, S_AN_G)[n], cf(h, S_AN_B)[n]) * sc;
      s_in[n] = v;
      mf(h, S_AN)[(long)b * A + n] = v;
    }
    row_stats(s_pre + 512, T, mean, rstd);
    if (t == 0) { mf(h, S_TN_STATS)[2 * b] = mean; mf(h, S_TN_STATS)[2 * b + 1] = rstd; }
    const float st = cf(h, S_T_SCALE)[0] * (h.p[S_KEEP_T] ? cf(h, S_KEEP_T)[0] : 1.f);
    for (int n = t; n < T; n += HT) {
      const float v = fmaf((s_pre[512 + n] - mean) * rstd, cf(h, S_TN_G)[n], cf(h, S_TN_B)[n]) * st;
      s_in[512 + n] = v;
      mf(h, S_TN)[(long)b * T + n] = v;
    }
  }
  __syncthreads();

  // the two projectors: Linear -> LayerNorm -> ReLU -> Dropout
  matvec(cf(h, S_WPA), cf(h, S_BPA), s_in, s_pre, A, Hd);
  matvec(cf(h, S_WPT), cf(h, S_BPT), s_in + 512, s_pre + 512, T, Hd);
  __syncthreads();
  float* Pg = mf(h, S_P) + (long)b * 2 * Hd;
  ln_relu_drop(s_pre, Hd, cf(h, S_GPA), cf(h, S_BEPA), s_P, Pg, mf(h, S_PA_PRE), mf(h, S_PA_STATS), b, d.drop_p,
               seed + 1);
  ln_relu_drop(s_pre + 512, Hd, cf(h, S_GPT), cf(h, S_BEPT), s_P + Hd, Pg + Hd, mf(h, S_PT_PRE), mf(h, S_PT_STATS), b,
               d.drop_p, seed + 2);

  // attention over the two modalities and the weighted sum
  matvec(cf(h, S_WATT), cf(h, S_BATT), s_P, s_pre, 2 * Hd, 2);
  __syncthreads();
  {
    const float a0 = s_pre[0], a1 = s_pre[1];
    const float m = fmaxf(a0, a1);
    const float e0 = expf(a0 - m), e1 = expf(a1 - m);
    const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
    if (t == 0) {
      mf(h, S_ATT)[2 * b] = a0; mf(h, S_ATT)[2 * b + 1] = a1;
      mf(h, S_WTS)[2 * b] = w0; mf(h, S_WTS)[2 * b + 1] = w1;
    }
    for (int n = t; n < Hd; n += HT) {
      const float v = s_P[n] * w0 + s_P[Hd + n] * w1;
      s_x[n] = v;
      mf(h, S_FUSED)[(long)b * Hd + n] = v;
    }
  }
  __syncthreads();

  // classifier
  matvec(cf(h, S_W0), cf(h, S_B0), s_x, s_pre, Hd, d.N0);
  __syncthreads();
  ln_relu_drop(s_pre, d.N0, cf(h, S_G0), cf(h, S_BE0), s_x, mf(h, S_H1) + (long)b * d.N0, mf(h, S_C0_PRE),
               mf(h, S_C0_STATS), b, d.drop_p, seed + 3);
  matvec(cf(h, S_W1), cf(h, S_B1), s_x, s_pre, d.N0, d.N1);
  __syncthreads();
  ln_relu_drop(s_pre, d.N1, cf(h, S_G1), cf(h, S_BE1), s_x, mf(h, S_H2) + (long)b * d.N1, mf(h, S_C1_PRE),
               mf(h, S_C1_STATS), b, 0.5f * d.drop_p, seed + 4);
  matvec(cf(h, S_W2), cf(h, S_B2), s_x, s_pre, d.N1, d.C);
  __syncthreads();
  for (int n = t; n < d.C; n += HT) mf(h, S_LOGITS)[(long)b * d.C + n] = s_pre[n];
  // with labels: the gradient of the mean cross-entropy w.r.t. this utterance's logits, (softmax - onehot) / B
  // (multimodal_model.py:452-456 and its autograd), so the backward launch can follow without a loss kernel between
  const long long* labels = static_cast<const long long*>(h.p[S_LABELS]);
  float* dl = mf(h, S_DLOGITS);
  if (labels && dl && t == 0) {
    float m = s_pre[0];
    for (int c = 1; c < d.C; ++c) m = fmaxf(m, s_pre[c]);
    float sum = 0.f;
    for (int c = 0; c < d.C; ++c) sum += expf(s_pre[c] - m);
    const float inv = 1.0f / sum;
    const long long y = labels[b];
    for (int c = 0; c < d.C; ++c) {
      const float p = expf(s_pre[c] - m) * inv;
      float g = (p - (c == y ? 1.f : 0.f)) / (float)d.B;
      if (d.grad_scale != 1.f) g *= d.grad_scale;
      dl[(long)b * d.C + c] = g;
    }
  }
}

__global__ __launch_bounds__(HT) void head_both_bwd_kernel(const HeadPtrs h, const HeadDims d) {
  __shared__ float s_d[1024];     // gradient w.r.t. the current layer's output
  __shared__ float s_g[1024];     // gradient w.r.t. the current layer's linear output
  __shared__ float s_dP[1024];    // gradient w.r.t. the projected audio | text
  __shared__ __attribute__((aligned(16))) float part[4 * HT];
  const int b = blockIdx.x, t = threadIdx.x;
  const int A = d.A, T = d.T, Hd = d.Hd, N0 = d.N0, N1 = d.N1, C = d.C;
  const unsigned long long seed = step_seed(h, d.seed);

  // logits layer (plain Linear): its dPre is the incoming gradient itself
  for (int n = t; n < C; n += HT) s_g[n] = cf(h, S_DLOGITS)[(long)b * C + n];
  __syncthreads();
  matvec_t(cf(h, S_W2), s_g, s_d, part, N1, C, false);
  ln_relu_drop_bwd(s_d, s_g, part, N1, cf(h, S_C1_PRE), cf(h, S_C1_STATS), cf(h, S_G1), cf(h, S_BE1),
                   mf(h, S_C1_DLN), mf(h, S_C1_DPRE), b, 0.5f * d.drop_p, seed + 4);
  matvec_t(cf(h, S_W1), s_g, s_d, part, N0, N1, false);
  ln_relu_drop_bwd(s_d, s_g, part, N0, cf(h, S_C0_PRE), cf(h, S_C0_STATS), cf(h, S_G0), cf(h, S_BE0),
                   mf(h, S_C0_DLN), mf(h, S_C0_DPRE), b, d.drop_p, seed + 3);
  matvec_t(cf(h, S_W0), s_g, s_d, part, Hd, N0, false);      // s_d = d fused

  // weighted sum and the softmax over the two attention scores
  {
    const float* Pg = cf(h, S_P) + (long)b * 2 * Hd;
    const float w0 = cf(h, S_WTS)[2 * b], w1 = cf(h, S_WTS)[2 * b + 1];
    for (int n = t; n < Hd; n += HT) {
      const float df = s_d[n];
      part[n] = df * Pg[n];
      part[512 + n] = df * Pg[Hd + n];
      s_dP[n] = df * w0;
      s_dP[Hd + n] = df * w1;
    }
    __syncthreads();
    const float d0 = wave_row_sum(part, Hd), d1 = wave_row_sum(part + 512, Hd);
    const float dot = w0 * d0 + w1 * d1;
    const float g0 = w0 * (d0 - dot), g1 = w1 * (d1 - dot);
    if (t == 0) {
      s_g[0] = g0; s_g[1] = g1;
      mf(h, S_ATT_DPRE)[2 * b] = g0; mf(h, S_ATT_DPRE)[2 * b + 1] = g1;
    }
    __syncthreads();
  }
  matvec_t(cf(h, S_WATT), s_g, s_dP, part, 2 * Hd, 2, true);

  // projectors, then the feature normaliser
  ln_relu_drop_bwd(s_dP, s_g, part, Hd, cf(h, S_PA_PRE), cf(h, S_PA_STATS), cf(h, S_GPA), cf(h, S_BEPA),
                   mf(h, S_PA_DLN), mf(h, S_PA_DPRE), b, d.drop_p, seed + 1);
  matvec_t(cf(h, S_WPA), s_g, s_d, part, A, Hd, false);       // s_d[0..A) = d an
  ln_relu_drop_bwd(s_dP + Hd, s_g, part, Hd, cf(h, S_PT_PRE), cf(h, S_PT_STATS), cf(h, S_GPT), cf(h, S_BEPT),
                   mf(h, S_PT_DLN), mf(h, S_PT_DPRE), b, d.drop_p, seed + 2);
  matvec_t(cf(h, S_WPT), s_g, s_d + 512, part, T, Hd, false);  // s_d[512..512+T) = d tn

#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    const int N = which ? T : A;
    const float* dy = s_d + (which ? 512 : 0);
    const float* x = cf(h, which ? S_TFEAT : S_AFEAT) + (long)b * N;
    const float* gam = cf(h, which ? S_TN_G : S_AN_G);
    const float* bet = cf(h, which ? S_TN_B : S_AN_B);
    const float* stats = cf(h, which ? S_TN_STATS : S_AN_STATS);
    const float* keep = cf(h, which ? S_KEEP_T : S_KEEP_A);
    float* dln = mf(h, which ? S_TN_DLN : S_AN_DLN) + (long)b * N;
    float* dx = mf(h, which ? S_D_TFEAT : S_D_AFEAT) + (long)b * N;
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const float kp = keep ? keep[0] : 1.f;
    const float sc = cf(h, which ? S_T_SCALE : S_A_SCALE)[0] * kp;
    for (int n = t; n < N; n += HT) {
      const float dv = dy[n];
      const float xh = (x[n] - mean) * rstd;
      const float dl = dv * sc;
      dln[n] = dl;
      s_g[n] = dl;
      s_dP[n] = dv * kp * fmaf(xh, gam[n], bet[n]);
      const float dg = dl * gam[n];
      part[n] = dg;
      part[512 + n] = dg * xh;
    }
    __syncthreads();
    const float s1 = wave_row_sum(part, N) / (float)N;
    const float s2 = wave_row_sum(part + 512, N) / (float)N;
    const float ds = wave_row_sum(s_dP, N);
    if (t == 0) mf(h, which ? S_T_DSPART : S_A_DSPART)[b] = ds;
    for (int n = t; n < N; n += HT) {
      const float xh = (x[n] - mean) * rstd;
      dx[n] = rstd * (s_g[n] * gam[n] - s1 - xh * s2);
    }
    __syncthreads();
  }
}

// Parameter gradients of the whole head in one launch (sums over the B utterances, fixed order).
// blockIdx.y selects the layer: 0..5 dense layers (pa, pt, att, c0, c1, c2), 6 / 7 the audio / text normaliser.
struct WLayer {
  int dpre, dln, pre, stats, x, ldx, dw, db, dg, dbe, K, N;
};

__global__ __launch_bounds__(256) void head_both_wgrad_kernel(const HeadPtrs h, const HeadDims d) {
  const int B = d.B, A = d.A, T = d.T, Hd = d.Hd, N0 = d.N0, N1 = d.N1, C = d.C;
  const int L = blockIdx.y;
  if (L >= 6) {
    const bool txt = L == 7;
    const int N = txt ? T : A;
    const float* dLn = cf(h, txt ? S_TN_DLN : S_AN_DLN);
    const float* x = cf(h, txt ? S_TFEAT : S_AFEAT);
    const float* stats = cf(h, txt ? S_TN_STATS : S_AN_STATS);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) {
      float sg = 0.f, sb = 0.f;
      for (int b = 0; b < B; ++b) {
        const float dv = dLn[(long)b * N + n];
        sg = fmaf(dv, (x[(long)b * N + n] - stats[2 * b]) * stats[2 * b + 1], sg);
        sb += dv;
      }
      mf(h, txt ? S_DTN_G : S_DAN_G)[n] = sg;
      mf(h, txt ? S_DTN_B : S_DAN_B)[n] = sb;
    }
    if (n == 0) {
      const float* part = cf(h, txt ? S_T_DSPART : S_A_DSPART);
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += part[b];
      mf(h, txt ? S_DT_SCALE : S_DA_SCALE)[0] = s;
    }
    return;
  }
  WLayer w;
  switch (L) {
    case 0: w = {S_PA_DPRE, S_PA_DLN, S_PA_PRE, S_PA_STATS, S_AN, A, S_DWPA, S_DBPA, S_DGPA, S_DBEPA, A, Hd}; break;
    case 1: w = {S_PT_DPRE, S_PT_DLN, S_PT_PRE, S_PT_STATS, S_TN, T, S_DWPT, S_DBPT, S_DGPT, S_DBEPT, T, Hd}; break;
    case 2: w = {S_ATT_DPRE, -1, -1, -1, S_P, 2 * Hd, S_DWATT, S_DBATT, -1, -1, 2 * Hd, 2}; break;
    case 3: w = {S_C0_DPRE, S_C0_DLN, S_C0_PRE, S_C0_STATS, S_FUSED, Hd, S_DW0, S_DB0, S_DG0, S_DBE0, Hd, N0}; break;
    case 4: w = {S_C1_DPRE, S_C1_DLN, S_C1_PRE, S_C1_STATS, S_H1, N0, S_DW1, S_DB1, S_DG1, S_DBE1, N0, N1}; break;
    default: w = {S_DLOGITS, -1, -1, -1, S_H2, N1, S_DW2, S_DB2, -1, -1, N1, C}; break;
  }
  const float* dPre = cf(h, w.dpre);
  const float* X = cf(h, w.x);
  const long nk = (long)w.N * w.K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nk + w.N; i += (long)gridDim.x * blockDim.x) {
    if (i < nk) {
      const int n = (int)(i / w.K), k = (int)(i - (long)n * w.K);
      float a = 0.f;
#pragma unroll 8
      for (int b = 0; b < B; ++b) a = fmaf(dPre[(long)b * w.N + n], X[(long)b * w.ldx + k], a);
      mf(h, w.dw)[i] = a;
    } else {
      const int n = (int)(i - nk);
      float sb = 0.f, sg = 0.f, sbt = 0.f;
      for (int b = 0; b < B; ++b) {
        sb += dPre[(long)b * w.N + n];
        if (w.dg >= 0) {
          const float xh = (cf(h, w.pre)[(long)b * w.N + n] - cf(h, w.stats)[2 * b]) * cf(h, w.stats)[2 * b + 1];
          const float dv = cf(h, w.dln)[(long)b * w.N + n];
          sg = fmaf(dv, xh, sg);
          sbt += dv;
        }
      }
      mf(h, w.db)[n] = sb;
      if (w.dg >= 0) { mf(h, w.dg)[n] = sg; mf(h, w.dbe)[n] = sbt; }
    }
  }
}

int fill(HeadPtrs& hp, HeadDims& hd, const void* const* ptrs, int nptrs, int first_required, int last_required, int B,
         int A, int T, int Hd, int N0, int N1, int C, float drop_p, unsigned long long seed, float grad_scale = 1.f) {
  if (!ptrs || nptrs != S_COUNT) return MMSA_ERR_ARG;
  if (B <= 0 || B > 65535 || A <= 0 || T <= 0 || Hd <= 0 || N0 <= 0 || N1 <= 0 || C <= 0) return MMSA_ERR_ARG;
  if (A > 512 || T > 512 || Hd > 512 || N0 > 512 || N1 > 512 || C > 512) return MMSA_ERR_LIMIT;
  if ((A | T | Hd | N0 | N1) & 3) return MMSA_ERR_ARG;        // 16-byte pieces of the weight rows
  if (!(drop_p >= 0.f && drop_p < 1.f)) return MMSA_ERR_ARG;
  for (int i = 0; i < S_COUNT; ++i) hp.p[i] = const_cast<void*>(ptrs[i]);
  for (int i = first_required; i <= last_required; ++i)
    if (!hp.p[i] && i != S_KEEP_A && i != S_KEEP_T && i != S_LABELS && i != S_STEP) return MMSA_ERR_ARG;
  static const int weights[] = {S_WPA, S_WPT, S_WATT, S_W0, S_W1, S_W2};
  for (int s : weights)
    if (reinterpret_cast<uintptr_t>(hp.p[s]) & 15) return MMSA_ERR_ARG;
  hd = HeadDims{B, A, T, Hd, N0, N1, C, drop_p, seed, grad_scale};
  return MMSA_OK;
}

}  // namespace

MMSA_EXPORT int mmsa_head_both_slots() { return S_COUNT; }

// Forward of the fusion head: slots a_feat .. b2 are read, slots an .. logits are written.  With the labels and
// dlogits slots set it also writes dlogits = grad_scale * (softmax(logits) - onehot(labels)) / B.
MMSA_EXPORT int mmsa_head_both_fwd(const void* const* ptrs_host, int nptrs, int B, int A, int T, int Hd, int N0, int N1,
                                   int C, float drop_p, unsigned long long seed, float grad_scale, void* stream) {
  HeadPtrs hp;
  HeadDims hd;
  const int rc = fill(hp, hd, ptrs_host, nptrs, S_AFEAT, S_LOGITS, B, A, T, Hd, N0, N1, C, drop_p, seed, grad_scale);
  if (rc != MMSA_OK) return rc;
  hipLaunchKernelGGL(head_both_fwd_kernel, dim3(B), dim3(HT), 0, mmsa_stream(stream), hp, hd);
  return mmsa_launch_status();
}

// Data-gradient backward: reads dlogits and what the forward saved; writes the per-layer dPre / dLn rows, the two
// per-utterance scale-gradient partials and d_afeat / d_tfeat.
MMSA_EXPORT int mmsa_head_both_bwd(const void* const* ptrs_host, int nptrs, int B, int A, int T, int Hd, int N0, int N1,
                                   int C, float drop_p, unsigned long long seed, void* stream) {
  HeadPtrs hp;
  HeadDims hd;
  const int rc = fill(hp, hd, ptrs_host, nptrs, S_AFEAT, S_D_TFEAT, B, A, T, Hd, N0, N1, C, drop_p, seed);
  if (rc != MMSA_OK) return rc;
  hipLaunchKernelGGL(head_both_bwd_kernel, dim3(B), dim3(HT), 0, mmsa_stream(stream), hp, hd);
  return mmsa_launch_status();
}

// Parameter gradients (all slots must be set): must follow mmsa_head_both_bwd in stream order.
MMSA_EXPORT int mmsa_head_both_bwd_params(const void* const* ptrs_host, int nptrs, int B, int A, int T, int Hd, int N0,
                                          int N1, int C, void* stream) {
  HeadPtrs hp;
  HeadDims hd;
  const int rc = fill(hp, hd, ptrs_host, nptrs, S_AFEAT, S_COUNT - 1, B, A, T, Hd, N0, N1, C, 0.f, 0);
  if (rc != MMSA_OK) return rc;
  hipLaunchKernelGGL(head_both_wgrad_kernel, dim3(96, 8), dim3(256), 0, mmsa_stream(stream), hp, hd);
  return mmsa_launch_status();
}

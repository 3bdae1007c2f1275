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
static_assert(LPR == 16, "the mat-vec adds a row's partial sums over one 16-lane row of the cross-lane data path");

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

// Timing aid (builds with -DMMSA_LAB_HEAD_STAMPS only): workgroup 0 leaves the 100 MHz clock at stage boundaries in
// the buffer of the audio projector's weight gradient, which neither the forward nor the data backward touches.
#ifdef MMSA_LAB_HEAD_STAMPS
#define HEAD_STAMP(i)                                                                                   \
  do {                                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0 && h.p[S_DWPA]) {                                           \
      static_cast<unsigned long long*>(h.p[S_DWPA])[i] = wall_clock64();                                \
      static_cast<unsigned long long*>(h.p[S_DWPA])[32 + (i)] = clock64();                              \
    }                                                                                                   \
  } while (0)
#else
#define HEAD_STAMP(i) do {} while (0)
#endif

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
struct RowPieces {
  f32x4 u[NIT], v[NIT];
};

// A pass pair (rows n0 .. n0 + 127) in two halves, so that a caller can put other work between the request for the
// weights and their use.  All of a row pair's pieces are requested before any is used: one round trip to memory per
// pass instead of one per step (within a wave that holds rows the loads are unconditional, from a clamped offset: a
// load under a per-lane condition would be waited for on its own).
template <int NIT>
__device__ __forceinline__ void rows_request_at(RowPieces<NIT>& r, const float* __restrict__ W, int K, int N, int n0,
                                                int it0) {
  const int t = threadIdx.x, sub = t % LPR, rloc = t / LPR;
  constexpr int RP = HT / LPR;
  const int wave_row0 = (t >> 6) * (64 / LPR);          // first row of this wave in a pass (uniform per wave)
  const int na = n0 + rloc, nb = n0 + RP + rloc;
  const bool wave_a = n0 + wave_row0 < N, wave_b = n0 + RP + wave_row0 < N;   // does this wave hold any row of the pass
  const float* wa = W + (long)min(na, N - 1) * K;
  const float* wb = W + (long)min(nb, N - 1) * K;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = min(4 * sub + (it0 + it) * 4 * LPR, K - 4);
    r.u[it] = r.v[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wave_a) r.u[it] = *reinterpret_cast<const f32x4*>(wa + k);
    if (wave_b) r.v[it] = *reinterpret_cast<const f32x4*>(wb + k);
  }
}

template <int NIT>
__device__ __forceinline__ void rows_accumulate_at(const RowPieces<NIT>& r, const float* x, float& a, float& b, int K,
                                                   int it0) {
  const int sub = threadIdx.x % LPR;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = 4 * sub + (it0 + it) * 4 * LPR;
    const bool in = k < K;
    const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (in ? k : 0));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xe = in ? x4[e] : 0.f;
      a = fmaf(r.u[it][e], xe, a);
      b = fmaf(r.v[it][e], xe, b);
    }
  }
}

__device__ __forceinline__ void rows_store(float a, float b, const float* bias, float* out, int N, int n0) {
  const int t = threadIdx.x, sub = t % LPR, rloc = t / LPR;
  const int na = n0 + rloc, nb = n0 + HT / LPR + rloc;
  a = row16_sum(a);
  b = row16_sum(b);
  if (sub == 0) {
    if (na < N) out[na] = a + (bias ? bias[na] : 0.f);
    if (nb < N) out[nb] = b + (bias ? bias[nb] : 0.f);
  }
}

template <int NIT>
__device__ __forceinline__ void rows_request(RowPieces<NIT>& r, const float* __restrict__ W, int K, int N, int n0) {
  rows_request_at(r, W, K, N, n0, 0);
}

// The request without a test per wave (rows beyond the matrix are asked for at the last row and never stored):
// straight-line code whose registers the compiler can plan, for the requests that stay in flight across other stages.
template <int NIT>
__device__ __forceinline__ void rows_request_ahead(RowPieces<NIT>& r, const float* __restrict__ W, int K, int N, int n0) {
  const int t = threadIdx.x, sub = t % LPR, rloc = t / LPR;
  const float* wa = W + (long)min(n0 + rloc, N - 1) * K;
  const float* wb = W + (long)min(n0 + HT / LPR + rloc, N - 1) * K;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int k = min(4 * sub + it * 4 * LPR, K - 4);
    r.u[it] = *reinterpret_cast<const f32x4*>(wa + k);
    r.v[it] = *reinterpret_cast<const f32x4*>(wb + k);
  }
}

template <int NIT>
__device__ __forceinline__ void rows_finish(const RowPieces<NIT>& r, const float* bias, const float* x, float* out, int K,
                                            int N, int n0) {
  float a = 0.f, b = 0.f;
  rows_accumulate_at(r, x, a, b, K, 0);
  rows_store(a, b, bias, out, N, n0);
}

template <int NIT>
__device__ __forceinline__ void matvec_steps(const float* __restrict__ W, const float* __restrict__ bias, const float* x,
                                             float* out, int K, int N) {
  for (int n0 = 0; n0 < N; n0 += 2 * (HT / LPR)) {
    RowPieces<NIT> r;
    rows_request(r, W, K, N, n0);
    rows_finish(r, bias, x, out, K, N, n0);
  }
}

__device__ __forceinline__ void matvec(const float* __restrict__ W, const float* __restrict__ bias, const float* x,
                                       float* out, int K, int N) {
  const int nit = (K + 4 * LPR - 1) / (4 * LPR);   // uniform
  if (nit <= 1) matvec_steps<1>(W, bias, x, out, K, N);
  else if (nit == 2) matvec_steps<2>(W, bias, x, out, K, N);
  else if (nit <= 4) matvec_steps<4>(W, bias, x, out, K, N);
  else if (nit <= 6) matvec_steps<6>(W, bias, x, out, K, N);
  else if (nit <= 8) matvec_steps<8>(W, bias, x, out, K, N);
  else {
    // rows longer than 512 columns (the attention layer over two projections wider than 256): eight steps at a time
    for (int n0 = 0; n0 < N; n0 += 2 * (HT / LPR)) {
      float a = 0.f, b = 0.f;
      for (int it0 = 0; it0 < nit; it0 += 8) {
        RowPieces<8> r;
        rows_request_at(r, W, K, N, n0, it0);
        rows_accumulate_at(r, x, a, b, K, it0);
      }
      rows_store(a, b, bias, out, N, n0);
    }
  }
}

// The same for a matrix already in LDS: the pieces are read as they are used (no registers set aside for a batch of
// requests), in the same order, so the sums are those of matvec.
__device__ __forceinline__ void matvec_lds(const float* W, const float* bias, const float* x, float* out, int K, int N) {
  const int nit = (K + 4 * LPR - 1) / (4 * LPR);
  const int wave_row0 = (threadIdx.x >> 6) * (64 / LPR);
  for (int n0 = 0; n0 < N; n0 += 2 * (HT / LPR)) {
    if (n0 + wave_row0 >= N) break;             // this wave holds no row of the pass (uniform)
    float a = 0.f, b = 0.f;
    for (int it = 0; it < nit; ++it) {
      RowPieces<1> r;
      rows_request_at(r, W, K, N, n0, it);
      rows_accumulate_at(r, x, a, b, K, it);
    }
    rows_store(a, b, bias, out, N, n0);
  }
}

// dx[k] (= or +=) sum_n g[n] W[n][k] for k < K (g, dx in LDS; K a multiple of 4, rows of W 16-byte aligned).  A thread
// takes four consecutive k (16-byte pieces of the rows of W, coalesced), KQ = K / 4 rounded up to whole waves threads
// span a row, and the sum over n is split over G = HT / KQ thread groups (whole waves, so their row tests are uniform);
// sixteen row pieces are in flight per thread -- four times the bytes per round trip of a dword per thread, which is
// what the two projectors' 256 x 256 and 256 x 384 matrices need (one and two round trips instead of four and eight).
// The groups' partial sums meet in `part` ([4 * HT] floats of LDS) and are added in group order.
template <int NP>
struct ColPieces {
  f32x4 w[NP];
};

// One round of the transposed mat-vec (rows n0 + i G, i < NP, of this thread's group) as request and use.
template <int NP>
__device__ __forceinline__ void cols_request(ColPieces<NP>& c, const float* __restrict__ W, int K, int N, int n0) {
  const int KQ = ((K >> 2) + 63) & ~63;
  const int G = HT / KQ;
  const int kq = threadIdx.x % KQ;
  const bool active = 4 * kq < K;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    c.w[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (n0 + i * G < N && active) c.w[i] = *reinterpret_cast<const f32x4*>(W + (long)(n0 + i * G) * K + 4 * kq);
  }
}

// The same without a test per piece: rows and columns beyond the matrix are asked for at a clamped place (cols_use
// multiplies such a row by zero and cols_reduce never reads such a column).  Straight-line code whose registers the
// compiler can plan, for the requests that stay in flight across other stages.
template <int NP>
__device__ __forceinline__ void cols_request_ahead(ColPieces<NP>& c, const float* __restrict__ W, int K, int N, int n0) {
  const int KQ = ((K >> 2) + 63) & ~63;
  const int G = HT / KQ;
  const int kq = min((int)(threadIdx.x % KQ), (K >> 2) - 1);
#pragma unroll
  for (int i = 0; i < NP; ++i)
    c.w[i] = *reinterpret_cast<const f32x4*>(W + (long)min(n0 + i * G, N - 1) * K + 4 * kq);
}

template <int NP>
__device__ __forceinline__ void cols_use(const ColPieces<NP>& c, const float* g, f32x4& acc, int K, int N, int n0) {
  const int G = HT / (((K >> 2) + 63) & ~63);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int n = n0 + i * G;
    const float gn = n < N ? g[n] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaf(gn, c.w[i][e], acc[e]);
  }
}

// The groups' partial sums meet in `part` and are added in group order.
__device__ __forceinline__ void cols_reduce(const f32x4 acc, float* dx, float* part, int K, bool accumulate) {
  const int t = threadIdx.x;
  const int KQ = ((K >> 2) + 63) & ~63;
  const int G = HT / KQ;
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

// NP: row pieces in flight per thread (a thread meets its rows in ascending order whatever NP is, so the sums do not
// depend on it): 16 for a matrix in memory, fewer for one already in LDS, where the registers buy nothing.
template <int NP = 16>
__device__ __forceinline__ void matvec_t(const float* __restrict__ W, const float* g, float* dx, float* part, int K,
                                         int N, bool accumulate) {
  const int KQ = ((K >> 2) + 63) & ~63;
  const int G = HT / KQ;
  const int grp = threadIdx.x / KQ;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = grp; n0 < N; n0 += NP * G) {   // uniform per wave
    ColPieces<NP> c;
    cols_request(c, W, K, N, n0);
    cols_use(c, g, acc, K, N, n0);
  }
  cols_reduce(acc, dx, part, K, accumulate);
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

// The head's stages are short and the workgroup's sixteen waves share four vector ALUs, so a stage's time is its
// instruction count times the waves that run it.  Stages over a row of N <= 512 elements therefore run on the waves
// that hold an element (one per thread) and the others go straight to the barrier; the row sums are computed by each
// of those waves for itself.
__device__ __forceinline__ bool wave_holds(int n, int N) { return (n & ~63) < N; }   // uniform per wave

// y = dropout(relu(LN(pre) * gamma + beta)); pre (LDS, N <= 512) is also stored to HBM with its statistics
__device__ __forceinline__ void ln_relu_drop(const float* pre, int N, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float* y_lds, float* y_g, float* pre_g,
                                             float* stats_g, int b, float drop_p, unsigned long long seed) {
  const int n = threadIdx.x;
  if (wave_holds(n, N)) {
    float mean, rstd;
    row_stats(pre, N, mean, rstd);
    if (n == 0) {
      stats_g[2 * b] = mean;
      stats_g[2 * b + 1] = rstd;
    }
    if (n < N) {
      const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
      const float p = pre[n];
      pre_g[(long)b * N + n] = p;
      float v = fmaxf(fmaf((p - mean) * rstd, gamma[n], beta[n]), 0.f);
      if (drop_p > 0.f) v = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? v * ks : 0.f;
      y_lds[n] = v;
      y_g[n] = v;
    }
  }
  __syncthreads();
}

// gradient through dropout, ReLU and LayerNorm of one Linear+LN+ReLU layer: d (LDS, gradient w.r.t. the layer output)
// -> g (LDS, gradient w.r.t. the linear output); dLn and dPre rows go to HBM for the parameter-gradient kernel.
// pre_row: this utterance's row of the saved linear outputs; tmp: 1024 floats of LDS scratch; N <= 512.
__device__ __forceinline__ void ln_relu_drop_bwd(const float* d_in, float* g, float* tmp, int N, const float* pre_row,
                                                 const float mean, const float rstd, const float* gamma,
                                                 const float* beta, float* dln_g, float* dpre_g, int b, float drop_p,
                                                 unsigned long long seed) {
  const int n = threadIdx.x;
  const bool on = wave_holds(n, N);
  float xh = 0.f, dg = 0.f;
  if (on && n < N) {
    const float ks = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
    float d = d_in[n];
    if (drop_p > 0.f) d = hash_uniform(seed, (unsigned long long)b * N + n) >= drop_p ? d * ks : 0.f;
    xh = (pre_row[n] - mean) * rstd;
    const float y = fmaf(xh, gamma[n], beta[n]);
    d = y > 0.f ? d : 0.f;
    dln_g[(long)b * N + n] = d;
    dg = d * gamma[n];
    tmp[n] = dg;
    tmp[512 + n] = dg * xh;
  }
  __syncthreads();
  if (on) {
    const float s1 = wave_row_sum(tmp, N) / (float)N;
    const float s2 = wave_row_sum(tmp + 512, N) / (float)N;
    if (n < N) {
      const float v = rstd * (dg - s1 - xh * s2);
      g[n] = v;
      dpre_g[(long)b * N + n] = v;
    }
  }
  __syncthreads();
}

// logits of utterance b (LDS) -> HBM; with labels also the gradient of the mean cross-entropy w.r.t. them,
// (softmax - onehot) / B (multimodal_model.py:452-456 and its autograd), so the backward launch can follow without a
// loss kernel between.  One wave, a class per lane.
__device__ __forceinline__ void write_logits(const HeadPtrs& h, const HeadDims& d, const float* s_pre, int b) {
  const int t = threadIdx.x;
  if (t >= 64) return;
  for (int n = t; n < d.C; n += 64) mf(h, S_LOGITS)[(long)b * d.C + n] = s_pre[n];
  const long long* labels = static_cast<const long long*>(h.p[S_LABELS]);
  float* dl = mf(h, S_DLOGITS);
  if (labels && dl) {
    // every lane of the wave is enabled here (the tests above are uniform); lanes beyond C carry -inf and 0
    float m = -INFINITY;
    for (int c = t; c < d.C; c += 64) m = fmaxf(m, s_pre[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = t; c < d.C; c += 64) sum += expf(s_pre[c] - m);
    const float inv = 1.0f / wave_sum(sum);
    const long long y = labels[b];
    for (int c = t; c < d.C; c += 64) {
      const float p = expf(s_pre[c] - m) * inv;
      float g = (p - (c == y ? 1.f : 0.f)) / (float)d.B;
      if (d.grad_scale != 1.f) g *= d.grad_scale;
      dl[(long)b * d.C + c] = g;
    }
  }
}

__global__ __launch_bounds__(HT) void head_both_fwd_kernel(const HeadPtrs h, const HeadDims d) {
  __shared__ __attribute__((aligned(16))) float s_in[1024];    // normalised audio | text features
  __shared__ __attribute__((aligned(16))) float s_pre[1024];   // linear outputs of the current layer
  __shared__ __attribute__((aligned(16))) float s_P[1024];     // projected audio | text
  __shared__ __attribute__((aligned(16))) float s_x[512];      // fused / hidden vectors
  const int b = blockIdx.x, t = threadIdx.x;
  const int A = d.A, T = d.T, Hd = d.Hd;
  const unsigned long long seed = step_seed(h, d.seed);
  HEAD_STAMP(0);

  // feature normaliser: LayerNorm * scale * modality keep flag
  for (int n = t; n < A; n += HT) s_pre[n] = cf(h, S_AFEAT)[(long)b * A + n];
  for (int n = t; n < T; n += HT) s_pre[512 + n] = cf(h, S_TFEAT)[(long)b * T + n];
  __syncthreads();
  HEAD_STAMP(1);
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
  HEAD_STAMP(2);

  // the two projectors: Linear -> LayerNorm -> ReLU -> Dropout
  matvec(cf(h, S_WPA), cf(h, S_BPA), s_in, s_pre, A, Hd);
  matvec(cf(h, S_WPT), cf(h, S_BPT), s_in + 512, s_pre + 512, T, Hd);
  __syncthreads();
  HEAD_STAMP(3);
  float* Pg = mf(h, S_P) + (long)b * 2 * Hd;
  ln_relu_drop(s_pre, Hd, cf(h, S_GPA), cf(h, S_BEPA), s_P, Pg, mf(h, S_PA_PRE), mf(h, S_PA_STATS), b, d.drop_p,
               seed + 1);
  HEAD_STAMP(4);
  ln_relu_drop(s_pre + 512, Hd, cf(h, S_GPT), cf(h, S_BEPT), s_P + Hd, Pg + Hd, mf(h, S_PT_PRE), mf(h, S_PT_STATS), b,
               d.drop_p, seed + 2);
  HEAD_STAMP(5);

  // attention over the two modalities and the weighted sum
  matvec(cf(h, S_WATT), cf(h, S_BATT), s_P, s_pre, 2 * Hd, 2);
  __syncthreads();
  HEAD_STAMP(6);
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
  HEAD_STAMP(7);

  // classifier
  matvec(cf(h, S_W0), cf(h, S_B0), s_x, s_pre, Hd, d.N0);
  __syncthreads();
  HEAD_STAMP(8);
  ln_relu_drop(s_pre, d.N0, cf(h, S_G0), cf(h, S_BE0), s_x, mf(h, S_H1) + (long)b * d.N0, mf(h, S_C0_PRE),
               mf(h, S_C0_STATS), b, d.drop_p, seed + 3);
  HEAD_STAMP(9);
  matvec(cf(h, S_W1), cf(h, S_B1), s_x, s_pre, d.N0, d.N1);
  __syncthreads();
  HEAD_STAMP(10);
  ln_relu_drop(s_pre, d.N1, cf(h, S_G1), cf(h, S_BE1), s_x, mf(h, S_H2) + (long)b * d.N1, mf(h, S_C1_PRE),
               mf(h, S_C1_STATS), b, 0.5f * d.drop_p, seed + 4);
  HEAD_STAMP(11);
  matvec(cf(h, S_W2), cf(h, S_B2), s_x, s_pre, d.N1, d.C);
  __syncthreads();
  HEAD_STAMP(12);
  write_logits(h, d, s_pre, b);
  HEAD_STAMP(13);
}

__global__ __launch_bounds__(HT) void head_both_bwd_kernel(const HeadPtrs h, const HeadDims d) {
  __shared__ float s_d[1024];     // gradient w.r.t. the current layer's output
  __shared__ float s_g[1024];     // gradient w.r.t. the current layer's linear output
  __shared__ float s_dP[1024];    // gradient w.r.t. the projected audio | text
  __shared__ __attribute__((aligned(16))) float part[4 * HT];
  const int b = blockIdx.x, t = threadIdx.x;
  const int A = d.A, T = d.T, Hd = d.Hd, N0 = d.N0, N1 = d.N1, C = d.C;
  const unsigned long long seed = step_seed(h, d.seed);
  HEAD_STAMP(0);

  auto row = [&](int slot, int N) { return cf(h, slot) + (long)b * N; };      // this utterance's row of a saved tensor
  auto stat = [&](int slot, int i) { return cf(h, slot)[2 * b + i]; };          // its mean (0) / reciprocal deviation (1)

  // logits layer (plain Linear): its dPre is the incoming gradient itself
  for (int n = t; n < C; n += HT) s_g[n] = cf(h, S_DLOGITS)[(long)b * C + n];
  __syncthreads();
  HEAD_STAMP(1);
  matvec_t(cf(h, S_W2), s_g, s_d, part, N1, C, false);
  HEAD_STAMP(2);
  ln_relu_drop_bwd(s_d, s_g, part, N1, row(S_C1_PRE, N1), stat(S_C1_STATS, 0), stat(S_C1_STATS, 1),
                   cf(h, S_G1), cf(h, S_BE1),
                   mf(h, S_C1_DLN), mf(h, S_C1_DPRE), b, 0.5f * d.drop_p, seed + 4);
  HEAD_STAMP(3);
  matvec_t(cf(h, S_W1), s_g, s_d, part, N0, N1, false);
  HEAD_STAMP(4);
  ln_relu_drop_bwd(s_d, s_g, part, N0, row(S_C0_PRE, N0), stat(S_C0_STATS, 0), stat(S_C0_STATS, 1),
                   cf(h, S_G0), cf(h, S_BE0),
                   mf(h, S_C0_DLN), mf(h, S_C0_DPRE), b, d.drop_p, seed + 3);
  HEAD_STAMP(5);
  matvec_t(cf(h, S_W0), s_g, s_d, part, Hd, N0, false);      // s_d = d fused
  HEAD_STAMP(6);

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
  HEAD_STAMP(7);
  matvec_t(cf(h, S_WATT), s_g, s_dP, part, 2 * Hd, 2, true);
  HEAD_STAMP(8);

  // projectors, then the feature normaliser
  ln_relu_drop_bwd(s_dP, s_g, part, Hd, row(S_PA_PRE, Hd), stat(S_PA_STATS, 0), stat(S_PA_STATS, 1),
                   cf(h, S_GPA), cf(h, S_BEPA),
                   mf(h, S_PA_DLN), mf(h, S_PA_DPRE), b, d.drop_p, seed + 1);
  HEAD_STAMP(9);
  matvec_t(cf(h, S_WPA), s_g, s_d, part, A, Hd, false);       // s_d[0..A) = d an
  HEAD_STAMP(10);
  ln_relu_drop_bwd(s_dP + Hd, s_g, part, Hd, row(S_PT_PRE, Hd), stat(S_PT_STATS, 0), stat(S_PT_STATS, 1),
                   cf(h, S_GPT), cf(h, S_BEPT),
                   mf(h, S_PT_DLN), mf(h, S_PT_DPRE), b, d.drop_p, seed + 2);
  HEAD_STAMP(11);
  matvec_t(cf(h, S_WPT), s_g, s_d + 512, part, T, Hd, false);  // s_d[512..512+T) = d tn
  HEAD_STAMP(12);

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
    HEAD_STAMP(13 + which);
  }
}

// ---- the same two kernels with every operand requested ahead of its use ------------------------------------------
//
// The head is a chain of some fourteen dependent stages, and in the kernels above most stages open with a request to
// memory that cannot be issued earlier only because of where it stands in the program: a layer's LayerNorm vectors and
// bias, an utterance's saved rows, the next layer's weights.  A round trip costs 1-2 us when the weights have been
// pushed out of the L2 by the convolutions, which is the case inside a training step.  None of these addresses depend
// on the data, so the kernels below ask for
//   * every short vector (parameters, this utterance's saved rows) and the three small matrices (attention, second and
//     third classifier layer) at kernel start, into LDS, in one round trip, and
//   * each pass of the three large matrices a stage or two ahead of its use, into registers,
// with the arithmetic of every stage, and so every result bit, unchanged.  They need the dimensions of the reference
// model or smaller (see staged_ok); other shapes take the kernels above.

constexpr int SM_WATT = 2048;    // floats of LDS for the attention matrix [2][2 Hd]
constexpr int SM_W1 = 8192;      // ... the second classifier layer [N1][N0]
constexpr int SM_W2 = 1024;      // ... the logits layer [C][N1]
constexpr int SM_SLOT = 512;     // ... each short vector (every one of them has at most 512 elements)

// The table of pointers is a kernel argument, and the compiler fetches each pointer from the argument segment where it
// is first used, a few at a time, waiting for each fetch: nine scalar round trips in a row, 2.5 us, before the first
// barrier of the forward.  A use of all the pointers a phase needs in one statement makes them one fetch.
// (One statement per group: the compiler waits before each such statement for what it uses.)
#define HEAD_P(i) "s"(h.p[i])
#define NEED_SLOTS_6(a0, a1, a2, a3, a4, a5) \
  asm volatile("" ::HEAD_P(a0), HEAD_P(a1), HEAD_P(a2), HEAD_P(a3), HEAD_P(a4), HEAD_P(a5))
#define NEED_SLOTS_13(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12)                                     \
  asm volatile("" ::HEAD_P(a0), HEAD_P(a1), HEAD_P(a2), HEAD_P(a3), HEAD_P(a4), HEAD_P(a5), HEAD_P(a6), HEAD_P(a7), \
               HEAD_P(a8), HEAD_P(a9), HEAD_P(a10), HEAD_P(a11), HEAD_P(a12))
#define NEED_SLOTS_24(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, \
                      a21, a22, a23)                                                                                 \
  asm volatile("" ::HEAD_P(a0), HEAD_P(a1), HEAD_P(a2), HEAD_P(a3), HEAD_P(a4), HEAD_P(a5), HEAD_P(a6), HEAD_P(a7),   \
               HEAD_P(a8), HEAD_P(a9), HEAD_P(a10), HEAD_P(a11), HEAD_P(a12), HEAD_P(a13), HEAD_P(a14), HEAD_P(a15), \
               HEAD_P(a16), HEAD_P(a17), HEAD_P(a18), HEAD_P(a19), HEAD_P(a20), HEAD_P(a21), HEAD_P(a22), HEAD_P(a23))

// Short vectors -> LDS, all requests in flight together, in as few instructions as it takes (the sixteen waves share
// four vector ALUs, so every instruction here is paid four times over): the eight waves whose threads can hold an
// element ask for every vector from a clamped index and store to a fixed place per vector, with no test per vector.
// Tests, even wave-uniform ones, would also split the requests over as many basic blocks, and the compiler then
// fetches each vector's pointer from the kernel arguments inside its block and waits for it there: twenty scalar
// round trips one after the other against one.
template <int NE>
__device__ __forceinline__ void stage_request(float (&v)[NE], const float* const (&src)[NE], const int (&cnt)[NE]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int e = 0; e < NE; ++e) v[e] = src[e][min(t, cnt[e] - 1)];
}
template <int NE>
__device__ __forceinline__ void stage_store(const float (&v)[NE], float* dst) {
  const int t = threadIdx.x;   // < SM_SLOT
#pragma unroll
  for (int e = 0; e < NE; ++e) dst[e * SM_SLOT + t] = v[e];
}

// A small matrix (n4 16-byte pieces, at most TRIPS per thread) -> LDS
template <int TRIPS>
__device__ __forceinline__ void small_request(f32x4 (&q)[TRIPS], const float* __restrict__ W, int n4) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < TRIPS; ++j) q[j] = reinterpret_cast<const f32x4*>(W)[min(t + j * HT, n4 - 1)];
}
template <int TRIPS>
__device__ __forceinline__ void small_store(const f32x4 (&q)[TRIPS], float* dst, int n4) {
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < TRIPS; ++j)
    if (t + j * HT < n4) reinterpret_cast<f32x4*>(dst)[t + j * HT] = q[j];
}

__global__ __launch_bounds__(HT) void head_both_fwd_staged_kernel(const HeadPtrs h, const HeadDims d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* s_in = lds;                 // [1024] normalised audio | text features
  float* s_pre = s_in + 1024;        // [1024] linear outputs of the current layer
  float* s_P = s_pre + 1024;         // [1024] projected audio | text
  float* s_x = s_P + 1024;           // [512]  fused / hidden vectors
  float* s_watt = s_x + 512;
  float* s_w1 = s_watt + SM_WATT;
  float* s_w2 = s_w1 + SM_W1;
  float* s_par = s_w2 + SM_W2;
  const int b = blockIdx.x, t = threadIdx.x;
  const int A = d.A, T = d.T, Hd = d.Hd, N0 = d.N0, N1 = d.N1, C = d.C;
  HEAD_STAMP(0);
  NEED_SLOTS_24(S_AN_G, S_AN_B, S_TN_G, S_TN_B, S_BPA, S_GPA, S_BEPA, S_BPT, S_GPT, S_BEPT, S_BATT, S_B0, S_G0, S_BE0,
                S_B1, S_G1, S_BE1, S_B2, S_AFEAT, S_TFEAT, S_WATT, S_W1, S_W2, S_WPA);

  enum { P_AN_G, P_AN_B, P_TN_G, P_TN_B, P_BPA, P_GPA, P_BEPA, P_BPT, P_GPT, P_BEPT, P_BATT, P_B0, P_G0, P_BE0,
         P_B1, P_G1, P_BE1, P_B2, P_AF, P_TF, NE };
  const float* const src[NE] = {cf(h, S_AN_G), cf(h, S_AN_B), cf(h, S_TN_G), cf(h, S_TN_B), cf(h, S_BPA),
                                cf(h, S_GPA), cf(h, S_BEPA), cf(h, S_BPT), cf(h, S_GPT), cf(h, S_BEPT),
                                cf(h, S_BATT), cf(h, S_B0), cf(h, S_G0), cf(h, S_BE0), cf(h, S_B1),
                                cf(h, S_G1), cf(h, S_BE1), cf(h, S_B2), cf(h, S_AFEAT) + (long)b * A,
                                cf(h, S_TFEAT) + (long)b * T};
  const int cnt[NE] = {A, A, T, T, Hd, Hd, Hd, Hd, Hd, Hd, 2, N0, N0, N0, N1, N1, N1, C, A, T};
  static_assert(NE == 20, "FWD_STAGED_LDS holds twenty short vectors");
  RowPieces<4> pa0, pa1;
  unsigned long long seed;
  float sc, st;
  {
    float pv[NE];
    f32x4 q_att[1], q_w1[2], q_w2[1];
    if (t < SM_SLOT) stage_request(pv, src, cnt);
    small_request(q_att, cf(h, S_WATT), Hd);
    small_request(q_w1, cf(h, S_W1), (N0 * N1) >> 2);
    small_request(q_w2, cf(h, S_W2), (N1 * C) >> 2);
    rows_request_ahead(pa0, cf(h, S_WPA), A, Hd, 0);
    // the scalars now: each is a pointer from the kernel arguments and then a value through it, two scalar round
    // trips that pass while the vector requests above are in flight
    NEED_SLOTS_6(S_STEP, S_A_SCALE, S_T_SCALE, S_KEEP_A, S_KEEP_T, S_WPT);
    seed = step_seed(h, d.seed);
    sc = cf(h, S_A_SCALE)[0] * (h.p[S_KEEP_A] ? cf(h, S_KEEP_A)[0] : 1.f);
    st = cf(h, S_T_SCALE)[0] * (h.p[S_KEEP_T] ? cf(h, S_KEEP_T)[0] : 1.f);
    if (t < SM_SLOT) stage_store(pv, s_par);
    small_store(q_att, s_watt, Hd);
    small_store(q_w1, s_w1, (N0 * N1) >> 2);
    small_store(q_w2, s_w2, (N1 * C) >> 2);
  }
  rows_request_ahead(pa1, cf(h, S_WPA), A, Hd, 128);
  auto par = [&](int e) { return s_par + e * SM_SLOT; };
  __syncthreads();
  HEAD_STAMP(1);

  // feature normaliser: LayerNorm * scale * modality keep flag; the audio row on the first eight waves, the text row
  // on the other eight, side by side
  {
    const bool txt = t >= 512;             // uniform per wave
    const int n = t & 511, N = txt ? T : A;
    if (wave_holds(n, N)) {
      const float* x = par(txt ? P_TF : P_AF);
      float mean, rstd;
      row_stats(x, N, mean, rstd);
      float* stats = mf(h, txt ? S_TN_STATS : S_AN_STATS);
      if (n == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
      if (n < N) {
        const float v = fmaf((x[n] - mean) * rstd, par(txt ? P_TN_G : P_AN_G)[n], par(txt ? P_TN_B : P_AN_B)[n]) *
                        (txt ? st : sc);
        s_in[(txt ? 512 : 0) + n] = v;
        mf(h, txt ? S_TN : S_AN)[(long)b * N + n] = v;
      }
    }
  }
  __syncthreads();
  HEAD_STAMP(2);

  // the two projectors: Linear -> LayerNorm -> ReLU -> Dropout; the text projector's rows arrive under the audio
  // projector's LayerNorm, the first classifier layer's under the text projector's
  float* Pg = mf(h, S_P) + (long)b * 2 * Hd;
  rows_finish(pa0, par(P_BPA), s_in, s_pre, A, Hd, 0);
  RowPieces<6> pt0, pt1;
  rows_request_ahead(pt0, cf(h, S_WPT), T, Hd, 0);
  rows_finish(pa1, par(P_BPA), s_in, s_pre, A, Hd, 128);
  __syncthreads();
  HEAD_STAMP(3);
  // (one pass pair ahead is enough: its 192 KB take the CU's 64 bytes a clock longer to arrive than this LayerNorm
  // takes to run, so the memory pipe does not idle, and the LayerNorm keeps its registers)
  ln_relu_drop(s_pre, Hd, par(P_GPA), par(P_BEPA), s_P, Pg, mf(h, S_PA_PRE), mf(h, S_PA_STATS), b, d.drop_p, seed + 1);
  HEAD_STAMP(4);
  rows_request_ahead(pt1, cf(h, S_WPT), T, Hd, 128);
  rows_finish(pt0, par(P_BPT), s_in + 512, s_pre + 512, T, Hd, 0);
  RowPieces<4> w0;
  rows_request_ahead(w0, cf(h, S_W0), Hd, N0, 0);
  rows_finish(pt1, par(P_BPT), s_in + 512, s_pre + 512, T, Hd, 128);
  __syncthreads();
  ln_relu_drop(s_pre + 512, Hd, par(P_GPT), par(P_BEPT), s_P + Hd, Pg + Hd, mf(h, S_PT_PRE), mf(h, S_PT_STATS), b,
               d.drop_p, seed + 2);
  HEAD_STAMP(5);

  // attention over the two modalities and the weighted sum
  matvec_lds(s_watt, par(P_BATT), s_P, s_pre, 2 * Hd, 2);
  __syncthreads();
  HEAD_STAMP(6);
  if (wave_holds(t, Hd)) {
    const float a0 = s_pre[0], a1 = s_pre[1];
    const float m = fmaxf(a0, a1);
    const float e0 = expf(a0 - m), e1 = expf(a1 - m);
    const float w0a = e0 / (e0 + e1), w1a = e1 / (e0 + e1);
    if (t == 0) {
      mf(h, S_ATT)[2 * b] = a0; mf(h, S_ATT)[2 * b + 1] = a1;
      mf(h, S_WTS)[2 * b] = w0a; mf(h, S_WTS)[2 * b + 1] = w1a;
    }
    if (t < Hd) {
      const float v = s_P[t] * w0a + s_P[Hd + t] * w1a;
      s_x[t] = v;
      mf(h, S_FUSED)[(long)b * Hd + t] = v;
    }
  }
  __syncthreads();
  HEAD_STAMP(7);

  // classifier
  rows_finish(w0, par(P_B0), s_x, s_pre, Hd, N0, 0);
  __syncthreads();
  HEAD_STAMP(8);
  ln_relu_drop(s_pre, N0, par(P_G0), par(P_BE0), s_x, mf(h, S_H1) + (long)b * N0, mf(h, S_C0_PRE), mf(h, S_C0_STATS), b,
               d.drop_p, seed + 3);
  HEAD_STAMP(9);
  matvec_lds(s_w1, par(P_B1), s_x, s_pre, N0, N1);
  __syncthreads();
  HEAD_STAMP(10);
  ln_relu_drop(s_pre, N1, par(P_G1), par(P_BE1), s_x, mf(h, S_H2) + (long)b * N1, mf(h, S_C1_PRE), mf(h, S_C1_STATS), b,
               0.5f * d.drop_p, seed + 4);
  HEAD_STAMP(11);
  matvec_lds(s_w2, par(P_B2), s_x, s_pre, N1, C);
  __syncthreads();
  HEAD_STAMP(12);
  write_logits(h, d, s_pre, b);
  HEAD_STAMP(13);
}

__global__ __launch_bounds__(HT) void head_both_bwd_staged_kernel(const HeadPtrs h, const HeadDims d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* s_d = lds;                  // [1024] gradient w.r.t. the current layer's output
  float* s_g = s_d + 1024;           // [1024] gradient w.r.t. the current layer's linear output (audio side)
  float* s_g2 = s_g + 1024;          // [512]  the same of the text projector
  float* s_dP = s_g2 + 512;          // [1024] gradient w.r.t. the projected audio | text
  float* part = s_dP + 1024;         // [4 HT]
  float* s_watt = part + 4 * HT;
  float* s_w1 = s_watt + SM_WATT;
  float* s_w2 = s_w1 + SM_W1;
  float* s_par = s_w2 + SM_W2;
  const int b = blockIdx.x, t = threadIdx.x;
  const int A = d.A, T = d.T, Hd = d.Hd, N0 = d.N0, N1 = d.N1, C = d.C;
  HEAD_STAMP(0);
  NEED_SLOTS_24(S_G1, S_BE1, S_G0, S_BE0, S_GPA, S_BEPA, S_GPT, S_BEPT, S_AN_G, S_AN_B, S_TN_G, S_TN_B, S_DLOGITS,
                S_C1_PRE, S_C0_PRE, S_PA_PRE, S_PT_PRE, S_P, S_AFEAT, S_TFEAT, S_WATT, S_W1, S_W2, S_W0);
  enum { P_G1, P_BE1, P_G0, P_BE0, P_GPA, P_BEPA, P_GPT, P_BEPT, P_AN_G, P_AN_B, P_TN_G, P_TN_B,
         P_DLOGITS, P_C1_PRE, P_C0_PRE, P_PA_PRE, P_PT_PRE, P_P, P_AF, P_TF, NE };
  const float* const src[NE] = {cf(h, S_G1), cf(h, S_BE1), cf(h, S_G0), cf(h, S_BE0), cf(h, S_GPA), cf(h, S_BEPA),
                                cf(h, S_GPT), cf(h, S_BEPT), cf(h, S_AN_G), cf(h, S_AN_B), cf(h, S_TN_G),
                                cf(h, S_TN_B), cf(h, S_DLOGITS) + (long)b * C, cf(h, S_C1_PRE) + (long)b * N1,
                                cf(h, S_C0_PRE) + (long)b * N0, cf(h, S_PA_PRE) + (long)b * Hd,
                                cf(h, S_PT_PRE) + (long)b * Hd, cf(h, S_P) + (long)b * 2 * Hd,
                                cf(h, S_AFEAT) + (long)b * A, cf(h, S_TFEAT) + (long)b * T};
  const int cnt[NE] = {N1, N1, N0, N0, Hd, Hd, Hd, Hd, A, A, T, T, C, N1, N0, Hd, Hd, 2 * Hd, A, T};
  static_assert(NE == 20, "BWD_STAGED_LDS holds twenty short vectors");
  float pv[NE];
  f32x4 q_att[1], q_w1[2], q_w2[1];
  if (t < SM_SLOT) stage_request(pv, src, cnt);
  small_request(q_att, cf(h, S_WATT), Hd);
  small_request(q_w1, cf(h, S_W1), (N0 * N1) >> 2);
  small_request(q_w2, cf(h, S_W2), (N1 * C) >> 2);
  // first classifier layer (N0 <= 128 rows over the 16 thread groups of a 256-wide row: 8 pieces a thread)
  const int grp0 = t / (((Hd >> 2) + 63) & ~63);
  ColPieces<8> c0;
  cols_request_ahead(c0, cf(h, S_W0), Hd, N0, grp0);
  // the scalars now: each is a pointer from the kernel arguments and then a value through it, two scalar round trips
  // that pass while the vector requests above are in flight
  NEED_SLOTS_13(S_STEP, S_C1_STATS, S_C0_STATS, S_PA_STATS, S_PT_STATS, S_AN_STATS, S_TN_STATS, S_WTS, S_KEEP_A,
                S_KEEP_T, S_A_SCALE, S_T_SCALE, S_WPA);
  const unsigned long long seed = step_seed(h, d.seed);
  auto stat = [&](int slot, int i) { return cf(h, slot)[2 * b + i]; };
  const float c1_mean = stat(S_C1_STATS, 0), c1_rstd = stat(S_C1_STATS, 1);
  const float c0_mean = stat(S_C0_STATS, 0), c0_rstd = stat(S_C0_STATS, 1);
  const float pa_mean = stat(S_PA_STATS, 0), pa_rstd = stat(S_PA_STATS, 1);
  const float pt_mean = stat(S_PT_STATS, 0), pt_rstd = stat(S_PT_STATS, 1);
  const float an_mean = stat(S_AN_STATS, 0), an_rstd = stat(S_AN_STATS, 1);
  const float tn_mean = stat(S_TN_STATS, 0), tn_rstd = stat(S_TN_STATS, 1);
  const float w0a = stat(S_WTS, 0), w1a = stat(S_WTS, 1);
  const float kpa = h.p[S_KEEP_A] ? cf(h, S_KEEP_A)[0] : 1.f, kpt = h.p[S_KEEP_T] ? cf(h, S_KEEP_T)[0] : 1.f;
  const float sca = cf(h, S_A_SCALE)[0] * kpa, sct = cf(h, S_T_SCALE)[0] * kpt;
  if (t < SM_SLOT) stage_store(pv, s_par);
  small_store(q_att, s_watt, Hd);
  small_store(q_w1, s_w1, (N0 * N1) >> 2);
  small_store(q_w2, s_w2, (N1 * C) >> 2);
  auto par = [&](int e) { return s_par + e * SM_SLOT; };
  __syncthreads();
  HEAD_STAMP(1);

  // logits layer (plain Linear): its dPre is the incoming gradient itself
  matvec_t<4>(s_w2, par(P_DLOGITS), s_d, part, N1, C, false);
  HEAD_STAMP(2);
  ln_relu_drop_bwd(s_d, s_g, part, N1, par(P_C1_PRE), c1_mean, c1_rstd, par(P_G1), par(P_BE1), mf(h, S_C1_DLN),
                   mf(h, S_C1_DPRE), b, 0.5f * d.drop_p, seed + 4);
  HEAD_STAMP(3);
  matvec_t<4>(s_w1, s_g, s_d, part, N0, N1, false);
  HEAD_STAMP(4);
  ln_relu_drop_bwd(s_d, s_g, part, N0, par(P_C0_PRE), c0_mean, c0_rstd, par(P_G0), par(P_BE0), mf(h, S_C0_DLN),
                   mf(h, S_C0_DPRE), b, d.drop_p, seed + 3);
  HEAD_STAMP(5);
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    cols_use(c0, s_g, acc, Hd, N0, grp0);
    cols_reduce(acc, s_d, part, Hd, false);                    // s_d = d fused
  }
  HEAD_STAMP(6);
  // the audio projector's matrix (Hd <= 256 rows over the 16 groups of a 256-wide row: one round of 16 pieces)
  const int grpa = t / (((A >> 2) + 63) & ~63);
  ColPieces<16> ca;
  cols_request_ahead(ca, cf(h, S_WPA), A, Hd, grpa);

  // weighted sum and the softmax over the two attention scores
  {
    const bool on = wave_holds(t, Hd);
    if (on && t < Hd) {
      const float* Pl = par(P_P);
      const float df = s_d[t];
      part[t] = df * Pl[t];
      part[512 + t] = df * Pl[Hd + t];
      s_dP[t] = df * w0a;
      s_dP[Hd + t] = df * w1a;
    }
    __syncthreads();
    if (t < 64) {
      const float d0 = wave_row_sum(part, Hd), d1 = wave_row_sum(part + 512, Hd);
      const float dot = w0a * d0 + w1a * d1;
      const float g0 = w0a * (d0 - dot), g1 = w1a * (d1 - dot);
      if (t == 0) {
        s_g[0] = g0; s_g[1] = g1;
        mf(h, S_ATT_DPRE)[2 * b] = g0; mf(h, S_ATT_DPRE)[2 * b + 1] = g1;
      }
    }
    __syncthreads();
  }
  HEAD_STAMP(7);
  matvec_t<2>(s_watt, s_g, s_dP, part, 2 * Hd, 2, true);
  HEAD_STAMP(8);

  // projectors, then the feature normaliser
  ln_relu_drop_bwd(s_dP, s_g, part, Hd, par(P_PA_PRE), pa_mean, pa_rstd, par(P_GPA), par(P_BEPA), mf(h, S_PA_DLN),
                   mf(h, S_PA_DPRE), b, d.drop_p, seed + 1);
  HEAD_STAMP(9);
  f32x4 acc_a = {0.f, 0.f, 0.f, 0.f};
  cols_use(ca, s_g, acc_a, A, Hd, grpa);
  // the text projector's matrix: rounds of 16 pieces, the first in flight under its LayerNorm backward
  const int KQt = ((T >> 2) + 63) & ~63, Gt = HT / KQt, grpt = t / KQt;
  ColPieces<16> ct;
  cols_request_ahead(ct, cf(h, S_WPT), T, Hd, grpt);
  cols_reduce(acc_a, s_d, part, A, false);                     // s_d[0..A) = d an
  HEAD_STAMP(10);
  ln_relu_drop_bwd(s_dP + Hd, s_g2, part, Hd, par(P_PT_PRE), pt_mean, pt_rstd, par(P_GPT), par(P_BEPT), mf(h, S_PT_DLN),
                   mf(h, S_PT_DPRE), b, d.drop_p, seed + 2);
  HEAD_STAMP(11);
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    cols_use(ct, s_g2, acc, T, Hd, grpt);
    for (int n0 = grpt + 16 * Gt; n0 < Hd; n0 += 16 * Gt) {   // uniform per wave
      cols_request(ct, cf(h, S_WPT), T, Hd, n0);
      cols_use(ct, s_g2, acc, T, Hd, n0);
    }
    cols_reduce(acc, s_d + 512, part, T, false);               // s_d[512..512+T) = d tn
  }
  HEAD_STAMP(12);

  // feature normaliser: the audio row on the first eight waves, the text row on the other eight, side by side
  {
    const bool txt = t >= 512;             // uniform per wave
    const int n = t & 511, N = txt ? T : A;
    const bool on = wave_holds(n, N);
    const float* x = par(txt ? P_TF : P_AF);
    const float* gam = par(txt ? P_TN_G : P_AN_G);
    const float mean = txt ? tn_mean : an_mean, rstd = txt ? tn_rstd : an_rstd;
    float* sums = part + (txt ? 2048 : 0);          // three rows of 512
    float xh = 0.f, dl = 0.f;
    if (on && n < N) {
      const float kp = txt ? kpt : kpa;
      const float dv = s_d[(txt ? 512 : 0) + n];
      xh = (x[n] - mean) * rstd;
      dl = dv * (txt ? sct : sca);
      mf(h, txt ? S_TN_DLN : S_AN_DLN)[(long)b * N + n] = dl;
      const float dg = dl * gam[n];
      sums[n] = dg;
      sums[512 + n] = dg * xh;
      sums[1024 + n] = dv * kp * fmaf(xh, gam[n], par(txt ? P_TN_B : P_AN_B)[n]);
    }
    __syncthreads();
    if (on) {
      const float s1 = wave_row_sum(sums, N) / (float)N;
      const float s2 = wave_row_sum(sums + 512, N) / (float)N;
      const float ds = wave_row_sum(sums + 1024, N);
      if (n == 0) mf(h, txt ? S_T_DSPART : S_A_DSPART)[b] = ds;
      if (n < N) mf(h, txt ? S_D_TFEAT : S_D_AFEAT)[(long)b * N + n] = rstd * (dl * gam[n] - s1 - xh * s2);
    }
  }
  HEAD_STAMP(13);
  HEAD_STAMP(14);
}

constexpr size_t FWD_STAGED_LDS = (1024 * 3 + 512 + SM_WATT + SM_W1 + SM_W2 + 20 * SM_SLOT) * sizeof(float);
constexpr size_t BWD_STAGED_LDS = (1024 * 3 + 512 + 4 * HT + SM_WATT + SM_W1 + SM_W2 + 20 * SM_SLOT) * sizeof(float);

// Do the staged kernels' fixed pass counts and LDS regions hold these dimensions?
bool staged_ok(const HeadDims& d) {
  const char* e = getenv("MMSA_HEAD_STAGED");   // 0: always the plain kernels (read at every launch: a test can switch)
  if (e && e[0] == '0') return false;
  if (d.A > 256 || d.T > 384 || d.Hd > 256 || d.N0 > 128) return false;
  if (4 * d.Hd > SM_WATT || d.N0 * d.N1 > SM_W1 || d.N1 * d.C > SM_W2) return false;
  if (d.N1 > SM_SLOT || d.C > SM_SLOT) return false;            // (the other short vectors are bounded above)
  return true;
}

template <typename K>
int launch_staged(K kernel, size_t lds, PerDeviceOnce& attr_set, int B, hipStream_t s, const HeadPtrs& hp, const HeadDims& hd) {
  if (attr_set.pending()) {   // raising the dynamic-LDS limit is sticky per kernel and device
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    attr_set.done();
  }
  hipLaunchKernelGGL(kernel, dim3(B), dim3(HT), lds, s, hp, hd);
  return mmsa_launch_status();
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
  if (staged_ok(hd)) {
    static PerDeviceOnce attr_set;
    return launch_staged(head_both_fwd_staged_kernel, FWD_STAGED_LDS, attr_set, B, mmsa_stream(stream), hp, hd);
  }
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
  if (staged_ok(hd)) {
    static PerDeviceOnce attr_set;
    return launch_staged(head_both_bwd_staged_kernel, BWD_STAGED_LDS, attr_set, B, mmsa_stream(stream), hp, hd);
  }
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

// Text branch kernels other than the dense contraction (which conv.hip provides):
// token-embedding gather, bias + ReLU + max-over-time with argmax, and the backward pass, which is sparse:
// only the arg-max window of each (utterance, filter) carries gradient.
//
// Reference semantics: `CNN implementation/multimodal_model.py:149` (Embedding, padding_idx 0: the padding row
// takes part in the forward like any other row but receives no data gradient), `:186-197` (Conv1d -> ReLU ->
// max over time -> concat).  The gather moves rows verbatim (bit-exact).
#include "mmsa_common.h"

namespace {

// one wave per token row; E floats per row (E % 4 == 0)
__global__ __launch_bounds__(256) void gather_rows_kernel(const long long* __restrict__ ids,
                                                          const float* __restrict__ table, float* __restrict__ out,
                                                          long ntok, int E, long vocab, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= ntok) return;
  long long id = ids[row];
  const bool ok = id >= 0 && id < vocab;
  if (!ok) {
    if (lane == 0 && err) *err = 1;
  }
  const f32x4* src = reinterpret_cast<const f32x4*>(table + (ok ? id : 0) * E);
  f32x4* dst = reinterpret_cast<f32x4*>(out + row * E);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < E / 4; i += 64) dst[i] = ok ? src[i] : zero;
}

// y[b][t][f] raw conv output (row length F), valid t < Lout.  feat[b*ldf + f] = max_t relu(y + bias),
// arg[b*ldf + f] = first t attaining it.
__global__ void bias_relu_maxtime_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                         float* __restrict__ feat, int* __restrict__ arg, int B, int L, int Lout,
                                         int F, int ldf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * F) return;
  const int b = i / F, f = i - b * F;
  const float bs = bias[f];
  float best = -1.f;
  int bt = 0;
  for (int t = 0; t < Lout; ++t) {
    const float v = fmaxf(y[((long)b * L + t) * F + f] + bs, 0.f);
    if (v > best) { best = v; bt = t; }
  }
  feat[(long)b * ldf + f] = best;
  arg[(long)b * ldf + f] = bt;
}

// dW[f][c][j] = sum_b g[b][f] * emb[b][t*(b,f)+j][c];  dbias[f] = sum_b g[b][f];  g = dfeat where feat > 0.
// grid = F blocks, block = E threads (c).
template <int K>
__global__ void text_bwd_w_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat,
                                  const int* __restrict__ arg, const float* __restrict__ emb, float* __restrict__ dW,
                                  float* __restrict__ dbias, int B, int L, int E, int ldf) {
  const int f = blockIdx.x, c = threadIdx.x;
  float acc[K];
#pragma unroll
  for (int j = 0; j < K; ++j) acc[j] = 0.f;
  float gb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float ft = feat[(long)b * ldf + f];
    if (ft > 0.f) {  // uniform across the block
      const float g = dfeat[(long)b * ldf + f];
      const int t0 = arg[(long)b * ldf + f];
      gb += g;
      if (c < E) {
        const float* e = emb + ((long)b * L + t0) * E + c;
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = fmaf(g, e[(long)j * E], acc[j]);
      }
    }
  }
  if (c < E) {
#pragma unroll
    for (int j = 0; j < K; ++j) dW[((long)f * E + c) * K + j] = acc[j];
  }
  if (c == 0) dbias[f] = gb;
}

struct TextConvs {
  int nconv;
  int k[4];
  int feat_off[4];
  const float* wp[4];  // packed [j][F][E]
};

// dEmb rows accumulated in LDS per utterance, then scatter-added into the table gradient (skip padding rows).
// grid = B, block = E threads; LDS = TL*E floats + TL ints.
__global__ void text_bwd_emb_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat,
                                    const int* __restrict__ arg, const long long* __restrict__ ids, TextConvs cv,
                                    float* __restrict__ dtable, int L, int E, int F, int ldf, long padding_idx, int TL) {
  extern __shared__ float acc[];  // [TL][E]
  int* touched = reinterpret_cast<int*>(acc + (long)TL * E);
  const int b = blockIdx.x, c = threadIdx.x;
  for (int t0 = 0; t0 < L; t0 += TL) {
    for (int t = 0; t < TL; ++t) acc[(long)t * E + c] = 0.f;
    if (c < TL) touched[c] = 0;
    __syncthreads();
    for (int ci = 0; ci < cv.nconv; ++ci) {
      const int K = cv.k[ci];
      const float* wp = cv.wp[ci];
      for (int f = 0; f < F; ++f) {
        const long o = (long)b * ldf + cv.feat_off[ci] + f;
        if (feat[o] > 0.f) {  // uniform
          const float g = dfeat[o];
          const int ts = arg[o];
          for (int j = 0; j < K; ++j) {
            const int t = ts + j - t0;
            if (t >= 0 && t < TL) {
              acc[(long)t * E + c] = fmaf(g, wp[((long)j * F + f) * E + c], acc[(long)t * E + c]);
              touched[t] = 1;  // same value from every thread
            }
          }
        }
      }
    }
    __syncthreads();
    for (int t = 0; t < TL && t0 + t < L; ++t) {
      if (touched[t]) {
        const long long id = ids[(long)b * L + t0 + t];
        if (id != padding_idx) atomicAdd(dtable + id * E + c, acc[(long)t * E + c]);
      }
    }
    __syncthreads();
  }
}

}  // namespace

// out[ntok][E] = table[ids[ntok]][:].  err (optional device int) is set to 1 if any id is outside [0, vocab).
MMSA_EXPORT int mmsa_embedding_gather(const long long* ids, const float* table, float* out, long ntok, int E, long vocab,
                                      int* err, void* stream) {
  MMSA_REQUIRE(ids && table && out && ntok > 0 && E > 0 && E % 4 == 0 && vocab > 0);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(mmsa_cdiv(ntok, 4)), dim3(256), 0, mmsa_stream(stream), ids, table, out,
                     ntok, E, vocab, err);
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_bias_relu_maxtime(const float* y, const float* bias, float* feat, int* arg, int B, int L, int Lout,
                                       int F, int ldf, void* stream) {
  MMSA_REQUIRE(y && bias && feat && arg && B > 0 && L > 0 && Lout > 0 && Lout <= L && F > 0 && ldf >= F);
  hipLaunchKernelGGL(bias_relu_maxtime_kernel, dim3(mmsa_cdiv((long)B * F, 256)), dim3(256), 0, mmsa_stream(stream), y,
                     bias, feat, arg, B, L, Lout, F, ldf);
  return mmsa_launch_status();
}

// Backward of the whole text branch.  dfeat/feat/arg: [B][ldf] with conv i's filters at columns feat_off[i]..+F.
// emb: gathered rows [B][L][E].  wp[i]: forward-packed weights [k_i][F][E].  dW[i]: [F][E][k_i] (overwritten),
// dbias[i]: [F] (overwritten).  dtable: [vocab][E], ACCUMULATED into (caller zeroes it).
MMSA_EXPORT int mmsa_text_bwd(const float* dfeat, const float* feat, const int* arg, const float* emb,
                              const long long* ids, int nconv, const int* ksize, const int* feat_off,
                              const float* const* wp, float* const* dW, float* const* dbias, float* dtable, int B, int L,
                              int E, int F, int ldf, long padding_idx, void* stream) {
  MMSA_REQUIRE(dfeat && feat && arg && emb && ids && ksize && feat_off && wp && dW && dbias && dtable);
  MMSA_REQUIRE(nconv >= 1 && nconv <= 4 && B > 0 && L > 0 && E > 0 && E <= 1024 && E % 64 == 0 && F > 0);
  hipStream_t s = mmsa_stream(stream);
  TextConvs cv;
  cv.nconv = nconv;
  for (int i = 0; i < 4; ++i) {
    cv.k[i] = i < nconv ? ksize[i] : 0;
    cv.feat_off[i] = i < nconv ? feat_off[i] : 0;
    cv.wp[i] = i < nconv ? wp[i] : nullptr;
  }
  for (int i = 0; i < nconv; ++i) {
    const float* df = dfeat + feat_off[i];
    const float* ft = feat + feat_off[i];
    const int* ag = arg + feat_off[i];
    switch (ksize[i]) {
      case 2: hipLaunchKernelGGL(text_bwd_w_kernel<2>, dim3(F), dim3(E), 0, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 3: hipLaunchKernelGGL(text_bwd_w_kernel<3>, dim3(F), dim3(E), 0, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 4: hipLaunchKernelGGL(text_bwd_w_kernel<4>, dim3(F), dim3(E), 0, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 5: hipLaunchKernelGGL(text_bwd_w_kernel<5>, dim3(F), dim3(E), 0, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 6: hipLaunchKernelGGL(text_bwd_w_kernel<6>, dim3(F), dim3(E), 0, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 7: hipLaunchKernelGGL(text_bwd_w_kernel<7>, dim3(F), dim3(E), 0, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      default: return MMSA_ERR_ARG;
    }
  }
  // rows of dEmb held in LDS at a time: as many as fit in 128 KiB
  int TL = (128 * 1024) / (E * (int)sizeof(float));
  if (TL > L) TL = L;
  if (TL > E) TL = E;  // the touched[] flags are cleared by threads c < TL
  MMSA_REQUIRE(TL >= 1);
  const size_t shmem = (size_t)TL * E * sizeof(float) + (size_t)TL * sizeof(int);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(text_bwd_emb_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
    return MMSA_ERR_LAUNCH;
  hipLaunchKernelGGL(text_bwd_emb_kernel, dim3(B), dim3(E), shmem, s, dfeat, feat, arg, ids, cv, dtable, L, E, F, ldf,
                     padding_idx, TL);
  return mmsa_launch_status();
}

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
struct MaxTimeMulti {
  int n;
  int Lout[4];
  const float* y[4];
  const float* bias[4];
};

__global__ void bias_relu_maxtime_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                         float* __restrict__ feat, int* __restrict__ arg, int B, int L, int Lout,
                                         int F, int ldf, MaxTimeMulti multi) {
  if (multi.n > 0) {   // uniform: problem blockIdx.z writes columns z*F .. z*F+F-1
    const int z = blockIdx.z;
    y = multi.y[z];
    bias = multi.bias[z];
    Lout = multi.Lout[z];
    feat += z * F;
    arg += z * F;
  }
  // grid = (ceil(F/64), B); 64 filters per block, four thread groups scan interleaved time steps, then the
  // (value, first index) pairs are merged in LDS: larger value wins, equal values keep the earlier step.
  __shared__ float sv[4][64];
  __shared__ int si[4][64];
  const int fl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + fl, b = blockIdx.y;
  float best = -1.f;
  int bt = 0;
  if (f < F) {
    const float bs = bias[f];
    const float* yp = y + (long)b * L * F + f;
#pragma unroll 4
    for (int t = grp; t < Lout; t += 4) {
      const float v = fmaxf(yp[(long)t * F] + bs, 0.f);
      if (v > best) { best = v; bt = t; }
    }
  }
  sv[grp][fl] = best;
  si[grp][fl] = bt;
  __syncthreads();
  if (grp == 0 && f < F) {
#pragma unroll
    for (int g = 1; g < 4; ++g) {
      const float v = sv[g][fl];
      const int t = si[g][fl];
      if (v > best || (v == best && t < bt)) { best = v; bt = t; }
    }
    feat[(long)b * ldf + f] = best;
    arg[(long)b * ldf + f] = bt;
  }
}

// dW[f][c][j] = sum_b g[b][f] * emb[b][t*(b,f)+j][c];  dbias[f] = sum_b g[b][f];  g = dfeat where feat > 0.
// grid = F blocks, block = E threads (c).  The per-utterance (g, t*) pairs are staged in LDS first so the
// embedding loads are unconditional and independent (the compiler keeps several in flight).
template <int K>
__global__ void text_bwd_w_kernel(const float* __restrict__ dfeat, const float* __restrict__ feat,
                                  const int* __restrict__ arg, const float* __restrict__ emb, float* __restrict__ dW,
                                  float* __restrict__ dbias, int B, int L, int E, int ldf) {
  extern __shared__ float sm[];
  float* gs = sm;                                  // [B]
  int* ts = reinterpret_cast<int*>(sm + B);        // [B]
  const int f = blockIdx.x, c = threadIdx.x;
  for (int b = c; b < B; b += blockDim.x) {
    const float ft = feat[(long)b * ldf + f];
    gs[b] = ft > 0.f ? dfeat[(long)b * ldf + f] : 0.f;
    ts[b] = arg[(long)b * ldf + f];
  }
  __syncthreads();
  float acc[K];
#pragma unroll
  for (int j = 0; j < K; ++j) acc[j] = 0.f;
  float gb = 0.f;
#pragma unroll 4
  for (int b = 0; b < B; ++b) {
    const float g = gs[b];
    const float* e = emb + ((long)b * L + ts[b]) * E + c;
    gb += g;
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = fmaf(g, e[(long)j * E], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < K; ++j) dW[((long)f * E + c) * K + j] = acc[j];
  if (c == 0) dbias[f] = gb;
}

struct TextConvs {
  int nconv;
  int k[4];
  int feat_off[4];
  const float* wp[4];  // packed [j][F][E]
};

// Groups the token positions of a batch by token id, so that the gradients of a token that occurs several times
// can be summed in a fixed order (ascending position) instead of meeting in float atomics in arrival order.
// One workgroup; thread i owns positions i, i + 1024, ... in every phase.
//   table [vocab][2] ints, all zero between launches (the kernel clears what it touched): running count | segment start
//   work  [5][ntok] ints: arrival index | unsorted members | seg_cnt | seg_off | members
// Result: for the FIRST position p of every distinct valid id, seg_cnt[p] = number of positions holding that id and
// members[seg_off[p] .. + seg_cnt[p]) = those positions in ascending order; seg_cnt = 0 at every other position.
// Padding tokens and ids outside the table belong to no group.  The arrival order of the counting atomics and the
// placement of the segments vary from run to run; the sorted segments and seg_cnt do not.
// Everything the phases hand to each other is written and read by waves of this one workgroup, which share a
// first-level cache: workgroup scope is enough (device-scope loads would each make the round trip to memory, 30 of
// them in a row for a token that every utterance holds).
__device__ __forceinline__ int coherent_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(1024) void token_groups_kernel(const long long* __restrict__ ids, long ntok, long vocab,
                                                            long padding_idx, int* table, int* work) {
  __shared__ int total;
  int* arrival = work;
  int* unsorted = work + ntok;
  int* seg_cnt = work + 2 * ntok;
  int* seg_off = work + 3 * ntok;
  int* members = work + 4 * ntok;
  const int tid = threadIdx.x;
  if (tid == 0) total = 0;
  for (long p = tid; p < ntok; p += 1024) {
    const long long id = ids[p];
    const bool valid = id >= 0 && id < vocab && id != padding_idx;
    arrival[p] = valid ? atomicAdd(table + 2 * id, 1) : -1;
    seg_cnt[p] = 0;
  }
  __syncthreads();
  for (long p = tid; p < ntok; p += 1024) {      // whoever arrived first reserves the id's segment
    if (arrival[p] != 0) continue;
    int* e = table + 2 * ids[p];
    const int c = coherent_load(e);
    __hip_atomic_store(e + 1, atomicAdd(&total, c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (long p = tid; p < ntok; p += 1024) {
    const int a = arrival[p];
    if (a < 0) continue;
    const int off = coherent_load(table + 2 * ids[p] + 1);
    __hip_atomic_store(unsorted + off + a, (int)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (long p = tid; p < ntok; p += 1024) {      // rank of p among the positions of its id
    if (arrival[p] < 0) continue;
    const int* e = table + 2 * ids[p];
    const int c = coherent_load(e), off = coherent_load(e + 1);
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += coherent_load(unsorted + off + j) < (int)p;
    members[off + rank] = (int)p;
    arrival[p] = rank;
    if (rank == 0) {
      seg_cnt[p] = c;
      seg_off[p] = off;
    }
  }
  __syncthreads();
  for (long p = tid; p < ntok; p += 1024) {
    if (arrival[p] != 0) continue;
    int* e = table + 2 * ids[p];
    e[0] = 0;
    e[1] = 0;
  }
}

// Gradient of one gathered embedding row: grid = (L, B), one row (b, t) per workgroup.
//   dEmb[b][t][c] = sum over convs i, filters f, taps j with arg[b][i,f] + j == t of  g[b][i,f] * W_i[f][c][j]
// Phase 1: the threads test the utterance's (conv, filter) pairs in parallel and compact the matching ones into
// an LDS list (ballot + prefix count, so the list order is the pair order: deterministic).  Phase 2: thread c
// walks the list four entries at a time, so four coalesced weight-row loads are in flight per step.
// Rows of padding tokens and of tokens outside the vocabulary are skipped.  A token that occurs once in the batch
// stores its row straight into the table gradient; the rows of a token that occurs several times go to drow[b][t]
// and text_sum_rows_kernel adds them up in position order.
__global__ __launch_bounds__(256) void text_bwd_emb_kernel(const float* __restrict__ dfeat,
                                                           const float* __restrict__ feat,
                                                           const int* __restrict__ arg,
                                                           const long long* __restrict__ ids, TextConvs cv,
                                                           float* __restrict__ dtable, float* __restrict__ drow,
                                                           const int* __restrict__ seg_cnt, int L, int E, int F,
                                                           int ldf, long padding_idx, long vocab,
                                                           int* __restrict__ err) {
  extern __shared__ float sm[];
  const int nf = cv.nconv * F;
  float* lg = sm;                                       // [nf + 4] gradient of the matching pair
  int* lci = reinterpret_cast<int*>(sm + nf + 4);       // [nf + 4] which conv
  int* loff = lci + nf + 4;                             // [nf + 4] offset of the weight row inside that conv's pack
  __shared__ int wave_cnt[4];
  const int t = blockIdx.x, b = blockIdx.y;
  const long pos = (long)b * L + t;
  const long long id = ids[pos];
  if (id == padding_idx) return;                        // uniform: the whole block leaves
  if (id < 0 || id >= vocab) {                          // no row of the table belongs to this token: write nothing
    if (threadIdx.x == 0 && err) *err = 1;
    return;
  }
  const bool alone = seg_cnt[pos] == 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int count = 0;
  for (int base = 0; base < nf; base += 256) {
    const int e = base + tid;
    bool flag = false;
    float g = 0.f;
    int ci = 0, off = 0;
    if (e < nf) {
      ci = e / F;
      const int f = e - ci * F;
      const long o = (long)b * ldf + cv.feat_off[ci] + f;
      if (feat[o] > 0.f) {
        const int j = t - arg[o];
        if (j >= 0 && j < cv.k[ci]) {
          g = dfeat[o];
          off = (j * F + f) * E;
          flag = g != 0.f;
        }
      }
    }
    const unsigned long long mask = __ballot(flag);
    const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[w] = __popcll(mask);
    __syncthreads();
    int before = count, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int n = wave_cnt[k];
      if (k < w) before += n;
      total += n;
    }
    if (flag) {
      lg[before + prefix] = g;
      lci[before + prefix] = ci;
      loff[before + prefix] = off;
    }
    count += total;
    __syncthreads();
  }
  if (count == 0 && alone) return;                      // the table gradient was cleared before this launch
  if (tid < 4) {                                        // pad the list to a multiple of four with null entries
    lg[count + tid] = 0.f;
    lci[count + tid] = 0;
    loff[count + tid] = 0;
  }
  __syncthreads();
  float* dst = alone ? dtable + id * E : drow + pos * E;
  for (int c = tid; c < E; c += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = 0; i < count; i += 4) {
      const float w0 = cv.wp[lci[i]][loff[i] + c];
      const float w1 = cv.wp[lci[i + 1]][loff[i + 1] + c];
      const float w2 = cv.wp[lci[i + 2]][loff[i + 2] + c];
      const float w3 = cv.wp[lci[i + 3]][loff[i + 3] + c];
      a0 = fmaf(lg[i], w0, a0);
      a1 = fmaf(lg[i + 1], w1, a1);
      a2 = fmaf(lg[i + 2], w2, a2);
      a3 = fmaf(lg[i + 3], w3, a3);
    }
    dst[c] = (a0 + a1) + (a2 + a3);
  }
}

// dtable[id] = drow[m_0] + drow[m_1] + ... (left to right, m ascending) for every id that occurs more than once:
// grid = (L, B), the workgroup of the id's first position does the sum, every other one leaves at once.
__global__ __launch_bounds__(256) void text_sum_rows_kernel(const long long* __restrict__ ids,
                                                            const float* __restrict__ drow,
                                                            const int* __restrict__ seg_cnt,
                                                            const int* __restrict__ seg_off,
                                                            const int* __restrict__ members,
                                                            float* __restrict__ dtable, int L, int E) {
  const long pos = (long)blockIdx.y * L + blockIdx.x;
  const int n = seg_cnt[pos];
  if (n < 2) return;
  const int* m = members + seg_off[pos];
  float* dst = dtable + ids[pos] * E;
  for (int c = threadIdx.x; c < E; c += 256) {
    float acc = drow[(long)m[0] * E + c];
    int i = 1;
    for (; i + 4 <= n; i += 4) {                        // four loads in flight, added in order
      const float r0 = drow[(long)m[i] * E + c], r1 = drow[(long)m[i + 1] * E + c];
      const float r2 = drow[(long)m[i + 2] * E + c], r3 = drow[(long)m[i + 3] * E + c];
      acc = (((acc + r0) + r1) + r2) + r3;
    }
    for (; i < n; ++i) acc += drow[(long)m[i] * E + c];
    dst[c] = acc;
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
  MMSA_REQUIRE(B <= 65535);
  hipLaunchKernelGGL(bias_relu_maxtime_kernel, dim3(mmsa_cdiv(F, 64), B), dim3(256), 0, mmsa_stream(stream), y,
                     bias, feat, arg, B, L, Lout, F, ldf, MaxTimeMulti{});
  return mmsa_launch_status();
}

// The same for n (<= 4) convolution outputs at once: problem i reads y[i] / bias[i] with Lout[i] valid steps and
// writes feature columns i*F .. i*F+F-1.  y / bias / Lout are host arrays.
MMSA_EXPORT int mmsa_bias_relu_maxtime_multi(const float* const* y, const float* const* bias, const int* Lout, int n,
                                             float* feat, int* arg, int B, int L, int F, int ldf, void* stream) {
  MMSA_REQUIRE(y && bias && Lout && feat && arg && n >= 1 && n <= 4 && B > 0 && B <= 65535 && L > 0 && F > 0);
  MMSA_REQUIRE(ldf >= n * F);
  MaxTimeMulti m;
  m.n = n;
  for (int i = 0; i < 4; ++i) {
    m.y[i] = i < n ? y[i] : nullptr;
    m.bias[i] = i < n ? bias[i] : nullptr;
    m.Lout[i] = i < n ? Lout[i] : 0;
    if (i < n) MMSA_REQUIRE(y[i] && bias[i] && Lout[i] > 0 && Lout[i] <= L);
  }
  hipLaunchKernelGGL(bias_relu_maxtime_kernel, dim3(mmsa_cdiv(F, 64), B, n), dim3(256), 0, mmsa_stream(stream), y[0],
                     bias[0], feat, arg, B, L, Lout[0], F, ldf, m);
  return mmsa_launch_status();
}

// Groups the ntok token positions by token id for mmsa_text_bwd (depends on the ids only: it can run any time between
// the arrival of the ids and the backward pass).  groups: 5*ntok ints [arrival | unsorted | seg_cnt | seg_off | members];
// id_table: 2*vocab ints that must be zero at the first call and are left zero by every call.
MMSA_EXPORT int mmsa_token_groups(const long long* ids, long ntok, long vocab, long padding_idx, int* groups,
                                  int* id_table, void* stream) {
  MMSA_REQUIRE(ids && groups && id_table && ntok > 0 && ntok < (1l << 30) && vocab > 0);
  hipLaunchKernelGGL(token_groups_kernel, dim3(1), dim3(1024), 0, mmsa_stream(stream), ids, ntok, vocab, padding_idx,
                     id_table, groups);
  return mmsa_launch_status();
}

// Backward of the whole text branch.  dfeat/feat/arg: [B][ldf] with conv i's filters at columns feat_off[i]..+F.
// emb: gathered rows [B][L][E].  wp[i]: forward-packed weights [k_i][F][E].  dW[i]: [F][E][k_i] (overwritten),
// dbias[i]: [F] (overwritten).  dtable: [vocab][E]: the rows of the batch's tokens are overwritten, every other
// row is left alone (the caller clears the table beforehand).  A token id outside [0, vocab) has no row: its
// gradient is dropped and *err (optional) is set to 1.  The result does not depend on the order in which workgroups
// run: the rows of a token that occurs several times are summed by position.  groups: what mmsa_token_groups left
// for these ids (5*B*L ints); drow: [B*L][E] floats of scratch.
MMSA_EXPORT int mmsa_text_bwd(const float* dfeat, const float* feat, const int* arg, const float* emb,
                              const long long* ids, int nconv, const int* ksize, const int* feat_off,
                              const float* const* wp, float* const* dW, float* const* dbias, float* dtable,
                              float* drow, const int* groups, int B, int L, int E, int F, int ldf,
                              long padding_idx, long vocab, int* err, void* stream) {
  MMSA_REQUIRE(dfeat && feat && arg && emb && ids && ksize && feat_off && wp && dW && dbias && dtable && vocab > 0);
  MMSA_REQUIRE(drow && groups && (long)B * L < (1l << 30));
  MMSA_REQUIRE(nconv >= 1 && nconv <= 4 && B > 0 && B <= 65535 && L > 0 && E > 0 && E <= 1024 && E % 64 == 0 && F > 0);
  hipStream_t s = mmsa_stream(stream);
  const long ntok = (long)B * L;
  TextConvs cv;
  cv.nconv = nconv;
  for (int i = 0; i < 4; ++i) {
    cv.k[i] = i < nconv ? ksize[i] : 0;
    cv.feat_off[i] = i < nconv ? feat_off[i] : 0;
    cv.wp[i] = i < nconv ? wp[i] : nullptr;
  }
  const size_t wsh = (size_t)B * 8;   // (g, t*) per utterance
  MMSA_REQUIRE(wsh <= 48 * 1024);
  for (int i = 0; i < nconv; ++i) {
    const float* df = dfeat + feat_off[i];
    const float* ft = feat + feat_off[i];
    const int* ag = arg + feat_off[i];
    switch (ksize[i]) {
      case 2: hipLaunchKernelGGL(text_bwd_w_kernel<2>, dim3(F), dim3(E), wsh, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 3: hipLaunchKernelGGL(text_bwd_w_kernel<3>, dim3(F), dim3(E), wsh, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 4: hipLaunchKernelGGL(text_bwd_w_kernel<4>, dim3(F), dim3(E), wsh, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 5: hipLaunchKernelGGL(text_bwd_w_kernel<5>, dim3(F), dim3(E), wsh, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 6: hipLaunchKernelGGL(text_bwd_w_kernel<6>, dim3(F), dim3(E), wsh, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      case 7: hipLaunchKernelGGL(text_bwd_w_kernel<7>, dim3(F), dim3(E), wsh, s, df, ft, ag, emb, dW[i], dbias[i], B, L, E, ldf); break;
      default: return MMSA_ERR_ARG;
    }
  }
  const size_t shmem = (size_t)3 * (nconv * F + 4) * sizeof(float);
  MMSA_REQUIRE(shmem <= 48 * 1024);
  const int* seg_cnt = groups + 2 * ntok;
  hipLaunchKernelGGL(text_bwd_emb_kernel, dim3(L, B), dim3(256), shmem, s, dfeat, feat, arg, ids, cv, dtable, drow,
                     seg_cnt, L, E, F, ldf, padding_idx, vocab, err);
  hipLaunchKernelGGL(text_sum_rows_kernel, dim3(L, B), dim3(256), 0, s, ids, drow, seg_cnt, groups + 3 * ntok,
                     groups + 4 * ntok, dtable, L, E);
  return mmsa_launch_status();
}

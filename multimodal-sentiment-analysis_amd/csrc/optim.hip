// Optimiser-side kernels over the flat parameter / gradient buffers:
//  * per-tensor L2 norms (the reference adds 0.001 * sum_p ||p||_2 to the loss in 'both' mode,
//    `CNN implementation/multimodal_model.py:456-462`) and the matching gradient 0.001 * p / ||p||,
//  * global-norm gradient clipping (`torch.nn.utils.clip_grad_norm_(max_norm=1.0)`,
//    `CNN implementation/multimodal_train.py:194`): coefficient min(1, max_norm / (total + 1e-6)),
//  * AdamW with decoupled weight decay (`multimodal_train.py:481-485,195`; torch defaults).
//
// Tensors live back to back in one flat fp32 buffer (each start aligned to 256 floats, gaps hold zeros).
// A host-built chunk table maps fixed-size chunks to tensors so that a workgroup never straddles two tensors:
//   chunk_off[c] (start, floats), chunk_len[c], chunk_tensor[c]; tensor_first_chunk[t] .. tensor_first_chunk[t+1].
// Reductions are two-stage with fixed summation order (no atomics).  All step-dependent scalars stay in device
// memory (state[]), so a whole training step can be replayed from a captured HIP graph without host reads.
#include "mmsa_common.h"

namespace {

// state[] layout (floats): 0 step count, 1 bias_correction1, 2 sqrt(bias_correction2), 3 clip coefficient,
//                          4 total grad norm, 5 sum of parameter norms
enum { ST_STEP = 0, ST_BC1 = 1, ST_BC2S = 2, ST_CLIP = 3, ST_GNORM = 4, ST_PNORM = 5 };

// One tensor of the flat buffer may be declared row-sparse: a [rows][width] table (the token embedding) whose data
// gradient `a` is zero outside the rows ids[q] of the positions q with first[q] > 0 (what mmsa_token_groups leaves: the
// first occurrence of every distinct valid id).  Its sum of squares is then taken over those rows only:
//   sum_i (a_i + c b_i)^2 = c^2 ||b||^2 + sum over touched rows of a_i (a_i + 2 c b_i),   c = alpha / ||b||,
// where c^2 ||b||^2 = alpha^2 needs no pass over b: 8 MB of rows instead of the 62 MB of table and gradient.
struct SparseRows {
  int tensor;                 // index of the row-sparse tensor, -1: none
  int first_chunk, nchunks;   // its chunks
  int width;                  // floats per row (a multiple of 4)
  long n;                     // positions
  const long long* ids;       // [n] row of each position
  const int* first;           // [n] > 0 where the position is the first one holding its row
};

// sum of squares per chunk of (a[i] + alpha * b[i] * inv_norm[tensor]) -- b optional
__global__ __launch_bounds__(256) void chunk_sumsq_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ tnorm, float alpha,
                                                          const long long* __restrict__ chunk_off,
                                                          const int* __restrict__ chunk_len,
                                                          const int* __restrict__ chunk_tensor,
                                                          double* __restrict__ chunk_ss, const SparseRows sp,
                                                          int as_adamw) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  const float* pa = a + chunk_off[c];
  const int n = chunk_len[c];
  float coef = 0.f;
  const float* pb = nullptr;
  if (b) {
    const float nm = tnorm[chunk_tensor[c]];
    coef = nm > 0.f ? alpha / nm : 0.f;
    pb = b + chunk_off[c];
  }
  if (sp.tensor >= 0 && chunk_tensor[c] == sp.tensor) {   // uniform
    // the tensor's chunks share its positions evenly; one wave per position, fixed order throughout
    const int j = c - sp.first_chunk;
    const long per = (sp.n + sp.nchunks - 1) / sp.nchunks;
    const long lo = j * per, hi = lo + per < sp.n ? lo + per : sp.n;
    const long base = chunk_off[sp.first_chunk];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double s = 0.0;
    for (long q = lo + w; q < hi; q += 4) {
      if (sp.first[q] <= 0) continue;
      const long row = base + sp.ids[q] * sp.width;
      float acc = 0.f;
      for (int i = 4 * lane; i < sp.width; i += 256) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(a + row + i);
        if (b) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(b + row + i);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc = fmaf(gv[k], fmaf(2.f * coef, pv[k], gv[k]), acc);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc = fmaf(gv[k], gv[k], acc);
        }
      }
      s += (double)acc;
    }
    if (j == 0 && threadIdx.x == 0 && coef > 0.f) s += (double)alpha * (double)alpha;
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) chunk_ss[c] = s;
    return;
  }
  const int n4 = n >> 2;
  if (as_adamw) {
    // the sums of squares of the parameters, term by term in the order in which adamw_kernel leaves them behind for
    // the next step: a run that recomputes them (after a checkpoint is loaded) continues with the same bits as one
    // that did not
    const f32x4* q4 = reinterpret_cast<const f32x4*>(pa);
    double ss = 0.0;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = q4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) ss += (double)v[k] * (double)v[k];
    }
    for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) ss += (double)pa[i] * (double)pa[i];
    ss = block_sum_d(ss, red);
    if (threadIdx.x == 0) chunk_ss[c] = ss;
    return;
  }
  // 16-byte loads, four fp32 partial sums per thread (a few dozen terms each), combined in double
  const f32x4* a4 = reinterpret_cast<const f32x4*>(pa);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(pb);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 v = a4[i];
    if (pb) {
      const f32x4 w = b4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaf(coef, w[k], v[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = fmaf(v[k], v[k], acc[k]);
  }
  double s = ((double)acc[0] + (double)acc[1]) + ((double)acc[2] + (double)acc[3]);
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
    float v = pa[i];
    if (pb) v = fmaf(coef, pb[i], v);
    s += (double)v * (double)v;
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) chunk_ss[c] = s;
}

// One thread: advances the step count and derives the bias corrections and the clip coefficient from the total norm.
// (The bias corrections are formed in double from the double betas, as torch.optim.AdamW forms them on the host:
// with fp32 betas 1 - 0.999f^t is 1.3e-5 off at t = 1.)
__device__ __forceinline__ void step_scalars(float* __restrict__ state, double beta1, double beta2, float max_norm,
                                             int do_clip, float gnorm) {
  const float t = state[ST_STEP] + 1.f;
  state[ST_STEP] = t;
  state[ST_BC1] = (float)(1.0 - pow(beta1, (double)t));
  state[ST_BC2S] = (float)sqrt(1.0 - pow(beta2, (double)t));
  float coef = 1.f;
  if (do_clip) coef = fminf(max_norm / (gnorm + 1e-6f), 1.f);
  state[ST_CLIP] = coef;
}

struct StepScalars {   // state == nullptr: tensor_norms_kernel leaves the step scalars alone
  float* state;
  double beta1, beta2;
  float max_norm;
  int do_clip;
};

// per-tensor norms from chunk sums; out_sum = sum_t norm_t (fp32, registration order), out_l2 = sqrt(sum_t norm_t^2).
// One workgroup, on the step's serial tail between the chunk sums and AdamW, so it is written for latency: every chunk
// sum and the chunk table are requested at once into LDS (one round trip to memory, where a wave per tensor used to
// make two per tensor, six tensors in a row), the bias corrections, which do not depend on the norms, are worked out
// by a thread of the last wave while those requests are in flight, and the sums over lanes stay on the vector ALU.
constexpr int TN_LDS_CHUNKS = 4096;
__global__ __launch_bounds__(1024) void tensor_norms_kernel(const double* __restrict__ chunk_ss,
                                                            const int* __restrict__ tensor_first_chunk, int ntensors,
                                                            int nchunks, float* __restrict__ tnorm,
                                                            float* __restrict__ out_sum, float* __restrict__ out_l2,
                                                            const StepScalars sc) {
  __shared__ double css[TN_LDS_CHUNKS];
  __shared__ int first[1025];
  __shared__ float nrm[1024];
  __shared__ float scal[3];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool staged = nchunks <= TN_LDS_CHUNKS;   // uniform
  if (staged) {
    double cv[TN_LDS_CHUNKS / 1024];
#pragma unroll
    for (int j = 0; j < TN_LDS_CHUNKS / 1024; ++j) cv[j] = chunk_ss[min(tid + j * 1024, nchunks - 1)];
    const int f0 = tensor_first_chunk[min(tid, ntensors)];
    const int f1 = tensor_first_chunk[ntensors];
    if (sc.state && tid == 1023) {
      const float t = sc.state[ST_STEP] + 1.f;
      scal[0] = t;
      scal[1] = (float)(1.0 - pow(sc.beta1, (double)t));
      scal[2] = (float)sqrt(1.0 - pow(sc.beta2, (double)t));
    }
#pragma unroll
    for (int j = 0; j < TN_LDS_CHUNKS / 1024; ++j)
      if (tid + j * 1024 < nchunks) css[tid + j * 1024] = cv[j];
    first[tid] = f0;
    if (tid == 0) first[ntensors] = f1;
    __syncthreads();
    // one wave per tensor (16 waves take tensors round-robin), lanes stride over the tensor's chunk sums
    for (int t = w; t < ntensors; t += 16) {
      double s = 0.0;
      for (int c = first[t] + lane; c < first[t + 1]; c += 64) s += css[c];
      s = wave_sum_d(s);
      if (lane == 0) {
        const float v = (float)sqrt(s);
        nrm[t] = v;
        if (tnorm) tnorm[t] = v;
      }
    }
  } else {
    if (sc.state && tid == 1023) {
      const float t = sc.state[ST_STEP] + 1.f;
      scal[0] = t;
      scal[1] = (float)(1.0 - pow(sc.beta1, (double)t));
      scal[2] = (float)sqrt(1.0 - pow(sc.beta2, (double)t));
    }
    for (int t = w; t < ntensors; t += 16) {
      double s = 0.0;
      for (int c = tensor_first_chunk[t] + lane; c < tensor_first_chunk[t + 1]; c += 64) s += chunk_ss[c];
      s = wave_sum_d(s);
      if (lane == 0) {
        const float v = (float)sqrt(s);
        nrm[t] = v;
        if (tnorm) tnorm[t] = v;
      }
    }
  }
  __syncthreads();
  if (w == 0) {
    double sq = 0.0;
    for (int t = lane; t < ntensors; t += 64) sq += (double)nrm[t] * (double)nrm[t];
    const float l2 = (float)sqrt(wave_sum_d(sq));
    if (lane == 0) {
      if (out_sum) {            // the regulariser's value: added up in registration order, as the reference's loop does
        float acc = 0.f;
        for (int t = 0; t < ntensors; ++t) acc += nrm[t];
        out_sum[0] = acc;
      }
      if (out_l2) out_l2[0] = l2;
      if (sc.state) {
        sc.state[ST_STEP] = scal[0];
        sc.state[ST_BC1] = scal[1];
        sc.state[ST_BC2S] = scal[2];
        sc.state[ST_CLIP] = sc.do_clip ? fminf(sc.max_norm / (l2 + 1e-6f), 1.f) : 1.f;
      }
    }
  }
}

__global__ void l2_grad_add_kernel(float* __restrict__ g, const float* __restrict__ p, const float* __restrict__ tnorm,
                                   float lambda, const float* __restrict__ scale,
                                   const long long* __restrict__ chunk_off, const int* __restrict__ chunk_len,
                                   const int* __restrict__ chunk_tensor) {
  const int c = blockIdx.x;
  const float nm = tnorm[chunk_tensor[c]];
  if (!(nm > 0.f)) return;
  const float coef = (scale ? lambda * scale[0] : lambda) / nm;
  float* pg = g + chunk_off[c];
  const float* pp = p + chunk_off[c];
  const int n = chunk_len[c];
  for (int i = threadIdx.x; i < n; i += blockDim.x) pg[i] = fmaf(coef, pp[i], pg[i]);
}

__global__ void step_scalars_kernel(float* __restrict__ state, double beta1, double beta2, float max_norm, int do_clip) {
  step_scalars(state, beta1, beta2, max_norm, do_clip, state[ST_GNORM]);
}

__global__ void scale_inplace_by_state_kernel(float* __restrict__ g, long n, const float* __restrict__ state, int idx) {
  const float coef = state[idx];
  if (coef == 1.f) return;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) g[i] *= coef;
}

__device__ __forceinline__ void adamw_update(float& p, float g, float& m, float& v, float coef, float clip, float decay,
                                             float omb1, float beta2, float omb2, float bc2s, float eps,
                                             float step_size) {
  // omb1 / omb2: 1 - beta rounded to fp32 from the double difference (0.1f, 0.001f), the scalars torch's lerp_ and
  // addcmul_ receive; 1.f - 0.999f would be 0.00099998713
  const float gr = fmaf(coef, p, g) * clip;      // data gradient + lambda * p / ||p||, then the clip coefficient
  p *= decay;
  m = m + omb1 * (gr - m);
  v = v * beta2 + omb2 * gr * gr;
  p -= step_size * (m / (sqrtf(v) / bc2s + eps));
}

// One workgroup per chunk of the flat buffer (a chunk never straddles tensors): p, m, v updated in place from
// (g + lambda * p / ||p||_tensor) * clip.  The regulariser's gradient is formed on the fly, so it is never stored.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const long long* __restrict__ chunk_off,
                                                    const int* __restrict__ chunk_len,
                                                    const int* __restrict__ chunk_tensor,
                                                    const float* __restrict__ tnorm, float lambda,
                                                    const float* __restrict__ state, float lr, float omb1, float beta2,
                                                    float omb2, float eps, float wd,
                                                    double* __restrict__ chunk_ss_new) {
  __shared__ double red[16];
  double ss = 0.0;   // sum of squares of the UPDATED parameters of this chunk (next step's regulariser norms)
  const float clip = state[ST_CLIP], bc1 = state[ST_BC1], bc2s = state[ST_BC2S];
  const float step_size = lr / bc1;
  const float decay = 1.f - lr * wd;
  const int c = blockIdx.x;
  const long off = chunk_off[c];
  const int n = chunk_len[c];
  float coef = 0.f;
  if (tnorm && lambda != 0.f) {
    const float nm = tnorm[chunk_tensor[c]];
    coef = nm > 0.f ? lambda / nm : 0.f;
  }
  const int n4 = n >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(p + off);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g + off);
  f32x4* m4 = reinterpret_cast<f32x4*>(m + off);
  f32x4* v4 = reinterpret_cast<f32x4*>(v + off);
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
    const f32x4 gg = g4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adamw_update(pk, gg[k], mk, vk, coef, clip, decay, omb1, beta2, omb2, bc2s, eps, step_size);
      pp[k] = pk;
      mm[k] = mk;
      vv[k] = vk;
      ss += (double)pk * (double)pk;
    }
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
    adamw_update(p[off + i], g[off + i], m[off + i], v[off + i], coef, clip, decay, omb1, beta2, omb2, bc2s, eps, step_size);
    ss += (double)p[off + i] * (double)p[off + i];
  }
  if (chunk_ss_new) {   // uniform
    ss = block_sum_d(ss, red);
    if (threadIdx.x == 0) chunk_ss_new[c] = ss;
  }
}

__global__ void add_scaled_scalar_kernel(float* __restrict__ dst, const float* __restrict__ a,
                                         const float* __restrict__ b, float alpha) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = a[0] + alpha * b[0];
}

// diagnostic: the GPU's constant-rate real-time counter (100 MHz) at the moment this one-thread kernel runs
__global__ void stamp_kernel(unsigned long long* __restrict__ buf, int slot) {
  buf[slot] = __builtin_amdgcn_s_memrealtime();
}

// diagnostic: one thread waits until `ticks` ticks of the real-time counter have passed (bounded: gives up after
// 2^22 polls, a few hundred milliseconds, whatever the counter does)
__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < (1 << 22); ++i) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

}  // namespace

// Diagnostic: keeps `stream` busy for about `microseconds` (at most 10 000).  Enqueued ahead of an event pair around
// a launch, it lets the host enqueue the pair and the launch before the GPU gets there, so that the pair times the
// kernel and not the host's launch gap.
MMSA_EXPORT int mmsa_spin(int microseconds, void* stream) {
  MMSA_REQUIRE(microseconds > 0 && microseconds <= 10000);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, mmsa_stream(stream), (unsigned long long)microseconds * 100ull);
  return mmsa_launch_status();
}

// Diagnostic: buf[slot] <- the GPU's real-time counter (ticks of 10 ns) when the stream reaches this point.  Placed
// between the launches of a captured sequence it shows when each stream really gets there during an unprofiled
// replay (a profiler's timeline changes the concurrency it reports on).
MMSA_EXPORT int mmsa_stamp(unsigned long long* buf, int slot, void* stream) {
  MMSA_REQUIRE(buf && slot >= 0);
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, mmsa_stream(stream), buf, slot);
  return mmsa_launch_status();
}

// Per-tensor L2 norms of the flat buffer `p` -> tnorm[ntensors]; sum of norms -> state[5] (and *sum_out if given).
MMSA_EXPORT int mmsa_param_norms(const float* p, const long long* chunk_off, const int* chunk_len,
                                 const int* chunk_tensor, int nchunks, const int* tensor_first_chunk, int ntensors,
                                 double* chunk_ss, float* tnorm, float* sum_out, int chunk_sums_ready, void* stream) {
  MMSA_REQUIRE(p && chunk_off && chunk_len && chunk_tensor && tensor_first_chunk && chunk_ss && tnorm);
  MMSA_REQUIRE(nchunks > 0 && ntensors > 0 && ntensors <= 1024);
  hipStream_t s = mmsa_stream(stream);
  if (!chunk_sums_ready)   // otherwise chunk_ss already holds the sums (written by the previous mmsa_adamw_step)
    hipLaunchKernelGGL(chunk_sumsq_kernel, dim3(nchunks), dim3(256), 0, s, p, (const float*)nullptr,
                       (const float*)nullptr, 0.f, chunk_off, chunk_len, chunk_tensor, chunk_ss, SparseRows{-1}, 1);
  hipLaunchKernelGGL(tensor_norms_kernel, dim3(1), dim3(1024), 0, s, chunk_ss, tensor_first_chunk, ntensors, nchunks,
                     tnorm, sum_out, (float*)nullptr, StepScalars{});
  return mmsa_launch_status();
}

// g += lambda * p / ||p|| per tensor (tensors with zero norm are left alone, as torch.norm's gradient is 0 there)
// (scale: optional device scalar that multiplies lambda)
MMSA_EXPORT int mmsa_l2_grad_add(float* g, const float* p, const float* tnorm, float lambda, const float* scale,
                                 const long long* chunk_off, const int* chunk_len, const int* chunk_tensor, int nchunks,
                                 void* stream) {
  MMSA_REQUIRE(g && p && tnorm && chunk_off && chunk_len && chunk_tensor && nchunks > 0);
  hipLaunchKernelGGL(l2_grad_add_kernel, dim3(nchunks), dim3(256), 0, mmsa_stream(stream), g, p, tnorm, lambda, scale,
                     chunk_off, chunk_len, chunk_tensor);
  return mmsa_launch_status();
}

// total gradient norm = || (||g_1||, ..., ||g_T||) ||_2  -> state[4].  When p / tnorm are given the norm is taken of
// g + lambda * p / ||p|| (the regulariser's gradient folded in without materialising it).
MMSA_EXPORT int mmsa_grad_norm(const float* g, const float* p, const float* tnorm, float lambda,
                               const long long* chunk_off, const int* chunk_len, const int* chunk_tensor, int nchunks,
                               const int* tensor_first_chunk, int ntensors, double* chunk_ss, float* state,
                               void* stream) {
  MMSA_REQUIRE(g && chunk_off && chunk_len && chunk_tensor && tensor_first_chunk && chunk_ss && state);
  MMSA_REQUIRE(nchunks > 0 && ntensors > 0 && ntensors <= 1024);
  MMSA_REQUIRE((p == nullptr) == (tnorm == nullptr));
  hipStream_t s = mmsa_stream(stream);
  hipLaunchKernelGGL(chunk_sumsq_kernel, dim3(nchunks), dim3(256), 0, s, g, p, tnorm, lambda, chunk_off, chunk_len,
                     chunk_tensor, chunk_ss, SparseRows{-1}, 0);
  hipLaunchKernelGGL(tensor_norms_kernel, dim3(1), dim3(1024), 0, s, chunk_ss, tensor_first_chunk, ntensors, nchunks,
                     (float*)nullptr, (float*)nullptr, state + ST_GNORM, StepScalars{});
  return mmsa_launch_status();
}

// mmsa_grad_norm that also derives, in its second launch, the step scalars mmsa_adamw_step would derive (step
// count + 1, bias corrections, clip coefficient): follow it with mmsa_adamw_step(..., scalars_ready = 1).
// sparse_tensor >= 0 declares that tensor a [rows][sparse_width] table whose gradient is zero outside the rows
// sparse_ids[q] of the sparse_n positions q with sparse_first[q] > 0 (mmsa_token_groups' seg_cnt): its share of the
// norm is then taken from those rows alone; sparse_chunk0 / sparse_nchunks: the tensor's range in the chunk table.
MMSA_EXPORT int mmsa_grad_norm_finish(const float* g, const float* p, const float* tnorm, float lambda,
                                      const long long* chunk_off, const int* chunk_len, const int* chunk_tensor,
                                      int nchunks, const int* tensor_first_chunk, int ntensors, double* chunk_ss,
                                      float* state, double beta1, double beta2, float max_norm, int do_clip,
                                      int sparse_tensor, int sparse_chunk0, int sparse_nchunks, int sparse_width,
                                      long sparse_n, const long long* sparse_ids, const int* sparse_first,
                                      void* stream) {
  MMSA_REQUIRE(g && chunk_off && chunk_len && chunk_tensor && tensor_first_chunk && chunk_ss && state);
  MMSA_REQUIRE(nchunks > 0 && ntensors > 0 && ntensors <= 1024);
  MMSA_REQUIRE((p == nullptr) == (tnorm == nullptr));
  SparseRows sp{-1};
  if (sparse_tensor >= 0) {
    MMSA_REQUIRE(sparse_tensor < ntensors && sparse_chunk0 >= 0 && sparse_nchunks > 0);
    MMSA_REQUIRE(sparse_chunk0 + sparse_nchunks <= nchunks && sparse_width > 0 && sparse_width % 4 == 0);
    MMSA_REQUIRE(sparse_n > 0 && sparse_ids && sparse_first);
    sp = SparseRows{sparse_tensor, sparse_chunk0, sparse_nchunks, sparse_width, sparse_n, sparse_ids, sparse_first};
  }
  hipStream_t s = mmsa_stream(stream);
  hipLaunchKernelGGL(chunk_sumsq_kernel, dim3(nchunks), dim3(256), 0, s, g, p, tnorm, lambda, chunk_off, chunk_len,
                     chunk_tensor, chunk_ss, sp, 0);
  hipLaunchKernelGGL(tensor_norms_kernel, dim3(1), dim3(1024), 0, s, chunk_ss, tensor_first_chunk, ntensors, nchunks,
                     (float*)nullptr, (float*)nullptr, state + ST_GNORM,
                     StepScalars{state, beta1, beta2, max_norm, do_clip});
  return mmsa_launch_status();
}

// One AdamW step over the chunks of the flat buffers.  state[0] (step count) advances by one; the clip coefficient
// is derived from state[4] (call mmsa_grad_norm first) when do_clip != 0.  With tnorm given, lambda * p / ||p|| is
// added to every gradient element on the fly (pass null / 0 for plain AdamW).  scalars_ready != 0: the step count and
// the coefficients in state[] were already advanced for this step by mmsa_grad_norm_finish.
MMSA_EXPORT int mmsa_adamw_step(float* p, const float* g, float* m, float* v, const long long* chunk_off,
                                const int* chunk_len, const int* chunk_tensor, int nchunks, const float* tnorm,
                                float lambda, float* state, float lr, double beta1, double beta2, float eps,
                                float weight_decay, float max_norm, int do_clip, double* chunk_ss_new,
                                int scalars_ready, void* stream) {
  MMSA_REQUIRE(p && g && m && v && state && chunk_off && chunk_len && chunk_tensor && nchunks > 0);
  hipStream_t s = mmsa_stream(stream);
  if (!scalars_ready)
    hipLaunchKernelGGL(step_scalars_kernel, dim3(1), dim3(1), 0, s, state, beta1, beta2, max_norm, do_clip);
  hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, s, p, g, m, v, chunk_off, chunk_len, chunk_tensor, tnorm,
                     lambda, state, lr, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay,
                     chunk_ss_new);
  return mmsa_launch_status();
}

// g *= state[idx]  (used to apply the clip coefficient in place for callers that run their own optimiser)
MMSA_EXPORT int mmsa_scale_by_state(float* g, long n, const float* state, int idx, void* stream) {
  MMSA_REQUIRE(g && state && n > 0 && idx >= 0 && idx < 8);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_inplace_by_state_kernel, dim3((unsigned)blocks), dim3(256), 0, mmsa_stream(stream), g, n, state,
                     idx);
  return mmsa_launch_status();
}

// dst[0] = a[0] + alpha * b[0]   (loss = cross-entropy + 0.001 * sum of parameter norms, on device)
MMSA_EXPORT int mmsa_scalar_axpy(float* dst, const float* a, const float* b, float alpha, void* stream) {
  MMSA_REQUIRE(dst && a && b);
  hipLaunchKernelGGL(add_scaled_scalar_kernel, dim3(1), dim3(64), 0, mmsa_stream(stream), dst, a, b, alpha);
  return mmsa_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Up to four device-to-device copies in one launch (a step's inputs -- waveforms or spectrograms, token ids,
// labels, the modality switches -- into the fixed buffers a replayed launch sequence reads).
namespace {

struct CopySet {
  const unsigned* src[4];
  unsigned* dst[4];
  long words[4];
};

__global__ __launch_bounds__(256) void copy_multi_kernel(const CopySet cs) {
  const int k = blockIdx.y;
  const long n = cs.words[k];
  const unsigned* __restrict__ s = cs.src[k];
  unsigned* __restrict__ d = cs.dst[k];
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if ((((unsigned long long)s | (unsigned long long)d) & 15) == 0) {
    const long n4 = n >> 2;
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* d4 = reinterpret_cast<uint4*>(d);
    for (long j = i; j < n4; j += stride) d4[j] = s4[j];
    for (long j = (n4 << 2) + i; j < n; j += stride) d[j] = s[j];
  } else {
    for (; i < n; i += stride) d[i] = s[i];
  }
}

}  // namespace

// srcs / dsts: host arrays of n (<= 4) device pointers, bytes: host array of n sizes (multiples of 4)
MMSA_EXPORT int mmsa_copy_multi(const void* const* srcs, void* const* dsts, const long* bytes, int n, void* stream) {
  MMSA_REQUIRE(srcs && dsts && bytes && n >= 1 && n <= 4);
  CopySet cs{};
  long most = 0;
  for (int k = 0; k < n; ++k) {
    MMSA_REQUIRE(srcs[k] && dsts[k] && bytes[k] >= 0 && bytes[k] % 4 == 0);
    MMSA_REQUIRE((((unsigned long long)srcs[k] | (unsigned long long)dsts[k]) & 3) == 0);
    cs.src[k] = static_cast<const unsigned*>(srcs[k]);
    cs.dst[k] = static_cast<unsigned*>(dsts[k]);
    cs.words[k] = bytes[k] / 4;
    if (cs.words[k] > most) most = cs.words[k];
  }
  if (most == 0) return MMSA_OK;
  int blocks = (int)((most / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(copy_multi_kernel, dim3(blocks, n), dim3(256), 0, mmsa_stream(stream), cs);
  return mmsa_launch_status();
}


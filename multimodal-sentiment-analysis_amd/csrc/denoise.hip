// Spectral-gating noise reduction of a batch of waveforms: what `CNN implementation/preprocess.py:142-148` obtains
// per utterance on the CPU from `noisereduce.reduce_noise(y, sr)` with the package's defaults (non-stationary gating).
//
// The signal is seen through a zero-padded chunk (padding 30 000 samples each side), as the package does:
//   1. nr_stft     STFT, n_fft 1024, hop 256, periodic Hann, zero boundary extension of 512, scaled by 1/sum(window)
//   2. nr_gate     per frequency bin along time: |X| smoothed by a forward and a backward one-pole filter (steady
//                  state initial conditions, as scipy's filtfilt without padding; segments of the time axis run in
//                  parallel and exchange their end states), then
//                  mask = sigmoid(((|X| - smooth) / smooth - thresh) * slope)
//   3. nr_synth    mask smoothed by the separable triangular filter (time taps, then frequency taps, zero outside),
//                  X * mask, inverse 1024-point real FFT, synthesis window
//   4. nr_ola      overlap-add of the four segments covering each sample, divided by the summed squared window,
//                  padding removed
// All four are bound by HBM traffic (about 13 MB per 3-second utterance through the workspace); the transforms are
// the packed 512-point FFT of fft512.h, one wavefront per frame.
#include "mmsa_common.h"
#include "fft512.h"

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 256;
constexpr int NB = 513;      // one-sided bins
constexpr int PITCH = 516;   // row pitch (elements) of the per-frame spectra and masks
constexpr int FPB = 4;       // frames per workgroup, one wavefront each
constexpr int NT_MAX = 33, NF_MAX = 65, HALO = 32;

struct NrPlan {
  long N;      // chunk length with padding
  int nseg;    // STFT frames of the chunk
  int f_lo;    // first frame that overlaps the output samples
  int nfr;     // frames that overlap the output samples
};

__host__ __device__ inline NrPlan nr_plan(long L, int pad) {
  NrPlan p;
  p.N = L + 2L * pad;
  p.nseg = (int)(1 + p.N / HOP);
  const long p0 = (long)pad + NFFT / 2;              // first output sample in boundary-extended coordinates
  const long p1 = p0 + L - 1;
  p.f_lo = p0 >= NFFT ? (int)((p0 - (NFFT - 1) + HOP - 1) / HOP) : 0;
  long f_hi = p1 / HOP;
  if (f_hi > p.nseg - 1) f_hi = p.nseg - 1;
  p.nfr = (int)(f_hi - p.f_lo + 1);
  return p;
}

// Valid length of row b (clamped to what the buffers were sized for, whatever the caller's array holds).
__device__ __forceinline__ long row_len(const int* __restrict__ lens, int b, long Lmax) {
  const long l = lens[b];
  return l < 1 ? 1 : (l > Lmax ? Lmax : l);
}

__device__ __forceinline__ void load_twiddles(float2* tws, const float* __restrict__ tw512, int t) {
  const float2 w = make_float2(tw512[2 * t], tw512[2 * t + 1]);
  tws[t] = w;
  tws[t + 256] = make_float2(-w.x, -w.y);
}

__global__ __launch_bounds__(256) void nr_stft_kernel(const float* __restrict__ wave, long stride, long T, long c0,
                                                      int pad, long L, const int* __restrict__ lens, int nseg,
                                                      const float* __restrict__ window,
                                                      const float* __restrict__ tw512,
                                                      const float* __restrict__ tw1024, float2* __restrict__ spec) {
  __shared__ float seg[(FPB - 1) * HOP + NFFT];
  __shared__ float2 fa[FPB][FFT512_LD];
  __shared__ float2 fb[FPB][FFT512_LD];
  __shared__ float2 tws[512];
  const int t = threadIdx.x, fi = t >> 6, j = t & 63;
  const int b = blockIdx.y, f0 = blockIdx.x * FPB;
  load_twiddles(tws, tw512, t);
  float2 wv[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) wv[r] = *reinterpret_cast<const float2*>(window + 2 * (j + 64 * r));
  const float* w = wave + (long)b * stride;
  if (lens) { L = row_len(lens, b, L); T = L; }      // a shorter row: its own chunk, zeros behind it
  const long N = L + 2L * pad;
  {
    // samples the group's frames cover, in signal coordinates; a group that sees only padding writes zero spectra
    const long n0 = c0 - pad + (long)f0 * HOP - NFFT / 2, n1 = n0 + (FPB - 1) * HOP + NFFT;
    const long lo = max(0L, c0 - pad), hi = min(T, c0 - pad + N);
    if (n1 <= lo || n0 >= hi) {
      for (int i = t; i < FPB * NB; i += 256) {
        const int f = f0 + i / NB;
        if (f < nseg) spec[((long)b * nseg + f) * PITCH + i % NB] = make_float2(0.f, 0.f);
      }
      return;
    }
  }
  for (int i = t; i < (FPB - 1) * HOP + NFFT; i += 256) {
    const long e = (long)f0 * HOP + i - NFFT / 2;    // position in the padded chunk
    float v = 0.f;
    if (e >= 0 && e < N) {
      const long n = c0 - pad + e;                   // position in the signal
      if (n >= 0 && n < T) v = w[n];
    }
    seg[i] = v;
  }
  __syncthreads();
  float2 v[8];
  const float* sgm = seg + fi * HOP;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = j + 64 * r;
    v[r] = make_float2(sgm[2 * n] * wv[r].x, sgm[2 * n + 1] * wv[r].y);
  }
  fft512(v, fa[fi], fb[fi], tws, j);
  // one-sided spectrum of the real frame: X[k] = (Z[k]+conj(Z[512-k]))/2 + W^k (Z[k]-conj(Z[512-k]))/(2i)
  const int f = f0 + fi;
  if (f < nseg) {
    float2* row = spec + ((long)b * nseg + f) * PITCH;
    constexpr float scale = 1.0f / 512.0f;           // 1 / sum of the periodic Hann window
    for (int k = j; k < NB; k += 64) {
      const float2 zk = fa[fi][k & 511], zc = fa[fi][(512 - k) & 511];
      const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);
      const float orr = 0.5f * (zk.y + zc.y), oi = -0.5f * (zk.x - zc.x);
      const float c = tw1024[2 * k], sn = tw1024[2 * k + 1];
      row[k] = make_float2(scale * (er + (orr * c - oi * sn)), scale * (ei + (orr * sn + oi * c)));
    }
  }
}

__device__ __forceinline__ float cabs(float2 z) { return sqrtf(z.x * z.x + z.y * z.y); }

__device__ __forceinline__ double powi(double x, int n) {
  double p = 1.0;
#pragma unroll 1
  for (int i = 0; i < n; ++i) p *= x;
  return p;
}

// One workgroup per (utterance, 64 bins), parallel along time: wavefront w owns frames [w * seg, (w + 1) * seg).
// A first-order recurrence is linear in its incoming state, so each wavefront runs its segment from a zero state, the
// segment ends are exchanged through LDS and combined (omb^seg per full segment crossed), and the incoming state's
// decaying contribution omb^(distance) * state is added on the next sweep over the segment:
//   sweep A (ascending)   forward filter from zero                       -> gate, segment end -> LDS
//   sweep B (descending)  + forward carry, backward filter from zero     -> gate, segment end -> LDS
//   sweep C (ascending)   + backward carry, sigmoid gate against |X|     -> gate
// The recurrences run in double; the initial states are those of scipy's filtfilt without padding (steady state for
// the first / last sample).  Loads go out eight frames at a time, ahead of the dependent chain.
__global__ __launch_bounds__(1024) void nr_gate_kernel(const float2* __restrict__ spec, float* __restrict__ gate,
                                                       int nseg, const int* __restrict__ lens, int pad, double bc,
                                                       float thresh, float slope) {
  __shared__ double ends[16][64];
  __shared__ double lastv[64];
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const int sg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row addresses stay scalar
  const int nw = blockDim.x >> 6;
  const int k = min(blockIdx.x * 64 + lane, NB - 1);       // lanes past the last bin shadow it (same values stored)
  const long row0 = (long)blockIdx.y * nseg * PITCH;       // rows are laid out for the longest utterance ...
  if (lens) nseg = min(nseg, nr_plan(row_len(lens, blockIdx.y, 1L << 30), pad).nseg);   // ... filtered over their own
  const int seg = ((nseg + nw - 1) / nw + U - 1) / U * U;
  const int s0 = sg * seg;
  const int len = max(0, min(seg, nseg - s0));
  const int nact = (nseg + seg - 1) / seg;                 // wavefronts that own frames
  const float2* sp = spec + row0 + (long)s0 * PITCH + k;
  float* g = gate + row0 + (long)s0 * PITCH + k;
  const double omb = 1.0 - bc, iomb = 1.0 / omb;
  const double a_seg = powi(omb, seg), a_len = powi(omb, len), a_last = powi(omb, nseg - (nact - 1) * seg);
  const double x0 = (double)cabs(spec[row0 + k]);

  double y = 0.0;
  for (int i0 = 0; i0 < len; i0 += U) {
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = i0 + u < len ? cabs(sp[(long)(i0 + u) * PITCH]) : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u < len) {
        y = bc * (double)a[u] + omb * y;
        g[(long)(i0 + u) * PITCH] = (float)y;
      }
  }
  ends[sg][lane] = y;
  __syncthreads();
  double carry = x0;                                       // state ahead of frame 0
  for (int r = 0; r < sg; ++r) carry = ends[r][lane] + a_seg * carry;
  __syncthreads();                                         // every carry is read; `ends` may be reused

  double s = 0.0, p = a_len;                               // frame i of the segment sees omb^(i+1) of the carry
  for (int i0 = len - 1; i0 >= 0; i0 -= U) {
    float yl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) yl[u] = i0 - u >= 0 ? g[(long)(i0 - u) * PITCH] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 - u >= 0) {
        const double yf = (double)yl[u] + p * carry;
        p *= iomb;
        if (i0 - u == len - 1 && s0 + len == nseg) lastv[lane] = yf;   // state behind the last frame
        s = bc * yf + omb * s;
        g[(long)(i0 - u) * PITCH] = (float)s;
      }
  }
  ends[sg][lane] = s;
  __syncthreads();
  if (len > 0) {
    carry = lastv[lane];
    for (int r = nact - 1; r > sg; --r) carry = ends[r][lane] + (r == nact - 1 ? a_last : a_seg) * carry;
    p = a_len;                                             // frame i sees omb^(len - i) of the carry
    for (int i0 = 0; i0 < len; i0 += U) {
      float a[U], sl[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = i0 + u < len;
        a[u] = in ? cabs(sp[(long)(i0 + u) * PITCH]) : 0.f;
        sl[u] = in ? g[(long)(i0 + u) * PITCH] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u < len) {
          const double sm = (double)sl[u] + p * carry;
          p *= iomb;
          const float above = (float)(((double)a[u] - sm) / sm);
          g[(long)(i0 + u) * PITCH] = 1.0f / (1.0f + expf(-(above - thresh) * slope));
        }
    }
  }
}

__global__ __launch_bounds__(256) void nr_synth_kernel(const float2* __restrict__ spec, const float* __restrict__ gate,
                                                       int nseg, int f_lo, int nfr, const int* __restrict__ lens,
                                                       int pad, const float* __restrict__ vt, int nt,
                                                       const float* __restrict__ vf, int nf, float prop,
                                                       const float* __restrict__ window,
                                                       const float* __restrict__ tw512,
                                                       const float* __restrict__ tw1024, float* __restrict__ segs) {
  __shared__ float tm[FPB][NB + 2 * HALO + 3];
  __shared__ __attribute__((aligned(16))) float2 fab[2][FPB][FFT512_LD];
  float2 (*fa)[FFT512_LD] = fab[0];
  float2 (*fb)[FFT512_LD] = fab[1];
  // the masked spectrum lives in the transform's second buffer: every wavefront has consumed it by the first barrier
  // inside fft512, and the buffer is first written after that barrier
  float2 (*xs)[FFT512_LD] = fb;
  static_assert(FFT512_LD >= NB, "masked spectrum must fit in a transform row");
  // before that, both buffers hold the mask rows the group's frames share (FPB + nt - 1 of them)
  float* rows = reinterpret_cast<float*>(fab);
  constexpr int ROWS_MAX = (int)(sizeof(float2) * 2 * FPB * FFT512_LD / (sizeof(float) * PITCH));
  __shared__ float2 tws[512];
  __shared__ float vts[NT_MAX], vfs[NF_MAX];
  const int t = threadIdx.x, fi = t >> 6, j = t & 63;
  const int b = blockIdx.y;
  const int lf = blockIdx.x * FPB + fi;              // frame within the needed range
  const int f = f_lo + lf;
  int nseg_b = nseg, nfr_b = nfr;                    // this utterance's own frame counts
  if (lens) {
    const NrPlan pl = nr_plan(row_len(lens, b, 1L << 30), pad);
    nseg_b = min(nseg, pl.nseg);
    nfr_b = min(nfr, pl.nfr);
  }
  const bool live = lf < nfr_b;
  load_twiddles(tws, tw512, t);
  if (t < nt) vts[t] = vt[t];
  if (t >= 64 && t - 64 < nf) vfs[t - 64] = vf[t - 64];
  if (j < HALO) {
    tm[fi][j] = 0.f;
    tm[fi][HALO + NB + j] = 0.f;
  }
  const float* gb = gate + (long)b * nseg * PITCH;
  const int ht = nt >> 1, hf = nf >> 1;
  const int nrows = FPB + nt - 1;
  const bool staged = nrows <= ROWS_MAX;
  if (staged) {
    const int fr0 = f_lo + blockIdx.x * FPB - ht;    // frame of row 0
    for (int i = t; i < nrows * (PITCH / 4); i += 256) {
      const int r = i / (PITCH / 4), c = i - r * (PITCH / 4);
      const int fr = fr0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (fr >= 0 && fr < nseg_b) v = *reinterpret_cast<const f32x4*>(gb + (long)fr * PITCH + 4 * c);
      *reinterpret_cast<f32x4*>(rows + r * PITCH + 4 * c) = v;
    }
  }
  __syncthreads();
  for (int k = j; k < NB; k += 64) {
    float acc = 0.f;
    if (live && staged) {
      for (int q = 0; q < nt; ++q) acc = fmaf(vts[q], rows[(fi + q) * PITCH + k], acc);
    } else if (live) {
      for (int q = 0; q < nt; ++q) {
        const int fr = f + q - ht;
        if (fr >= 0 && fr < nseg_b) acc = fmaf(vts[q], gb[(long)fr * PITCH + k], acc);
      }
    }
    tm[fi][HALO + k] = acc;
  }
  __syncthreads();
  const float2* srow = spec + ((long)b * nseg + (live ? f : 0)) * PITCH;
  for (int k = j; k < NB; k += 64) {
    float acc = 0.f;
    const float* tp = &tm[fi][HALO + k - hf];
    for (int q = 0; q < nf; ++q) acc = fmaf(vfs[q], tp[q], acc);
    const float m = acc * prop + (1.0f - prop);
    float2 x = make_float2(0.f, 0.f);
    if (live) x = srow[k];
    xs[fi][k] = make_float2(x.x * m, x.y * m);
  }
  __syncthreads();
  // Inverse real transform through the packed 512-point one: Z[k] = E[k] + i O[k] with E = X[k] + conj(X[512-k]),
  // O = (X[k] - conj(X[512-k])) e^{+2 pi i k / 1024}; x[2n] + i x[2n+1] = IFFT512(Z)[n] / 1024, and the inverse is taken
  // as conj(FFT(conj Z)).
  float2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int k = j + 64 * r;
    const float2 x = xs[fi][k], y = xs[fi][512 - k];
    const float c = tw1024[2 * k], s = -tw1024[2 * k + 1];
    const float er = x.x + y.x, ei = x.y - y.y;
    const float dr = x.x - y.x, di = x.y + y.y;
    const float orr = dr * c - di * s, oi = dr * s + di * c;
    v[r] = make_float2(er - oi, -(ei + orr));
  }
  fft512(v, fa[fi], fb[fi], tws, j);
  if (live) {
    float2* out = reinterpret_cast<float2*>(segs + ((long)b * nfr + lf) * NFFT);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = j + 64 * r;
      const float2 F = fa[fi][n];
      const float2 w = *reinterpret_cast<const float2*>(window + 2 * n);
      // 1/1024 of the inverse transform times sum(window) = 512 of the 'spectrum' scaling, times the synthesis window
      out[n] = make_float2(0.5f * F.x * w.x, -0.5f * F.y * w.y);
    }
  }
}

__global__ __launch_bounds__(256) void nr_ola_kernel(const float* __restrict__ segs, int nfr, int f_lo, int nseg,
                                                     int pad, long L, const int* __restrict__ lens,
                                                     const float* __restrict__ window, float* __restrict__ out,
                                                     long out_stride, long c0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  const int b = blockIdx.y;
  if (lens) {
    const long len = row_len(lens, b, L);
    if (i >= len) {                                  // behind a shorter row: zeros
      out[(long)b * out_stride + c0 + i] = 0.f;
      return;
    }
    nseg = min(nseg, nr_plan(len, pad).nseg);
  }
  const long q = (long)pad + i;                      // position in the padded chunk
  const long p = q + NFFT / 2;                       // position in the boundary-extended chunk
  long fhi = p / HOP;
  if (fhi > nseg - 1) fhi = nseg - 1;
  const long flo = p >= NFFT ? (p - (NFFT - 1) + HOP - 1) / HOP : 0;
  float acc = 0.f, norm = 0.f;
  for (long f = flo; f <= fhi; ++f) {
    const int o = (int)(p - f * HOP);
    acc += segs[((long)b * nfr + (f - f_lo)) * NFFT + o];
    const float w = window[o];
    norm = fmaf(w, w, norm);
  }
  float v = acc / (norm > 1e-10f ? norm : 1.0f);
  if (q >= (long)(nseg - 1) * HOP) v = 0.f;          // past the last whole hop the inverse STFT holds nothing
  out[(long)b * out_stride + c0 + i] = v;
}

}  // namespace

// Frame counts and workspace need (floats per utterance) of one chunk of L output samples with `pad` samples of
// context on each side.  Host only.
MMSA_EXPORT int mmsa_denoise_plan(long L, int pad, int* nseg, int* f_lo, int* nfr, long* ws_floats) {
  MMSA_REQUIRE(L > 0 && pad >= 0 && nseg && f_lo && nfr && ws_floats);
  MMSA_REQUIRE(L + 2L * pad < (1L << 30) && L + 2L * pad >= NFFT);   // the padded chunk holds at least one frame
  const NrPlan p = nr_plan(L, pad);
  *nseg = p.nseg; *f_lo = p.f_lo; *nfr = p.nfr;
  *ws_floats = (long)p.nseg * PITCH * 3 + (long)p.nfr * NFFT;
  return MMSA_OK;
}

// wave: [batch] rows of T samples (row stride `stride`); the chunk [c0, c0+L) of every row is denoised into
// out[b][c0 .. c0+L) (row stride `out_stride`), reading up to `pad` samples of context on each side (zeros outside
// the signal).  lens: null, or [batch] valid lengths (1..T) of rows that hold utterances of different lengths; each
// is then processed as a signal of its own length (c0 must be 0 and L == T) and its output is zero behind it.  vt [nt] / vf [nf]: normalised time / frequency taps of the mask-smoothing filter (odd counts).
// b_iir: coefficient of the one-pole smoothing filter.  ws: batch * ws_floats floats of scratch, 16-byte aligned.
MMSA_EXPORT int mmsa_denoise_fwd(const float* wave, long stride, long T, long c0, long L, int pad, int batch,
                                 const int* lens, const float* window, const float* tw512, const float* tw1024, const float* vt, int nt,
                                 const float* vf, int nf, double b_iir, float thresh, float slope, float prop,
                                 float* ws, float* out, long out_stride, void* stream) {
  MMSA_REQUIRE(wave && window && tw512 && tw1024 && vt && vf && ws && out);
  MMSA_REQUIRE(batch > 0 && batch <= 65535 && T > 0 && L > 0 && c0 >= 0 && c0 + L <= T && pad >= 0);
  MMSA_REQUIRE(stride >= T && out_stride >= T && L + 2L * pad < (1L << 30) && L + 2L * pad >= NFFT);
  MMSA_REQUIRE(nt >= 1 && nt <= NT_MAX && (nt & 1) && nf >= 1 && nf <= NF_MAX && (nf & 1));
  MMSA_REQUIRE(b_iir > 0.0 && b_iir <= 1.0);
  MMSA_REQUIRE(!lens || (c0 == 0 && L == T));
  MMSA_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  const NrPlan p = nr_plan(L, pad);
  float2* spec = reinterpret_cast<float2*>(ws);
  float* gate = ws + (long)batch * p.nseg * PITCH * 2;
  float* segs = gate + (long)batch * p.nseg * PITCH;
  hipStream_t s = mmsa_stream(stream);
  hipLaunchKernelGGL(nr_stft_kernel, dim3(mmsa_cdiv(p.nseg, FPB), batch), dim3(256), 0, s, wave, stride, T, c0, pad,
                     L, lens, p.nseg, window, tw512, tw1024, spec);
  const int gwaves = p.nseg >= 16 * 8 ? 16 : mmsa_cdiv(p.nseg, 8);
  hipLaunchKernelGGL(nr_gate_kernel, dim3(mmsa_cdiv(NB, 64), batch), dim3(64 * gwaves), 0, s, spec, gate, p.nseg, lens,
                     pad, b_iir, thresh, slope);
  hipLaunchKernelGGL(nr_synth_kernel, dim3(mmsa_cdiv(p.nfr, FPB), batch), dim3(256), 0, s, spec, gate, p.nseg, p.f_lo,
                     p.nfr, lens, pad, vt, nt, vf, nf, prop, window, tw512, tw1024, segs);
  hipLaunchKernelGGL(nr_ola_kernel, dim3(mmsa_cdiv(L, 256), batch), dim3(256), 0, s, segs, p.nfr, p.f_lo, p.nseg, pad, L,
                     lens, window, out, out_stride, c0);
  return mmsa_launch_status();
}

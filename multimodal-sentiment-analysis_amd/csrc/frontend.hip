// Log-mel front end: waveform [B, n] -> normalised log-mel map [B, 1, 64, 130].
//
// Replaces, for a batch of utterances at once, what the reference computes per utterance on the CPU in
// `CNN implementation/preprocess.py:151-168` (pad/trim, MelSpectrogram, AmplitudeToDB, normalize, NaN scrub)
// and `:111-123` (z-score with unbiased std, clamp to +-6).
//
// Kernel 1 (logmel_db): one workgroup per (utterance, group of FPB consecutive frames).  The group's
//   (FPB+1)*512 reflect-padded samples are staged once in LDS (frames overlap by half), each frame is
//   windowed, packed as a 512-point complex sequence, transformed with a radix-8 Stockham FFT (8-point
//   DFTs in registers, three passes through LDS), unpacked to the 513 one-sided power bins, multiplied
//   by the (sparse, triangular) mel filterbank and converted to dB.
// Kernel 2 (logmel_norm): one workgroup per utterance: max -> (max-80) floor -> mean / unbiased std
//   (two-pass, double accumulators) -> (x-mean)/(std+1e-8) -> clamp -> NaN/Inf scrub, in place.
#include "mmsa_common.h"
#include "fft512.h"

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 512;
constexpr int NBINS = 513;
constexpr int NMEL = 64;
constexpr int FPB = 4;  // frames per workgroup: 64 threads each

// One workgroup = FPB consecutive frames of one utterance, one wavefront per frame.  After the group's samples are
// staged (one barrier), every wavefront runs on its own: it windows its frame, packs the 1024 real samples as a
// 512-point complex sequence, transforms it in place in its LDS row (radix-8 Stockham, 8-point DFTs in registers, no
// workgroup barrier: fft512_wave) and unpacks the 513 one-sided power bins into the same row.  A second barrier, then
// the (sparse, triangular) mel filterbank over all frames of the group and the conversion to dB.  30 KiB of LDS per
// workgroup, so five workgroups share a CU.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void logmel_db_kernel(
    const float* __restrict__ wave, long wave_stride, int n_valid, int n_target, int n_frames,
    const float* __restrict__ window,   // [1024]
    const float* __restrict__ tw512,    // [256][2]  cos, -sin of 2*pi*j/512
    const float* __restrict__ tw1024,   // [513][2]  cos, -sin of 2*pi*k/1024
    const float* __restrict__ melfb,    // [513][64]
    const int* __restrict__ melrange,   // [64][2]   first bin, one-past-last bin with non-zero weight
    float* __restrict__ out)            // [B][64][n_frames] dB
{
  // seg: the group's reflect-padded samples; once every wavefront has consumed them (second barrier) the dB values
  // of the group are collected in the same memory
  __shared__ float seg[(FPB + 1) * HOP];
  __shared__ float2 rows[FPB][FFT512_LD];   // per frame: FFT work row (padded), then its power spectrum
  __shared__ float2 twh[256];               // e^{-2 pi i m / 512}, m < 256
  static_assert(sizeof(float) * (NBINS + 3) <= sizeof(float2) * FFT512_LD, "the power spectrum must fit in the row");
  static_assert(NMEL * (FPB + 1) <= (FPB + 1) * HOP, "the dB values must fit in the sample buffer");
  float (*dbv)[FPB + 1] = reinterpret_cast<float (*)[FPB + 1]>(seg);

  const int t = threadIdx.x;
  twh[t] = make_float2(tw512[2 * t], tw512[2 * t + 1]);
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FPB;
  const float* w = wave + (long)b * wave_stride;
  const int fi = t >> 6, j = t & 63;

  // Constants this thread will need later are requested now, so that their memory latency passes behind the staging
  // and the transform: its window pair per packed sample, the unpack twiddles of its bins, its mel weights.
  constexpr int UNP = (NBINS + 63) / 64;           // unpack items per lane: bins j, j + 64, ..., 512 (lane 0 only)
  constexpr int MW = 12;                           // mel weights held in registers per lane (rest: read on demand)
  float2 wv[8], utw[UNP];
  float mwt[MW];
#pragma unroll
  for (int r = 0; r < 8; ++r) wv[r] = *reinterpret_cast<const float2*>(window + 2 * (j + 64 * r));
#pragma unroll
  for (int u = 0; u < UNP; ++u) utw[u] = *reinterpret_cast<const float2*>(tw1024 + 2 * min(j + 64 * u, NBINS - 1));
  const int mel_m = t >> 2, mel_q = t & 3;
  const int mel_lo = melrange[2 * mel_m], mel_hi = melrange[2 * mel_m + 1];
#pragma unroll
  for (int u = 0; u < MW; ++u) {
    const int bin = min(mel_lo + mel_q + 4 * u, NBINS - 1);
    mwt[u] = melfb[bin * NMEL + mel_m];
  }

  // stage reflect-padded samples: padded index q = f0*512 + i  <->  sample n = q - 512
  for (int i = t; i < (FPB + 1) * HOP; i += 256) {
    int n = f0 * HOP + i - NFFT / 2;
    if (n < 0) n = -n;
    if (n >= n_target) n = 2 * (n_target - 1) - n;
    float v = 0.f;
    if (n >= 0 && n < n_valid) v = w[n];
    seg[i] = v;
  }
  __syncthreads();

  float2* row = rows[fi];
  {
    float2 v[8];
    // z[n] = (x[2n], x[2n+1]) * window
    const float* sgm = seg + fi * HOP;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = j + 64 * r;
      v[r] = make_float2(sgm[2 * n] * wv[r].x, sgm[2 * n + 1] * wv[r].y);
    }
    fft512_wave(v, row, twh, j);
  }
  // unpack the real-input spectrum: X[k] = (Z[k]+conj(Z[512-k]))/2 + W^k (Z[k]-conj(Z[512-k]))/(2i).  The power
  // values go into the row the spectrum is read from, so every lane first takes all its pairs.
  {
    float2 zk[UNP], zc[UNP];
#pragma unroll
    for (int u = 0; u < UNP; ++u) {
      const int k = min(j + 64 * u, NBINS - 1);
      zk[u] = row[k & 511];
      zc[u] = row[(512 - k) & 511];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float* power = reinterpret_cast<float*>(row);
#pragma unroll
    for (int u = 0; u < UNP; ++u) {
      const int k = j + 64 * u;
      if (k >= NBINS) break;
      const float er = 0.5f * (zk[u].x + zc[u].x), ei = 0.5f * (zk[u].y - zc[u].y);   // even part
      const float dr = 0.5f * (zk[u].x - zc[u].x), di = 0.5f * (zk[u].y + zc[u].y);   // (Z - conj Z')/2
      // odd part = (dr + i di) / i = di - i dr
      const float orr = di, oi = -dr;
      const float c = utw[u].x, sn = utw[u].y;
      const float xr = er + (orr * c - oi * sn);
      const float xi = ei + (orr * sn + oi * c);
      power[k] = xr * xr + xi * xi;
    }
  }
  __syncthreads();
  // mel: 4 lanes per filter; each filter weight is fetched once and applied to all frames of the group, then dB
  {
    const int m = mel_m, q = mel_q;
    float acc[FPB];
#pragma unroll
    for (int f = 0; f < FPB; ++f) acc[f] = 0.f;
    auto power = [&](int f, int bin) { return reinterpret_cast<const float*>(rows[f])[bin]; };
#pragma unroll
    for (int u = 0; u < MW; ++u) {
      const int bin = mel_lo + q + 4 * u;
      if (bin < mel_hi) {
#pragma unroll
        for (int f = 0; f < FPB; ++f) acc[f] = fmaf(mwt[u], power(f, bin), acc[f]);
      }
    }
    for (int bin = mel_lo + q + 4 * MW; bin < mel_hi; bin += 4) {   // filters wider than 4 * MW bins
      const float wgt = melfb[bin * NMEL + m];
#pragma unroll
      for (int f = 0; f < FPB; ++f) acc[f] = fmaf(wgt, power(f, bin), acc[f]);
    }
#pragma unroll
    for (int f = 0; f < FPB; ++f) {
      float a = acc[f];
      a += dpp_mov<DPP_XOR1>(a);     // the four lanes of a filter (every lane is enabled here)
      a += dpp_mov<DPP_XOR2>(a);
      if (q == 0) dbv[m][f] = 10.0f * log10f(fmaxf(a, 1e-10f));
    }
  }
  __syncthreads();
  for (int i = t; i < NMEL * FPB; i += 256) {
    const int m = i / FPB, f = i - m * FPB;
    if (f0 + f < n_frames) out[((long)b * NMEL + m) * n_frames + f0 + f] = dbv[m][f];
  }
}

// specmask (optional): per utterance {first mel row, rows, first frame, frames} zero-filled after the dB floor and
// before the statistics (preprocess.py:92-109,162-165); a zero count disables that mask.
__global__ __launch_bounds__(1024) void logmel_norm_kernel(float* __restrict__ io, int n_elem, int n_frames,
                                                           float top_db, const int* __restrict__ specmask) {
  extern __shared__ float vals[];
  __shared__ float redf[16];
  __shared__ double redd[16];
  const int t = threadIdx.x;
  float* p = io + (long)blockIdx.x * n_elem;
  float mx = -INFINITY;
  for (int i = t; i < n_elem; i += 1024) {
    const float v = p[i];
    vals[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, redf);
  const float floor_v = mx - top_db;
  int mf0 = 0, mfn = 0, mt0 = 0, mtn = 0;
  if (specmask) {
    const int* mk = specmask + 4 * blockIdx.x;
    mf0 = mk[0]; mfn = mk[1]; mt0 = mk[2]; mtn = mk[3];
  }
  // (one workgroup's sixteen waves share four vector ALUs: this kernel's time is its instruction count, so the index
  // division of the masks is only paid when there are masks)
  double s = 0.0;
  if (mfn > 0 || mtn > 0) {   // uniform
    for (int i = t; i < n_elem; i += 1024) {
      float v = fmaxf(vals[i], floor_v);
      const int row = i / n_frames, col = i - row * n_frames;
      if ((unsigned)(row - mf0) < (unsigned)mfn || (unsigned)(col - mt0) < (unsigned)mtn) v = 0.f;
      vals[i] = v;
      s += (double)v;
    }
  } else {
    for (int i = t; i < n_elem; i += 1024) {
      const float v = fmaxf(vals[i], floor_v);
      vals[i] = v;
      s += (double)v;
    }
  }
  s = block_sum_d(s, redd);
  const double mean = s / (double)n_elem;
  double q = 0.0;
  for (int i = t; i < n_elem; i += 1024) {
    const double d = (double)vals[i] - mean;
    q += d * d;
  }
  q = block_sum_d(q, redd);
  const float meanf = (float)mean;
  const float stdf = (float)sqrt(q / (double)(n_elem - 1));
  const float denom = stdf + 1e-8f;
  for (int i = t; i < n_elem; i += 1024) {
    float v = (vals[i] - meanf) / denom;
    v = fminf(fmaxf(v, -6.f), 6.f);
    if (!(fabsf(v) <= 6.f)) v = 0.f;  // NaN scrub (Inf cannot survive the clamp)
    p[i] = v;
  }
}

}  // namespace

// Host-side table builder (no GPU work): periodic Hann window and FFT twiddles, evaluated in double.
MMSA_EXPORT int mmsa_logmel_tables_host(float* window /*[1024]*/, float* tw512 /*[512]*/, float* tw1024 /*[1026]*/) {
  MMSA_REQUIRE(window && tw512 && tw1024);
  const double pi = 3.14159265358979323846;
  for (int n = 0; n < NFFT; ++n) window[n] = (float)(0.5 - 0.5 * cos(2.0 * pi * n / NFFT));
  for (int j = 0; j < 256; ++j) {
    tw512[2 * j] = (float)cos(2.0 * pi * j / 512.0);
    tw512[2 * j + 1] = (float)(-sin(2.0 * pi * j / 512.0));
  }
  for (int k = 0; k <= 512; ++k) {
    tw1024[2 * k] = (float)cos(2.0 * pi * k / 1024.0);
    tw1024[2 * k + 1] = (float)(-sin(2.0 * pi * k / 1024.0));
  }
  return MMSA_OK;
}

// wave: [batch] rows of `n_valid` samples, row stride `wave_stride` floats.  Samples beyond n_valid are
// zeros up to n_target (pad) and samples beyond n_target are ignored (trim).  out: [batch][1][64][n_frames].
// specmask: null, or [batch][4] ints {first mel row, rows, first frame, frames} of the training-time spectral masks.
MMSA_EXPORT int mmsa_logmel_fwd(const float* wave, long wave_stride, int n_valid, int n_target, int batch,
                                const float* window, const float* tw512, const float* tw1024, const float* melfb,
                                const int* melrange, float top_db, const int* specmask, float* out, void* stream) {
  MMSA_REQUIRE(wave && window && tw512 && tw1024 && melfb && melrange && out);
  MMSA_REQUIRE(batch > 0 && n_target >= NFFT && n_valid >= 0 && wave_stride >= 0);
  if (n_valid > n_target) n_valid = n_target;
  const int n_frames = 1 + n_target / HOP;
  const int n_elem = NMEL * n_frames;
  MMSA_REQUIRE((size_t)n_elem * sizeof(float) <= 60 * 1024);
  dim3 grid(mmsa_cdiv(n_frames, FPB), batch);
  hipLaunchKernelGGL(logmel_db_kernel, grid, dim3(256), 0, mmsa_stream(stream), wave, wave_stride, n_valid, n_target,
                     n_frames, window, tw512, tw1024, melfb, melrange, out);
  hipLaunchKernelGGL(logmel_norm_kernel, dim3(batch), dim3(1024), n_elem * sizeof(float), mmsa_stream(stream), out,
                     n_elem, n_frames, top_db, specmask);
  return mmsa_launch_status();
}

// ------------------------------------------------------------------------------------------------ resampling
// Band-limited sinc interpolation with a Hann window, the algorithm of torchaudio.transforms.Resample (defaults
// lowpass_filter_width 6, rolloff 0.99) that `CNN implementation/preprocess.py:131-134` applies per utterance, fused
// with the mono mix (`:127-129`) and the pad / trim to the model's input length (`:151-156`).
// With orig / new reduced by their gcd, output sample i*new + p is the dot product of filter p ([K] taps) with the
// zero-padded input window starting at sample i*orig - width.
namespace {

constexpr int RS_FRAMES = 8;    // input strides (of `orig` samples) per workgroup
constexpr int RS_THREADS = 512;

// Work item = four consecutive output phases of two consecutive frames: per tap one 16-byte read of the filter bank
// (rows padded to `ld` floats, a multiple of 4, so the read is aligned; the bank stays in L2) and two LDS reads feed
// eight multiply-adds.  The sum over taps runs in tap order with fused multiply-adds.
__global__ __launch_bounds__(RS_THREADS) void resample_kernel(const float* __restrict__ pcm, int channels, long T,
                                                              long stride_b, const float* __restrict__ kernT, int ld,
                                                              int orig, int nw, int width, int K, long n_res,
                                                              float* __restrict__ out, int n_out) {
  extern __shared__ float win[];   // mono-mixed, zero-padded input window of RS_FRAMES strides plus the filter tail
  const int b = blockIdx.y;
  const long i0 = (long)blockIdx.x * RS_FRAMES;
  const int nwin = RS_FRAMES * orig + K;
  const float* src = pcm + (long)b * stride_b;
  for (int i = threadIdx.x; i < nwin; i += RS_THREADS) {
    const long n = i0 * orig + i - width;
    float v = 0.f;
    if (n >= 0 && n < T) {
      for (int c = 0; c < channels; ++c) v += src[(long)c * T + n];
      if (channels > 1) v = v / (float)channels;
    }
    win[i] = v;
  }
  __syncthreads();
  const int groups = (nw + 3) >> 2;
  for (int item = threadIdx.x; item < groups * (RS_FRAMES / 2); item += RS_THREADS) {
    const int fp = item / groups, pg = item - fp * groups;
    const int p = 4 * pg;
    const float* x0 = win + (2 * fp) * orig;
    const float* x1 = x0 + orig;
    const float* kp = kernT + p;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(kp + (long)k * ld);
      const float u = x0[k], v = x1[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] = fmaf(w[e], u, a0[e]);
        a1[e] = fmaf(w[e], v, a1[e]);
      }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const long jbase = (i0 + 2 * fp + f) * nw + p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long j = jbase + e;            // output sample
        if (p + e < nw && j < n_out) out[(long)b * n_out + j] = j < n_res ? (f ? a1[e] : a0[e]) : 0.f;
      }
    }
  }
}

long rs_gcd(long a, long b) { while (b) { const long t = a % b; a = b; b = t; } return a; }

}  // namespace

// Filter bank of the resampler, built on the host in double precision (torchaudio's construction) and stored
// transposed, kernT[K][ld] with ld = new rounded up to a multiple of 4 (padding zero), so that consecutive output
// phases read consecutive floats in aligned 16-byte pieces.  Returns through the out parameters the reduced rates, the
// one-sided filter width and the tap count K = 2*width + orig; kernT may be null to query the sizes first.
MMSA_EXPORT int mmsa_resample_filters_host(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff,
                                           float* kernT, int* orig_out, int* new_out, int* width_out, int* K_out) {
  MMSA_REQUIRE(orig_freq > 0 && new_freq > 0 && lowpass_filter_width > 0 && rolloff > 0.0 && rolloff <= 1.0);
  MMSA_REQUIRE(orig_out && new_out && width_out && K_out);
  const long g = rs_gcd(orig_freq, new_freq);
  const int orig = (int)(orig_freq / g), nw = (int)(new_freq / g);
  const double base = (double)(orig < nw ? orig : nw) * rolloff;
  const int width = (int)ceil((double)lowpass_filter_width * orig / base);
  const int K = 2 * width + orig;
  *orig_out = orig; *new_out = nw; *width_out = width; *K_out = K;
  if (!kernT) return MMSA_OK;
  const double pi = 3.14159265358979323846, lpw = (double)lowpass_filter_width;
  const double scale = base / orig;
  const int ld = (nw + 3) & ~3;
  for (long i = 0; i < (long)K * ld; ++i) kernT[i] = 0.f;
  for (int p = 0; p < nw; ++p)
    for (int k = 0; k < K; ++k) {
      double t = (-(double)p / nw + (double)(k - width) / orig) * base;
      if (t < -lpw) t = -lpw;
      if (t > lpw) t = lpw;
      const double c = cos(t * pi / lpw / 2.0);
      const double window = c * c;
      t *= pi;
      const double s = t == 0.0 ? 1.0 : sin(t) / t;
      kernT[(long)k * ld + p] = (float)(s * window * scale);
    }
  return MMSA_OK;
}

// pcm: [batch][channels][T] (utterance stride `stride_b` floats) at the original rate -> out [batch][n_out] at the
// new rate: mono mix, resample (resampled length ceil(new*T/orig)), zero pad or trim to n_out.
MMSA_EXPORT int mmsa_resample_fwd(const float* pcm, int channels, long T, long stride_b, const float* kernT, int orig,
                                  int nw, int width, int K, float* out, int n_out, int batch, void* stream) {
  MMSA_REQUIRE(pcm && kernT && out && channels > 0 && T > 0 && batch > 0 && batch <= 65535 && n_out > 0);
  MMSA_REQUIRE(orig > 0 && nw > 0 && width > 0 && K == 2 * width + orig && stride_b >= (long)channels * T);
  MMSA_REQUIRE((reinterpret_cast<uintptr_t>(kernT) & 15) == 0);
  const int ld = (nw + 3) & ~3;
  const size_t sh = (size_t)(RS_FRAMES * orig + K) * sizeof(float);
  if (sh > 150 * 1024) return MMSA_ERR_LIMIT;
  if (sh > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(resample_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh) != hipSuccess)
    return MMSA_ERR_LAUNCH;
  const long n_res = ((long)nw * T + orig - 1) / orig;
  const int frames = mmsa_cdiv(n_out, nw);
  dim3 grid(mmsa_cdiv(frames, RS_FRAMES), batch);
  hipLaunchKernelGGL(resample_kernel, grid, dim3(RS_THREADS), sh, mmsa_stream(stream), pcm, channels, T, stride_b, kernT,
                     ld, orig, nw, width, K, n_res, out, n_out);
  return mmsa_launch_status();
}


// ------------------------------------------------------------------------------------------------ waveform augmentation
// Training-time waveform augmentations of `CNN implementation/preprocess.py:56-90,137-140` for a whole batch in one
// launch, fused with the pad / trim (`:151-156`).  The random decisions are made on the host (Python `random`, in the
// reference's draw order) and arrive as one row of four doubles per utterance:
//   [0] shift mode: 0 none, 1 right, 2 left      [1] shift in samples (0 with mode 1/2 silences the utterance, as the
//   reference's empty `[:-0]` slices do)         [2] 1 = add Gaussian noise       [3] resampling factor (0 = none)
// Composition: out[j] = aug[idx(j)], idx(j) = trunc(float(j * factor)) clamped to n_in-1 for j < ceil(n_in / factor),
// aug[i] = shifted[i] + noise_factor * N(0,1).  With `lens` (optional, one int per utterance) row b holds an utterance of
// lens[b] <= n_in samples and is augmented as a signal of that length (an empty one gives zeros).  The noise is a counter-based Box-Muller draw (seed, utterance, sample):
// the same distribution as the reference's torch.randn_like, not the same stream.
namespace {

__global__ __launch_bounds__(256) void augment_wave_kernel(const float* __restrict__ in, long in_stride, int n_in,
                                                           const int* __restrict__ lens,
                                                           const double* __restrict__ params, float noise_factor,
                                                           unsigned long long seed, float* __restrict__ out,
                                                           int n_target) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_target) return;
  if (lens) {
    const int n_row = lens[b];
    n_in = n_row < n_in ? n_row : n_in;
    if (n_in <= 0) {
      out[(long)b * n_target + j] = 0.f;
      return;
    }
  }
  const double* pr = params + 4 * b;
  const int mode = (int)pr[0];
  const long shift = (long)pr[1];
  const bool noisy = pr[2] != 0.0;
  const double factor = pr[3];
  const float* x = in + (long)b * in_stride;
  long idx = j, n_len = n_in;
  if (factor > 0.0) {
    n_len = (long)ceil((double)n_in / factor);
    // torch.arange(0, n, factor) evaluates i*factor in double, stores float32; .long() truncates
    idx = (long)(float)((double)j * factor);
    if (idx > n_in - 1) idx = n_in - 1;
  }
  float v = 0.f;
  if (j < n_len) {
    long src = idx;
    bool live = true;
    if (mode == 1) { src = idx - shift; live = shift > 0 && src >= 0; }
    if (mode == 2) { src = idx + shift; live = shift > 0 && src < n_in; }
    if (live) v = x[src];
    if (noisy) {
      const unsigned long long key = ((unsigned long long)b << 32) + (unsigned long long)idx;
      const float u1 = 1.0f - hash_uniform(seed, 2 * key), u2 = hash_uniform(seed, 2 * key + 1);   // u1 in (0,1]
      const float gauss = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
      {
#pragma clang fp contract(off)   // two roundings, as `audio + noise * factor` (:78-79), not one fused multiply-add
        const float scaled = gauss * noise_factor;
        v = v + scaled;
      }
    }
  }
  out[(long)b * n_target + j] = v;
}

}  // namespace

MMSA_EXPORT int mmsa_augment_wave(const float* in, long in_stride, int n_in, const int* lens, const double* params,
                                  float noise_factor, unsigned long long seed, float* out, int n_target, int batch,
                                  void* stream) {
  MMSA_REQUIRE(in && params && out && n_in > 0 && n_target > 0 && batch > 0 && batch <= 65535 && in_stride >= n_in);
  dim3 grid(mmsa_cdiv(n_target, 256), batch);
  hipLaunchKernelGGL(augment_wave_kernel, grid, dim3(256), 0, mmsa_stream(stream), in, in_stride, n_in, lens,
                     params, noise_factor, seed, out, n_target);
  return mmsa_launch_status();
}

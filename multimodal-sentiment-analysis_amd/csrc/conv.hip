// Convolutions of the audio and text branches as implicit GEMMs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, 64-wide wavefronts).
//
// Activations are channels-last: X[b][h][w][c] (the reference's NCHW tensors never materialise inside the
// branch; its input has one channel and its output is pooled to [B, C], so the layout is private).  The same
// kernel covers: the 3x3 / pad 1 convolutions and 1x1 shortcuts of `ResidualBlock`
// (`CNN implementation/multimodal_model.py:48-59,64-68`), their data gradients (a 3x3 convolution of the
// output gradient with the tap-mirrored, channel-transposed weights), and the text branch's width-3/4/5
// `Conv1d` over the gathered embeddings (`:152-155,186-188`; H = 1, taps along W, no padding).
//
//   Y[b, p, n] = sum over taps t, channels c of  X[b, p + off_t, c] * Wp[t][n][c]
//
// Forward / data-gradient kernel (conv_igemm): workgroup tile BM positions x BN output channels, K consumed in
// chunks of 32 input channels of one tap; both operand tiles are staged through LDS (rows of 32 floats padded
// to 36 so the 16-byte fragment reads are bank-conflict free) with register prefetch of the next chunk and two
// LDS buffers (one barrier per chunk).  Each lane reads 4 consecutive k per ds_read_b128 and feeds them to 4
// successive MFMAs; A and B use the same k permutation, so the contraction is unchanged.
// Optional epilogue: per-tile column sums and sums of squares (the BatchNorm batch statistics and the
// squeeze-excite global average come from these, so the raw output is not re-read).
//
// Weight-gradient kernel (conv_wgrad): D[cin][cout] += X[p+off][cin] * dY[p][cout] summed over positions.  In
// channels-last layout both MFMA operands are 128-byte contiguous channel runs of one position, so lanes load
// them straight from global memory (no LDS); each wave keeps all taps' 32x32 accumulators, the four waves of
// a workgroup take different image rows and are combined through LDS, and workgroups write partial slabs that
// a second kernel sums (fixed order: results are run-to-run identical) into the reference's
// [Cout][Cin][kh][kw] layout.
#include "mmsa_common.h"

namespace {

constexpr int MAX_TAPS = 9;
constexpr int KC = 32;        // channels per K chunk
constexpr int LDS_STRIDE = 36;  // floats per staged row (32 + 4 pad)

struct ConvGeom {
  int B, H, W, Cin, Cout, ntaps;
  int dh[MAX_TAPS], dw[MAX_TAPS];
};

// ------------------------------------------------------------------------------------------------
// weight packing:  W[Cout][Cin][ntaps] (reference layout, taps = kh*kw flattened)  ->
//   fwd  [t][Cout][Cin]   (N = Cout rows, K = Cin contiguous)
//   dgrad[t][Cin][Cout]   (N = Cin rows,  K = Cout contiguous), same tap index (offsets are negated by the caller)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ dgrad,
                                    int Cout, int Cin, int ntaps) {
  const long total = (long)Cout * Cin * ntaps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % ntaps);
    const long r = i / ntaps;
    const int ci = (int)(r % Cin);
    const int co = (int)(r / Cin);
    const float v = w[i];
    if (fwd) fwd[((long)t * Cout + co) * Cin + ci] = v;
    if (dgrad) dgrad[((long)t * Cin + ci) * Cout + co] = v;
  }
}

// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, bool STATS, bool ACCUM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const float* __restrict__ X, const float* __restrict__ Wp,
                                                         float* __restrict__ Y, float* __restrict__ stats,
                                                         ConvGeom g, int tiles_per_img) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold whole 32x32 MFMA tiles");
  constexpr int A_LD = BM / 32;  // float4 loads per thread for the A tile (256 threads x 8 float4 per row)
  constexpr int B_LD = BN / 32;

  extern __shared__ float lds[];
  constexpr int BUF_FLOATS = (BM + BN) * LDS_STRIDE;  // one staging buffer: A rows then B rows

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int HW = g.H * g.W;
  const int b = blockIdx.x / tiles_per_img;
  const int tile = blockIdx.x - b * tiles_per_img;
  const int p0 = tile * BM;
  const int n0 = blockIdx.y * BN;

  // per-thread staging coordinates
  const int c4 = t & 7;         // which float4 of the 32-channel chunk
  const int rbase = t >> 3;     // 0..31
  int ph[A_LD], pw[A_LD];
  bool pv[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int p = p0 + rbase + 32 * j;
    pv[j] = p < HW;
    ph[j] = p / g.W;
    pw[j] = p - ph[j] * g.W;
  }
  const float* Xb = X + (long)b * HW * g.Cin;
  const int chunks_per_tap = g.Cin / KC;
  const int nchunks = g.ntaps * chunks_per_tap;

  f32x4 ra[A_LD], rb[B_LD];
  auto load_chunk = [&](int ck) {
    const int tap = ck / chunks_per_tap;
    const int cin0 = (ck - tap * chunks_per_tap) * KC;
    const int dh = g.dh[tap], dw = g.dw[tap];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int hh = ph[j] + dh, ww = pw[j] + dw;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pv[j] && hh >= 0 && hh < g.H && ww >= 0 && ww < g.W)
        v = *reinterpret_cast<const f32x4*>(Xb + ((long)(hh * g.W + ww)) * g.Cin + cin0 + 4 * c4);
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int n = n0 + rbase + 32 * j;
      rb[j] = *reinterpret_cast<const f32x4*>(Wp + ((long)tap * g.Cout + n) * g.Cin + cin0 + 4 * c4);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      *reinterpret_cast<f32x4*>(lds + buf * BUF_FLOATS + (rbase + 32 * j) * LDS_STRIDE + 4 * c4) = ra[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      *reinterpret_cast<f32x4*>(lds + buf * BUF_FLOATS + (BM + rbase + 32 * j) * LDS_STRIDE + 4 * c4) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  const int li = lane & 31, lh = lane >> 5;
  for (int ck = 0; ck < nchunks; ++ck) {
    const int buf = ck & 1;
    if (ck + 1 < nchunks) load_chunk(ck + 1);
    const float* Ab = lds + buf * BUF_FLOATS + (wm * WTM + li) * LDS_STRIDE + 4 * lh;
    const float* Bb = lds + buf * BUF_FLOATS + (BM + wn * WTN + li) * LDS_STRIDE + 4 * lh;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_STRIDE + 8 * gk);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_STRIDE + 8 * gk);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    }
    if (ck + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Yb = Y + (long)b * HW * g.Cout;
  float csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) csum[j] = csq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int p = p0 + row;
        if (p < HW) {
          float v = acc[i][j][r];
          float* dst = Yb + (long)p * g.Cout + col;
          if (ACCUM) v += *dst;
          *dst = v;
          if (STATS) {
            csum[j] += v;
            csq[j] = fmaf(v, v, csq[j]);
          }
        }
      }
    }
  if (STATS) {
    // combine the two lane halves, then the WAVES_M waves sharing a column range (through LDS, fixed order)
    float* red = lds;  // safe: every wave passed the loop's final barrier
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      csum[j] += __shfl_xor(csum[j], 32, 64);
      csq[j] += __shfl_xor(csq[j], 32, 64);
      if (lh == 0) {
        red[(wm * 2 + 0) * BN + wn * WTN + j * 32 + li] = csum[j];
        red[(wm * 2 + 1) * BN + wn * WTN + j * 32 + li] = csq[j];
      }
    }
    __syncthreads();
    if (t < 2 * BN) {
      const int which = t / BN, col = t - which * BN;
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < WAVES_M; ++m) s += red[(m * 2 + which) * BN + col];
      // stats[(b*tiles + tile)][2][Cout]
      stats[((long)blockIdx.x * 2 + which) * g.Cout + n0 + col] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient.  grid = (tile pairs (cin32 x cout32), nsplit).  Rows (b,h) of the image are dealt to
// (split, wave) round-robin.  partial[split][tap][Cin][Cout].
template <int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                         float* __restrict__ partial, ConvGeom g, int w_valid) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kk = lane >> 5;
  const int cout_tiles = g.Cout / 32;
  const int cin0 = (blockIdx.x / cout_tiles) * 32;
  const int cout0 = (blockIdx.x % cout_tiles) * 32;
  const int nsplit = gridDim.y;
  const int rows = g.B * g.H;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int row = blockIdx.y * 4 + wave; row < rows; row += nsplit * 4) {
    const int b = row / g.H, h = row - b * g.H;
    const float* dYr = dY + ((long)row * g.W) * g.Cout + cout0 + li;
    const float* Xb = X + ((long)b * g.H * g.W) * g.Cin + cin0 + li;
    for (int w0 = 0; w0 < w_valid; w0 += 2) {
      const int w = w0 + kk;
      const bool wv = w < w_valid;
      const float dy = wv ? dYr[(long)w * g.Cout] : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int hh = h + g.dh[t], ww = w + g.dw[t];
        float xv = 0.f;
        if (wv && hh >= 0 && hh < g.H && ww >= 0 && ww < g.W) xv = Xb[((long)(hh * g.W + ww)) * g.Cin];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, dy, acc[t], 0, 0, 0);
      }
    }
  }

  // combine the 4 waves: waves 2,3 -> LDS; waves 0,1 add; wave 1 -> LDS; wave 0 adds and writes.
  float* bufA = lds;
  float* bufB = lds + NT * 1024;
  if (wave >= 2) {
    float* dst = (wave == 2 ? bufA : bufB);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wave < 2) {
    const float* src = (wave == 0 ? bufA : bufB);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] += src[(t * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) bufA[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* out = partial + (long)blockIdx.y * NT * g.Cin * g.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[t][r] + bufA[(t * 16 + r) * 64 + lane];
        const int ci = cin0 + (r & 3) + 8 * (r >> 2) + 4 * kk;   // D row
        out[((long)t * g.Cin + ci) * g.Cout + cout0 + li] = v;   // D col = lane & 31
      }
  }
}

// partial[split][tap][Cin][Cout] -> dW[Cout][Cin][ntaps]  (+= when accumulate)
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int nsplit, int ntaps,
                                    int Cin, int Cout, int accumulate) {
  const long total = (long)ntaps * Cin * Cout;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += partial[(long)k * total + i];
  const int co = (int)(i % Cout);
  const long r = i / Cout;
  const int ci = (int)(r % Cin);
  const int t = (int)(r / Cin);
  float* dst = dW + ((long)co * Cin + ci) * ntaps + t;
  *dst = accumulate ? (*dst + s) : s;
}

int fill_geom(ConvGeom& g, int B, int H, int W, int Cin, int Cout, int ntaps, const int* dh, const int* dw) {
  if (ntaps < 1 || ntaps > MAX_TAPS) return MMSA_ERR_ARG;
  g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ntaps = ntaps;
  for (int i = 0; i < MAX_TAPS; ++i) {
    g.dh[i] = i < ntaps ? dh[i] : 0;
    g.dw[i] = i < ntaps ? dw[i] : 0;
  }
  return MMSA_OK;
}

template <int BM, int BN, int WM_, int WN_>
int launch_igemm(const float* X, const float* Wp, float* Y, float* stats, const ConvGeom& g, int accumulate,
                 hipStream_t s) {
  const int HW = g.H * g.W;
  const int tiles = mmsa_cdiv(HW, BM);
  dim3 grid(g.B * tiles, g.Cout / BN);
  const size_t shmem = 2 * (size_t)(BM + BN) * LDS_STRIDE * sizeof(float);
  if (stats) {
    if (accumulate) return MMSA_ERR_ARG;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM_, WN_, true, false>), grid, dim3(256), shmem, s, X, Wp, Y, stats,
                       g, tiles);
  } else if (accumulate) {
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM_, WN_, false, true>), grid, dim3(256), shmem, s, X, Wp, Y, stats,
                       g, tiles);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM_, WN_, false, false>), grid, dim3(256), shmem, s, X, Wp, Y,
                       stats, g, tiles);
  }
  return mmsa_launch_status();
}

}  // namespace

MMSA_EXPORT int mmsa_conv_pack_weights(const float* w, float* fwd, float* dgrad, int Cout, int Cin, int ntaps,
                                       void* stream) {
  MMSA_REQUIRE(w && (fwd || dgrad) && Cout > 0 && Cin > 0 && ntaps > 0);
  const long total = (long)Cout * Cin * ntaps;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, mmsa_stream(stream), w, fwd, dgrad, Cout, Cin,
                     ntaps);
  return mmsa_launch_status();
}

// Rows of statistics the forward kernel writes for a given geometry: B * tiles_per_image, where the tile height
// (BM) is the one mmsa_conv_fwd picks for that geometry.
static int pick_bm(int B, int H, int W, int Cout) {
  // prefer 128-row tiles; fall back to 64 when the grid would leave most of the 256 CUs idle
  const long blocks128 = (long)B * mmsa_cdiv((long)H * W, 128) * (Cout / 64 > 0 ? Cout / 64 : 1);
  return blocks128 >= 384 ? 128 : 64;
}

MMSA_EXPORT int mmsa_conv_tile_rows(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MMSA_ERR_ARG;
  return pick_bm(B, H, W, Cout);
}

// Y[B,H,W,Cout] = conv(X[B,H,W,Cin], Wp[ntaps][Cout][Cin]) with per-tap offsets (dh, dw); out-of-image reads are 0.
// stats (optional): [B*ceil(H*W/BM)][2][Cout] column sums / sums of squares per tile, BM = mmsa_conv_tile_rows().
// accumulate != 0: Y += result (stats must be null).
MMSA_EXPORT int mmsa_conv_fwd(const float* X, const float* Wp, float* Y, float* stats, int B, int H, int W, int Cin,
                              int Cout, int ntaps, const int* dh, const int* dw, int accumulate, void* stream) {
  MMSA_REQUIRE(X && Wp && Y && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  MMSA_REQUIRE(Cin % KC == 0 && Cout % 32 == 0);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  hipStream_t s = mmsa_stream(stream);
  if (Cout % 64 != 0) {
    // 32-wide output (data gradient into a 32-channel tensor): 128 x 32 tiles, 4 x 1 waves
    if (stats) return MMSA_ERR_ARG;
    return launch_igemm<128, 32, 4, 1>(X, Wp, Y, nullptr, g, accumulate, s);
  }
  if (pick_bm(B, H, W, Cout) == 128) return launch_igemm<128, 64, 2, 2>(X, Wp, Y, stats, g, accumulate, s);
  return launch_igemm<64, 64, 2, 2>(X, Wp, Y, stats, g, accumulate, s);
}

MMSA_EXPORT long mmsa_conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int ntaps, int* nsplit_out) {
  if (B <= 0 || H <= 0 || Cin % 32 || Cout % 32 || ntaps < 1 || ntaps > MAX_TAPS) return MMSA_ERR_ARG;
  const int pairs = (Cin / 32) * (Cout / 32);
  const int rows = B * H;
  int nsplit = mmsa_cdiv(512, pairs);          // aim for >= 512 workgroups
  const int max_split = mmsa_cdiv(rows, 4);    // at least one row per wave
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  if (nsplit_out) *nsplit_out = nsplit;
  return (long)nsplit * ntaps * Cin * Cout;
}

// dW[Cout][Cin][ntaps] (= or +=) sum over b,h,w of X[b,h+dh,w+dw,cin] * dY[b,h,w,cout]; w < w_valid only.
MMSA_EXPORT int mmsa_conv_wgrad(const float* X, const float* dY, float* dW, float* workspace, int B, int H, int W,
                                int w_valid, int Cin, int Cout, int ntaps, const int* dh, const int* dw,
                                int accumulate, void* stream) {
  MMSA_REQUIRE(X && dY && dW && workspace && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && w_valid > 0 && w_valid <= W);
  MMSA_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0);
  MMSA_REQUIRE(ntaps == 1 || ntaps == 3 || ntaps == 4 || ntaps == 5 || ntaps == 9);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  int nsplit = 1;
  mmsa_conv_wgrad_workspace_floats(B, H, W, Cin, Cout, ntaps, &nsplit);
  hipStream_t s = mmsa_stream(stream);
  dim3 grid((Cin / 32) * (Cout / 32), nsplit);
  const size_t shmem = 2 * (size_t)ntaps * 1024 * sizeof(float);
  switch (ntaps) {
    case 1: hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), shmem, s, X, dY, workspace, g, w_valid); break;
    case 3: hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(256), shmem, s, X, dY, workspace, g, w_valid); break;
    case 4: hipLaunchKernelGGL(conv_wgrad_kernel<4>, grid, dim3(256), shmem, s, X, dY, workspace, g, w_valid); break;
    case 5: hipLaunchKernelGGL(conv_wgrad_kernel<5>, grid, dim3(256), shmem, s, X, dY, workspace, g, w_valid); break;
    default: {
      // 9 taps need 72 KiB of LDS for the wave combine
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<9>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
        return MMSA_ERR_LAUNCH;
      hipLaunchKernelGGL(conv_wgrad_kernel<9>, grid, dim3(256), shmem, s, X, dY, workspace, g, w_valid);
    }
  }
  const long total = (long)ntaps * Cin * Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, dW,
                     nsplit, ntaps, Cin, Cout, accumulate);
  return mmsa_launch_status();
}

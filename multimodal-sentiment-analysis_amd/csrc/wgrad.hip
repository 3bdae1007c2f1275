// Weight gradients of the residual blocks' convolutions (`CNN implementation/multimodal_model.py:48-59,64-68` under
// autograd), second generation: one workgroup per CU, operands streamed into LDS by DMA while the matrix cores work.
//
//   dW9[cout][cin][tap] = sum over b, h, w of  X[b, h + dh_tap, w + dw_tap, cin] * dY[b, h, w, cout]        (3x3, pad 1)
//   dW1[cout][cin]      = sum over b, h, w of  X[b, h, w, cin] * dY1[b, h, w, cout]                          (1x1 shortcut)
//
// The contraction runs over positions, of which there are 4 096 to 66 560 per layer at batch 32, while the output
// has only 1 to 32 tiles of 32 cin x 64 cout: the position range is split over workgroups (as many as the chip has
// CUs), each of which writes a partial slab, and a reduction sums the slabs in fixed order (run-to-run identical).
//
// Workgroup = 8 waves, two per SIMD.  At the top of a strip every wave issues its share of the `buffer_load ... lds`
// requests for the NEXT strip: R image rows of X with a zero halo ([R+2][W+2][32 cin]; halo and rows outside the image
// come from out-of-range buffer offsets, which read as zero) and of dY ([R][W][64 cout], and the shortcut's dY1
// alike), 1 KiB per wave instruction, into the other of two LDS buffers.  Then it works on the current strip with
// v_mfma_f32_32x32x2_f32: wave (jt, q, rh) owns output columns cout0 + 32 jt .. + 31, taps 0-4 (q = 0) or 5-8 (q = 1),
// i.e. five or four 32 x 32 accumulators that live in registers across all strips, and half rh of the strip's
// positions.  The two waves of a SIMD have opposite q, so every SIMD
// carries nine products per step and each wave's address arithmetic and LDS reads hide under its partner's MFMAs.
// Every operand is one conflict-free ds_read_b32 (a lane half reads 32 consecutive channels of one position); the
// reads of step s+1 are issued among the MFMAs of step s.  One barrier per strip hands the filled buffer over and
// the drained one back, so the DMA of strip k+1 runs under the matrix work of strip k (the first generation staged
// through registers between two barriers and left the matrix pipe idle for a quarter of the kernel).  At the end the
// rh halves are added through LDS.
//
// The 1x1 shortcut's gradient (same X, its own dY1) is a second pass over the workgroup's strips inside the same
// launch: the strips stream through the same two buffers once more with dY1 in dY's place, and every wave forms the
// single product X(centre) * dY1 for a quarter of each strip's positions.  (Staging dY1 beside dY in the first pass
// halves the strip height that fits in LDS -- one image row at width 65 -- and with it the 3x3 products' efficiency:
// 46 / 62 / 70 TFLOP/s for the three blocks' launches against 72 / 88 / 95 without the shortcut.)
//
// Reduction: wgrad_reduce_multi_kernel sums the slabs of any number of layers in one launch, into the reference's
// [Cout][Cin][kh][kw] layout.
#include "mmsa_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

struct Wg2Geom {
  int B, H, W, Cin, Cout;
  int dh[9], dw[9];
};

constexpr int WG2_MAXP = 14;          // LDS-DMA pieces (1 KiB) per wave and strip
constexpr int WG2_OOB = 0x7ffffff0;   // a byte offset no tensor reaches: the hardware answers with zeros
constexpr size_t WG2_LDS_MAX = 156 * 1024;
// LDS one workgroup may take for its two strip buffers.  MMSA_WGRAD2_LDS_KB overrides (tuning aid): a smaller budget
// means shorter strips (more halo re-reads), and room on the CU for the small kernels of the data-gradient chain.
static size_t wg2_lds_budget() {
  static long kb = -1;
  if (kb < 0) {
    const char* e = getenv("MMSA_WGRAD2_LDS_KB");
    kb = e ? atol(e) : 156;
    if (kb < 32 || kb > 156) kb = 156;
  }
  return (size_t)kb * 1024;
}
constexpr size_t WG2_PARK_BYTES = 4 * 5 * 16 * 64 * 4;
constexpr size_t WG2_PARK1_BYTES = 6 * 16 * 64 * 4;

struct Wg2Plan {
  int R, strips_per_img, nstrips, nsplit;
  int xpieces, ypieces;      // 1 KiB pieces per strip: X region, one dY region
  size_t shmem;
};

// Strip height: the largest R (rows per strip) whose two buffers fit in LDS and whose pieces fit the loader waves'
// tables; nsplit: workgroups per output tile, so that the launch has about one workgroup per CU.
Wg2Plan wg2_plan(int B, int H, int W, int Cin, int Cout, bool sc) {
  Wg2Plan p{};
  const int WP = W + 2;
  // among the heights whose two buffers fit, the one with the least work per image: every strip costs about one row
  // more than it holds (halo rows, hand-over), and rows past the image in the last strip are computed as zeros
  int best = 0;
  long best_cost = 0;
  for (int R = 1; R <= H && R <= 16; ++R) {
    const int xp = ((R + 2) * WP + 7) / 8, yp = (R * W + 3) / 4;
    const int pieces = xp + yp;
    if (2 * (size_t)pieces * 1024 > wg2_lds_budget() || (pieces + 7) / 8 > WG2_MAXP) break;
    const long cost = (long)((H + R - 1) / R) * (R + 1);
    if (!best || cost <= best_cost) { best = R; best_cost = cost; }
  }
  if (!best) return p;   // R = 0: two one-row strips do not fit (rows wider than about 120 positions)
  p.R = best;
  p.strips_per_img = (H + best - 1) / best;
  p.nstrips = B * p.strips_per_img;
  p.xpieces = ((p.R + 2) * WP + 7) / 8;
  p.ypieces = (p.R * W + 3) / 4;
  const int pairs = (Cin / 32) * (Cout / 64);
  static long wgs = -1;                                   // MMSA_WGRAD2_WGS overrides (tuning aid)
  if (wgs < 0) {
    const char* e = getenv("MMSA_WGRAD2_WGS");
    wgs = e ? atol(e) : 256;
    if (wgs < 32 || wgs > 1024) wgs = 256;
  }
  int nsplit = (int)wgs / pairs;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > p.nstrips) nsplit = p.nstrips;
  const int per = (p.nstrips + nsplit - 1) / nsplit;      // strips per workgroup, evenly dealt
  p.nsplit = (p.nstrips + per - 1) / per;
  p.shmem = 2 * (size_t)(p.xpieces + p.ypieces) * 1024;
  // the epilogue parks four waves' accumulators, and six waves' shortcut accumulators behind them
  const size_t park = WG2_PARK_BYTES + (sc ? WG2_PARK1_BYTES : 0);
  if (p.shmem < park) p.shmem = park;
  return p;
}

#ifndef WG2_ORDER
#define WG2_ORDER 0
#endif
#ifdef MMSA_LAB_STAMPS
// Diagnostic builds only (tools/wgrad2_lab.hip): per-workgroup cycle counts, written to a buffer of their own.
__device__ unsigned long long* g_wg2_stamps;
#endif

// WFIX: the map's width when it is one of the model's (16, 32, 65: the row pitch of the strip then folds into the
// LDS instructions' immediate offsets), 0 for any other width.
template <bool SC, int WFIX>
__global__ __launch_bounds__(512) void conv_wgrad2_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                          const float* __restrict__ dY1, float* __restrict__ partial,
                                                          float* __restrict__ partial1, const Wg2Geom g, int R,
                                                          int nstrips, int strips_per_img, int xpieces, int ypieces) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int t = threadIdx.x, lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  // Wave w8: output columns cout0 + 32 jt .., half `rh` of each strip's positions, tap group q.  Waves w8 and w8 + 4
  // share a SIMD: they get opposite tap groups (five and four products per step), so every SIMD carries nine.
#ifndef WG2_PAIR
#define WG2_PAIR 0
#endif
#if WG2_PAIR == 0
  const int jt = w8 & 1, rh = w8 >> 2, q = ((w8 >> 1) & 1) ^ rh;
#elif WG2_PAIR == 1
  const int jt = w8 & 1, rh = w8 >> 2, q = (w8 >> 1) & 1;
#elif WG2_PAIR == 2
  const int q = w8 & 1, jt = (w8 >> 1) & 1, rh = w8 >> 2;       // neighbours w8, w8 + 1 get opposite tap groups
#else
  const int q = (w8 & 1) ^ ((w8 >> 1) & 1), jt = (w8 >> 1) & 1, rh = w8 >> 2;
#endif
  const int W = WFIX ? WFIX : g.W;
  const int WP = W + 2;
  const int nsplit = gridDim.y;
  const int cgroups = g.Cout / 64;
  const int cin0 = ((int)blockIdx.x / cgroups) * 32;
  const int cout0 = ((int)blockIdx.x % cgroups) * 64;
  const int xfl = xpieces * 256, yfl = ypieces * 256;                 // floats per region
  const int buf_fl = xfl + yfl;
  const int first = blockIdx.y;
  const int count = (nstrips - first + nsplit - 1) / nsplit;          // strips first, first + nsplit, ...

  // ------------------------------------------------------------------------------------------ DMA plan
  const long xbytes = (long)g.B * g.H * W * g.Cin * 4, ybytes = (long)g.B * g.H * W * g.Cout * 4;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SC ? dY1 : dY), 0, (int)ybytes, 0x00020000);
  // Piece p of a strip (this wave: p = w8, w8 + 8, ...): X pieces cover 8 positions x 128 B, dY pieces 4 positions x
  // 256 B.  Per lane and piece, fixed for the whole kernel: the byte offset relative to the strip's first row, and the
  // image row relative to it (a value that fails every row test marks a lane that always reads zero: halo columns,
  // positions past the end of the strip).
  const int total = xpieces + ypieces;
  int rel[WG2_MAXP], row[WG2_MAXP];
#pragma unroll
  for (int i = 0; i < WG2_MAXP; ++i) {
    const int p = w8 + 8 * i;
    rel[i] = 0;
    row[i] = -(1 << 20);
    if (p < xpieces) {
      const int pos = 8 * p + (lane >> 3), c4 = lane & 7;
      const int r = pos / WP, c = pos - r * WP;
      if (r < R + 2 && c >= 1 && c <= W) {
        rel[i] = (((r - 1) * W + (c - 1)) * g.Cin + cin0 + 4 * c4) * 4;
        row[i] = r - 1;
      }
    } else if (p < total) {
      const int py = p - xpieces;
      const int pos = 4 * py + (lane >> 4), c4 = lane & 15;
      const int r = pos / W, c = pos - r * W;
      if (r < R) {
        rel[i] = ((r * W + c) * g.Cout + cout0 + 4 * c4) * 4;
        row[i] = r;
      }
    }
  }
  // strip number k of this workgroup -> buffer slot & 1 (ry: the gradient map that fills the dY region)
  auto request = [&](int k, int slot, const __amdgpu_buffer_rsrc_t& ry) {
    const int strip = first + k * nsplit;
    const int b = strip / strips_per_img;
    const int r0 = (strip - b * strips_per_img) * R;
    const int xbase = ((b * g.H + r0) * W) * g.Cin * 4;
    const int ybase = ((b * g.H + r0) * W) * g.Cout * 4;
    float* buf = lds + (slot & 1) * buf_fl;
#pragma unroll
    for (int i = 0; i < WG2_MAXP; ++i) {
      const int p = w8 + 8 * i;        // uniform
      if (p >= total) break;
      const bool ok = (unsigned)(r0 + row[i]) < (unsigned)g.H;
      if (p < xpieces) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(buf + p * 256), 16, ok ? xbase + rel[i] : WG2_OOB, 0, 0,
                                                 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void*)(buf + p * 256), 16, ok ? ybase + rel[i] : WG2_OOB, 0, 0,
                                                 0);
      }
    }
  };

  // ------------------------------------------------------------------------------------------ matrix plan
  const int li = lane & 31, kk = lane >> 5;
  // q = 0: taps 0..4 (row above: dw -1, 0, +1; centre row: dw -1, 0); q = 1: taps 5..8 (centre row: dw +1; row below:
  // dw -1, 0, +1).  The taps of one image
  // row are 128 bytes apart in the strip, so each step needs two X addresses (a full row of three taps, the centre
  // row) and constant offsets; a lane half (kk) handles one of the step's two positions.
  const int t_first = q == 0 ? 0 : 5;
  f32x16 acc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x16 acc1, acc1b;                        // the shortcut's 32 cin x 32 cout block (second pass), even / odd steps
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = acc1b[r] = 0.f;

  const int nsteps = (W + 1) >> 1;           // steps per image row (two positions each)
  const int T = R * nsteps;                  // steps per strip, row by row
  const int s_begin = rh == 0 ? 0 : (T + 1) / 2, s_end = rh == 0 ? (T + 1) / 2 : T;   // this wave's share
  const int nmine = s_end - s_begin;
  const int r_begin = s_begin / nsteps, st_begin = s_begin - r_begin * nsteps;
  const int lane_x = li * 4 + kk * 128;
  const int laneY = (jt * 32 + li) * 4 + kk * 256;
  const int dx_wrap = WP * 128 - (nsteps - 1) * 256, dy_wrap = W * 256 - (nsteps - 1) * 512;
  constexpr bool ODD = WFIX ? (WFIX & 1) != 0 : true;   // the last step of an odd-width row has one position only

  // One strip.  QSEL 0 / 1: taps 0-4 / taps 5-8.  Two operand sets alternate: the LDS
  // reads of step s+1 are issued among the MFMAs of step s.  Scalars walk the wave's steps row by row (xs / ys: byte
  // offsets of the step's first position).  One X address per step: the upper of the two image rows the wave's
  // taps touch, at dw = -1; the other row is one strip pitch further, its taps 128 bytes apart.
  auto strip = [&](const char* buf, auto qsel_tag) {
    constexpr int QSEL = decltype(qsel_tag)::value;
    constexpr int TC = QSEL == 1 ? 4 : 5;
    static_assert(QSEL == 0 || QSEL == 1, "tap group");
    const int pitch = WP * 32;                                   // floats between image rows of the strip
    const char* xb = buf + lane_x - 128 - (QSEL == 0 ? WP * 128 : 0);
    const char* yb = buf + xfl * 4 + laneY;
    int st = st_begin;
    int xs = ((r_begin + 1) * WP + 1) * 128 + st_begin * 256, ys = r_begin * W * 256 + st_begin * 512;
    struct Ops {
      float a[5], dy;
    };
    const float *pX, *pY;
    bool dead = false;
    auto addresses = [&]() {
      pX = reinterpret_cast<const float*>(xb + xs);
      pY = reinterpret_cast<const float*>(yb + ys);
      if (ODD) dead = (W & 1) && st == nsteps - 1 && kk == 1;   // the position past the end of an odd-width row
    };
    // the step's reads in three groups (issued between the MFMAs of the step before)
    auto read_y = [&](Ops& o) {
      o.dy = pY[0];
      o.a[0] = QSEL == 0 ? pX[0] : pX[64];
    };
    auto read_x1 = [&](Ops& o) {
      if (QSEL == 0) { o.a[1] = pX[32]; o.a[2] = pX[64]; }
      else { o.a[1] = pX[pitch]; o.a[2] = pX[pitch + 32]; }
    };
    auto read_x2 = [&](Ops& o) {
      if (QSEL == 0) { o.a[3] = pX[pitch]; o.a[4] = pX[pitch + 32]; }
      else { o.a[3] = pX[pitch + 64]; }
      if (ODD) o.dy = dead ? 0.f : o.dy;
    };
    auto advance = [&]() {
      const bool wrap = st == nsteps - 1;                // uniform: on to the next image row
      xs += wrap ? dx_wrap : 256;
      ys += wrap ? dy_wrap : 512;
      st = wrap ? 0 : st + 1;
    };
    auto mfma = [&](const Ops& o, int i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i], o.dy, acc[i], 0, 0, 0);
    };
    // one step: the MFMAs of `cur` with the reads of `nxt` (whose addresses are ready) between them, then the scalar
    // and address arithmetic of the step after `nxt` under the last MFMAs -- in exactly this order
    auto step = [&](const Ops& cur, Ops& nxt, bool move_on) {
      mfma(cur, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_y(nxt);
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      read_x1(nxt);
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 2);
      __builtin_amdgcn_sched_barrier(0);
      read_x2(nxt);
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 3);
      __builtin_amdgcn_sched_barrier(0);
      if (move_on) {
        advance();
        addresses();
      }
      __builtin_amdgcn_sched_barrier(0);
      if (TC == 5) mfma(cur, 4);
      __builtin_amdgcn_sched_barrier(0);
    };
    // Steps in pairs, so that the two operand sets swap roles without register moves.  The last pair is peeled: with
    // an odd number of steps its second half works on a re-read of the last step with dY forced to zero.  (A loop
    // with a conditional tail made the compiler keep several copies of the accumulators.)
    Ops o0, o1;
    addresses();
    read_y(o0);
    read_x1(o0);
    read_x2(o0);
    const int npairs = (nmine + 1) >> 1;
    const bool padded = (nmine & 1) != 0;
    if (nmine > 1) {
      advance();
      addresses();
    }
    for (int i = 0; i + 1 < npairs; ++i) {
      step(o0, o1, true);
      step(o1, o0, true);
    }
    step(o0, o1, false);          // reads step nmine - 1 (or, padded, step nmine - 1 again)
    if (padded) o1.dy = 0.f;
#pragma unroll
    for (int i = 0; i < TC; ++i) mfma(o1, i);
  };

#ifdef MMSA_LAB_STAMPS
  const unsigned long long lab_t0 = __builtin_amdgcn_s_memtime(), lab_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long lab_comp = 0, lab_wait = 0, lab_t;
#endif
  request(0, 0, rY);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // strip 0 has landed
#ifdef MMSA_LAB_STAMPS
  const unsigned long long lab_t1 = __builtin_amdgcn_s_memtime();
#endif
  // The strip loop once per tap group, chosen outside it (the choice is the same for a wave's whole life): with the
  // choice inside the loop the compiler kept one set of accumulators per branch and copied 80 registers at every join.
  auto strips = [&](auto qsel_tag) {
    for (int k = 0; k < count; ++k) {
#ifdef MMSA_LAB_STAMPS
      lab_t = __builtin_amdgcn_s_memtime();
#endif
      // the next strip streams into the other buffer (drained one barrier ago) under this strip's matrix work; behind
      // the last strip the shortcut's pass starts over at the workgroup's first strip
      if (k + 1 < count) request(k + 1, k + 1, rY);
      else if (SC) request(0, count, rY1);
      const char* buf = reinterpret_cast<const char*>(lds + (k & 1) * buf_fl);
      if (nmine > 0) strip(buf, qsel_tag);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef MMSA_LAB_STAMPS
      { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); lab_comp += n_ - lab_t; lab_t = n_; }
#endif
      __builtin_amdgcn_s_barrier();
#ifdef MMSA_LAB_STAMPS
      lab_wait += __builtin_amdgcn_s_memtime() - lab_t;
#endif
    }
  };
  if (q == 0) strips(std::integral_constant<int, 0>{});
  else strips(std::integral_constant<int, 1>{});

  // Second pass (shortcut): strip k of the pass sits in buffer (count + k) & 1, its dY region holds dY1.  Wave
  // (jt, ph) takes quarter ph = 2 q + rh of the strip's steps: one product X(centre) * dY1 per step.
  const int ph = 2 * q + rh;
  if (SC) {
    const int b_begin = (T * ph) / 4, b_end = (T * (ph + 1)) / 4;
    const int rb = b_begin / nsteps, sb = b_begin - rb * nsteps;
    for (int k = 0; k < count; ++k) {
      if (k + 1 < count) request(k + 1, count + k + 1, rY1);
      const char* buf = reinterpret_cast<const char*>(lds + ((count + k) & 1) * buf_fl);
      const char* xb = buf + lane_x;                       // centre tap: the position itself
      const char* yb = buf + xfl * 4 + laneY;
      int st = sb;
      int xs = ((rb + 1) * WP + 1) * 128 + sb * 256, ys = rb * W * 256 + sb * 512;
      // two steps per trip into two accumulators: the second product does not wait for the first, and the next
      // trip's operand reads are on their way while both run
      auto fetch = [&](float& a, float& d) {
        a = *reinterpret_cast<const float*>(xb + xs);
        d = *reinterpret_cast<const float*>(yb + ys);
        if (ODD) d = ((W & 1) && st == nsteps - 1 && kk == 1) ? 0.f : d;
        const bool wrap = st == nsteps - 1;
        xs += wrap ? dx_wrap : 256;
        ys += wrap ? dy_wrap : 512;
        st = wrap ? 0 : st + 1;
      };
      int i = b_begin;
      for (; i + 1 < b_end; i += 2) {
        float a0, d0, a1, d1;
        fetch(a0, d0);
        fetch(a1, d1);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, d0, acc1, 0, 0, 0);
        acc1b = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, d1, acc1b, 0, 0, 0);
      }
      if (i < b_end) {
        float a0, d0;
        fetch(a0, d0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, d0, acc1, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
#ifdef MMSA_LAB_STAMPS
  const unsigned long long lab_t2 = __builtin_amdgcn_s_memtime();
#endif

  // The two halves of each (jt, q) pair meet in LDS (all strips are consumed: the buffers are free): the waves of
  // the second half park their accumulators, the waves of the first half add them to their own, in this fixed order.
  float* park = lds + (jt + 2 * q) * (5 * 16 * 64);
  float* park1 = lds + 4 * (5 * 16 * 64) + jt * (3 * 16 * 64);      // the shortcut's quarters 1..3 of column half jt
  if (SC) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] += acc1b[r];
    if (ph != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) park1[((ph - 1) * 16 + r) * 64 + lane] = acc1[r];
    }
  }
  if (rh == 1) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) park[(i * 16 + r) * 64 + lane] = acc[i][r];
  }
  __syncthreads();
  if (rh == 1) return;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] += park[(i * 16 + r) * 64 + lane];

  // slab [tap][Cin][Cout] of this workgroup (D row = cin, D col = lane & 31 = cout); shortcut slab [Cin][Cout]
  float* out = partial + (long)blockIdx.y * 9 * g.Cin * g.Cout;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int tap = t_first + i;
    if (tap < 9) {
      float* dst = out + (long)tap * g.Cin * g.Cout;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cin0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        dst[(long)ci * g.Cout + cout0 + jt * 32 + li] = acc[i][r];
      }
    }
  }
  if (SC && ph == 0) {      // quarters in fixed order: 0 + 1 + 2 + 3
    float* dst = partial1 + (long)blockIdx.y * g.Cin * g.Cout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc1[r];
#pragma unroll
      for (int k = 0; k < 3; ++k) v += park1[(k * 16 + r) * 64 + lane];
      const int ci = cin0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      dst[(long)ci * g.Cout + cout0 + jt * 32 + li] = v;
    }
  }
#ifdef MMSA_LAB_STAMPS
  if (lane == 0 && g_wg2_stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = g_wg2_stamps + (((long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (jt + 2 * q)) * 8;
    o[0] = lab_t1 - lab_t0;                          // first strip's arrival
    o[1] = lab_comp;                                 // matrix loops
    o[2] = lab_wait;                                 // barriers behind them
    o[3] = __builtin_amdgcn_s_memtime() - lab_t2;    // reduction through LDS and slab stores
    o[4] = __builtin_amdgcn_s_memtime() - lab_t0;
    o[5] = __builtin_amdgcn_s_memrealtime() - lab_r0;   // 100 MHz ticks: o[4] / o[5] * 0.1 = the wave's clock in GHz
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// slabs[k][tap][Cin][Cout], k < nslabs  ->  dW[Cout][Cin][ntaps], for a table of layers in one launch.  256 outputs
// per workgroup (one 16-byte piece per lane and slab); four waves take interleaved slabs, four independent loads
// each per round, and are combined in fixed order through LDS.
struct ReduceDesc {
  const float* slabs;
  float* dW;
  int nslabs, ntaps, Cin, Cout;
  int block0, nblocks;     // this layer's workgroups are block0 .. block0 + nblocks - 1 of the launch
};

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const ReduceDesc* __restrict__ descs, int n) {
  __shared__ f32x4 red[4][64];
  int which = 0;
  for (int i = 1; i < n; ++i)
    if ((int)blockIdx.x >= descs[i].block0) which = i;     // uniform
  const ReduceDesc d = descs[which];
  const long block = (long)blockIdx.x - d.block0;
  const long total = (long)d.ntaps * d.Cin * d.Cout;       // a multiple of 4
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = block * 256 + 4 * lane;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s0 = zero, s1 = zero, s2 = zero, s3 = zero;
  if (i < total) {
    const float* src = d.slabs + i;
    int k = w;
    f32x4 t0 = zero, t1 = zero, t2 = zero, t3 = zero;
    for (; k + 28 < d.nslabs; k += 32) {          // eight 16-byte loads in flight per lane
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (long)k * total);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (long)(k + 4) * total);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (long)(k + 8) * total);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (long)(k + 12) * total);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(src + (long)(k + 16) * total);
      const f32x4 a5 = *reinterpret_cast<const f32x4*>(src + (long)(k + 20) * total);
      const f32x4 a6 = *reinterpret_cast<const f32x4*>(src + (long)(k + 24) * total);
      const f32x4 a7 = *reinterpret_cast<const f32x4*>(src + (long)(k + 28) * total);
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      t0 += a4; t1 += a5; t2 += a6; t3 += a7;
    }
    s0 += t0; s1 += t1; s2 += t2; s3 += t3;
    for (; k + 12 < d.nslabs; k += 16) {
      s0 += *reinterpret_cast<const f32x4*>(src + (long)k * total);
      s1 += *reinterpret_cast<const f32x4*>(src + (long)(k + 4) * total);
      s2 += *reinterpret_cast<const f32x4*>(src + (long)(k + 8) * total);
      s3 += *reinterpret_cast<const f32x4*>(src + (long)(k + 12) * total);
    }
    for (; k < d.nslabs; k += 4) s0 += *reinterpret_cast<const f32x4*>(src + (long)k * total);
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  const int t = threadIdx.x;
  const long io = block * 256 + t;
  if (io < total) {
    const float s = (red[0][t >> 2][t & 3] + red[1][t >> 2][t & 3]) + (red[2][t >> 2][t & 3] + red[3][t >> 2][t & 3]);
    const int co = (int)(io % d.Cout);
    const long r = io / d.Cout;
    const int ci = (int)(r % d.Cin);
    const int tp = (int)(r / d.Cin);
    d.dW[((long)co * d.Cin + ci) * d.ntaps + tp] = s;
  }
}

int wg2_fill(Wg2Geom& g, int B, int H, int W, int Cin, int Cout, const int* dh, const int* dw) {
  g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
  for (int i = 0; i < 9; ++i) {
    if (dh[i] != i / 3 - 1 || dw[i] != i % 3 - 1) return MMSA_ERR_ARG;   // row-major 3 x 3 offsets (kh * 3 + kw)
    g.dh[i] = dh[i];
    g.dw[i] = dw[i];
  }
  return MMSA_OK;
}

bool wg2_geometry_ok(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64) return false;
  const long big = (long)B * H * W * (Cin > Cout ? Cin : Cout) * 4;
  return big < 0x7ff00000L;     // 32-bit byte offsets inside one tensor
}

}  // namespace

// Partial slabs a launch of mmsa_conv_wgrad_slabs writes for this geometry: *nslabs_host slabs, each [9][Cin][Cout]
// (returned: floats of all of them) and, with the shortcut, as many [Cin][Cout] slabs.
MMSA_EXPORT long mmsa_conv_wgrad_slab_floats(int B, int H, int W, int Cin, int Cout, int with_shortcut,
                                             int* nslabs_host) {
  if (!wg2_geometry_ok(B, H, W, Cin, Cout)) return MMSA_ERR_ARG;
  const Wg2Plan p = wg2_plan(B, H, W, Cin, Cout, with_shortcut != 0);
  if (!p.R) return MMSA_ERR_LIMIT;
  if (nslabs_host) *nslabs_host = p.nsplit;
  return (long)p.nsplit * 9 * Cin * Cout;
}

// Partial sums of dW9 (3x3; dh / dw must be the row-major offsets of a 3 x 3 kernel: tap kh * 3 + kw at (kh - 1, kw - 1))
// from dY and, when dY_sc is given, of the
// 1x1 shortcut's dW1 from dY_sc (both convolutions read X), one slab per workgroup row: slabs [nslabs][9][Cin][Cout],
// slabs_sc [nslabs][Cin][Cout].  mmsa_conv_wgrad_reduce_multi turns slabs into gradients.
MMSA_EXPORT int mmsa_conv_wgrad_slabs(const float* X, const float* dY, const float* dY_sc, float* slabs, float* slabs_sc,
                                      int B, int H, int W, int Cin, int Cout, const int* dh, const int* dw,
                                      void* stream) {
  MMSA_REQUIRE(X && dY && slabs && dh && dw && (!dY_sc || slabs_sc));
  MMSA_REQUIRE(wg2_geometry_ok(B, H, W, Cin, Cout));
  Wg2Geom g;
  if (wg2_fill(g, B, H, W, Cin, Cout, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  const bool sc = dY_sc != nullptr;
  const Wg2Plan p = wg2_plan(B, H, W, Cin, Cout, sc);
  if (!p.R) return MMSA_ERR_LIMIT;
  dim3 grid((Cin / 32) * (Cout / 64), p.nsplit);
  hipStream_t s = mmsa_stream(stream);
  auto launch = [&](auto kernel) -> int {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)p.shmem) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    hipLaunchKernelGGL(kernel, grid, dim3(512), p.shmem, s, X, dY, sc ? dY_sc : dY, slabs, sc ? slabs_sc : slabs, g, p.R,
                       p.nstrips, p.strips_per_img, p.xpieces, p.ypieces);
    return MMSA_OK;
  };
  int rc;
  if (sc) {
    rc = W == 16 ? launch(conv_wgrad2_kernel<true, 16>) : W == 32 ? launch(conv_wgrad2_kernel<true, 32>)
       : W == 65 ? launch(conv_wgrad2_kernel<true, 65>) : launch(conv_wgrad2_kernel<true, 0>);
  } else {
    rc = W == 16 ? launch(conv_wgrad2_kernel<false, 16>) : W == 32 ? launch(conv_wgrad2_kernel<false, 32>)
       : W == 65 ? launch(conv_wgrad2_kernel<false, 65>) : launch(conv_wgrad2_kernel<false, 0>);
  }
  if (rc != MMSA_OK) return rc;
  return mmsa_launch_status();
}

MMSA_EXPORT int mmsa_conv_wgrad_reduce_desc_bytes() { return (int)sizeof(ReduceDesc); }

// descs: DEVICE array of n records {const float* slabs; float* dW; int nslabs, ntaps, Cin, Cout, block0, nblocks}
// (40 bytes each) with nblocks = ceil(ntaps * Cin * Cout / 256) and block0 = the sum of the earlier records' nblocks;
// total_blocks = the sum over all records.  dW[Cout][Cin][ntaps] is overwritten.
MMSA_EXPORT int mmsa_conv_wgrad_reduce_multi(const void* descs, int n, int total_blocks, void* stream) {
  MMSA_REQUIRE(descs && n > 0 && n <= 64 && total_blocks > 0);
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(total_blocks), dim3(256), 0, mmsa_stream(stream),
                     static_cast<const ReduceDesc*>(descs), n);
  return mmsa_launch_status();
}

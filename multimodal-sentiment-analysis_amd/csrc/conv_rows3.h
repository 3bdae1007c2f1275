// Implicit GEMM for the 3x3 / pad 1 convolutions (and their data gradients) that stages every input strip ONCE per
// row of three taps.  Included by conv.hip inside its anonymous namespace (uses ConvGeom, KC, MAX_TAPS).
//
// Why: on this chip every global -> LDS request instruction (1 KiB per wave) holds its SIMD's issue for about 110
// cycles, matrix instructions of the other resident wave included, so the number of requests per MFMA bounds the
// kernel.  The general kernel (conv_igemm_kernel) fetches a fresh [BM rows][32 channels] input tile for each of the
// nine taps.  The three taps of one kernel row (kh fixed, kw = -1, 0, +1) read the SAME run of flattened positions
// shifted by one: position p of tap kw is X[p + dh*W + kw], and a run of BM + 2 positions holds all three.  Here the
// K loop therefore walks  channel chunk -> tap row ("group") -> tap:  a group's strip of BM + 2 positions x 32
// channels is requested once (into a ring of NA strips), each of its taps then only needs its weight tile (ring of
// NB tiles, one per step), and the MFMA waves read the input fragments of tap kw from strip rows r + 1 + kw.
// Requests per 3 taps and 64 x 64 tile: 9 + 24 instead of 48; for the 128 x 32 tile 17 + 12 instead of 60.
//
// Zero padding: a strip row whose flattened position falls outside the image reads as zero (buffer range check), which
// covers the rows above / below the image once the column is valid.  A shifted read that would wrap into the
// neighbouring image row (w + kw outside [0, W)) is pointed at a strip row that always holds zeros instead (row BM + 2):
// the redirection is part of the per-lane fragment addresses computed once per workgroup, so the K loop has no
// masking and no per-tap validity plan, and a workgroup's first request goes out a few hundred cycles after it starts.
//
// Step sequences (uniform per workgroup):  3x3: per chunk three groups of three taps [+ one single-tap group reading
// the second operand pair X2 / Wp2];  1x1 (centre tap of the offset table, a problem of a multi-problem launch): per
// chunk one single-tap group.  A problem's K loop may cover a sub-range of the channel chunks (a K range split over
// two workgroups per tile).
//
// Pipeline: loader step s requests the weight tile of step s + 4 and, when a group begins at step s + 4, that group's
// strip ahead of it (single-tap-only sequence: the strip of step s + 3; its strips come one per step, and a request
// overwrites the strip NA groups back, so the NA - 1 groups between must span the look-ahead: NA = 4 strips for 64-row
// tiles, 3 for 128-row tiles, which never run that sequence).  It then waits until everything step s + 2 reads has
// landed -- requests complete in order, so that is a bound on the outstanding count, known statically for each
// position of the step sequence -- and joins the barrier.  The three step sequences are written out as three loops
// without cursors or branches inside: the loader wave's scalar bookkeeping sits on the critical path once the
// requests themselves are few.  The MFMA waves mirror conv_igemm_kernel: fragments of half a chunk are read behind
// the first MFMA of the previous half.

struct Conv3Args {
  int n;                       // problems of the launch (grid.z), at least 1
  int ntaps[4];                // 9: rows of three taps;  1: the centre tap alone
  int chunk0[4], chunk1[4];    // 32-channel chunks [chunk0, chunk1) a problem's K loop covers
  const float* Wp[4];          // packed weights [ntaps][Cout][Cin]
  float* Y[4];
  float* stats[4];             // STATS instantiations
  const float* bsx[4];         // STATS == 2: the map x [B,H,W,Cout] whose BatchNorm + ReLU the result is the gradient of,
  const float* bscoeff[4];     // ... and that BatchNorm's coefficient table [4][Cout] (mean, 1/std, scale, shift)
  const float* bnrelu[4];      // BNRELU instantiations: [4][Cout] table or null
  const float* X2;             // optional second operand pair (see conv_igemm_kernel)
  const float* Wp2;
  int second_z;                // problem whose K loop takes the second pair (-1: every 9-tap problem)
};

template <int N>
__device__ __forceinline__ void wait_vm_at_most() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// STATS: 0 none; 1 per-tile column sums and sums of squares of the result (BatchNorm batch statistics of a forward
// convolution); 2 the result dY is the gradient with respect to relu(bn(x)): per-tile column sums of g = dY [bn(x) > 0]
// and of g * xhat (the two reductions of the BatchNorm backward), so that the data-gradient convolution leaves them
// behind and no separate pass over dY and x is needed for them.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STATS, bool ACCUM, bool BNRELU>
__global__ __launch_bounds__(512) void conv_igemm3_kernel(const float* __restrict__ X, const ConvGeom g,
                                                          int tiles_per_img, const Conv3Args a) {
  static_assert(WAVES_M * WAVES_N == 4, "4 MFMA waves per workgroup (plus 4 loader waves)");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold whole 32x32 MFMA tiles");
  constexpr int SROWS = BM + 8;            // strip rows in LDS (BM + 2 used; whole 8-row request pieces)
  constexpr int NPIECE = SROWS / 8;        // request instructions per strip
  constexpr int AP0 = (NPIECE + 3) / 4;    // ... of which wave 0 issues AP0, the others AP0 or AP0 - 1
  constexpr int B_LD = BN / 32;            // request instructions per wave for one weight tile
  constexpr int STRIP = SROWS * KC, BBUF = BN * KC;
  constexpr int NA = BM == 64 ? 4 : 3, NB = 5, D = NB - 1;
  constexpr int OOB = 0x7ffffff0;

  extern __shared__ __attribute__((aligned(16))) float lds[];   // [NA][SROWS][32] strips, then [NB][BN][32] weight tiles
  LAB_STAMP(0);
  const int z = blockIdx.z;
  const bool p1 = a.ntaps[z] == 1;                                           // uniform
  const bool dual = !p1 && a.X2 != nullptr && (a.second_z < 0 || z == a.second_z);
  const float* __restrict__ Wp = a.Wp[z];
  float* __restrict__ Y = a.Y[z];
  const int chunk0 = a.chunk0[z];
  const int ngrp = p1 ? 1 : (dual ? 4 : 3);
  const int nsteps = (p1 ? 1 : (dual ? 10 : 9)) * (a.chunk1[z] - chunk0);
  // offsets of the three tap rows and the column step of the first tap of a row (constant indices: scalar loads)
  const int dh0 = g.dh[0], dh1 = g.dh[3], dh2 = g.dh[6], dwf = g.dw[0];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int HW = g.H * g.W;
  const int b = blockIdx.x / tiles_per_img;
  const int tile = blockIdx.x - b * tiles_per_img;
  const int p0 = tile * BM;
  const int n0 = blockIdx.y * BN;
  typedef __attribute__((address_space(3))) void lds_void;
  // Strip layout: row u holds flattened position p0 - 1 + u of the tap row's input line for u < BM + 2; the rows
  // behind them are always requested out of range and therefore hold zeros.
  const int f0 = p0 - 1;

  auto group_taps = [&](int k) { return (p1 || k == 3) ? 1 : 3; };   // uniform

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bx[TM][TN][16];   // STATS == 2: this lane's elements of x, requested before the K loop
  if (loader) {
    const float* Xb = X + (long)b * HW * g.Cin;
    const __amdgpu_buffer_rsrc_t rX =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0, HW * g.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wp), 0, (p1 ? 1 : 9) * g.Cout * g.Cin * 4, 0x00020000);
    const float* X2b = dual ? a.X2 + (long)b * HW * g.Cin : Xb;
    const __amdgpu_buffer_rsrc_t rX2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X2b), 0, HW * g.Cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dual ? a.Wp2 : Wp), 0, g.Cout * g.Cin * 4, 0x00020000);
    // Per-lane source plan.  Strip piece P (8 rows, P = wave, wave + 4, ...): lane l supplies strip row u = 8 P + (l >> 3),
    // 16-byte slot l & 7 holding piece (l & 7) ^ ((u >> 1) & 7) of that row's position in the tap row's input line.  A
    // position before the image gives a negative (= huge unsigned) offset, one past its end an offset beyond the
    // buffer: both read as zero, and so does the fixed negative offset of the zero rows behind the strip.
    const int lrow = lane >> 3, lslot = lane & 7;
    int aoff[AP0], boff[B_LD];
#pragma unroll
    for (int j = 0; j < AP0; ++j) {
      const int u = 8 * (wave + 4 * j) + lrow;
      aoff[j] = u < BM + 2 ? ((f0 + u) * g.Cin + 4 * (lslot ^ ((u >> 1) & 7))) * 4 : -(1 << 30);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int r = 8 * (wave + 4 * j) + lrow;
      boff[j] = ((n0 + r) * g.Cin + 4 * (lslot ^ ((r >> 1) & 7))) * 4;
    }
    const int row_bytes = g.W * g.Cin * 4;
    const int nchunks = a.chunk1[z] - chunk0;

    // Request helpers.  Uniform values are pinned to scalar registers (readfirstlane): where the compiler's divergence
    // analysis loses track of one, it wraps every request in a loop over the distinct values of its scalar offset.
    int slot_w = 0, buf_w = 0;   // ring positions the next strip / weight tile is written to
    // strip of tap row k (0..2; 3: the second operand pair) of channel chunk c (relative to chunk0); !real: past the end of
    // the K loop, the instructions still issue (the waits below count them) but fetch nothing
    auto strip = [&](int c, int k, bool real) {
      const bool second = k == 3;
      const int dh = p1 ? 0 : (k == 0 ? dh0 : (k == 1 ? dh1 : (k == 2 ? dh2 : 0)));
      const int dshift = __builtin_amdgcn_readfirstlane(dh * row_bytes);
      const int soff = __builtin_amdgcn_readfirstlane((chunk0 + c) * (KC * 4));
      float* base = lds + __builtin_amdgcn_readfirstlane(slot_w * STRIP + wave * 256);
#pragma unroll
      for (int j = 0; j < AP0; ++j) {
        if (wave + 4 * j < NPIECE) {   // uniform
          const int va = real ? aoff[j] + dshift : OOB;
          if (second)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, (lds_void*)(base + j * 1024), 16, va, soff, 0, 0);
          else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(base + j * 1024), 16, va, soff, 0, 0);
        }
      }
      slot_w = slot_w + 1 == NA ? 0 : slot_w + 1;
    };
    // weight tile of tap `tap` (0..8 of the problem's packed weights; -1: the second pair's) for channel chunk c
    auto wtile = [&](int c, int tap, bool real) {
      const bool second = tap < 0;
      const int sb = __builtin_amdgcn_readfirstlane(
          second ? (chunk0 + c) * (KC * 4) : ((tap * g.Cout * g.Cin + (chunk0 + c) * KC) * 4));
      float* base = lds + __builtin_amdgcn_readfirstlane(NA * STRIP + buf_w * BBUF + wave * 256);
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const int vb = real ? boff[j] : OOB;
        if (second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (lds_void*)(base + j * 1024), 16, vb, sb, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(base + j * 1024), 16, vb, sb, 0, 0);
      }
      buf_w = buf_w + 1 == NB ? 0 : buf_w + 1;
    };
    // End of loader step s: everything step s + 2 reads must have landed.  Requests complete in issue order, so that is
    // "at most N requests outstanding", N = this wave's requests issued after the last one step s + 2 needs: the weight
    // tiles of steps s + 3 and s + 4 plus NSTRIPS strips (those of groups that begin at s + 3 or s + 4; wave 0 issues
    // AP0 pieces per strip, the others AP0 - 1 when the strip's piece count is 1 modulo 4).  Then the barrier.
    const bool full = wave + 4 * (AP0 - 1) < NPIECE;   // this wave issues AP0 pieces per strip (uniform)
    auto publish = [&](auto nstrips_c) {
      constexpr int NS = decltype(nstrips_c)::value;
      if (full)
        wait_vm_at_most<2 * B_LD + NS * AP0>();
      else
        wait_vm_at_most<2 * B_LD + NS * (AP0 - 1)>();
      __builtin_amdgcn_s_barrier();
    };
    const std::integral_constant<int, 0> S0;
    const std::integral_constant<int, 1> S1;
    const std::integral_constant<int, 2> S2;
    LAB_STAMP_L(1);
    if (p1) {
      // single-tap groups: a strip per step, requested three steps ahead (the ring of NA = 4 strips allows no more)
      wtile(0, 0, true);
      strip(0, 0, true);
      wtile(1, 0, 1 < nchunks);
      strip(1, 0, 1 < nchunks);
      wtile(2, 0, 2 < nchunks);
      strip(2, 0, 2 < nchunks);
      wtile(3, 0, 3 < nchunks);
      publish(S1);
#pragma unroll 1
      for (int s = 0; s < nchunks; ++s) {
        strip(s + 3, 0, s + 3 < nchunks);
        wtile(s + 4, 0, s + 4 < nchunks);
        publish(S1);
      }
    } else if (!dual) {
      // groups of three taps, G = 3 per chunk; group index -> (chunk, tap row) kept for the groups one and two ahead
      const int G = 3 * nchunks;
      int c1 = 0, k1 = 1, c2 = 0, k2 = 2;   // groups 1 and 2 (every chunk has three groups)
      strip(0, 0, true);
      wtile(0, 0, true);
      wtile(0, 1, true);
      wtile(0, 2, true);
      strip(c1, k1, true);
      wtile(c1, 3 * k1, true);
      publish(S1);
#pragma unroll 1
      for (int grp = 0; grp < G; ++grp) {
        const bool r1 = grp + 1 < G, r2 = grp + 2 < G;
        wtile(c1, 3 * k1 + 1, r1);
        publish(S1);
        wtile(c1, 3 * k1 + 2, r1);
        publish(S0);
        strip(c2, k2, r2);
        wtile(c2, 3 * k2, r2);
        publish(S1);
        c1 = c2;
        k1 = k2;
        if (++k2 == 3) {
          k2 = 0;
          ++c2;
        }
      }
    } else {
      // per chunk: tap rows at steps 0, 3, 6 and the second pair's single tap at step 9
      strip(0, 0, true);
      wtile(0, 0, true);
      wtile(0, 1, true);
      wtile(0, 2, true);
      strip(0, 1, true);
      wtile(0, 3, true);
      publish(S1);
#pragma unroll 1
      for (int c = 0; c < nchunks; ++c) {
        const bool rn = c + 1 < nchunks;
        wtile(c, 4, true);
        publish(S1);
        wtile(c, 5, true);
        publish(S0);
        strip(c, 2, true);
        wtile(c, 6, true);
        publish(S1);
        wtile(c, 7, true);
        publish(S1);
        wtile(c, 8, true);
        publish(S0);
        strip(c, 3, true);
        wtile(c, -1, true);
        publish(S1);
        strip(c + 1, 0, rn);
        wtile(c + 1, 0, rn);
        publish(S2);
        wtile(c + 1, 1, rn);
        publish(S1);
        wtile(c + 1, 2, rn);
        publish(S0);
        strip(c + 1, 1, rn);
        wtile(c + 1, 3, rn);
        publish(S1);
      }
    }
    // the trailing requests may still be writing LDS, which the statistics epilogue reuses
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    const int li = lane & 31, lh = lane >> 5;
    // Fragment addresses.  Tap position J of a row group (J = 0, 1, 2: column offsets dwf, 0, -dwf; single-tap groups
    // read like J = 1) reads strip row  u = strip row of the lane's own position + offset;  pre[J][i][g4] is the float
    // offset of 16-byte piece 2 g4 + lh of that row inside a strip (XOR swizzle of the row applied).
    int pre[3][TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * WTM + i * 32 + li;
      const int wcol = (p0 + r) % g.W;
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        const int dw = dwf * (1 - J);
        const bool wraps = (dw < 0 && wcol == 0) || (dw > 0 && wcol == g.W - 1);
        const int u = wraps ? BM + 2 : r + 1 + dw;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) pre[J][i][g4] = u * KC + 4 * ((2 * g4 + lh) ^ ((u >> 1) & 7));
      }
    }
    const int swzb = (li >> 1) & 7;
    const int b_row = NA * STRIP + (wn * WTN + li) * KC;
    int pieceB[4];
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) pieceB[gk] = 4 * ((2 * gk + lh) ^ swzb);
    struct Frag {
      f32x4 a[2][TM], b[2][TN];
    };
    auto read_half = [&](Frag& f, int slot, const int (&pj)[TM][4], int bbuf, int half) {
      const float* As = lds + slot * STRIP;
      const float* Bb = lds + bbuf * BBUF + b_row;
#pragma unroll
      for (int gk = 0; gk < 2; ++gk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[gk][i] = *reinterpret_cast<const f32x4*>(As + pj[i][2 * half + gk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[gk][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * KC + pieceB[2 * half + gk]);
      }
    };
    auto mfma_steps = [&](const Frag& f, int s_begin, int s_end) {
#pragma unroll
      for (int s = s_begin; s < s_end; ++s) {
        const int gk = s >> 2, e = s & 3;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[gk][i][e], f.b[gk][j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    if (STATS == 2) {
      const float* xb = a.bsx[z] + (long)b * HW * g.Cout;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int p = p0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            bx[i][j][r] = p < HW ? xb[(long)p * g.Cout + n0 + wn * WTN + j * 32 + li] : 0.f;
          }
    }
    Frag f0, f1;
    int slot = 0, bbuf = 0;
    // One step: the 16 MFMAs of tap position J on strip `slot` and weight tile `bbuf`; behind the first MFMA of each
    // half go the LDS reads of the next half -- the second half of this step, then the first half of the next step
    // (tap position JN on strip `nslot`).
#ifndef MMSA_LAB_SKIP_TAPS
#define MMSA_LAB_SKIP_TAPS 0   // diagnostic builds (tools/igemm_lab.hip): 1 leaves out the matrix instructions of two of
#endif                         // every three taps while every load, fragment read and barrier stays -- what a kernel
                               // with a third of the multiplications could gain at best
    auto step = [&](auto Jc, auto JNc, int nslot) {
      constexpr int J = decltype(Jc)::value, JN = decltype(JNc)::value;
      constexpr bool LIVE = MMSA_LAB_SKIP_TAPS == 0 || (MMSA_LAB_SKIP_TAPS == 1 && J == 0);   // 2: no tap at all
      const int nbuf = bbuf + 1 == NB ? 0 : bbuf + 1;
      if (LIVE) mfma_steps(f0, 0, 1);
      read_half(f1, slot, pre[J], bbuf, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (LIVE) mfma_steps(f0, 1, 8);
      if (LIVE) mfma_steps(f1, 0, 1);
      read_half(f0, nslot, pre[JN], nbuf, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (LIVE) mfma_steps(f1, 1, 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      bbuf = nbuf;
      slot = nslot;
    };
    using std::integral_constant;
    const integral_constant<int, 0> J0;
    const integral_constant<int, 1> J1;
    const integral_constant<int, 2> J2;
    auto next_slot = [&]() { return slot + 1 == NA ? 0 : slot + 1; };

    __builtin_amdgcn_s_barrier();
    const int nchunks = a.chunk1[z] - chunk0;
    if (p1) {
      read_half(f0, 0, pre[1], 0, 0);
      LAB_STAMP(1);
#pragma unroll 1
      for (int c = 0; c < nchunks; ++c) step(J1, J1, next_slot());
    } else {
      read_half(f0, 0, pre[0], 0, 0);
      LAB_STAMP(1);
      // One loop per step sequence, each without a branch inside its body: where paths through the loop body merge,
      // the compiler parks the accumulator in another register tuple and copies all sixteen registers back (behind a
      // wait for the matrix pipe to drain) on every trip.
      if (!dual) {
#pragma unroll 1
        for (int grp = 0; grp < 3 * nchunks; ++grp) {
          step(J0, J1, slot);
          step(J1, J2, slot);
          step(J2, J0, next_slot());
        }
      } else {
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
          step(J0, J1, slot);
          step(J1, J2, slot);
          step(J2, J0, next_slot());
          step(J0, J1, slot);
          step(J1, J2, slot);
          step(J2, J0, next_slot());
          step(J0, J1, slot);
          step(J1, J2, slot);
          step(J2, J1, next_slot());   // the second pair's single tap follows
          step(J1, J0, next_slot());
        }
      }
    }
    __builtin_amdgcn_s_barrier();
  }

  LAB_STAMP(2);
  // epilogue (as conv_igemm_kernel): D[row][col]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int li = lane & 31, lh = lane >> 5;
  float* Yb = Y + (long)b * HW * g.Cout;
  const float* __restrict__ ep = BNRELU ? a.bnrelu[z] : nullptr;   // uniform
  float csum[TN], csq[TN], esc[TN], esh[TN], bmean[TN], bistd[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    csum[j] = csq[j] = 0.f;
    esc[j] = 1.f;
    esh[j] = 0.f;
    bmean[j] = bistd[j] = 0.f;
    const int col = n0 + wn * WTN + j * 32 + li;
    if (BNRELU && ep) {
      esc[j] = ep[2 * g.Cout + col];
      esh[j] = ep[3 * g.Cout + col];
    }
    if (STATS == 2) {
      const float* bc = a.bscoeff[z];
      bmean[j] = bc[col];
      bistd[j] = bc[g.Cout + col];
      esc[j] = bc[2 * g.Cout + col];
      esh[j] = bc[3 * g.Cout + col];
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int p = p0 + row;
        if (p < HW && !loader) {
          const float raw = acc[i][j][r];
          const float v = BNRELU && ep ? fmaxf(fmaf(raw, esc[j], esh[j]), 0.f) : raw;
          float* dst = Yb + (long)p * g.Cout + col;
          if (ACCUM) {
            unsafeAtomicAdd(dst, v);   // two writers at most onto a zeroed map: order-independent
          } else {
            *dst = v;
          }
          if (STATS == 1) {
            csum[j] += raw;
            csq[j] = fmaf(raw, raw, csq[j]);
          }
          if (STATS == 2) {   // the expressions of bn_relu_bwd_stats_kernel
            const float x = bx[i][j][r];
            const float gr = fmaf(x, esc[j], esh[j]) > 0.f ? raw : 0.f;
            csum[j] += gr;
            csq[j] = fmaf(gr, (x - bmean[j]) * bistd[j], csq[j]);
          }
        }
      }
    }
  if (STATS) {
    float* red = lds;   // safe: every wave passed the loop's final barrier
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float lo, hi;   // the two lane halves' values, the same pair in every lane
      swap32_pair(csum[j], lo, hi);
      csum[j] = lo + hi;
      swap32_pair(csq[j], lo, hi);
      csq[j] = lo + hi;
      if (lh == 0 && !loader) {
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
      float* st = a.stats[z];
      st[((long)which * g.Cout + n0 + col) * gridDim.x + blockIdx.x] = s;
    }
  }
  LAB_STAMP(3);
}

// Whether an offset table is the 3x3 pattern this kernel walks: taps in three rows of three, one row offset per row,
// column offsets (c, 0, -c) with c = +-1 in every row.
static bool rows3_pattern(int ntaps, const int* dh, const int* dw) {
  if (ntaps != 9) return false;
  const int c = dw[0];
  if (c != 1 && c != -1) return false;
  for (int k = 0; k < 3; ++k) {
    if (dh[3 * k] < -1 || dh[3 * k] > 1) return false;
    for (int j = 0; j < 3; ++j)
      if (dh[3 * k + j] != dh[3 * k] || dw[3 * k + j] != c * (1 - j)) return false;
  }
  return true;
}

// Which launches take this kernel.  Measured at batch 32 against the general kernel (tools/lab_rows3.sh, kernel alone):
// forward 32->64 36.4 -> 34.5 us, 64->64 61.8 -> 56.7, 64->128 29.1 -> 26.0, 128->128 49.7 -> 44.9 (108 TFLOP/s),
// 128->256 31.6 -> 26.1, 256->256 55.7 -> 46.0 (105 TFLOP/s); data gradient 64->32 (128 x 32 tile) 48.5 -> 35.6,
// 256->128 53.1 -> 42.4.  (The first versions of this kernel were SLOWER than the general one although they issued a
// third fewer requests: a loader wave that walks the step sequence with run-time cursors executes ~150 scalar
// instructions and a dozen branches per step, which is as long as the MFMA waves' step.)
// MMSA_IGEMM3 = 0: never; 1: 32-wide results only; 2 (default): every eligible launch.  Read at every launch, so a test
// can switch within one process.
static int rows3_mode() {
  const char* e = getenv("MMSA_IGEMM3");
  const int mode = e ? atoi(e) : 2;
  return mode < 0 || mode > 2 ? 2 : mode;
}

template <int BM, int BN, int WM_, int WN_, int STATS, bool ACCUM, bool BNRELU>
int launch_igemm3_inst(const float* X, const ConvGeom& g, const Conv3Args& a, hipStream_t s) {
  constexpr size_t shmem = ((BM == 64 ? 4 : 3) * (size_t)(BM + 8) + 5 * (size_t)BN) * KC * sizeof(float);
  static PerDeviceOnce attr_set;
  if (shmem > 48 * 1024 && attr_set.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm3_kernel<BM, BN, WM_, WN_, STATS, ACCUM, BNRELU>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    attr_set.done();
  }
  const int tiles = mmsa_cdiv((long)g.H * g.W, BM);
  dim3 grid(g.B * tiles, g.Cout / BN, a.n);
  hipLaunchKernelGGL((conv_igemm3_kernel<BM, BN, WM_, WN_, STATS, ACCUM, BNRELU>), grid, dim3(512), shmem, s, X, g, tiles,
                     a);
  return mmsa_launch_status();
}

// Picks the instantiation: statistics epilogue (every problem has a statistics buffer), atomic accumulation, folded
// BatchNorm + ReLU (any problem has a table).  Cout % 64 != 0 selects the 128 x 32 tile (no statistics there).
static int launch_igemm3(const float* X, const ConvGeom& g, const Conv3Args& a, bool accumulate, hipStream_t s) {
  if ((long)g.H * g.W * g.Cin * 4 >= (1L << 30)) return MMSA_ERR_LIMIT;   // byte offsets within an image are 32-bit
  const bool stats = a.stats[0] != nullptr;
  bool bnrelu = false;
  for (int i = 0; i < a.n; ++i) bnrelu = bnrelu || a.bnrelu[i];
  const bool bwd_stats = a.bsx[0] != nullptr;
  if (bwd_stats) {
    if (!stats || bnrelu || accumulate || g.Cout % 64 != 0 || a.n != 1 || !a.bscoeff[0]) return MMSA_ERR_ARG;
    return launch_igemm3_inst<64, 64, 2, 2, 2, false, false>(X, g, a, s);
  }
  if (g.Cout % 64 != 0) {
    if (stats || bnrelu) return MMSA_ERR_ARG;
    for (int i = 0; i < a.n; ++i)
      if (a.ntaps[i] != 9) return MMSA_ERR_ARG;   // the 128-row tile's ring of three strips cannot run single-tap groups
    if (accumulate) return launch_igemm3_inst<128, 32, 4, 1, 0, true, false>(X, g, a, s);
    return launch_igemm3_inst<128, 32, 4, 1, 0, false, false>(X, g, a, s);
  }
  if (bnrelu) {
    if (stats || accumulate) return MMSA_ERR_ARG;
    return launch_igemm3_inst<64, 64, 2, 2, 0, false, true>(X, g, a, s);
  }
  if (stats) {
    if (accumulate) return MMSA_ERR_ARG;
    return launch_igemm3_inst<64, 64, 2, 2, 1, false, false>(X, g, a, s);
  }
  if (accumulate) return launch_igemm3_inst<64, 64, 2, 2, 0, true, false>(X, g, a, s);
  return launch_igemm3_inst<64, 64, 2, 2, 0, false, false>(X, g, a, s);
}

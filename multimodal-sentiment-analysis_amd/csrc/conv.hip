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
// chunks of 32 input channels of one tap.  A workgroup is eight waves with two roles: waves 4-7 only issue
// loads, waves 0-3 only read LDS and issue MFMAs.  The loader waves bring both operand tiles from global memory
// straight into LDS (buffer loads with an LDS destination, 1 KiB per wave instruction; positions outside the
// image, channel tails and the padding halo use an out-of-range buffer offset, which the hardware answers with
// zeros), several chunks ahead of the consumers in a ring of IGEMM_NBUF buffers; one barrier per chunk hands a
// filled buffer over and takes a drained one back.  LDS rows are unpadded 128-byte runs of 32 channels; the
// 16-byte piece a lane needs is stored at slot (piece ^ ((row >> 1) & 7)), applied on the source address because
// the DMA's destination is lane-linear, which keeps the consumers' ds_read_b128 fragment reads conflict free.
// Each lane reads 4 consecutive k per ds_read_b128 and feeds them to 4 successive MFMAs; A and B use the same k
// permutation, so the contraction is unchanged.  Keeping vector-memory instructions out of the MFMA waves matters
// on this chip: a 1 KiB load costs the issuing SIMD about as long as two MFMAs, and at least two workgroups
// per CU are needed to cover it.
// Optional second operand pair (conv_fwd_dual): one more tap appended to the K loop that reads another tensor
// with weights of its own -- the shortcut's data gradient accumulated in the same launch as the 3x3 one.
// Optional epilogue: per-tile column sums and sums of squares (the BatchNorm batch statistics and the
// squeeze-excite global average come from these, so the raw output is not re-read), stored [2][Cout][tiles].
//
// Weight-gradient kernel (conv_wgrad): D[cin][cout] += X[p+off][cin] * dY[p][cout] summed over positions.
// A workgroup owns a (cout tile, cin tile) pair and a slab of image rows; strips of rows of X (with a zero halo)
// and dY are staged in LDS in batches of independent 16-byte loads, operands are conflict-free ds_read_b32, the
// four waves are (cout half) x (tap group) with accumulators living across strips, and the tap groups swap
// between workgroups that share a CU so their LDS phases interleave.  Workgroups write partial slabs that a
// second kernel sums (fixed order: results are run-to-run identical) into the reference's
// [Cout][Cin][kh][kw] layout.
#include "mmsa_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int MAX_TAPS = 9;
constexpr int KC = 32;        // channels per K chunk
#ifndef IGEMM_NBUF_VALUE
#define IGEMM_NBUF_VALUE 5
#endif
constexpr int IGEMM_NBUF = IGEMM_NBUF_VALUE;   // LDS buffers of the forward / data-gradient kernel (chunks in flight + 2)
// The 128 x 32 tile (data gradient into a 32-channel map) takes 20 KiB per buffer: a ring of three lets two workgroups
// share a CU, which is worth more there than the deeper prefetch (59 -> 55 us on the 32 x 32 x 65 x 64 layer).
constexpr int igemm_nbuf(int BM, int BN) { return BM == 128 && BN == 32 && IGEMM_NBUF > 3 ? 3 : IGEMM_NBUF; }

struct ConvGeom {
  int B, H, W, Cin, Cout, ntaps;
  int dh[MAX_TAPS + 1], dw[MAX_TAPS + 1];   // entries past ntaps are zero (the optional second operand's centre tap)
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

// The same re-layout for a whole table of weight tensors in one launch (blockIdx.y = tensor).
struct PackDesc {
  const float* w;
  float* fwd;
  float* dgrad;
  int Cout, Cin, ntaps, pad_;
};

// Workgroup = a 32 (cout) x 32 (cin) x ntaps tile of one tensor, staged in LDS: the reads are whole runs of 32 * ntaps
// consecutive floats per cout row, the writes runs of 32 consecutive cin (forward layout) and 32 consecutive cout
// (data-gradient layout).  blockIdx.x counts the tensor's tiles; workgroups past its tile count leave at once.
constexpr int PACK_T = 32;
constexpr int PACK_MAX_TILES = 64;   // per launch row: tensors with more tiles are walked with a stride

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackDesc* __restrict__ descs) {
  __shared__ float tile[PACK_T][PACK_T * MAX_TAPS + 1];
  const PackDesc d = descs[blockIdx.y];
  const int nt = d.ntaps;
  const int cin_tiles = (d.Cin + PACK_T - 1) / PACK_T, cout_tiles = (d.Cout + PACK_T - 1) / PACK_T;
  const int t = threadIdx.x;
  for (int tl = blockIdx.x; tl < cin_tiles * cout_tiles; tl += gridDim.x) {   // uniform
    const int co0 = (tl / cin_tiles) * PACK_T, ci0 = (tl % cin_tiles) * PACK_T;
    const int nci = min(PACK_T, d.Cin - ci0), nco = min(PACK_T, d.Cout - co0);
    const int run = nci * nt;                       // consecutive source floats per cout row
    for (int r = t >> 6; r < nco; r += 4) {         // one wave per row at a time
      const float* src = d.w + ((long)(co0 + r) * d.Cin + ci0) * nt;
      for (int j = t & 63; j < run; j += 64) tile[r][j] = src[j];
    }
    __syncthreads();
    const int c = t & 31;
    for (int tp = 0; tp < nt; ++tp) {
      if (d.fwd && c < nci)
        for (int r = t >> 5; r < nco; r += 8)       // 32 lanes = 32 consecutive cin of one (tap, cout) row
          d.fwd[((long)tp * d.Cout + co0 + r) * d.Cin + ci0 + c] = tile[r][c * nt + tp];
      if (d.dgrad && c < nco)
        for (int ci = t >> 5; ci < nci; ci += 8)    // 32 lanes = 32 consecutive cout of one (tap, cin) row
          d.dgrad[((long)tp * d.Cin + ci0 + ci) * d.Cout + co0 + c] = tile[c][ci * nt + tp];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Several convolutions of the same input in one launch (blockIdx.z = problem): per-problem packed weights, output
// and tap count (a prefix of the shared offset table).  Used for the text branch's three filter widths.
struct ConvMulti {
  int n;
  int ntaps[4];
  int tap0[4];        // problem i uses entries tap0[i] .. tap0[i] + ntaps[i] - 1 of the shared offset table
  const float* Wp[4];
  float* Y[4];
  float* stats[4];    // per-problem statistics (STATS instantiations)
  const float* bnrelu[4];   // per-problem [4][Cout] BatchNorm table (rows 2, 3 = scale, shift) applied with a ReLU to
                            // the result before it is stored, or null: inference, where the table is known beforehand
  // optional second operand pair appended to the K loop: Y += X2[b, p, :] . Wp2[:, :] (a 1x1 convolution of another
  // tensor with the same channel count, e.g. the shortcut's data gradient next to the 3x3 one)
  const float* X2;
  const float* Wp2;
  int second_z;       // with n > 0: the one problem whose K loop takes the second pair (-1: every problem)
};

#ifdef MMSA_LAB_STAMPS
// Diagnostic builds only (tools/igemm_lab.hip): per-workgroup clock stamps, written to a buffer of their own.
__device__ unsigned long long* g_lab_stamps;
#define LAB_STAMP(i)                                                                                   \
  do {                                                                                                 \
    if (threadIdx.x == 0 && g_lab_stamps) {                                                            \
      const long wg_ = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;           \
      g_lab_stamps[wg_ * 16 + (i)] = __builtin_amdgcn_s_memtime();                                     \
      g_lab_stamps[wg_ * 16 + 4 + (i)] = __builtin_amdgcn_s_memrealtime();                             \
    }                                                                                                  \
  } while (0)
// loader-side stamps (first lane of the first loader wave), slots 8 ..
#define LAB_STAMP_L(i)                                                                                 \
  do {                                                                                                 \
    if (threadIdx.x == 256 && g_lab_stamps) {                                                          \
      const long wg_ = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;           \
      g_lab_stamps[wg_ * 16 + 8 + (i)] = __builtin_amdgcn_s_memtime();                                 \
    }                                                                                                  \
  } while (0)
#else
#define LAB_STAMP(i)
#define LAB_STAMP_L(i)
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, bool STATS, bool ACCUM, int NB, bool BNRELU = false>
__global__ __launch_bounds__(512) void conv_igemm_kernel(const float* __restrict__ X, const float* __restrict__ Wp0,
                                                         float* __restrict__ Y0, float* __restrict__ stats,
                                                         const ConvGeom g, int tiles_per_img, const ConvMulti multi) {
  // kernel arguments are only read (never written: a modified by-value struct would be spilled to scratch)
  const bool many = multi.n > 0;   // uniform
  const int ntaps = many ? multi.ntaps[blockIdx.z] : g.ntaps;
  const int tap0 = many ? multi.tap0[blockIdx.z] : 0;
  const float* __restrict__ Wp = many ? multi.Wp[blockIdx.z] : Wp0;
  float* __restrict__ Y = many ? multi.Y[blockIdx.z] : Y0;
  static_assert(WAVES_M * WAVES_N == 4, "4 MFMA waves per workgroup (plus 4 loader waves)");
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold whole 32x32 MFMA tiles");
  constexpr int A_LD = BM / 32;  // LDS-DMA instructions per wave and chunk for the A tile (8 rows = 1 KiB each)
  constexpr int B_LD = BN / 32;
  constexpr int LD = A_LD + B_LD;
  constexpr int BUF_FLOATS = (BM + BN) * KC;   // one LDS buffer: A rows then B rows of one 32-channel chunk
  constexpr int NBUF = NB;                     // ring of LDS buffers; a chunk is requested NBUF-1 steps ahead

  extern __shared__ __attribute__((aligned(16))) float lds[];
  LAB_STAMP(0);

  const int t = threadIdx.x;
  const int lane = t & 63;
  // Waves 0-3 only read fragments from LDS and issue MFMAs; waves 4-7 only issue the global -> LDS requests.  A
  // request instruction holds its wave for of the order of 100 cycles and the matrix pipe of that SIMD idles
  // meanwhile unless another wave feeds it, so each SIMD hosts one wave of either kind.
  const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int HW = g.H * g.W;
  const int b = blockIdx.x / tiles_per_img;
  const int tile = blockIdx.x - b * tiles_per_img;
  const int p0 = tile * BM;
  const int n0 = blockIdx.y * BN;

  // Both operands go from global memory straight into LDS (buffer loads with the LDS destination: no staging
  // registers, no LDS store instructions).  One wave instruction moves 8 rows x 128 bytes into 1 KiB of LDS in lane
  // order, so the LDS image has unpadded 128-byte rows; bank conflicts of the 16-byte fragment reads are avoided by
  // an XOR swizzle applied on the SOURCE side: slot s of row r holds the 16-byte piece s ^ ((r >> 1) & 7) of that
  // row, and the fragment reads apply the same XOR.  A byte offset past the end of the buffer reads as zero, which
  // is how taps outside the image, and rows past the end of the last tile, become zeros.
  const float* Xb = X + (long)b * HW * g.Cin;
  const __amdgpu_buffer_rsrc_t rX =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0, HW * g.Cin * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp), 0, ntaps * g.Cout * g.Cin * 4, 0x00020000);
  const bool dual = multi.X2 != nullptr && (!many || multi.second_z < 0 || (int)blockIdx.z == multi.second_z);   // uniform
  const float* X2b = dual ? multi.X2 + (long)b * HW * g.Cin : Xb;
  const __amdgpu_buffer_rsrc_t rX2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X2b), 0, HW * g.Cin * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dual ? multi.Wp2 : Wp), 0, g.Cout * g.Cin * 4, 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  typedef __attribute__((address_space(3))) void lds_void;

  // Per-lane source plan: wave w fills row groups w, w+4, ... (8 rows each); lane l supplies row 8*group + (l >> 3),
  // slot l & 7.
  const int lrow = lane >> 3, lslot = lane & 7;
  int aoff[A_LD], boff[B_LD];
  unsigned amask[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int r = 8 * (wave + 4 * j) + lrow;
    const int p = p0 + r;
    const bool ok = p < HW;
    const int h = p / g.W, w = p - h * g.W;
    aoff[j] = (p * g.Cin + 4 * (lslot ^ ((r >> 1) & 7))) * 4;
    // validity of every entry of the offset table, then the problem's range of it: with constant indices the table
    // arrives in a few wide scalar loads ahead of the loop (indexed by tap0 + tp, every iteration waited for a
    // scalar load of its own: 6 000 cycles per workgroup before the first request went out)
    unsigned mall = 0;
#pragma unroll
    for (int ta = 0; ta < MAX_TAPS; ++ta) {
      const int hh = h + g.dh[ta], ww = w + g.dw[ta];   // no short-circuit: a branch per tap would fence the loads
      mall |= (unsigned)(((unsigned)hh < (unsigned)g.H) & ((unsigned)ww < (unsigned)g.W)) << ta;
    }
    unsigned m = (mall >> tap0) & ((1u << ntaps) - 1u);
    if (ok) m |= 1u << ntaps;   // the second operand's centre tap
    amask[j] = ok ? m : 0u;
  }
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int r = 8 * (wave + 4 * j) + lrow;
    boff[j] = ((n0 + r) * g.Cin + 4 * (lslot ^ ((r >> 1) & 7))) * 4;
  }
  const int nchunks = (ntaps + (dual ? 1 : 0)) * (g.Cin / KC);
  LAB_STAMP_L(0);

  // load cursor (uniform): which (tap, channel offset) the next request fetches; it stops at the last chunk, so the
  // trailing requests of the pipeline re-read that chunk harmlessly instead of branching.  The tap's offset is read
  // one chunk before it is needed (ntoff), so the scalar load is never waited for.
  int ltap = 0, lcin = 0, lcount = 0, wbuf = 0;
  int ndh = g.dh[tap0], ndw = g.dw[tap0];
  // one LDS-DMA instruction of the current request: pieces 0 .. A_LD-1 are A rows, the rest B rows
#ifndef MMSA_LAB_ABLATE
#define MMSA_LAB_ABLATE 0   // diagnostic builds (tools/igemm_lab.hip) drop pipeline stages to price them
#endif
  auto request_piece = [&](int j) {
    float* base = lds + wbuf * BUF_FLOATS + wave * 256;
    if (MMSA_LAB_ABLATE & 16) {   // every lane out of range: the instruction issues, nothing is fetched
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(base + j * 1024), 16, int(OOB), 0, 0, 0);
      return;
    }
    const bool second = ltap >= ntaps;   // uniform
    if (j < A_LD) {
      const int ntoff = (ndh * g.W + ndw) * g.Cin * 4;
      const int va = ((amask[j] >> ltap) & 1u) ? aoff[j] + ntoff : OOB;
      if (second)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX2, (lds_void*)(base + j * 1024), 16, int(va), int(lcin * 4), 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void*)(base + j * 1024), 16, int(va), int(lcin * 4), 0, 0);
    } else {
      if (second) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (lds_void*)(base + BM * KC + (j - A_LD) * 1024), 16,
                                                 int(boff[j - A_LD]), int(lcin * 4), 0, 0);
      } else {
        const int sb = (ltap * g.Cout * g.Cin + lcin) * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(base + BM * KC + (j - A_LD) * 1024), 16,
                                                 int(boff[j - A_LD]), int(sb), 0, 0);
      }
    }
  };
  auto advance_cursor = [&]() {
    wbuf = wbuf + 1 == NBUF ? 0 : wbuf + 1;
    if (lcount + 1 < nchunks) {   // uniform scalar bookkeeping
      ++lcount;
      lcin += KC;
      if (lcin == g.Cin) {
        lcin = 0;
        ++ltap;
        ndh = g.dh[tap0 + ltap];
        ndw = g.dw[tap0 + ltap];
      }
    }
  };
  auto request_next = [&]() {
#pragma unroll
    for (int j = 0; j < LD; ++j) request_piece(j);
    advance_cursor();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragments of half a chunk (16 of its 32 channels): lane (li, lh) holds, per 8-channel group gk, the 4 consecutive
  // channels 8*gk + 4*lh .. +3 of its row (piece 2*gk + lh) and feeds them to 4 successive MFMAs; A and B use the
  // same permutation of k.
  const int li = lane & 31, lh = lane >> 5;
  const int swz = (li >> 1) & 7;
  const int a_row = (wm * WTM + li) * KC;
  const int b_row = BM * KC + (wn * WTN + li) * KC;
  int piece[4];
#pragma unroll
  for (int gk = 0; gk < 4; ++gk) piece[gk] = 4 * ((2 * gk + lh) ^ swz);
  struct Frag {
    f32x4 a[2][TM], b[2][TN];
  };
  auto read_half = [&](Frag& f, int buf, int half) {
    const float* Ab = lds + buf * BUF_FLOATS + a_row;
    const float* Bb = lds + buf * BUF_FLOATS + b_row;
#pragma unroll
    for (int gk = 0; gk < 2; ++gk) {
      const int pc = piece[2 * half + gk];
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a[gk][i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * KC + pc);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[gk][j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * KC + pc);
    }
  };
  // the MFMAs of one k pair (step s = 4*gk + e of a half chunk)
  auto mfma_step = [&](const Frag& f, int s) {
    const int gk = s >> 2, e = s & 3;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[gk][i][e], f.b[gk][j][e], acc[i][j], 0, 0, 0);
  };
  auto mfma_steps = [&](const Frag& f, int s_begin, int s_end) {
#pragma unroll
    for (int s = s_begin; s < s_end; ++s) {
      mfma_step(f, s);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Pipeline (ring of NBUF LDS buffers, one barrier per chunk).  At the top of the step for chunk ck: chunks ck and
  // ck+1 are in LDS and visible, f0 holds the fragments of the first half of chunk ck, chunks ck+2 .. ck+NBUF-2 are
  // on their way.  MFMA waves: first-half MFMAs (reading the second-half fragments behind the first of them), then
  // second-half MFMAs (reading the first-half fragments of chunk ck+1 behind the first of them): every LDS read has
  // half a chunk of matrix work between its issue and its first use.  Loader waves: request chunk ck+NBUF-1 into
  // the buffer chunk ck-1 has left, then wait until at most the NBUF-3 youngest requests are outstanding (chunk
  // ck+2 has landed).  Both then join the barrier that publishes chunk ck+2 and retires the reads of chunk ck.
  Frag f0, f1;
  if (loader) {
    LAB_STAMP_L(1);
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) request_next();
    LAB_STAMP_L(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 3) * LD) : "memory");
    LAB_STAMP_L(3);
    __builtin_amdgcn_s_barrier();
    for (int ck = 0; ck < nchunks; ++ck) {
      if (!(MMSA_LAB_ABLATE & 1)) request_next();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 3) * LD) : "memory");
      if (!(MMSA_LAB_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    }
    // the trailing requests may still be writing LDS, which the statistics epilogue reuses
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    read_half(f0, 0, 0);
    LAB_STAMP(1);
    int rbuf = 0;
    for (int ck = 0; ck < nchunks; ++ck) {
      const int rnext = rbuf + 1 == NBUF ? 0 : rbuf + 1;
      // the LDS reads go out behind the first MFMA of each half: the matrix pipe is busy while they issue
      mfma_steps(f0, 0, 1);
      if (!(MMSA_LAB_ABLATE & 8)) read_half(f1, rbuf, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_steps(f0, 1, 8);
      mfma_steps(f1, 0, 1);
      if (!(MMSA_LAB_ABLATE & 8)) read_half(f0, rnext, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_steps(f1, 1, 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(MMSA_LAB_ABLATE & 4)) __builtin_amdgcn_s_barrier();
      rbuf = rnext;
    }
    __builtin_amdgcn_s_barrier();
  }

  LAB_STAMP(2);
  // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* Yb = Y + (long)b * HW * g.Cout;
  // BNRELU instantiations (multi-problem launches only): a problem with a coefficient table stores
  // max(scale * conv + shift, 0), the expression of bn_relu_apply_kernel
  const float* __restrict__ ep = BNRELU && many ? multi.bnrelu[blockIdx.z] : nullptr;   // uniform
  float csum[TN], csq[TN], esc[TN], esh[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    csum[j] = csq[j] = 0.f;
    esc[j] = 1.f;
    esh[j] = 0.f;
    if (BNRELU && ep) {
      const int col = n0 + wn * WTN + j * 32 + li;
      esc[j] = ep[2 * g.Cout + col];
      esh[j] = ep[3 * g.Cout + col];
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
            // a hardware float add at the L2: equal to read-add-write for a single writer, and with two writers onto
            // a zeroed map (the two halves of a split K range) still order-independent, since a + b == b + a
            unsafeAtomicAdd(dst, v);
          } else {
            *dst = v;
          }
          if (STATS) {
            csum[j] += raw;
            csq[j] = fmaf(raw, raw, csq[j]);
          }
        }
      }
    }
  if (STATS) {
    // combine the two lane halves, then the WAVES_M waves sharing a column range (through LDS, fixed order)
    float* red = lds;  // safe: every wave passed the loop's final barrier
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
      // stats[2][Cout][B*tiles]: the rows of one channel are contiguous for the kernel that sums them
      float* st = many ? multi.stats[blockIdx.z] : stats;
      st[((long)which * g.Cout + n0 + col) * gridDim.x + blockIdx.x] = s;
    }
  }
  LAB_STAMP(3);
}

#include "conv_rows3.h"

// ------------------------------------------------------------------------------------------------
// Weight gradient.  grid = (cin tiles of 32 x cout groups of 64, nsplit).  A workgroup walks over strips of R
// image rows (strip = blockIdx.y, += gridDim.y); per strip it stages the X rows r0-1 .. r0+R with a zero halo
// column on each side ([R+2][W+2][32 cin]) and the dY rows ([R][W][64 cout]) in LDS, so the inner loop has no
// bounds checks: every MFMA operand is one conflict-free ds_read_b32 (32 consecutive channels per lane half).
// Waves: (cout tile of 32) x (tap group: taps 0..4 / 5..8) for 3x3; for 1x1 the second index splits the rows
// instead and each half writes its own slab.  Accumulators stay in registers across strips; each workgroup
// writes one partial slab [tap][Cin][Cout] (two for 1x1), summed in fixed order by wgrad_reduce_kernel.
template <int NT>
__device__ __forceinline__ void conv_wgrad_body(const float* __restrict__ X, const float* __restrict__ dY,
                                                float* __restrict__ partial, const ConvGeom& g, int R, int nstrips,
                                                int strips_per_img, int nsplit) {
  if ((int)blockIdx.y >= nsplit) return;   // uniform (a paired launch sizes the grid for the larger problem)
  constexpr int MAXT = NT == 1 ? 1 : (NT + 1) / 2;
  extern __shared__ float lds[];
  const int WP = g.W + 2;
  float* Xs = lds;
  float* Ys = lds + (R + 2) * WP * 32;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, kk = lane >> 5;
  // 3x3: the two tap groups hold 5 and 4 taps.  Two workgroups share a CU; the second swaps the groups between its
  // wave pairs, so that (with the usual wave -> SIMD placement) every SIMD hosts one 5-tap and one 4-tap wave.
  const int swap = NT == 1 ? 0 : (int)(((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1);
  const int jt = wave & 1, q = (wave >> 1) ^ swap;
  const int cgroups = g.Cout / 64;
  const int cin0 = (blockIdx.x / cgroups) * 32;
  const int cout0 = (blockIdx.x % cgroups) * 64;
  const int t_first = NT == 1 ? 0 : (q == 0 ? 0 : MAXT);
  const int t_count = NT == 1 ? 1 : (q == 0 ? MAXT : NT - MAXT);
  int toff[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int tap = min(t_first + i, NT - 1);
    toff[i] = (g.dh[tap] * WP + g.dw[tap]) * 32;
  }
  f32x16 acc[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#ifdef MMSA_LAB_STAMPS
  unsigned long long lab_t0 = __builtin_amdgcn_s_memtime(), lab_load = 0, lab_comp = 0, lab_t;
  const unsigned long long lab_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int strip = blockIdx.y; strip < nstrips; strip += nsplit) {
#ifdef MMSA_LAB_STAMPS
    lab_t = __builtin_amdgcn_s_memtime();
#endif
    const int b = strip / strips_per_img;
    const int r0 = (strip - b * strips_per_img) * R;
    __syncthreads();  // the previous strip's reads are done
    // staging in batches of eight independent 16-byte loads per thread, so that one round trip to memory covers
    // eight pieces instead of one (requesting both strips whole before writing any of them is faster for the kernel
    // alone and slower inside the training step, where the larger bursts collide with the concurrent kernels)
    const float* Xb = X + (long)b * g.H * g.W * g.Cin + cin0;
    const int nx = (R + 2) * WP * 8;
    for (int i0 = t; i0 < nx; i0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int c4 = i & 7, pc = i >> 3;
        const int row = pc / WP, col = pc - row * WP;
        const int h = r0 - 1 + row, w = col - 1;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < nx && h >= 0 && h < g.H && w >= 0 && w < g.W)
          v[u] = *reinterpret_cast<const f32x4*>(Xb + ((long)h * g.W + w) * g.Cin + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        if (i < nx) *reinterpret_cast<f32x4*>(Xs + (long)i * 4) = v[u];
      }
    }
    const float* Yb = dY + (long)b * g.H * g.W * g.Cout + cout0;
    const int ny = R * g.W * 16;
    for (int i0 = t; i0 < ny; i0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int c4 = i & 15, pc = i >> 4;
        const int row = pc / g.W, w = pc - row * g.W;
        const int h = r0 + row;
        v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (i < ny && h < g.H) v[u] = *reinterpret_cast<const f32x4*>(Yb + ((long)h * g.W + w) * g.Cout + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        if (i < ny) *reinterpret_cast<f32x4*>(Ys + (long)i * 4) = v[u];
      }
    }
    __syncthreads();
#ifdef MMSA_LAB_STAMPS
    { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); lab_load += n_ - lab_t; lab_t = n_; }
#endif
    // One MFMA step covers two positions (k = lane half).  Two operand sets alternate: the LDS reads of step s+1 go
    // out before the MFMAs of step s and are first used a whole step later, so their latency hides behind the matrix
    // pipe.  Past the end of a row (odd W, or the extra step of an odd step count) the dY operand is zero and the X
    // address is clamped.  The loop is specialised on the wave's tap count (5 or 4 of the nine taps; 1 for 1x1).
    const int nsteps = (g.W + 1) >> 1;
    auto rows = [&](auto tc_tag) {
      constexpr int TC = decltype(tc_tag)::value;
      for (int r = (NT == 1 ? q : 0); r < R; r += (NT == 1 ? 2 : 1)) {
        const float* yrow = Ys + (r * g.W) * 64 + jt * 32 + li;
        const float* xrow = Xs + ((r + 1) * WP + 1) * 32 + li;
        float dy0, dy1, a0[TC], a1[TC];
        auto fetch = [&](int st, float& dy, float (&a)[TC]) {
          const int w = 2 * st + kk;
          const bool v = w < g.W;
          const int wc = v ? w : g.W - 1;
          dy = 0.f;
          if (v) dy = yrow[wc * 64];
#pragma unroll
          for (int i = 0; i < TC; ++i) a[i] = xrow[wc * 32 + toff[i]];
        };
        auto fma_step = [&](const float dy, const float (&a)[TC]) {
#pragma unroll
          for (int i = 0; i < TC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], dy, acc[i], 0, 0, 0);
        };
        fetch(0, dy0, a0);
        for (int st = 0; st < nsteps; st += 2) {
          fetch(st + 1, dy1, a1);
          __builtin_amdgcn_sched_barrier(0);
          fma_step(dy0, a0);
          __builtin_amdgcn_sched_barrier(0);
          fetch(st + 2, dy0, a0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < nsteps) fma_step(dy1, a1);   // uniform
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (NT == 1 || t_count == MAXT)
      rows(std::integral_constant<int, MAXT>{});
    else
      rows(std::integral_constant<int, (NT == 1 ? 1 : NT - MAXT)>{});
#ifdef MMSA_LAB_STAMPS
    { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); lab_comp += n_ - lab_t; }
#endif
  }

#ifdef MMSA_LAB_STAMPS
  const unsigned long long lab_t2 = __builtin_amdgcn_s_memtime();
#endif
  const int slab = NT == 1 ? (int)blockIdx.y * 2 + q : (int)blockIdx.y;
  float* out = partial + (long)slab * NT * g.Cin * g.Cout;
#pragma unroll
  for (int i = 0; i < MAXT; ++i)
    if (i < t_count) {
      const int tap = t_first + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cin0 + (r & 3) + 8 * (r >> 2) + 4 * kk;                          // D row
        out[((long)tap * g.Cin + ci) * g.Cout + cout0 + jt * 32 + li] = acc[i][r];      // D col = lane & 31
      }
    }
#ifdef MMSA_LAB_STAMPS
  if (threadIdx.x == 0 && g_lab_stamps) {
    const long wg_ = (long)blockIdx.y * gridDim.x + blockIdx.x;
    g_lab_stamps[wg_ * 8 + 0] = lab_load;
    g_lab_stamps[wg_ * 8 + 1] = lab_comp;
    g_lab_stamps[wg_ * 8 + 2] = __builtin_amdgcn_s_memtime() - lab_t2;
    g_lab_stamps[wg_ * 8 + 3] = __builtin_amdgcn_s_memtime() - lab_t0;
    g_lab_stamps[wg_ * 8 + 4] = lab_r0;
    g_lab_stamps[wg_ * 8 + 5] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                            float* __restrict__ partial, ConvGeom g, int R, int nstrips,
                                                            int strips_per_img) {
  conv_wgrad_body<NT>(X, dY, partial, g, R, nstrips, strips_per_img, (int)gridDim.y);
}

// The 3x3 and the 1x1 weight gradient of a residual block's two input convolutions (same input X, different output
// gradients) in one launch: blockIdx.z selects the problem, each with its own strip plan.
struct WgradArgs {
  const float* dY;
  float* partial;
  int R, nstrips, strips_per_img, nsplit;
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_pair_kernel(const float* __restrict__ X, const ConvGeom g9,
                                                                 const ConvGeom g1, const WgradArgs a9,
                                                                 const WgradArgs a1) {
  if (blockIdx.z == 0)
    conv_wgrad_body<9>(X, a9.dY, a9.partial, g9, a9.R, a9.nstrips, a9.strips_per_img, a9.nsplit);
  else
    conv_wgrad_body<1>(X, a1.dY, a1.partial, g1, a1.R, a1.nstrips, a1.strips_per_img, a1.nsplit);
}

// partial[slab][tap][Cin][Cout] -> dW[Cout][Cin][ntaps]  (+= when accumulate).  256 outputs per workgroup (four
// consecutive ones per lane, read as one 16-byte piece per slab); the NW (4 or 16, by the number of slabs) waves take
// interleaved slabs (four independent loads each per round) and are combined in fixed order through LDS.
constexpr int RED_OUT = 256;   // outputs per workgroup of the reduction
template <int NW>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, float* __restrict__ dW, int nslabs,
                                                  int ntaps, int Cin, int Cout, int accumulate, long block) {
  __shared__ f32x4 red[NW][64];
  const long total = (long)ntaps * Cin * Cout;         // a multiple of 4 (Cout is a multiple of 64)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = block * RED_OUT + 4 * lane;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 s0 = zero, s1 = zero, s2 = zero, s3 = zero;
  if (i < total) {
    const float* src = partial + i;
    int k = w;
    for (; k + 3 * NW < nslabs; k += 4 * NW) {
      s0 += *reinterpret_cast<const f32x4*>(src + (long)k * total);
      s1 += *reinterpret_cast<const f32x4*>(src + (long)(k + NW) * total);
      s2 += *reinterpret_cast<const f32x4*>(src + (long)(k + 2 * NW) * total);
      s3 += *reinterpret_cast<const f32x4*>(src + (long)(k + 3 * NW) * total);
    }
    for (; k < nslabs; k += NW) s0 += *reinterpret_cast<const f32x4*>(src + (long)k * total);
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  // 256 threads finish one output each: thread t takes component t & 3 of lane t >> 2
  const int t = threadIdx.x;
  const long io = block * RED_OUT + t;
  if (t < RED_OUT && io < total) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) s += red[k][t >> 2][t & 3];
    const int co = (int)(io % Cout);
    const long r = io / Cout;
    const int ci = (int)(r % Cin);
    const int tp = (int)(r / Cin);
    float* dst = dW + ((long)co * Cin + ci) * ntaps + tp;
    *dst = accumulate ? (*dst + s) : s;
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                               int nslabs, int ntaps, int Cin, int Cout, int accumulate) {
  wgrad_reduce_body<NW>(partial, dW, nslabs, ntaps, Cin, Cout, accumulate, (long)blockIdx.x);
}

// the reductions of a paired launch: the first blocks9 workgroups sum the 3x3 slabs, the rest the 1x1 slabs
template <int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_reduce_pair_kernel(const float* __restrict__ partial9,
                                                                    float* __restrict__ dW9, int nslabs9,
                                                                    const float* __restrict__ partial1,
                                                                    float* __restrict__ dW1, int nslabs1, int blocks9,
                                                                    int Cin, int Cout) {
  if ((int)blockIdx.x < blocks9)
    wgrad_reduce_body<NW>(partial9, dW9, nslabs9, 9, Cin, Cout, 0, (long)blockIdx.x);
  else
    wgrad_reduce_body<NW>(partial1, dW1, nslabs1, 1, Cin, Cout, 0, (long)blockIdx.x - blocks9);
}

constexpr size_t WGRAD_LDS_MAX = 150 * 1024;   // LDS a weight-gradient workgroup may ask for

struct WgradPlan {
  int R, strips_per_img, nstrips, nsplit, nslabs;
  size_t shmem;
};

WgradPlan plan_wgrad(int B, int H, int W, int Cin, int Cout, int ntaps) {
  WgradPlan p;
  int R = (128 + W / 2) / W;                       // about 128 positions per strip
  if (R < 1) R = 1;
  if (R > H) R = H;
  if (ntaps == 1 && R > 1 && (R & 1)) R -= 1;      // the two row halves of the 1x1 variant stay balanced
  p.R = R;
  p.strips_per_img = (H + R - 1) / R;
  p.nstrips = B * p.strips_per_img;
  const int pairs = (Cin / 32) * (Cout / 64);
  const long slab_bytes = (long)ntaps * Cin * Cout * 4 * (ntaps == 1 ? 2 : 1);
  static long wgs = -1;                                           // MMSA_WGRAD_WGS overrides (tuning aid)
  if (wgs < 0) {
    const char* e = getenv("MMSA_WGRAD_WGS");
    wgs = e ? atol(e) : 512;
    if (wgs < 64) wgs = 512;
  }
  long max_split = wgs / pairs;                                   // default: about two workgroups per CU
  static long budget_mb = -1;                                     // MMSA_WGRAD_BUDGET_MB overrides (tuning aid)
  if (budget_mb < 0) {
    const char* e = getenv("MMSA_WGRAD_BUDGET_MB");
    budget_mb = e ? atol(e) : 48;
    if (budget_mb < 1) budget_mb = 48;
  }
  const long budget = (budget_mb << 20) / slab_bytes;             // cap on the partial slabs (default 48 MiB)
  if (max_split > budget) max_split = budget;
  if (max_split < 1) max_split = 1;
  if (max_split > p.nstrips) max_split = p.nstrips;
  const int per = (int)((p.nstrips + max_split - 1) / max_split);  // strips per workgroup, evenly dealt
  p.nsplit = (p.nstrips + per - 1) / per;
  p.nslabs = p.nsplit * (ntaps == 1 ? 2 : 1);
  p.shmem = ((size_t)(R + 2) * (W + 2) * 32 + (size_t)R * W * 64) * sizeof(float);
  return p;
}

int fill_geom(ConvGeom& g, int B, int H, int W, int Cin, int Cout, int ntaps, const int* dh, const int* dw) {
  if (ntaps < 1 || ntaps > MAX_TAPS) return MMSA_ERR_ARG;
  g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ntaps = ntaps;
  for (int i = 0; i <= MAX_TAPS; ++i) {
    g.dh[i] = i < ntaps ? dh[i] : 0;
    g.dw[i] = i < ntaps ? dw[i] : 0;
  }
  return MMSA_OK;
}

template <int BM, int BN, int WM_, int WN_, bool STATS, bool ACCUM, int NB, bool BNRELU = false>
int launch_igemm_ring(const float* X, const float* Wp, float* Y, float* stats, const ConvGeom& g, int tiles, dim3 grid,
                      const ConvMulti& multi, hipStream_t s) {
  constexpr size_t shmem = NB * (size_t)(BM + BN) * KC * sizeof(float);
  static PerDeviceOnce attr_set;   // raising the dynamic-LDS limit is sticky per kernel and device
  if (shmem > 48 * 1024 && attr_set.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM_, WN_, STATS, ACCUM, NB, BNRELU>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    attr_set.done();
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM_, WN_, STATS, ACCUM, NB, BNRELU>), grid, dim3(512), shmem, s, X, Wp,
                     Y, stats, g, tiles, multi);
  return mmsa_launch_status();
}

// Ring depth per launch.  The 64 x 64 tile takes 16 KiB per buffer: five buffers (two workgroups per CU, deeper
// prefetch) unless a ring of three (three workgroups per CU) saves a whole round of workgroups over the chip's 256
// CUs -- the 2080-position maps at batch 32, whose 1056 workgroups need three rounds of 512 but two of 768 (5-9 %
// faster there; on the smaller maps, which fit one round either way, the shallower ring loses 3-8 %).  On grids of many
// rounds (inference batches) the third workgroup pays only where the K loop is short (kchunks = 32-channel chunks per
// tile <= 18: a workgroup's 15-20 k cycle start-up is then as long as its loop and a third resident workgroup hides
// it: 80 vs 68, 91 vs 85 TFLOP/s at batch 256); with longer loops the deeper ring wins (98 vs 95, 105 vs 92).
int pick_ring(int BM, int BN, long blocks, int kchunks) {
  const int nb = igemm_nbuf(BM, BN);
  if (BM == 64 && BN == 64 && nb > 3) {
    static int ring3 = -1;   // MMSA_IGEMM_RING3=0 keeps five buffers everywhere (A/B aid)
    if (ring3 < 0) {
      const char* e = getenv("MMSA_IGEMM_RING3");
      ring3 = e ? atoi(e) : 1;
    }
    if (ring3 && (blocks + 767) / 768 < (blocks + 511) / 512 && (kchunks <= 18 || blocks <= 1536)) return 3;
  }
  if (BM == 128 && BN == 64 && nb > 3) {
    static int ring128 = -1;   // MMSA_IGEMM_RING128=3: two 128-row workgroups per CU (A/B aid)
    if (ring128 < 0) {
      const char* e = getenv("MMSA_IGEMM_RING128");
      ring128 = e ? atoi(e) : 5;
    }
    if (ring128 == 3) return 3;
  }
  return nb;
}

template <int BM, int BN, int WM_, int WN_, bool STATS, bool ACCUM, bool BNRELU = false>
int launch_igemm_one(const float* X, const float* Wp, float* Y, float* stats, const ConvGeom& g, int tiles, dim3 grid,
                     const ConvMulti& multi, hipStream_t s) {
  constexpr int NB = igemm_nbuf(BM, BN);
  if constexpr (BN == 64 && NB > 3) {
    const int kchunks = (g.ntaps + (multi.X2 ? 1 : 0)) * (g.Cin / KC);
    if (pick_ring(BM, BN, (long)grid.x * grid.y * grid.z, kchunks) == 3)
      return launch_igemm_ring<BM, BN, WM_, WN_, STATS, ACCUM, 3, BNRELU>(X, Wp, Y, stats, g, tiles, grid, multi, s);
  }
  return launch_igemm_ring<BM, BN, WM_, WN_, STATS, ACCUM, NB, BNRELU>(X, Wp, Y, stats, g, tiles, grid, multi, s);
}

template <int BM, int BN, int WM_, int WN_>
int launch_igemm(const float* X, const float* Wp, float* Y, float* stats, const ConvGeom& g, int accumulate,
                 const ConvMulti& multi, hipStream_t s) {
  const int HW = g.H * g.W;
  const int tiles = mmsa_cdiv(HW, BM);
  dim3 grid(g.B * tiles, g.Cout / BN, multi.n > 0 ? multi.n : 1);
  if (stats) {
    if (accumulate) return MMSA_ERR_ARG;
    return launch_igemm_one<BM, BN, WM_, WN_, true, false>(X, Wp, Y, stats, g, tiles, grid, multi, s);
  }
  if (accumulate) return launch_igemm_one<BM, BN, WM_, WN_, false, true>(X, Wp, Y, stats, g, tiles, grid, multi, s);
  return launch_igemm_one<BM, BN, WM_, WN_, false, false>(X, Wp, Y, stats, g, tiles, grid, multi, s);
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
  static int forced = -1;   // MMSA_CONV_BM=64|128 overrides (tuning aid)
  if (forced < 0) {
    const char* e = getenv("MMSA_CONV_BM");
    forced = e ? atoi(e) : 0;
  }
  if (forced == 64 || forced == 128) return forced;
  // 64-row tiles always: at batch 32 they are 15-20 % faster on the 2080-position maps and 2x faster on the 128/512-
  // position ones (fine-grained tail, two or three workgroups per CU); at batch 256, where the tail no longer matters,
  // the 128-row tile with its one workgroup per CU (five buffers of 24 KiB) reaches 56-93 TFLOP/s where 64 rows reach
  // 80-105, and with two per CU (three buffers) it only draws level (75-105).  MMSA_CONV_BM=128 selects it for A/B.
  (void)B; (void)H; (void)W; (void)Cout;
  return 64;
}

// Whether the launchers below hand this geometry / offset table to the strip-sharing kernel (conv_rows3.h).
static bool uses_rows3(int B, int H, int W, int Cout, int ntaps, const int* dh, const int* dw) {
  const int mode = rows3_mode();
  if (!mode || !rows3_pattern(ntaps, dh, dw)) return false;
  return Cout % 64 != 0 || (mode == 2 && pick_bm(B, H, W, Cout) == 64);
}

MMSA_EXPORT int mmsa_conv_shares_tap_rows(int B, int H, int W, int Cout, int ntaps, const int* dh, const int* dw) {
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && ntaps >= 1 && ntaps <= MAX_TAPS && dh && dw);
  return uses_rows3(B, H, W, Cout, ntaps, dh, dw) ? 1 : 0;
}

// LDS ring depth (last template argument of the kernel instantiation) that a launch of `nproblems` convolutions with
// this geometry uses: part of the kernel's name in profiler output.
MMSA_EXPORT int mmsa_conv_ring_buffers(int B, int H, int W, int Cout, int nproblems, int kchunks) {
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && nproblems > 0 && kchunks > 0);
  const int bm = Cout % 64 ? 128 : pick_bm(B, H, W, Cout), bn = Cout % 64 ? 32 : 64;
  const long blocks = (long)B * mmsa_cdiv((long)H * W, bm) * (Cout / bn) * nproblems;
  return pick_ring(bm, bn, blocks, kchunks);
}

MMSA_EXPORT int mmsa_conv_tile_rows(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return MMSA_ERR_ARG;
  return pick_bm(B, H, W, Cout);
}

// Y[B,H,W,Cout] = conv(X[B,H,W,Cin], Wp[ntaps][Cout][Cin]) with per-tap offsets (dh, dw); out-of-image reads are 0.
// stats (optional): [2][Cout][B*ceil(H*W/BM)] column sums / sums of squares per tile, BM = mmsa_conv_tile_rows().
// accumulate != 0: Y += result (stats must be null).
MMSA_EXPORT int mmsa_conv_fwd(const float* X, const float* Wp, float* Y, float* stats, int B, int H, int W, int Cin,
                              int Cout, int ntaps, const int* dh, const int* dw, int accumulate, void* stream) {
  MMSA_REQUIRE(X && Wp && Y && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  MMSA_REQUIRE(Cin % KC == 0 && Cout % 32 == 0);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  hipStream_t s = mmsa_stream(stream);
  if (uses_rows3(B, H, W, Cout, ntaps, dh, dw)) {
    if (Cout % 64 != 0 && stats) return MMSA_ERR_ARG;
    Conv3Args a{};
    a.n = 1;
    a.ntaps[0] = 9; a.chunk0[0] = 0; a.chunk1[0] = Cin / KC;
    a.Wp[0] = Wp; a.Y[0] = Y; a.stats[0] = stats;
    a.second_z = -1;
    return launch_igemm3(X, g, a, accumulate != 0, s);
  }
  if (Cout % 64 != 0) {
    // 32-wide output (data gradient into a 32-channel tensor): 128 x 32 tiles, 4 x 1 waves
    if (stats) return MMSA_ERR_ARG;
    return launch_igemm<128, 32, 4, 1>(X, Wp, Y, nullptr, g, accumulate, ConvMulti{}, s);
  }
  if (pick_bm(B, H, W, Cout) == 128)
    return launch_igemm<128, 64, 2, 2>(X, Wp, Y, stats, g, accumulate, ConvMulti{}, s);
  return launch_igemm<64, 64, 2, 2>(X, Wp, Y, stats, g, accumulate, ConvMulti{}, s);
}

// mmsa_conv_fwd for a data gradient whose result Y is the gradient with respect to relu(bn(x)) (x [B,H,W,Cout], coeff
// that BatchNorm's [4][Cout] table): also leaves in stats ([2][Cout][B*tiles], the layout of mmsa_conv_fwd's
// statistics) the per-tile column sums of g = Y [bn(x) > 0] and of g * xhat, which mmsa_bn_relu_bwd_finish turns into
// the BatchNorm's parameter gradients and the gradient with respect to x.  3x3 offset tables and Cout % 64 == 0 only
// (mmsa_conv_shares_tap_rows tells): MMSA_ERR_ARG otherwise.
MMSA_EXPORT int mmsa_conv_fwd_bnrelu_bwd_stats(const float* X, const float* Wp, float* Y, float* stats, const float* x,
                                               const float* coeff, int B, int H, int W, int Cin, int Cout, int ntaps,
                                               const int* dh, const int* dw, void* stream) {
  MMSA_REQUIRE(X && Wp && Y && stats && x && coeff && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % KC == 0 && Cout % 64 == 0);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  MMSA_REQUIRE(uses_rows3(B, H, W, Cout, ntaps, dh, dw));
  Conv3Args a{};
  a.n = 1;
  a.ntaps[0] = 9; a.chunk0[0] = 0; a.chunk1[0] = Cin / KC;
  a.Wp[0] = Wp; a.Y[0] = Y; a.stats[0] = stats; a.bsx[0] = x; a.bscoeff[0] = coeff;
  a.second_z = -1;
  return launch_igemm3(X, g, a, false, mmsa_stream(stream));
}

// Y[B,H,W,Cout] = conv(X, Wp) + X2 . Wp2 in one launch: X2 [B,H,W,Cin] is a second input with the same channel count
// and Wp2 [Cout][Cin] its 1x1 weights (e.g. the data gradient of a residual block's input: the 3x3 branch plus the
// 1x1 shortcut).
MMSA_EXPORT int mmsa_conv_fwd_dual(const float* X, const float* Wp, const float* X2, const float* Wp2, float* Y, int B,
                                   int H, int W, int Cin, int Cout, int ntaps, const int* dh, const int* dw,
                                   void* stream) {
  MMSA_REQUIRE(X && Wp && X2 && Wp2 && Y && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  MMSA_REQUIRE(Cin % KC == 0 && Cout % 32 == 0);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  hipStream_t s = mmsa_stream(stream);
  if (uses_rows3(B, H, W, Cout, ntaps, dh, dw)) {
    Conv3Args a{};
    a.n = 1;
    a.ntaps[0] = 9; a.chunk0[0] = 0; a.chunk1[0] = Cin / KC;
    a.Wp[0] = Wp; a.Y[0] = Y;
    a.X2 = X2; a.Wp2 = Wp2; a.second_z = -1;
    return launch_igemm3(X, g, a, false, s);
  }
  ConvMulti m{};
  m.X2 = X2;
  m.Wp2 = Wp2;
  if (Cout % 64 != 0) return launch_igemm<128, 32, 4, 1>(X, Wp, Y, nullptr, g, 0, m, s);
  if (pick_bm(B, H, W, Cout) == 128) return launch_igemm<128, 64, 2, 2>(X, Wp, Y, nullptr, g, 0, m, s);
  return launch_igemm<64, 64, 2, 2>(X, Wp, Y, nullptr, g, 0, m, s);
}

// The same contraction with its K range split over two workgroups per output tile (taps 0..4 | taps 5..8 and the second
// pair), for geometries whose tiles cover less than half of the chip: Y MUST BE ZERO on entry; both halves add their
// part with hardware float atomics -- two addends per element, so the result does not depend on their order.
// mmsa_conv_fwd_dual_prefers_split tells whether the split pays for a geometry (at most one tile per CU of the chip;
// measured +0.9 % on the training step at batch 32, where the two smaller maps qualify).
MMSA_EXPORT int mmsa_conv_fwd_dual_prefers_split(int B, int H, int W, int Cout) {
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0);
  static int limit = -1;   // MMSA_DUAL_SPLIT=N: split up to N tiles (0 never; A/B aid)
  if (limit < 0) {
    const char* e = getenv("MMSA_DUAL_SPLIT");
    limit = e ? atoi(e) : 256;
  }
  if (!limit || Cout % 64 != 0 || pick_bm(B, H, W, Cout) != 64) return 0;
  const long blocks = (long)B * mmsa_cdiv((long)H * W, 64) * (Cout / 64);
  return blocks <= limit ? 1 : 0;
}

MMSA_EXPORT int mmsa_conv_fwd_dual_split(const float* X, const float* Wp, const float* X2, const float* Wp2,
                                         float* Y_zeroed, int B, int H, int W, int Cin, int Cout, int ntaps,
                                         const int* dh, const int* dw, void* stream) {
  MMSA_REQUIRE(X && Wp && X2 && Wp2 && Y_zeroed && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ntaps == 9);
  MMSA_REQUIRE(Cin % KC == 0 && Cout % 64 == 0);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  if (Cin / KC >= 2 && uses_rows3(B, H, W, Cout, ntaps, dh, dw)) {
    // the two workgroups of a tile take the lower and the upper half of the channel chunks (all nine taps and the
    // second pair each), so both walk whole rows of taps
    Conv3Args a{};
    a.n = 2;
    const int half = (Cin / KC + 1) / 2;
    for (int i = 0; i < 2; ++i) {
      a.ntaps[i] = 9; a.chunk0[i] = i ? half : 0; a.chunk1[i] = i ? Cin / KC : half;
      a.Wp[i] = Wp; a.Y[i] = Y_zeroed;
    }
    a.X2 = X2; a.Wp2 = Wp2; a.second_z = -1;
    return launch_igemm3(X, g, a, true, mmsa_stream(stream));
  }
  ConvMulti m{};
  m.n = 2;
  m.ntaps[0] = 5; m.tap0[0] = 0; m.Wp[0] = Wp;
  m.ntaps[1] = 4; m.tap0[1] = 5; m.Wp[1] = Wp + 5L * Cout * Cin;
  m.Y[0] = m.Y[1] = Y_zeroed;
  m.X2 = X2;
  m.Wp2 = Wp2;
  m.second_z = 1;
  return launch_igemm<64, 64, 2, 2>(X, Wp, Y_zeroed, nullptr, g, 1, m, mmsa_stream(stream));
}

// n (<= 4) convolutions of the same input X in one launch: problem i uses packed weights Wp[i] ([ntaps[i]][Cout][Cin]),
// entries tap0[i] .. tap0[i] + ntaps[i] - 1 of the shared offset table (tap0 null: every problem starts at entry 0) and
// writes Y[i] ([B,H,W,Cout]); stats (null, or n pointers) receives each problem's per-tile column sums / sums of
// squares in the layout of mmsa_conv_fwd; bnrelu (null, or n pointers, each null or a [4][Cout] BatchNorm coefficient
// table whose rows 2 and 3 hold scale and shift): problem i stores max(scale * conv + shift, 0) instead of the
// convolution -- for inference, where the table does not depend on the convolution's own statistics (stats must be
// null then).  Wp / Y / ntaps / tap0 / stats / bnrelu are host arrays.  ntable = entries of dh / dw.
MMSA_EXPORT int mmsa_conv_fwd_multi(const float* X, const float* const* Wp, float* const* Y, const int* ntaps,
                                    const int* tap0, float* const* stats, const float* const* bnrelu, int n, int B,
                                    int H, int W, int Cin, int Cout, int ntable, const int* dh, const int* dw,
                                    void* stream) {
  MMSA_REQUIRE(X && Wp && Y && ntaps && dh && dw && n >= 1 && n <= 4);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % KC == 0 && Cout % 64 == 0);
  MMSA_REQUIRE(ntable >= 1 && ntable <= MAX_TAPS);
  ConvMulti m{};
  m.n = n;
  bool any_bnrelu = false;
  for (int i = 0; i < n; ++i) {
    const int t0 = tap0 ? tap0[i] : 0;
    MMSA_REQUIRE(ntaps[i] >= 1 && t0 >= 0 && t0 + ntaps[i] <= ntable && Wp[i] && Y[i]);
    MMSA_REQUIRE(!stats || stats[i]);
    m.ntaps[i] = ntaps[i];
    m.tap0[i] = t0;
    m.Wp[i] = Wp[i];
    m.Y[i] = Y[i];
    m.stats[i] = stats ? stats[i] : nullptr;
    m.bnrelu[i] = bnrelu ? bnrelu[i] : nullptr;
    any_bnrelu = any_bnrelu || m.bnrelu[i];
  }
  MMSA_REQUIRE(!(any_bnrelu && stats));   // the folded form is for inference: no statistics wanted there
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntable, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  {
    // every problem either the whole 3x3 table or its centre tap alone: the strip-sharing kernel takes the launch
    bool rows3 = uses_rows3(B, H, W, Cout, ntable, dh, dw);
    for (int i = 0; rows3 && i < n; ++i) {
      const bool whole = m.ntaps[i] == 9 && m.tap0[i] == 0;
      const bool centre = m.ntaps[i] == 1 && dh[m.tap0[i]] == 0 && dw[m.tap0[i]] == 0;
      rows3 = whole || centre;
    }
    if (rows3) {
      Conv3Args a{};
      a.n = n;
      for (int i = 0; i < n; ++i) {
        a.ntaps[i] = m.ntaps[i]; a.chunk0[i] = 0; a.chunk1[i] = Cin / KC;
        a.Wp[i] = m.Wp[i]; a.Y[i] = m.Y[i]; a.stats[i] = m.stats[i]; a.bnrelu[i] = m.bnrelu[i];
      }
      a.second_z = -1;
      return launch_igemm3(X, g, a, false, mmsa_stream(stream));
    }
  }
  float* st0 = stats ? stats[0] : nullptr;
  if (any_bnrelu) {
    const int tiles = mmsa_cdiv(H * W, 64);
    return launch_igemm_one<64, 64, 2, 2, false, false, true>(X, m.Wp[0], m.Y[0], nullptr, g, tiles,
                                                              dim3(B * tiles, Cout / 64, n), m, mmsa_stream(stream));
  }
  if (stats && pick_bm(B, H, W, Cout) == 128)
    return launch_igemm<128, 64, 2, 2>(X, m.Wp[0], m.Y[0], st0, g, 0, m, mmsa_stream(stream));
  return launch_igemm<64, 64, 2, 2>(X, m.Wp[0], m.Y[0], st0, g, 0, m, mmsa_stream(stream));
}

MMSA_EXPORT int mmsa_conv_pack_desc_bytes() { return (int)sizeof(PackDesc); }

// descs: DEVICE array of n records {const float* w; float* fwd; float* dgrad; int Cout, Cin, ntaps, pad} (40 bytes each)
MMSA_EXPORT int mmsa_conv_pack_weights_multi(const void* descs, int n, void* stream) {
  MMSA_REQUIRE(descs && n > 0 && n <= 65535);
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(PACK_MAX_TILES, n), dim3(256), 0, mmsa_stream(stream),
                     static_cast<const PackDesc*>(descs));
  return mmsa_launch_status();
}

MMSA_EXPORT long mmsa_conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int ntaps, int* nsplit_out) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64 || (ntaps != 1 && ntaps != 9))
    return MMSA_ERR_ARG;
  const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, ntaps);
  if (p.shmem > WGRAD_LDS_MAX) return MMSA_ERR_LIMIT;   // a one-row strip of this map does not fit in LDS
  if (nsplit_out) *nsplit_out = p.nslabs;
  return (long)p.nslabs * ntaps * Cin * Cout;
}

// First half of the weight gradient  dW[Cout][Cin][ntaps] = sum over b,h,w of X[b,h+dh,w+dw,cin] * dY[b,h,w,cout]:
// partial slabs into `workspace` (mmsa_conv_wgrad_reduce finishes the job).  ntaps is 9 (offsets within one pixel of
// the centre) or 1 (the centre), the two shapes the residual blocks need; w_valid must equal W.
MMSA_EXPORT int mmsa_conv_wgrad(const float* X, const float* dY, float* workspace, int B, int H, int W, int w_valid,
                                int Cin, int Cout, int ntaps, const int* dh, const int* dw, void* stream) {
  MMSA_REQUIRE(X && dY && workspace && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && w_valid == W);
  MMSA_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0);
  MMSA_REQUIRE(ntaps == 1 || ntaps == 9);
  for (int i = 0; i < ntaps; ++i) MMSA_REQUIRE(dh[i] >= -1 && dh[i] <= 1 && dw[i] >= -1 && dw[i] <= 1);
  ConvGeom g;
  if (fill_geom(g, B, H, W, Cin, Cout, ntaps, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, ntaps);
  if (p.shmem > WGRAD_LDS_MAX) return MMSA_ERR_LIMIT;
  hipStream_t s = mmsa_stream(stream);
  dim3 grid((Cin / 32) * (Cout / 64), p.nsplit);
  if (ntaps == 9) {
    if (p.shmem > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<9>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.shmem) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_wgrad_kernel<9>, grid, dim3(256), p.shmem, s, X, dY, workspace, g, p.R, p.nstrips,
                       p.strips_per_img);
  } else {
    if (p.shmem > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.shmem) != hipSuccess)
      return MMSA_ERR_LAUNCH;
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), p.shmem, s, X, dY, workspace, g, p.R, p.nstrips,
                       p.strips_per_img);
  }
  return mmsa_launch_status();
}

// Both weight gradients that read a residual block's input X in one launch pair: dW9[Cout][Cin][9] from dY9 (the 3x3
// convolution, offsets dh / dw) and dW1[Cout][Cin][1] from dY1 (the 1x1 shortcut).  ws9 / ws1 are the slab workspaces of
// mmsa_conv_wgrad_workspace_floats for 9 and 1 taps; results are identical to two separate mmsa_conv_wgrad +
// mmsa_conv_wgrad_reduce sequences.
MMSA_EXPORT int mmsa_conv_wgrad_pair(const float* X, const float* dY9, const float* dY1, float* ws9, float* ws1,
                                     float* dW9, float* dW1, int B, int H, int W, int Cin, int Cout, const int* dh,
                                     const int* dw, void* stream) {
  MMSA_REQUIRE(X && dY9 && dY1 && ws9 && ws1 && dW9 && dW1 && dh && dw);
  MMSA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin % 32 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0);
  for (int i = 0; i < 9; ++i) MMSA_REQUIRE(dh[i] >= -1 && dh[i] <= 1 && dw[i] >= -1 && dw[i] <= 1);
  ConvGeom g9, g1;
  const int zero = 0;
  if (fill_geom(g9, B, H, W, Cin, Cout, 9, dh, dw) != MMSA_OK) return MMSA_ERR_ARG;
  if (fill_geom(g1, B, H, W, Cin, Cout, 1, &zero, &zero) != MMSA_OK) return MMSA_ERR_ARG;
  const WgradPlan p9 = plan_wgrad(B, H, W, Cin, Cout, 9), p1 = plan_wgrad(B, H, W, Cin, Cout, 1);
  const size_t shmem = p9.shmem > p1.shmem ? p9.shmem : p1.shmem;
  if (shmem > WGRAD_LDS_MAX) return MMSA_ERR_LIMIT;
  hipStream_t s = mmsa_stream(stream);
  if (shmem > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pair_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
    return MMSA_ERR_LAUNCH;
  const WgradArgs a9{dY9, ws9, p9.R, p9.nstrips, p9.strips_per_img, p9.nsplit};
  const WgradArgs a1{dY1, ws1, p1.R, p1.nstrips, p1.strips_per_img, p1.nsplit};
  dim3 grid((Cin / 32) * (Cout / 64), p9.nsplit > p1.nsplit ? p9.nsplit : p1.nsplit, 2);
  hipLaunchKernelGGL(conv_wgrad_pair_kernel, grid, dim3(256), shmem, s, X, g9, g1, a9, a1);
  const int blocks9 = (int)((9L * Cin * Cout + RED_OUT - 1) / RED_OUT);
  const int blocks1 = (int)(((long)Cin * Cout + RED_OUT - 1) / RED_OUT);
  if (p9.nslabs > 64 || p1.nslabs > 64)
    hipLaunchKernelGGL(wgrad_reduce_pair_kernel<16>, dim3(blocks9 + blocks1), dim3(1024), 0, s, ws9, dW9, p9.nslabs, ws1,
                       dW1, p1.nslabs, blocks9, Cin, Cout);
  else
    hipLaunchKernelGGL(wgrad_reduce_pair_kernel<4>, dim3(blocks9 + blocks1), dim3(256), 0, s, ws9, dW9, p9.nslabs, ws1,
                       dW1, p1.nslabs, blocks9, Cin, Cout);
  return mmsa_launch_status();
}

// Second half of the weight gradient: sums the partial slabs mmsa_conv_wgrad left in `workspace` (fixed order) into
// dW[Cout][Cin][ntaps] (= or +=).  Must follow mmsa_conv_wgrad with the same geometry on the same stream.
MMSA_EXPORT int mmsa_conv_wgrad_reduce(const float* workspace, float* dW, int B, int H, int W, int Cin, int Cout,
                                       int ntaps, int accumulate, void* stream) {
  MMSA_REQUIRE(workspace && dW && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  MMSA_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && (ntaps == 1 || ntaps == 9));
  const WgradPlan p = plan_wgrad(B, H, W, Cin, Cout, ntaps);
  const long total = (long)ntaps * Cin * Cout;
  if (p.nslabs > 64)
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((total + RED_OUT - 1) / RED_OUT)), dim3(1024), 0,
                       mmsa_stream(stream), workspace, dW, p.nslabs, ntaps, Cin, Cout, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((total + RED_OUT - 1) / RED_OUT)), dim3(256), 0,
                       mmsa_stream(stream),
                       workspace, dW, p.nslabs, ntaps, Cin, Cout, accumulate);
  return mmsa_launch_status();
}

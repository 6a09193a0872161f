// conv_nhwc_bf16.hip -- convolution (+ bias, ReLU) on bf16 tensors stored channels-last, for gfx950 (BASELINE config 5), hiprtc-specialised.
//
//   out[img][oy][ox][out_coff + oc] = relu( bias[oc] + sum_{ky,kx,c} in[img][oy*SY+ky-PY][ox*SX+kx-PX][c] * filts[oc][ky][kx][c] )
//
//   in     img:y:x:chan           bf16   (chan a multiple of 8; the layout pass pads with zero channels)
//   filts  out_chan:y:x:in_chan   bf16
//   biases out_chan               float
//   out    img:y:x:chan           bf16 (OUT_F32 = 0) or float (OUT_F32 = 1); may be a channel slice [out_coff, out_coff + OC) of a wider tensor
//
// This is the layout the matrix cores want: the contraction index k = (ky, kx, c) is CONTIGUOUS in both operands, so a lane's MFMA
// fragment (8 consecutive k of one row) is one 16-byte chunk in HBM, in LDS and in the register file, and nothing is converted,
// transposed or gathered element by element on the way.  The reference reaches its own fast kernels the same way: k1conv / tconv run on
// transposed copies of `in` / `filts` made by separate xpose functions outside the timed call, and write the next layer's transposed
// format directly (src/rtc_prof.cc:92-121 `<arg>_ref` dims + run_xpose; src/rtc_fwd.cc:229-243,495-503); its only reduced-precision
// precedent is 16-bit storage with fp32 math (src/cnn_codegen.cc:440-449).  Here: bf16 storage, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Implicit GEMM, no patch: D[i = oc][j = pel] with pel = (img, oy, ox) flattened.  A K step is BK consecutive k = BK/8 chunks; for
// K step s the lane that owns LDS slot (row, chunk) of the pel image loads chunk (s*BK/8 + chunk) of that pel's k axis, i.e. 8 channels of
// input position (oy*SY+ky-PY, ox*SX+kx-PX) -- an address that is affine in (ky, kx, c) per pel -- or nothing (zero) when the position is
// padding.  Both images are filled by `buffer_load_dwordx4 ... lds` (global -> LDS without passing through registers: no staging VGPRs,
// no ds_write instructions; out-of-range lanes -- padding, tile edges, the K tail -- are given an offset past the buffer, read as zero and
// land in LDS as zero).  A wave instruction fills 1 KB of an image lane-linearly, so the LDS image is row-major [row][BK] with the chunk
// position XOR-swizzled through the SOURCE address (slot (row, p) holds chunk p ^ f(row)) and the same XOR on the fragment read: the
// ds_read_b128 of a 16-lane group then covers all 64 banks (f(row) = (row / rows-per-256-B) mod chunks-per-row).
// Epilogue: bias + ReLU in the accumulator layout (a lane holds 4 consecutive out_chans of one pel per register quad), converted to bf16,
// written to an LDS tile [pel][oc] and read back as 16-byte row chunks: every global store instruction writes whole 16-byte runs of
// consecutive out_chans, a tile row is BI*2 contiguous bytes.
//
// Numerics: bf16 products are exact in fp32; the MFMA sums 16 of them per instruction in an order of its own, so results are NOT
// bit-comparable with a CPU loop; parity is stated against the oracle fed the same bf16 operands (tests), unpinned by construction.
//
// -D parameters: KNAME BI BJ BK(32|64; f32: 16|32) WI WJ MINW CIN KH KW SY SX PY PX CH CW COH COW RELU OUT_F32 NBUF(2..8) [IN_F32 SPLITK KSL]

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef GROUP_I
#define GROUP_I 8
#endif
#ifndef RELU
#define RELU 0
#endif
#ifndef OUT_F32
#define OUT_F32 0
#endif
#ifndef IN_F32
#define IN_F32 0  // 1: float tensors (in / filts / out), exact fp32 MFMA (v_mfma_f32_32x32x2_f32): every output is ONE ascending-k fma chain -- bit-identical to the
#endif            // reference's per-thread fmaf loop (and to gemm_conv_f32.hip).  A 16-byte chunk is 4 k; a lane reads its row's chunk with one ds_read_b128 and
                  // feeds two MFMAs from it (lanes 0-31 supply k = 4c, 4c+2, lanes 32-63 k = 4c+1, 4c+3: the instruction adds its two k in ascending order).
                  // Used by hip_conv for the shapes whose operands are k-contiguous in the REFERENCE layout already -- output 1x1, kernel == whole input
                  // (AlexNet fc6-fc8: in[img][K], filts[oc][K]) -- where the tile-starved, one-workgroup-per-CU launch needs many K steps in flight
                  // (a deep LDS-DMA ring) rather than registers.  Requires OUT_F32, no split-K (it would break the chain).
#ifndef SPLITK
#define SPLITK 0 // 1: the grid is tiles x p.splitk; slice s runs K steps [s*kt_per, (s+1)*kt_per) and stores its raw fp32 partial tile to slab s of p.ws
#endif           // ([pel][oc], row pitch Mi); bodahip_nhwc_splitk_reduce (REDUCE_ONLY) sums the slabs and applies bias / ReLU / the output type.  For
                 // tile-starved layers with a long K (7x7-map layers at 64 images, fully-connected layers): this path has no summation order to keep.
#ifndef KSL
#define KSL 1    // > 1: K SLICES REDUCED INSIDE THE LAUNCH (round 5).  The grid is tiles x KSL; workgroup bid runs tile bid / KSL, K steps [s * kt_per, (s + 1) * kt_per) with
#endif           // s = bid % KSL, stores its raw fp32 accumulators to its slab of the call's slice workspace (p.ws: tile tickets first, slabs from p.ws + p.ws_slab on; a
                 // slab is the accumulator registers thread by thread -- 16-byte stores, fully coalesced, no layout change), publishes them (release fence, agent scope:
                 // the slices of a tile run on different XCDs, i.e. behind different L2s) and takes a ticket (one atomic add per workgroup).  The workgroup that draws the
                 // LAST ticket of its tile acquires, sums the KSL slabs in slice order -- the same order whoever arrives last: run-to-run deterministic --, resets the
                 // ticket for the next launch / graph replay, and runs the ordinary epilogue.  One kernel, no second launch, no shared scratch: legal for members of a
                 // hip_conv_nhwc_set and for calls that overlap in an edge-free graph.  For tile-starved layers with a long K (7x7 / 14x14 maps at 64 images).
#ifndef ABLATE
#define ABLATE 0   // measurement only (wrong results): 1 = no operand loads, 2 = no fragment reads / MFMAs, 3 = fragment reads but no MFMAs,
                   // 4 = no K loop at all (prologue + epilogue), 5 = return at once (launch floor), 6 = K loop but no epilogue
#endif
#ifndef INTERLEAVE
#define INTERLEAVE 0  // 1: spread the next step's LDS-DMA pieces between the MFMA groups of the current step instead of issuing them back to back at the top of
#endif                // the step.  Measured on MI355X (ResNet-50 / GoogLeNet lists at 64 images): 35 % SLOWER (kernel time 1.51 -> 2.06 ms, res4 3x3 41 -> 64 us):
                      // an LDS-DMA issued among MFMAs and ds_reads costs far more than one issued in a run of its own.  Kept for the record, off.
#ifndef GROUPS
#define GROUPS 0 // 1: HORIZONTALLY FUSED convolutions -- up to four convolutions that read the same `in` with the same kernel geometry (an inception module's 1x1 /
#endif           // 3x3-reduce / 5x5-reduce convs; a ResNet stage's branch1 + branch2a) run as ONE launch: `filts` / `bias` are the members' filters / biases
                 // stacked along out_chan, every member padded with zero rows to a multiple of GP out_chans (GP a multiple of BI, so a tile row belongs to
                 // exactly one member), and each member's tile rows are stored to that member's own destination (tensor, channel count, channel offset) --
                 // second kernel argument grp_args_t.  The input is read once and the launch has the members' tiles together (they are tile-starved and
                 // launch-bound one by one).  Same MFMA chain per output as the separate launches: bit-identical results.  No split-K.
#ifndef NBUF
#define NBUF 2   // LDS ring depth: the loads of K step s + NBUF - 1 are issued before the MFMAs of step s (NBUF - 2 steps of loads stay in flight across a barrier)
#endif

// BODAHIP_AS_MEMBER (set by the host when it builds a hip_conv_nhwc_set kernel, native_kernels.cc: conv_nhwc_set): this file is then included once per member
// specialisation inside a namespace of its own, and instead of a kernel it defines the same body as   __device__ void KNAME(gemm_args_t const &p, int bid, char *smem)
// -- one workgroup of a larger grid running tile `bid` of this member, in LDS the wrapper kernel owns.  Nothing else changes: a member computes exactly what
// its own launch computes.
#ifdef BODAHIP_AS_MEMBER
#define BODAHIP_BID member_bid
#else
#define BODAHIP_BID blockIdx.x
#endif
#ifndef BODAHIP_ARGS_DEFINED
struct gemm_args_t { // identical to gemm_conv_f32.hip (one host-side struct); I = filts, J = in
  float const *I; float const *J; float *D; float const *bias;
  int Mi, Nj, K;
  int ldI, ldJ, ldD;
  int C, H, W, OH, OW;
  int tiles_i, tiles_j;
  int splitk, kt_per;
  float *ws; long ws_slab;
  unsigned I_bytes, J_bytes;
  unsigned D_bytes;
  int out_ctot, out_coff;
  int const *ktab; int ktab_n;
  long bsI, bsJ, bsD;
};

struct grp_args_t { // GROUPS: member m owns fused out_chans [oc0[m], oc0[m] + noc[m]) (oc0[m] multiples of the pad granularity; rows up to oc0[m+1] are zero padding)
  int n; int oc0[4]; int noc[4];
  void *D[4]; unsigned D_bytes[4]; int ctot[4]; int coff[4];
};
#endif // BODAHIP_ARGS_DEFINED

#ifdef REDUCE_ONLY
// out[pel][out_coff + oc] = cvt( relu( bias[oc] + sum_s ws[s][pel][oc] ) );  args: ws, ws_slab (floats per slab), splitk, D, Mi = OC, Nj = pels, out_ctot / out_coff.
// One thread per 4 consecutive out_chans (OC % 4 == 0) or per element.
extern "C" __global__ __launch_bounds__(256) void KNAME(gemm_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  bool const v4 = (p.Mi % 4 == 0) && (((p.out_ctot | p.out_coff) & 3) == 0);
  long const n = v4 ? (long)p.Nj * p.Mi / 4 : (long)p.Nj * p.Mi;
  if (idx >= n) return;
  if (v4) {
    long const e0 = idx * 4; int const pel = (int)(e0 / p.Mi), oc = (int)(e0 - (long)pel * p.Mi);
    f32x4 a = *reinterpret_cast<f32x4 const *>(p.bias + oc);
    for (int s = 0; s < p.splitk; ++s) a += *reinterpret_cast<f32x4 const *>(p.ws + (long)s * p.ws_slab + e0);
    if (RELU) { for (int e = 0; e < 4; ++e) a[e] = (a[e] > 0.f) ? a[e] : 0.f; }
    long const o = (long)pel * p.out_ctot + p.out_coff + oc;
#if OUT_F32
    *reinterpret_cast<f32x4 *>(p.D + o) = a;
#else
    bf16x4 b; b[0] = (__bf16)a[0]; b[1] = (__bf16)a[1]; b[2] = (__bf16)a[2]; b[3] = (__bf16)a[3];
    *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(p.D) + o) = b;
#endif
  } else {
    int const pel = (int)(idx / p.Mi), oc = (int)(idx - (long)pel * p.Mi);
    float a = p.bias[oc];
    for (int s = 0; s < p.splitk; ++s) a += p.ws[(long)s * p.ws_slab + idx];
    if (RELU) a = (a > 0.f) ? a : 0.f;
    long const o = (long)pel * p.out_ctot + p.out_coff + oc;
#if OUT_F32
    p.D[o] = a;
#else
    reinterpret_cast<__bf16 *>(p.D)[o] = (__bf16)a;
#endif
  }
}
#else

namespace {
constexpr int kNW = WI * WJ, kNT = kNW * 64;
constexpr int kTI = BI / (WI * 32), kTJ = BJ / (WJ * 32);
static_assert(BI % (WI * 32) == 0 && BJ % (WJ * 32) == 0, "tile must be a multiple of the 32x32 MFMA tile per wave");
constexpr int kEB = IN_F32 ? 4 : 2;           // bytes per element
constexpr int kCK = 16 / kEB;                 // k per 16-byte chunk
static_assert(IN_F32 ? (BK == 16 || BK == 32) : (BK == 32 || BK == 64), "BK: 32 | 64 (bf16), 16 | 32 (f32)");
static_assert(CIN % kCK == 0, "channels-last tensors carry a whole number of 16-byte chunks per position");
static_assert(!IN_F32 || (OUT_F32 && !SPLITK && KSL == 1), "the exact fp32 variant writes float and never splits K");
static_assert(KSL >= 1 && KSL <= 32 && !(SPLITK && KSL > 1), "K slices: 1..32, one mechanism at a time");
constexpr int kRowB = BK * kEB;               // bytes per LDS row
constexpr int kCPR = kRowB / 16;              // 16-byte chunks per LDS row
constexpr int kRP = 256 / kRowB;              // LDS rows per 256 bytes (one pass over the 64 banks)
constexpr int kCG = CIN / kCK;                // chunks per tap
constexpr int kTaps = KH * KW;
constexpr int kKC = kTaps * kCG;              // chunks along k
constexpr int kNK = (kKC + kCPR - 1) / kCPR;  // K steps
constexpr bool kFast = (kCG % kCPR == 0);     // every K step lies inside one tap
constexpr bool kNoPad = (PY == 0 && PX == 0 && (COH - 1) * SY + KH <= CH && (COW - 1) * SX + KW <= CW); // no tap ever leaves the plane
constexpr int kIInst = BI * kCPR / 64, kJInst = BJ * kCPR / 64;   // 1-KB wave instructions per image
static_assert((BI * kCPR) % 64 == 0 && (BJ * kCPR) % 64 == 0, "an image must be a whole number of 1-KB wave loads");
constexpr int kISlots = (kIInst + kNW - 1) / kNW, kJSlots = (kJInst + kNW - 1) / kNW;
static_assert(NBUF >= 2 && NBUF <= 8, "ring depth 2..8");
static_assert((NBUF - 2) * (kISlots + kJSlots) <= 56, "loads kept in flight must fit the 6-bit vmcnt");
static_assert(NBUF == 2 || (kIInst % kNW == 0 && kJInst % kNW == 0), "a ring deeper than 2 counts loads per wave: every wave must issue the same number");
constexpr int kLoadsPerStep = kISlots + kJSlots;   // (per wave, when they divide evenly)
constexpr int kIImg = BI * kRowB, kJImg = BJ * kRowB;            // bytes
constexpr int kEPitch = BI * 2 + 16;                               // epilogue tile [pel][oc] bf16, rows de-phased by 4 banks
constexpr int kStage = NBUF * (kIImg + kJImg), kEpi = (OUT_F32 || SPLITK) ? 0 : BJ * kEPitch;
constexpr int kSmem = kStage > kEpi ? kStage : kEpi;
constexpr int kOOB = (int)0x80000000;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void *lds_t;
__device__ __forceinline__ rsrc_t make_rsrc(void const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ constexpr int swz(int row) { return (row / kRP) & (kCPR - 1); }
} // namespace

#ifdef BODAHIP_AS_MEMBER
static_assert(!SPLITK && !IN_F32, "a member of a set: one kernel (K slices only in their in-launch form, KSL)");
constexpr int member_smem_bytes = kSmem, member_threads = WI * WJ * 64, member_minw = MINW;
__device__ __forceinline__ void KNAME(gemm_args_t const &p, grp_args_t const &q, int const member_bid, char *const smem) {   // (q: read by the GROUPS form only)
#elif GROUPS
static_assert(!SPLITK && !IN_F32, "fused convolutions: bf16 tensors, K slices only in their in-launch form (KSL)");
extern "C" __global__ __launch_bounds__(WI * WJ * 64, MINW) void KNAME(gemm_args_t const p, grp_args_t const q) {
  __shared__ __attribute__((aligned(1024))) char smem[kSmem];
#else
extern "C" __global__ __launch_bounds__(WI * WJ * 64, MINW) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(1024))) char smem[kSmem];
#endif
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave / WJ, wj = wave % WJ;
  if (ABLATE == 5) return;

  int tile_i, tile_j; // XCD-aware workgroup -> tile map (as gemm_conv_f32.hip)
  {
#if SPLITK
    int const bid = BODAHIP_BID / p.splitk;
#elif KSL > 1
    int const bid = BODAHIP_BID / KSL;
#else
    int const bid = BODAHIP_BID;
#endif
    int const nb = p.tiles_i * p.tiles_j;
    int const qn = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (qn + 1) : rr * (qn + 1) + (xcd - rr) * qn) + idx;
    int const group_sz = GROUP_I * p.tiles_j, gid = nid / group_sz, first_i = gid * GROUP_I;
    int const gsz = min(p.tiles_i - first_i, GROUP_I), in_g = nid - gid * group_sz;
    tile_i = first_i + in_g % gsz; tile_j = in_g / gsz;
  }
  int const i0 = tile_i * BI, j0 = tile_j * BJ;
  rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);

  // ---- this thread's LDS slots: slot s of an image = chunk ((s*kNW + wave)*64 + lane) of the row-major [row][kCPR] image
  int ibase[kISlots], ichunk[kISlots];        // filters: byte offset of row oc (k = 0) or OOB; logical chunk of the slot
  int jbase[kJSlots], jchunk[kJSlots], jyx[kJSlots];   // input: byte offset of (img, oy*SY-PY, ox*SX-PX, c = 0); logical chunk; (iy0 << 16) | (ix0 & 0xffff)
#pragma unroll
  for (int s = 0; s < kISlots; ++s) {
    int const ci = (s * kNW + wave) * 64 + lane, row = ci / kCPR, pos = ci % kCPR;
    ichunk[s] = pos ^ swz(row);
    ibase[s] = (i0 + row < p.Mi) ? (int)((unsigned)(i0 + row) * (unsigned)(kKC * 16)) : kOOB;
  }
#pragma unroll
  for (int s = 0; s < kJSlots; ++s) {
    int const ci = (s * kNW + wave) * 64 + lane, row = ci / kCPR, pos = ci % kCPR;
    jchunk[s] = pos ^ swz(row);
    int const pel = j0 + row;
    int const img = pel / (COH * COW), rem = pel - img * (COH * COW), oy = rem / COW, ox = rem - oy * COW;
    int const iy0 = oy * SY - PY, ix0 = ox * SX - PX;
    bool const ok = pel < p.Nj;
    jbase[s] = ((img * CH + iy0) * CW + ix0) * (CIN * kEB);
    jyx[s] = ok ? ((iy0 << 16) | (ix0 & 0xffff)) : (int)0x80008000;   // (a row of no image: every tap fails the range test)
    if (kNoPad && !ok) jbase[s] = kOOB;
  }

  // One K step's operand loads are kISlots + kJSlots 1-KB pieces per wave.  `piece` issues piece pc of step `step` into ring slot `buf`.
  struct step_ctx_t { int ky, kx, cg0; };
  auto step_ctx = [&](int step) {
    step_ctx_t c{0, 0, step * kCPR};
    if constexpr (kFast) { int const tap = step / (kCG / kCPR); c.cg0 = (step - tap * (kCG / kCPR)) * kCPR; c.ky = tap / KW; c.kx = tap - c.ky * KW; }
    return c;
  };
  auto piece = [&](int step, step_ctx_t const &c, int buf, int pc) {
    if (ABLATE == 1) return;
    char *const Ib = smem + buf * (kIImg + kJImg), *const Jb = Ib + kIImg;
    if (pc < kISlots) {
      int const s = pc, q = s * kNW + wave;
      if (kIInst % kNW != 0 && q >= kIInst) return;
      int const kc = step * kCPR + ichunk[s];
      int off = ibase[s] + kc * 16;
      if ((kKC % kCPR != 0 && kc >= kKC) || ibase[s] == kOOB) off = kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rI, (lds_t)(Ib + q * 1024), 16, off, 0, 0, 0);
    } else {
      int const s = pc - kISlots, q = s * kNW + wave;
      if (kJInst % kNW != 0 && q >= kJInst) return;
      int off; bool ok = true;
      int const iy0 = jyx[s] >> 16, ix0 = (int)(short)(jyx[s] & 0xffff);
      if constexpr (kFast) {
        off = jbase[s] + ((c.ky * CW + c.kx) * CIN + (c.cg0 + jchunk[s]) * kCK) * kEB;
        if constexpr (!kNoPad) ok = ((unsigned)(iy0 + c.ky) < (unsigned)CH) && ((unsigned)(ix0 + c.kx) < (unsigned)CW);
      } else {
        int const kc = step * kCPR + jchunk[s];
        int const t = kc / kCG, cg = kc - t * kCG, y = t / KW, x = t - y * KW;
        off = jbase[s] + ((y * CW + x) * CIN + cg * kCK) * kEB;
        ok = (kc < kKC);
        if constexpr (!kNoPad) ok = ok && ((unsigned)(iy0 + y) < (unsigned)CH) && ((unsigned)(ix0 + x) < (unsigned)CW);
      }
      if (kNoPad) { if (jbase[s] == kOOB) ok = false; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_t)(Jb + q * 1024), 16, ok ? off : kOOB, 0, 0, 0);
    }
  };
  constexpr int kPieces = kISlots + kJSlots;
  auto stage = [&](int step, int buf) {
    step_ctx_t const c = step_ctx(step);
#pragma unroll
    for (int pc = 0; pc < kPieces; ++pc) piece(step, c, buf, pc);
  };

  f32x16 acc[kTI][kTJ];
#pragma unroll
  for (int a = 0; a < kTI; ++a)
#pragma unroll
    for (int b = 0; b < kTJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment reads: lane l holds row (l & 31), k-chunk (2*kk + (l >> 5)) of each 16-deep MFMA step, at the swizzled chunk position
  int const h = lane >> 5, fsw = swz(lane & 31);
#if IN_F32
  constexpr int kKK = kCPR;   // one fragment read (4 k) per chunk, two MFMAs each
  int xo[kKK];
#pragma unroll
  for (int kk = 0; kk < kKK; ++kk) xo[kk] = (kk ^ fsw) * 16;
#else
  constexpr int kKK = BK / 16;
  int xo[kKK];
#pragma unroll
  for (int kk = 0; kk < kKK; ++kk) xo[kk] = ((2 * kk + h) ^ fsw) * 16;
#endif
  int const arow = (wi * (kTI * 32) + (lane & 31)) * kRowB, brow = (wj * (kTJ * 32) + (lane & 31)) * kRowB;

  // K loop over an NBUF-deep LDS ring.  Step s: issue the loads of step s + NBUF - 1 into the buffer step s - 1 has just released, run the
  // MFMAs of step s, wait until this wave's loads of step s + 1 have landed (a COUNTED vmcnt: the younger NBUF - 2 steps stay in flight),
  // then one barrier -- after it every wave's share of step s + 1 is in LDS and every wave is done reading step s.  The barrier is the
  // bare instruction: __syncthreads() would drain vmcnt to 0 while LDS-DMA is in flight.
  auto wait_loads = [&](int in_flight_steps) {   // (the count is an immediate: one case per depth)
    switch (in_flight_steps < 0 ? 0 : in_flight_steps) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerStep) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF > 3 ? 2 * kLoadsPerStep : 0) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF > 4 ? 3 * kLoadsPerStep : 0) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF > 5 ? 4 * kLoadsPerStep : 0) : "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF > 6 ? 5 * kLoadsPerStep : 0) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF > 7 ? 6 * kLoadsPerStep : 0) : "memory"); break;
    }
  };
  auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };   // (the step's last MFMAs, which the compiler may sink below it, only read registers)
#if SPLITK
  int const k_begin = (int)(BODAHIP_BID % p.splitk) * p.kt_per, nk = max(0, min(kNK, k_begin + p.kt_per) - k_begin);   // this slice's K steps
#elif KSL > 1
  int const k_begin = (int)(BODAHIP_BID % KSL) * p.kt_per, nk = max(0, min(kNK, k_begin + p.kt_per) - k_begin);
#else
  constexpr int k_begin = 0, nk = (ABLATE == 4) ? 0 : kNK;
#endif
#pragma unroll
  for (int s0 = 0; s0 < NBUF - 1; ++s0) if (s0 < nk) stage(k_begin + s0, s0);
  wait_loads((nk < NBUF - 1 ? nk : NBUF - 1) - 1);
  barrier();
  int cur = 0;
  for (int step = 0; step < nk; ++step) {
    int const nxt = (cur == 0) ? NBUF - 1 : cur - 1;   // == (step + NBUF - 1) % NBUF
    bool const more = step + NBUF - 1 < nk;
    step_ctx_t const nc = step_ctx(k_begin + step + NBUF - 1);
#if !INTERLEAVE
    if (more) stage(k_begin + step + NBUF - 1, nxt);
#endif
    char const *const Ib = smem + cur * (kIImg + kJImg), *const Jb = Ib + kIImg;
    constexpr int kPP = (kPieces + kKK - 1) / kKK;   // pieces of the next step issued per MFMA group (INTERLEAVE only)
#pragma unroll
    for (int kk = 0; kk < kKK; ++kk) {
      if (ABLATE == 2) break;
#if IN_F32
      f32x4 a[kTI], b[kTJ];
#pragma unroll
      for (int t = 0; t < kTI; ++t) a[t] = *reinterpret_cast<f32x4 const *>(Ib + arow + t * (32 * kRowB) + xo[kk]);
#pragma unroll
      for (int t = 0; t < kTJ; ++t) b[t] = *reinterpret_cast<f32x4 const *>(Jb + brow + t * (32 * kRowB) + xo[kk]);
#pragma unroll
      for (int half = 0; half < 2; ++half) {      // k = 4kk + 2*half (lanes 0-31) | + 1 (lanes 32-63)
        float av[kTI], bw[kTJ];
#pragma unroll
        for (int t = 0; t < kTI; ++t) av[t] = h ? a[t][2 * half + 1] : a[t][2 * half];
#pragma unroll
        for (int t = 0; t < kTJ; ++t) bw[t] = h ? b[t][2 * half + 1] : b[t][2 * half];
#pragma unroll
        for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
          for (int tb = 0; tb < kTJ; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta], bw[tb], acc[ta][tb], 0, 0, 0);
      }
#else
      bf16x8 a[kTI], b[kTJ];
#pragma unroll
      for (int t = 0; t < kTI; ++t) a[t] = *reinterpret_cast<bf16x8 const *>(Ib + arow + t * (32 * kRowB) + xo[kk]);
#pragma unroll
      for (int t = 0; t < kTJ; ++t) b[t] = *reinterpret_cast<bf16x8 const *>(Jb + brow + t * (32 * kRowB) + xo[kk]);
#if INTERLEAVE
      if (more) {
#pragma unroll
        for (int pc = kk * kPP; pc < (kk + 1) * kPP && pc < kPieces; ++pc) piece(k_begin + step + NBUF - 1, nc, nxt, pc);
      }
#endif
#pragma unroll
      for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
        for (int tb = 0; tb < kTJ; ++tb) {
          if (ABLATE == 3) { asm volatile("" ::"v"(a[ta]), "v"(b[tb])); continue; }   // (keeps the reads alive)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
        }
#if INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);   // keep the groups in program order: reads | loads | MFMAs
#endif
#endif
    }
    // loads still wanted in flight after this wait: those of steps step + 2 .. step + NBUF - 1 that exist
    int const newest = (nk - 1 < step + NBUF - 1) ? nk - 1 : step + NBUF - 1;
    int const fl = newest - (step + 1);
    wait_loads(NBUF == 2 ? 0 : fl);
    // ... and this wave's fragment reads of the step must have RETURNED before it releases the slot: an LDS-DMA write of another wave does not queue behind ds_reads
    // that were merely issued (found with the multi-problem launch, tools/multi_stress.py: 32x128 tiles / 1x4 waves / ring of two -- intermittently wrong outputs
    // without this wait, none in 240 launches with it; the same hazard exists here for every ring depth, the slot refilled at the top of step s is the one step s-1 read)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    cur = (cur + 1 == NBUF) ? 0 : cur + 1;
  }
  if (ABLATE == 6) { if (acc[0][0][0] == 123.456f) p.D[0] = 1.f; return; }
#if KSL > 1
  {   // ---- in-launch reduction of the K slices (see KSL at the top of conv_nhwc_bf16.hip).  Publish / combine as /opt/skills/guides/cdna_hip_programming.md prescribes for a
      // split-K seam on gfx950: write-through (sc1) 16-byte slab stores, every wave drains vmcnt, barrier, ONE relaxed agent-scope ticket per workgroup; the last
      // arriver reads the slabs with sc1 loads (L1-bypassing: the per-XCD L2s are not coherent with each other, a slab written write-through is read from the fabric).
    int const tile_id = (int)BODAHIP_BID / KSL, slice = (int)BODAHIP_BID % KSL;
    constexpr int kQ = kTI * kTJ * 4;                                  // accumulator quads per thread
    constexpr int kSlabB = kQ * kNT * 16;                           // bytes per slab: the accumulator registers, thread by thread (fully coalesced 16-byte accesses)
    rsrc_t const rW = make_rsrc(p.ws + p.ws_slab + (long)tile_id * (long)(KSL * (kSlabB / 4)), (unsigned)(KSL * kSlabB));
    unsigned *const ticket = reinterpret_cast<unsigned *>(p.ws) + tile_id;
#pragma unroll
    for (int a = 0; a < kTI; ++a)
#pragma unroll
      for (int b = 0; b < kTJ; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float const x0 = acc[a][b][4 * g], x1 = acc[a][b][4 * g + 1], x2 = acc[a][b][4 * g + 2], x3 = acc[a][b][4 * g + 3];
          f32x4 v; v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rW, slice * kSlabB + (((a * kTJ + b) * 4 + g) * kNT + tid) * 16, 0, 16);
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier" ::: "memory");       // every wave's share of the slab has left (write-through: acknowledged by the fabric)
    unsigned *const flag = reinterpret_cast<unsigned *>(smem);                     // (the operand images are dead: every wave is past the K loop's last barrier)
    if (tid == 0) *flag = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier" ::: "memory");
    bool const last = (*flag == (unsigned)(KSL - 1));
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");                 // (the flag is read before the epilogue reuses the LDS)
    if (!last) return;
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch / graph replay
    // the sum runs over the slabs in slice order whoever arrived last (this workgroup's own slab is read back like the others): run-to-run deterministic
#pragma unroll
    for (int a = 0; a < kTI; ++a)
#pragma unroll
      for (int b = 0; b < kTJ; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int const o = (((a * kTJ + b) * 4 + g) * kNT + tid) * 16;
          f32x4 sum = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, o, 0, 16));
#pragma unroll
          for (int s = 1; s < KSL; ++s) sum += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, s * kSlabB + o, 0, 16));
          float const y0 = sum[0], y1 = sum[1], y2 = sum[2], y3 = sum[3];
          acc[a][b][4 * g] = y0; acc[a][b][4 * g + 1] = y1; acc[a][b][4 * g + 2] = y2; acc[a][b][4 * g + 3] = y3;
        }
  }
#endif
  // ---- epilogue.  C/D layout of the 32x32 MFMA family: column j = lane & 31, rows i = 8*g + 4*(lane >> 5) + e for register 4*g + e:
  // a lane holds 4 consecutive out_chans of one pel per register quad
#if SPLITK
  rsrc_t const rD = make_rsrc(p.ws + (long)(BODAHIP_BID % p.splitk) * p.ws_slab, (unsigned)p.Nj * (unsigned)p.Mi * 4u), rB = make_rsrc(p.bias, 0u);  // (no bias here: every load reads 0)
  int const o_ctot = p.Mi, o_coff = 0;
#elif GROUPS
  // the member this tile row belongs to (workgroup-uniform); from here on `e_i0` / `e_Mi` are the tile's first out_chan and the out_chan count INSIDE the member
  int gm = 0;
#pragma unroll
  for (int m = 1; m < 4; ++m) if (m < q.n && i0 >= q.oc0[m]) gm = m;
  rsrc_t const rD = make_rsrc(q.D[gm], q.D_bytes[gm]), rB = make_rsrc(p.bias + q.oc0[gm], (unsigned)q.noc[gm] * 4u);
  int const o_ctot = q.ctot[gm], o_coff = q.coff[gm];
#else
  rsrc_t const rD = make_rsrc(p.D, p.D_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
  int const o_ctot = p.out_ctot, o_coff = p.out_coff;
#endif
#if GROUPS
  int const e_i0 = i0 - q.oc0[gm], e_Mi = q.noc[gm];
#else
  int const e_i0 = i0, e_Mi = p.Mi;
#endif
  constexpr bool kRelu = RELU && !SPLITK;
  f32x4 bv[kTI][4];
#pragma unroll
  for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int const oc = e_i0 + wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h;
      if (oc + 4 <= e_Mi) bv[ta][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, oc * 4, 0, 0));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[ta][g][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, (oc + e < e_Mi) ? (oc + e) * 4 : kOOB, 0, 0));
      }
    }
#if OUT_F32 || SPLITK
  // fp32 output (tests / the last layer of a net; the partial tiles of a K slice): 16-byte stores of 4 consecutive out_chans straight from the accumulator layout
#pragma unroll
  for (int tb = 0; tb < kTJ; ++tb) {
    int const pel = j0 + wj * (kTJ * 32) + tb * 32 + (lane & 31);
    unsigned const rowoff = ((unsigned)pel * (unsigned)o_ctot + (unsigned)o_coff) * 4u;
#pragma unroll
    for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int const oc = e_i0 + wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h;
        float x[4];   // (scalars, not elements of a vector: this hipcc mis-compiles element-wise bit-casts of a vector's lanes -- DESIGN.md section 3.1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = acc[ta][tb][4 * g + e] + bv[ta][g][e]; if (kRelu) x[e] = (x[e] > 0.f) ? x[e] : 0.f; }
        if (pel < p.Nj) {
          if (oc + 4 <= e_Mi && ((o_ctot | o_coff) & 3) == 0) {
            f32x4 v; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rD, (int)(rowoff + (unsigned)oc * 4u), 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (oc + e < e_Mi) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x[e]), rD, (int)(rowoff + (unsigned)(oc + e) * 4u), 0, 0);
          }
        }
      }
  }
#else
  {
    char *const E = smem;   // (every wave is past the last barrier of the K loop: the operand images are dead)
#pragma unroll
    for (int tb = 0; tb < kTJ; ++tb) {
      int const prow = wj * (kTJ * 32) + tb * 32 + (lane & 31);
#pragma unroll
      for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { float x = acc[ta][tb][4 * g + e] + bv[ta][g][e]; if (kRelu) x = (x > 0.f) ? x : 0.f; v[e] = (__bf16)x; }
          *reinterpret_cast<bf16x4 *>(E + prow * kEPitch + (wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h) * 2) = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    constexpr int kRowChunks = BI / 8, kChunks = BJ * kRowChunks;
    bool const vec_ok = ((o_ctot | o_coff) & 7) == 0;   // 16-byte aligned rows and slices
#pragma unroll
    for (int n = 0; n < (kChunks + kNT - 1) / kNT; ++n) {
      int const idx = tid + n * kNT;
      if (kChunks % kNT != 0 && idx >= kChunks) break;
      int const prow = idx / kRowChunks, cc = idx - prow * kRowChunks;
      int const pel = j0 + prow, oc = e_i0 + cc * 8;
      if (pel >= p.Nj || oc >= e_Mi) continue;
      u32x4 const v = *reinterpret_cast<u32x4 const *>(E + prow * kEPitch + cc * 16);
      unsigned const off = ((unsigned)pel * (unsigned)o_ctot + (unsigned)o_coff + (unsigned)oc) * 2u;
      if (vec_ok && oc + 8 <= e_Mi) __builtin_amdgcn_raw_buffer_store_b128(v, rD, (int)off, 0, 0);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (oc + e < e_Mi) __builtin_amdgcn_raw_buffer_store_b16((short)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu), rD, (int)(off + 2u * e), 0, 0);
      }
    }
  }
#endif
}
#endif // REDUCE_ONLY

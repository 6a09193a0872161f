// conv_nhwc_patch_bf16.hip -- KH x KW (stride 1 in x) convolution on channels-last bf16 tensors from an LDS INPUT PATCH, for gfx950 (BASELINE config 5).
// Same contract as bodahip_conv_nhwc_bf16 (kernels/conv_nhwc_bf16.hip: in / out img:y:x:chan bf16, biases float, bias + ReLU epilogue, optional channel slice
// of a wider output, float output) except for the filters, which arrive as
//       filts  in_grp : y : x : out_chan : in_chan8     bf16      F'[g][ky][kx][oc][8] = filts[oc][ky][kx][8g .. 8g+8)
// (the layout pass writes that form for the layers that take this kernel: `hip_conv_nhwc` binds by the dims of its `filts`).
//
// Why: the implicit-GEMM kernel re-reads every input position once per tap -- for a 3x3 layer the pel-side operand stream is 9x the tensor, and on the
// 3x3 layers of ResNet-50 / GoogLeNet at 64 images that L2 -> LDS stream (measured ~22 TB/s chip-wide) takes as long as the MFMAs and overlaps poorly with
// them.  Here a K step is CG groups of 8 channels x ALL taps: the LDS holds, per group, the zero-padded input rows the tile's output positions touch
// ("slots" of W + 2 PX chunks; consecutive output rows of an image share slots; a tile that crosses into the next image starts a new slot group -- the
// geometry of gemm_conv_f32.hip's patch mode and of conv_patch_bf16.hip), each input chunk is loaded ONCE per K step with a 16-byte load, and a lane's MFMA B
// fragment for tap (ky, kx) is one ds_read_b128 at   patch[g][(slot(j) + ky) * Wp + ox(j) + kx].   The contraction index is ordered (group, ky, kx, channel
// in group) in both operands.  The patch is register-staged (the global loads of step s+1 fly under the MFMAs of step s); lanes of a staging load are laid out
// position-major (CG consecutive lanes = the CG x 16 contiguous bytes of one position), the LDS image group-major with a group pitch that is 2 (mod 16) chunks, so
// the transposing ds_write_b128s are conflict-free, and a slot pitch that keeps the fragment reads conflict-free across row ends (WPITCH below).
// The FILTERS take one of two paths: by default (ADIRECT=1, see the option below) every wave loads its own MFMA A fragments straight from global memory through a
// register ring and the LDS holds only the patch, double-buffered; with ADIRECT=0 they are register-staged into the same single LDS image as the patch.
//
// Numerics: as conv_nhwc_bf16.hip (fp32 accumulate in the MFMA's own order; parity stated against the oracle on bf16-rounded operands).
//
// -D parameters: KNAME BI BJ WI WJ MINW CG CIN KH KW SY PY PX CH CW COH COW RELU OUT_F32 [ADIRECT PF BPF WPITCH DBUF ABLATE POOL KSL]       (SX == 1)

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
#ifndef DBUF
#define DBUF 0     // 1: two LDS images -- step s+1 is stored into the other image right after the MFMAs of step s were issued (one barrier per K step instead of two,
#endif             // a wave's staging stores overlap the other waves' MFMAs); costs twice the LDS
#ifndef ADIRECT
#define ADIRECT 0  // 1: the FILTER operand does not pass through the LDS at all.  A filter chunk has no reuse across taps (the patch has KH*KW-fold reuse): staging it only
#endif             // shares it between the WJ waves of a tile row, at the price of most of the ds_write traffic of a K step (a 128 x 36-slot filter image is 18 of the 23
                   // 16-byte stores per thread) and of the LDS space that limits the workgroups per CU.  Here every wave loads its own MFMA A fragments straight from
                   // global memory -- in F'[g][ky][kx][oc][8] the fragment of k-iteration s is 2 x 512 contiguous bytes, rows (s, s+1) of a [k-slot][out_chan] matrix
                   // -- PF k-iterations ahead through a register ring (counted vmcnt, loads return in order), waves of the same tile row meet in the L1.  The LDS
                   // holds only the patch, double-buffered: one barrier per K step.  Wants wave tiles wide in pels (kTJ = 4: one 1-KB A load per four MFMAs).
#ifndef PF
#define PF 8       // ADIRECT: k-iterations of filter fragments in flight
#endif
#ifndef BPF
#define BPF 1      // ADIRECT: k-iterations the patch fragment reads (ds_read_b128) run ahead of their MFMAs
#endif
#ifndef ABLATE
#define ABLATE 0   // measurement only (wrong results): 1 = no operand loads, 2 = no fragment reads / MFMAs, 3 = no LDS stores of the staged operands, 4 = no K loop at all
#endif
#ifndef KSL
#define KSL 1      // > 1: K slices reduced inside the launch -- see conv_nhwc_bf16.hip (same protocol, same workspace layout); ADIRECT form only
#endif
#ifndef POOL
#define POOL 0     // 1: MAX POOLING FUSED INTO THE CONSUMING 1x1 CONVOLUTION (an inception module's pool -> pool-projection pair, test/rtc/pool.cucl + a k1conv in the
#endif             // reference: two functions, src/rtc_fwd.cc:545-549):   out[oc][pel] = relu( bias + sum_c F[oc][c] * max_{ky,kx in KH x KW} in[pel + (ky,kx) - pad][c] ).
                   // The patch of a K step is what the KH x KW window needs -- exactly the rows this kernel stages for a KH x KW convolution -- and a lane's B fragment is the
                   // element-wise maximum of the KH*KW patch reads instead of one of them; the filters are 1x1 (F'[g][0][0][oc][8]: one k-slot per channel group).  The
                   // maximum is taken on the bf16 bit patterns as signed 16-bit integers (v_pk_max_i16), which orders NON-NEGATIVE values like their floats, and window
                   // positions outside the plane read the patch's zero padding: both are right exactly when the input is non-negative (the output of a ReLU, of a pooling
                   // or concatenation of such) -- the caller's obligation (boda_amd/conv_pipe.py checks the producers).  Per output the MFMA chain is the 1x1 convolution's
                   // own (ascending 16-channel groups): bit-identical to pooling and convolution run apart.

// BODAHIP_AS_MEMBER: see conv_nhwc_bf16.hip -- this file as one member of a hip_conv_nhwc_set kernel ( __device__ void KNAME(gemm_args_t const &p, int bid, char *smem) ).
#ifdef BODAHIP_AS_MEMBER
#define BODAHIP_BID member_bid
#else
#define BODAHIP_BID blockIdx.x
#endif
#ifndef BODAHIP_ARGS_DEFINED
struct gemm_args_t { // identical to gemm_conv_f32.hip (one host-side struct); I = F', J = in
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
#endif // BODAHIP_ARGS_DEFINED

namespace {
constexpr int kNT = WI * WJ * 64;
constexpr int kTI = BI / (WI * 32);
constexpr int kTJ = BJ / (WJ * 32);
static_assert(BI % (WI * 32) == 0 && BJ % (WJ * 32) == 0, "tile must be a multiple of the MFMA tile per wave");
static_assert(CIN % 8 == 0, "channels-last tensors carry whole 16-byte chunks per position");
#ifndef WPITCH
#define WPITCH 1   // 1: slot pitch padded so that consecutive output positions read consecutive-modulo-16 chunks across row ends (bank-conflict-free fragment reads)
#endif
constexpr int kTaps = KH * KW, kWr = CW + 2 * PX;                  // columns of a slot that hold data
// The MFMA B fragment of 32 consecutive output positions is one ds_read_b128 per lane, served in groups of 16 lanes that must cover 16 different chunks modulo 16
// (64 banks x 4 bytes).  Consecutive positions of one row are consecutive chunks; at a row end the chunk index jumps by SY * pitch - COW, so with a slot pitch that
// makes that jump a multiple of 16 the chunk index stays  position + const (mod 16)  across the rows of a tile.  Measured before (AlexNet conv3, 13 x 13 maps,
// 256 images): SQ_LDS_BANK_CONFLICT = 48 % of SQ_LDS_IDX_ACTIVE, the LDS busy 62 % of the kernel.  The pad columns are never loaded, stored or read.
constexpr int wpitch() { for (int p = kWr; p < kWr + 16; ++p) if ((SY * p - COW) % 16 == 0) return p; return kWr; }
constexpr int kWp = WPITCH ? wpitch() : kWr;
constexpr int kNCG = CIN / 8;                                      // channel groups of the tensor
constexpr int kNKT = (kNCG + CG - 1) / CG;                         // K steps
static_assert(!POOL || ADIRECT, "the fused-pooling form reads its filter fragments directly");
static_assert(KSL == 1 || (ADIRECT && KSL <= 32), "K slices: the direct-filter form, at most 32");
constexpr int kNPr = POOL ? CG : CG * kTaps;                       // k-slots (of 8 channels) per K step (POOL: one per channel group -- the window is reduced before the MFMA) ...
constexpr int kNP = kNPr + (kNPr & 1);                             // ... padded to whole MFMAs (two slots each): an odd count gets one all-zero filter slot
static_assert(KH >= SY, "patch slots assume overlapping or abutting windows in y");
constexpr int kRowsMax = (BJ - 2) / COW + 2;                       // output rows a BJ-pel tile can touch
constexpr int kSegFull = (COH - 1) * SY + KH;                      // slots of a whole image
constexpr int kSegMax0 = (COH - 1 + kRowsMax - 1) / COH + 1;       // images a tile can touch
constexpr int kSegMax = kSegMax0 < kRowsMax ? kSegMax0 : kRowsMax;
constexpr int kSlots = (kRowsMax - kSegMax) * SY + kSegMax * KH;   // slots per channel group (upper bound over tile positions)
constexpr int kCS = kSlots * kWp;                                  // input positions (16-byte chunks) per channel group
constexpr int kCSp = kCS + ((2 - kCS % 16) + 16) % 16;             // group pitch in chunks: 2 (mod 16) -> the CG chunks of a position land 8 banks apart
constexpr int kLoadN = kSlots * kWr * CG;                          // patch chunks loaded per K step
constexpr int kPE = (kLoadN + kNT - 1) / kNT;                      // ... per thread
constexpr int kIE = (kNP * BI + kNT - 1) / kNT;                    // filter chunks per thread per K step
constexpr int kEPitch = BI * 2 + 16;                               // epilogue tile [pel][oc] bf16, rows de-phased by 4 banks
constexpr int kAImg = ADIRECT ? 0 : kNP * BI;                      // chunks of the filter image
constexpr int kImgC = kAImg + CG * kCSp;                            // chunks of one operand image pair
constexpr int kOpB = 16 * kImgC * ((DBUF || ADIRECT) ? 2 : 1), kEpiB = OUT_F32 ? 0 : BJ * kEPitch;
constexpr int kSmem = kOpB > kEpiB ? kOpB : kEpiB;
constexpr int kOOB = (int)0x80000000;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(void const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ u32x4 bload4(rsrc_t r, int voff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0); }
// chunk offset of k-slot q = (g, ky, kx) inside the patch, relative to a lane's output position
constexpr int slot_off(int q) { return (q >= kNPr) ? 0 : (POOL ? q * kCSp : ((q / kTaps) * kCSp + ((q % kTaps) / KW) * kWp + (q % KW))); } // (the pad slot reads tap 0: its filter row is zero)
typedef short s16x8 __attribute__((ext_vector_type(8)));
// the B fragment of one output position and k-slot: the patch chunk at `at` -- or, POOL, the maximum over the window whose tap (0, 0) is at `at`
__device__ __forceinline__ bf16x8 frag_b(u32x4 const *Js, int at) {
#if POOL
  s16x8 m = __builtin_bit_cast(s16x8, Js[at]);
#pragma unroll
  for (int t = 1; t < kTaps; ++t) m = __builtin_elementwise_max(m, __builtin_bit_cast(s16x8, Js[at + (t / KW) * kWp + (t % KW)]));
  return __builtin_bit_cast(bf16x8, m);
#else
  return __builtin_bit_cast(bf16x8, Js[at]);
#endif
}
} // namespace

#ifdef BODAHIP_AS_MEMBER
constexpr int member_smem_bytes = kSmem, member_threads = WI * WJ * 64, member_minw = MINW;
__device__ __forceinline__ void KNAME(gemm_args_t const &p, grp_args_t const &, int const member_bid, char *const smem) {   // (one member signature for both kernel files)
#else
extern "C" __global__ __launch_bounds__(WI * WJ * 64, MINW) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) char smem[kSmem];
#endif
  u32x4 *const Is0 = reinterpret_cast<u32x4 *>(smem);              // A operand: [k-slot][out_chan] chunks
  u32x4 *const Js0 = Is0 + kAImg;                                 // input patch: [group][slot][padded column] chunks, group pitch kCSp
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave / WJ, wj = wave % WJ;

  int tile_i, tile_j; // XCD-aware workgroup -> tile map (as gemm_conv_f32.hip)
  {
    int const bid = (int)BODAHIP_BID / KSL, nb = p.tiles_i * p.tiles_j;
    int const q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int const group_sz = GROUP_I * p.tiles_j, gid = nid / group_sz, first_i = gid * GROUP_I;
    int const gsz = min(p.tiles_i - first_i, GROUP_I), in_g = nid - gid * group_sz;
    tile_i = first_i + in_g % gsz; tile_j = in_g / gsz;
  }
  int const i0 = tile_i * BI, j0 = tile_j * BJ;

  // ---- this thread's patch chunks: element el = (position, group within the K step), position-major (CG consecutive lanes read CG x 16 contiguous bytes)
  int const R0 = j0 / COW, img0 = R0 / COH, oy0 = R0 - img0 * COH;   // first output row of the tile (workgroup-uniform)
  int const seg0 = (COH - 1 - oy0) * SY + KH;                        // slots of the first image's part
  int const n_img = p.Nj / (COH * COW);
  int pgoff[kPE], pdst[kPE];
#pragma unroll
  for (int e = 0; e < kPE; ++e) {
    int const el = tid + e * kNT, pos = el / CG, g = el - pos * CG, s = pos / kWr, col = pos - s * kWr, ix = col - PX;
    int const s2 = s - seg0, im2 = s2 / kSegFull;
    int const img = (s < seg0) ? img0 : (img0 + 1 + im2);
    int const iy = (s < seg0) ? (oy0 * SY - PY + s) : (s2 - im2 * kSegFull - PY);
    bool const ok = (el < kLoadN) && (img < n_img) && ((unsigned)iy < (unsigned)CH) && ((unsigned)ix < (unsigned)CW);
    pgoff[e] = ok ? (((img * CH + iy) * CW + ix) * (CIN * 2) + g * 16) : kOOB;
    pdst[e] = g * kCSp + s * kWp + col;
  }
  int bj[kTJ]; // MFMA B operand: chunk index of this lane's output position (tap (0,0), group 0)
#pragma unroll
  for (int t = 0; t < kTJ; ++t) {
    int const jg = min(j0 + wj * (kTJ * 32) + t * 32 + (lane & 31), p.Nj - 1);
    int const R = jg / COW, ox = jg - R * COW, img = R / COH, oy = R - img * COH;
    int const slot = (img == img0) ? ((oy - oy0) * SY) : (seg0 + (img - img0 - 1) * kSegFull + oy * SY);
    bj[t] = slot * kWp + ox;
  }
  bool const hi = (lane >> 5) != 0; // lanes 32-63 supply the odd k-slot of an MFMA

  f32x16 acc[kTI][kTJ];
#pragma unroll
  for (int a = 0; a < kTI; ++a)
#pragma unroll
    for (int b = 0; b < kTJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);
  u32x4 rp[kPE], rf[kIE];
  auto load_step = [&](int kt) {
    if (ABLATE == 1) { for (int e = 0; e < kPE; ++e) rp[e] = u32x4{0u, 0u, 0u, 0u}; for (int e = 0; e < kIE; ++e) rf[e] = u32x4{0u, 0u, 0u, 0u}; return; }
    int const cg0 = kt * CG;
#pragma unroll
    for (int e = 0; e < kPE; ++e) {
      int const el = tid + e * kNT, g = el % CG;
      bool const ok = (pgoff[e] != kOOB) && (kNCG % CG == 0 || cg0 + g < kNCG);
      rp[e] = bload4(rJ, ok ? (pgoff[e] + cg0 * 16) : kOOB);
    }
#pragma unroll
    for (int e = 0; e < kIE; ++e) {
      int const el = tid + e * kNT, q = el / BI, i = el - q * BI;             // k-slot q = (g, tap) of this step, out_chan i0 + i
      int const g = q / kTaps, tap = q - g * kTaps, cg = cg0 + g;
      bool const ok = (el < kNP * BI) && (q < kNPr) && (cg < kNCG) && (i0 + i < p.Mi);
      rf[e] = bload4(rI, ok ? (int)((((unsigned)cg * kTaps + tap) * (unsigned)p.Mi + (unsigned)(i0 + i)) * 16u) : kOOB);
    }
  };
  auto store_step = [&](int buf) {
    if (ABLATE == 3) { asm volatile("" ::"v"(rp[0]), "v"(rf[0])); return; }
    u32x4 *const Is = Is0 + buf * kImgC, *const Js = Js0 + buf * kImgC;
#pragma unroll
    for (int e = 0; e < kPE; ++e) if (((e + 1) * kNT <= kLoadN) || (tid + e * kNT < kLoadN)) Js[pdst[e]] = rp[e];
#pragma unroll
    for (int e = 0; e < kIE; ++e) { int const el = tid + e * kNT; if (((e + 1) * kNT <= kNP * BI) || (el < kNP * BI)) Is[el] = rf[e]; }
  };

#if ADIRECT
  {
    constexpr int kN = kNP / 2, kPF = (PF < kN) ? PF : kN;          // MFMA k-iterations per K step; fragments in flight
    int aoff[kTI];                                                   // byte offset of this lane's chunk in k-slot row (hi) of a k-iteration
#pragma unroll
    for (int t = 0; t < kTI; ++t) { int const oc = i0 + wi * (kTI * 32) + t * 32 + (lane & 31); aoff[t] = (oc < p.Mi) ? ((hi ? p.Mi : 0) + oc) * 16 : kOOB; }
    int const rowB = p.Mi * 16;
    // k-slot q of K step kt is row kt * kNPr + q of the [k-slot][out_chan] filter matrix; rows past its end (the ragged last step, the step after the last) are past
    // the buffer and read as zero.  (The offset goes through the VGPR: an SGPR offset is not range-checked.)
    auto load_a = [&](int kt, int s, u32x4 *dst) {
      int const ro = (kt * kNPr + 2 * s) * rowB;
#pragma unroll
      for (int t = 0; t < kTI; ++t) {
        int const vo = ((kNPr & 1) && s == kN - 1 && hi) ? kOOB : aoff[t];   // (the zero pad slot of an odd step)
        dst[t] = (ABLATE == 1) ? u32x4{0u, 0u, 0u, 0u} : bload4(rI, vo + ro);
      }
    };
    auto load_patch = [&](int kt) {
      int const cg0 = kt * CG;
#pragma unroll
      for (int e = 0; e < kPE; ++e) {
        int const el = tid + e * kNT, g = el % CG;
        bool const ok = (pgoff[e] != kOOB) && (kNCG % CG == 0 || cg0 + g < kNCG);
        rp[e] = (ABLATE == 1) ? u32x4{0u, 0u, 0u, 0u} : bload4(rJ, ok ? (pgoff[e] + cg0 * 16) : kOOB);
      }
    };
    auto store_patch = [&](int buf) {
      u32x4 *const Js = Js0 + buf * kImgC;
#pragma unroll
      for (int e = 0; e < kPE; ++e) if (((e + 1) * kNT <= kLoadN) || (tid + e * kNT < kLoadN)) Js[pdst[e]] = rp[e];
    };
    u32x4 cur[kN][kTI], nxt[kPF][kTI];
#if KSL > 1
    int const kt0 = ((int)BODAHIP_BID % KSL) * p.kt_per, kt1 = min(kNKT, kt0 + p.kt_per);   // this slice's K steps
#else
    constexpr int kt0 = 0, kt1 = (ABLATE == 4) ? 0 : kNKT;
#endif
    load_patch(kt0);
#pragma unroll
    for (int s = 0; s < kPF; ++s) load_a(kt0, s, cur[s]);
    store_patch(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      u32x4 const *const Js = Js0 + ((kt - kt0) & 1) * kImgC;
      if (kt + 1 < kt1) load_patch(kt + 1);
      constexpr int kBD = (BPF < kN) ? BPF : kN, kBR = kBD + 1;   // patch fragments are read kBD k-iterations ahead, through a ring of kBR sets
      bf16x8 b[kBR][kTJ];
#pragma unroll
      for (int d = 0; d < kBD; ++d)
#pragma unroll
        for (int t = 0; t < kTJ; ++t) b[d][t] = frag_b(Js, bj[t] + (hi ? slot_off(2 * d + 1) : slot_off(2 * d)));
#pragma unroll
      for (int s = 0; s < kN; ++s) {
        if (s + kPF < kN) load_a(kt, s + kPF, cur[s + kPF]); else load_a(kt + 1, s + kPF - kN, nxt[s + kPF - kN]);
        if (s + kBD < kN) {
          int const jo = hi ? slot_off(2 * (s + kBD) + 1) : slot_off(2 * (s + kBD));
#pragma unroll
          for (int t = 0; t < kTJ; ++t) b[(s + kBD) % kBR][t] = frag_b(Js, bj[t] + jo);
        }
        if (ABLATE != 2) {
#pragma unroll
          for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
            for (int tb = 0; tb < kTJ; ++tb)
              acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[s][ta]), b[s % kBR][tb], acc[ta][tb], 0, 0, 0);
        } else asm volatile("" ::"v"(cur[s][0]), "v"(b[s % kBR][0]));
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt + 1 < kt1) store_patch(((kt - kt0) & 1) ^ 1);
#pragma unroll
      for (int s = 0; s < kPF; ++s)
#pragma unroll
        for (int t = 0; t < kTI; ++t) cur[s][t] = nxt[s][t];
      __syncthreads();
    }
  }
#else
  load_step(0);
#if DBUF
  store_step(0);
  __syncthreads();
#endif
  for (int kt = 0; kt < ((ABLATE == 4) ? 0 : kNKT); ++kt) {
#if DBUF
    int const cur = kt & 1;
#else
    int const cur = 0;
    store_step(0);
    __syncthreads();
#endif
    if (kt + 1 < kNKT) load_step(kt + 1);      // next step's loads fly under this step's MFMAs
    u32x4 const *const Is = Is0 + cur * kImgC, *const Js = Js0 + cur * kImgC;
    u32x4 const *const Ic = Is + wi * (kTI * 32) + (lane & 31);
#pragma unroll
    for (int s = 0; s < ((ABLATE == 2) ? 0 : kNP / 2); ++s) {
      int const jo = hi ? slot_off(2 * s + 1) : slot_off(2 * s);
      bf16x8 a[kTI], b[kTJ];
#pragma unroll
      for (int t = 0; t < kTI; ++t) a[t] = __builtin_bit_cast(bf16x8, Ic[(2 * s + (hi ? 1 : 0)) * BI + t * 32]);
#pragma unroll
      for (int t = 0; t < kTJ; ++t) b[t] = __builtin_bit_cast(bf16x8, Js[bj[t] + jo]);
#pragma unroll
      for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
        for (int tb = 0; tb < kTJ; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
#if DBUF
    if (kt + 1 < kNKT) store_step(cur ^ 1);    // (the other image: its last readers passed the previous barrier)
#endif
    __syncthreads();                           // every wave is done reading before the next step overwrites the images (DBUF: and the next image is complete)
  }
#endif // ADIRECT

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
  // ---- epilogue (as conv_nhwc_bf16.hip).  C/D layout of the 32x32 MFMA family: column j = lane & 31, rows i = 8*g + 4*(lane >> 5) + e for register 4*g + e
  int const h = lane >> 5;
  rsrc_t const rD = make_rsrc(p.D, p.D_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
  int const o_ctot = p.out_ctot, o_coff = p.out_coff;
  f32x4 bv[kTI][4];
#pragma unroll
  for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int const oc = i0 + wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h;
      if (oc + 4 <= p.Mi) bv[ta][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, oc * 4, 0, 0));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[ta][g][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, (oc + e < p.Mi) ? (oc + e) * 4 : kOOB, 0, 0));
      }
    }
#if OUT_F32
#pragma unroll
  for (int tb = 0; tb < kTJ; ++tb) {
    int const pel = j0 + wj * (kTJ * 32) + tb * 32 + (lane & 31);
    unsigned const rowoff = ((unsigned)pel * (unsigned)o_ctot + (unsigned)o_coff) * 4u;
#pragma unroll
    for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int const oc = i0 + wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h;
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = acc[ta][tb][4 * g + e] + bv[ta][g][e]; if (RELU) x[e] = (x[e] > 0.f) ? x[e] : 0.f; }
        if (pel < p.Nj) {
          if (oc + 4 <= p.Mi && ((o_ctot | o_coff) & 3) == 0) {
            f32x4 v; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rD, (int)(rowoff + (unsigned)oc * 4u), 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (oc + e < p.Mi) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x[e]), rD, (int)(rowoff + (unsigned)(oc + e) * 4u), 0, 0);
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
          for (int e = 0; e < 4; ++e) { float x = acc[ta][tb][4 * g + e] + bv[ta][g][e]; if (RELU) x = (x > 0.f) ? x : 0.f; v[e] = (__bf16)x; }
          *reinterpret_cast<bf16x4 *>(E + prow * kEPitch + (wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h) * 2) = v;
        }
    }
    __syncthreads();
    constexpr int kRowChunks = BI / 8, kChunks = BJ * kRowChunks;
    bool const vec_ok = ((o_ctot | o_coff) & 7) == 0;   // 16-byte aligned rows and slices
#pragma unroll
    for (int n = 0; n < (kChunks + kNT - 1) / kNT; ++n) {
      int const idx = tid + n * kNT;
      if (kChunks % kNT != 0 && idx >= kChunks) break;
      int const prow = idx / kRowChunks, cc = idx - prow * kRowChunks;
      int const pel = j0 + prow, oc = i0 + cc * 8;
      if (pel >= p.Nj || oc >= p.Mi) continue;
      u32x4 const v = *reinterpret_cast<u32x4 const *>(E + prow * kEPitch + cc * 16);
      unsigned const off = ((unsigned)pel * (unsigned)o_ctot + (unsigned)o_coff + (unsigned)oc) * 2u;
      if (vec_ok && oc + 8 <= p.Mi) __builtin_amdgcn_raw_buffer_store_b128(v, rD, (int)off, 0, 0);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (oc + e < p.Mi) __builtin_amdgcn_raw_buffer_store_b16((short)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu), rD, (int)(off + 2u * e), 0, 0);
      }
    }
  }
#endif
}

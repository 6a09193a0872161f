// conv_patch_bf16.hip -- bf16-operand / fp32-accumulate convolution from an LDS input patch, for KH x KW kernels with stride 1 in x
// (gfx950, hiprtc-specialised; BASELINE config 5).  Same contract as bodahip_conv_bf16 (fp32 NCHW / OIHW tensors in HBM, operands
// rounded to bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16, bias + ReLU epilogue); what changes is the operand path.
//
// The table-gather bf16 kernel (gemm_conv_bf16.hip) loads one dword per im2col element, and with 16x less matrix time per k than
// fp32 that instruction stream is all it does (3x3 layers: 115-240 TF/s).  Here the contraction index is ordered
//       k = (in_chan group of 8, ky, kx, in_chan within the group)
// -- any order is legal as long as both operands use it -- so that the 8 consecutive k an MFMA lane supplies are 8 CHANNELS of one
// tap.  The LDS then holds, per group of 8 channels, the zero-padded input rows the tile's output positions touch, channel-innermost
// (one 16-byte chunk = 8 bf16 channels of one input position): lane l's B fragment for tap (ky,kx) is ONE ds_read_b128 at
//       patch[group][ (slot(j) + ky) * Wp + ox(j) + kx ]            (slot geometry exactly as in gemm_conv_f32.hip's patch mode)
// and staging a group costs one coalesced dword load per input element instead of KH*KW gathered ones.  The filters are re-laid-out
// once per call by bodahip_filt_bf16 into F'[group][tap][out_chan][8] bf16, so the A operand is staged with plain 16-byte loads.
// One K step = CG channel groups x all taps = CG*KH*KW MFMA k-slots of 8 (two slots per 32x32x16 MFMA: lanes 0-31 / 32-63).
//
// Numerics: as gemm_conv_bf16.hip (bf16 products are exact in fp32; the summation order differs from any CPU loop): parity is
// stated against the oracle fed bf16-rounded operands (tests: mrd < 5e-4) -- unpinned by construction, the reference has no bf16.
//
// -D parameters: KNAME BI BJ WI WJ MINW CG KH KW SY PY PX CH CW COH COW RELU        (SX == 1, in_chan % 8 == 0)

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef GROUP_I
#define GROUP_I 8
#endif
#ifndef RELU
#define RELU 0
#endif
#ifndef SWAPST
#define SWAPST ((BJ / (WJ * 32)) % 2 == 0)
#endif

struct gemm_args_t { // identical to gemm_conv_f32.hip (one host-side struct); I = F' (re-laid-out bf16 filters), J = in
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

#ifdef S2D_ONLY
// Space-to-depth front end for strided convolutions on few input channels (the conv1 layers: 11x11/4 or 7x7/2 on 3 channels), which
// the patch kernel cannot take directly (stride 1 in x, channels in groups of 8): an s x s block of input pixels becomes s*s channels,
//   in2[b][c*s*s + dy*s + dx][Y][X] = in[b][c][s*Y + dy - Pry][s*X + dx - Prx]       (zero outside; Pr = pad rounded up to a multiple of s)
//   f2 [oc][c*s*s + dy*s + dx][a][b] = filts[oc][c][s*a + dy - (Pry - PY)][s*b + dx - (Prx - PX)]   (zero outside)
// and the layer is the stride-1, unpadded ceil((K + Pr - P)/s)-square convolution of in2 with f2 -- same outputs, term for term.
struct s2d_args_t {
  float const *src; float *dst;
  int B, C, H, W, OC, KH, KW;       // source tensor geometry (filters: OC, C, KH, KW)
  int S, C2, H2, W2;                // block size; channels (padded to a multiple of 8), rows, cols of the result (filters: rows = KHb, cols = KWb)
  int oy, ox;                       // input: Pr (rounded-up pad) per axis; filters: Pr - P per axis
  int filt;                         // 0: input tensor, 1: filters
  int mode, c2_lo;                  // mode 0: one thread per result element with channel >= c2_lo (filters; the zero pad channels of the input)
};                                  // mode 1: one thread per (b, c, Y, dy, X): S consecutive source pixels -> S channel planes (reads and writes coalesced)
extern "C" __global__ __launch_bounds__(256) void KNAME(s2d_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  int const SH = p.filt ? p.KH : p.H, SW = p.filt ? p.KW : p.W, ss = p.S * p.S;
  if (p.mode == 1) {
    long const n = (long)p.B * p.C * p.H2 * p.S * p.W2;
    if (idx >= n) return;
    int const X = (int)(idx % p.W2); long r = idx / p.W2;
    int const dy = (int)(r % p.S); r /= p.S;
    int const Y = (int)(r % p.H2); r /= p.H2;
    int const c = (int)(r % p.C), b = (int)(r / p.C);
    int const y = p.S * Y + dy - p.oy, x0 = p.S * X - p.ox;
    bool const yok = (unsigned)y < (unsigned)SH;
    float const *row = p.src + (((long)b * p.C + c) * SH + (yok ? y : 0)) * SW;
    float *d = p.dst + ((((long)b * p.C2 + (long)c * ss + dy * p.S) * p.H2 + Y) * p.W2 + X);
    long const plane = (long)p.H2 * p.W2;
    for (int dx = 0; dx < p.S; ++dx) { int const x = x0 + dx; d[dx * plane] = (yok && (unsigned)x < (unsigned)SW) ? row[x] : 0.f; }
    return;
  }
  int const nc = p.C2 - p.c2_lo;
  long const n = (long)(p.filt ? p.OC : p.B) * nc * p.H2 * p.W2;
  if (idx >= n) return;
  int const X = (int)(idx % p.W2), Y = (int)((idx / p.W2) % p.H2), c2 = p.c2_lo + (int)((idx / ((long)p.W2 * p.H2)) % nc), b = (int)(idx / ((long)p.W2 * p.H2 * nc));
  int const c = c2 / ss, dy = (c2 / p.S) % p.S, dx = c2 % p.S;
  int const y = p.S * Y + dy - p.oy, x = p.S * X + dx - p.ox;
  float v = 0.f;
  if (c < p.C && (unsigned)y < (unsigned)SH && (unsigned)x < (unsigned)SW) v = p.src[(((long)b * p.C + c) * SH + y) * SW + x];
  p.dst[(((long)b * p.C2 + c2) * p.H2 + Y) * p.W2 + X] = v;
}
#elif defined(FILT_ONLY)
// F'[(cg*taps + tap)*OC + oc] (16-byte chunks) = bf16( filts[oc][8*cg + e][tap] ), e = 0..7; channels past C are zero.
// args: I = filts (fp32 OIHW), D = F' (as float*), Mi = OC, C = in_chans, K = taps
extern "C" __global__ __launch_bounds__(256) void KNAME(gemm_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  int const ncg = (p.C + 7) / 8;
  if (idx >= (long)ncg * p.K * p.Mi) return;
  int const oc = (int)(idx % p.Mi), tap = (int)((idx / p.Mi) % p.K), cg = (int)(idx / ((long)p.Mi * p.K));
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) { int const c = 8 * cg + e; v[e] = (__bf16)((c < p.C) ? p.I[((long)oc * p.C + c) * p.K + tap] : 0.f); }
  reinterpret_cast<bf16x8 *>(p.D)[idx] = v;
}
#else

namespace {
constexpr int kNT = WI * WJ * 64;
constexpr int kTI = BI / (WI * 32);
constexpr int kTJ = BJ / (WJ * 32);
static_assert(BI % (WI * 32) == 0 && BJ % (WJ * 32) == 0, "tile must be a multiple of the MFMA tile per wave");
constexpr int kTaps = KH * KW, kWp = CW + 2 * PX;
constexpr int kNPr = CG * kTaps;                                   // k-slots (of 8 channels) per K step ...
constexpr int kNP = kNPr + (kNPr & 1);                             // ... padded to whole MFMAs (two slots each): an odd count gets one all-zero filter slot
static_assert(KH >= SY, "patch slots assume overlapping or abutting windows in y");
constexpr int kRowsMax = (BJ - 2) / COW + 2;                       // output rows a BJ-pel tile can touch
constexpr int kSegFull = (COH - 1) * SY + KH;                      // slots of a whole image
constexpr int kSegMax0 = (COH - 1 + kRowsMax - 1) / COH + 1;       // images a tile can touch
constexpr int kSegMax = kSegMax0 < kRowsMax ? kSegMax0 : kRowsMax;
constexpr int kSlots = (kRowsMax - kSegMax) * SY + kSegMax * KH;   // slots per channel group (upper bound over tile positions)
constexpr int kCS = kSlots * kWp;                                  // input positions (16-byte chunks) per channel group
constexpr int kPE = (kCS * CG + kNT - 1) / kNT;                    // patch chunks per thread per K step
constexpr int kIE = (kNP * BI + kNT - 1) / kNT;                    // filter chunks per thread per K step
constexpr int kOOB = (int)0x80000000;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(void const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float bload1(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0)); }
// chunk offset of k-slot q = (g, ky, kx) inside the patch, relative to a lane's output position
constexpr int slot_off(int q) { return (q >= kNPr) ? 0 : ((q / kTaps) * kCS + ((q % kTaps) / KW) * kWp + (q % KW)); } // (the pad slot reads tap 0: its filter row is zero)
} // namespace

extern "C" __global__ __launch_bounds__(WI * WJ * 64, MINW) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) bf16x8 Is[kNP * BI];     // A operand: [k-slot][out_chan] chunks
  __shared__ __attribute__((aligned(16))) bf16x8 Js[CG * kCS];     // input patch: [group][slot][padded column] chunks
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave / WJ, wj = wave % WJ;

  int tile_i, tile_j; // XCD-aware workgroup -> tile map (as gemm_conv_f32.hip)
  {
    int const bid = blockIdx.x, nb = p.tiles_i * p.tiles_j;
    int const q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int const group_sz = GROUP_I * p.tiles_j, gid = nid / group_sz, first_i = gid * GROUP_I;
    int const gsz = min(p.tiles_i - first_i, GROUP_I), in_g = nid - gid * group_sz;
    tile_i = first_i + in_g % gsz; tile_j = in_g / gsz;
  }
  int const i0 = tile_i * BI, j0 = tile_j * BJ;

  // ---- this thread's patch chunks (fixed input positions; the channel group advances per K step) and filter chunks
  int const R0 = j0 / COW, img0 = R0 / COH, oy0 = R0 - img0 * COH;   // first output row of the tile (workgroup-uniform)
  int const seg0 = (COH - 1 - oy0) * SY + KH;                        // slots of the first image's part
  int const n_img = p.Nj / (COH * COW);
  int const plane4 = CH * CW * 4;
  int pgoff[kPE], pgrp[kPE];
#pragma unroll
  for (int e = 0; e < kPE; ++e) {
    int const el = tid + e * kNT, g = el / kCS, pos = el - g * kCS, s = pos / kWp, ix = pos - s * kWp - PX;
    int const s2 = s - seg0, im2 = s2 / kSegFull;
    int const img = (s < seg0) ? img0 : (img0 + 1 + im2);
    int const iy = (s < seg0) ? (oy0 * SY - PY + s) : (s2 - im2 * kSegFull - PY);
    bool const ok = (el < kCS * CG) && (img < n_img) && ((unsigned)iy < (unsigned)CH) && ((unsigned)ix < (unsigned)CW);
    pgoff[e] = ok ? (((img * p.C * CH + iy) * CW + ix) * 4) : kOOB;
    pgrp[e] = g;
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
  int const ncg = p.C / 8;                       // channel groups (host guarantees C % 8 == 0)
  int const nkt = (ncg + CG - 1) / CG;
  float rp[kPE][8]; f32x4 rf[kIE];
  auto load_step = [&](int kt) {
    int const cg0 = kt * CG;
#pragma unroll
    for (int e = 0; e < kPE; ++e) {
      int const cg = cg0 + pgrp[e];
      int const off = (cg < ncg && pgoff[e] != kOOB) ? (pgoff[e] + cg * 8 * plane4) : kOOB;
#pragma unroll
      for (int c = 0; c < 8; ++c) rp[e][c] = bload1(rJ, off, c * plane4);
    }
#pragma unroll
    for (int e = 0; e < kIE; ++e) {
      int const el = tid + e * kNT, q = el / BI, i = el - q * BI;             // k-slot q = (g, tap) of this step, out_chan i0 + i
      int const g = q / kTaps, tap = q - g * kTaps, cg = cg0 + g;
      bool const ok = (el < kNP * BI) && (q < kNPr) && (cg < ncg) && (i0 + i < p.Mi);
      rf[e] = bload4(rI, ok ? (int)((((unsigned)cg * kTaps + tap) * (unsigned)p.Mi + (unsigned)(i0 + i)) * 16u) : kOOB);
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int e = 0; e < kPE; ++e) {
      int const el = tid + e * kNT;
      if (el < kCS * CG) { bf16x8 v;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (__bf16)rp[e][c];
        Js[el] = v; }
    }
#pragma unroll
    for (int e = 0; e < kIE; ++e) {
      int const el = tid + e * kNT;
      if (el < kNP * BI) Is[el] = __builtin_bit_cast(bf16x8, rf[e]);
    }
  };

  load_step(0);
  for (int kt = 0; kt < nkt; ++kt) {
    store_step();
    __syncthreads();
    if (kt + 1 < nkt) load_step(kt + 1);      // next step's loads fly under this step's MFMAs
    bf16x8 const *const Ic = Is + wi * (kTI * 32) + (lane & 31);
#pragma unroll
    for (int s = 0; s < kNP / 2; ++s) {
      int const jo = hi ? slot_off(2 * s + 1) : slot_off(2 * s);
      bf16x8 a[kTI], b[kTJ];
#pragma unroll
      for (int t = 0; t < kTI; ++t) a[t] = Ic[(2 * s + (hi ? 1 : 0)) * BI + t * 32];
#pragma unroll
      for (int t = 0; t < kTJ; ++t) b[t] = Js[bj[t] + jo];
#pragma unroll
      for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
        for (int tb = 0; tb < kTJ; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __syncthreads();                           // every wave is done reading before the next step overwrites the images
  }

  // ---- epilogue (as gemm_conv_bf16.hip): column j = lane&31, row i = (r&3) + 8*(r>>2) + 4*(lane>>5)
  {
    rsrc_t const rD = make_rsrc(p.D, p.D_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
    unsigned const S4 = (unsigned)(COH * COW) * 4u;
    int const ib = i0 + wi * (kTI * 32) + 4 * (lane >> 5);
    auto rowc = [](int ta, int r) { return ta * 32 + (r & 3) + 8 * (r >> 2); };
    unsigned const ipart = (unsigned)ib * S4;
    float bv[kTI][16];
#pragma unroll
    for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[ta][r] = bload1(rB, (ib + rowc(ta, r)) * 4, 0);
#if SWAPST
    // paired 256-byte row stores (see gemm_conv_f32.hip): bias / ReLU in the MFMA layout, then v_permlane32_swap so that one register
    // holds 64 consecutive pels of row i and the other of row i+4
    auto store_all = [&](bool const edge) {
      int const ibl = i0 + wi * (kTI * 32);
#pragma unroll
      for (int tp = 0; tp < kTJ / 2; ++tp) {
        int const jg = j0 + wj * (kTJ * 32) + tp * 64 + lane;
        int const OHW = COH * COW;
        int const img = jg / OHW, pel = jg - img * OHW;
        unsigned const jpart = (jg < p.Nj) ? ((((unsigned)img * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel) * 4u + (unsigned)ibl * S4) : 0x80000000u;
#pragma unroll
        for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float va = acc[ta][2 * tp][r], vb = acc[ta][2 * tp + 1][r];
            va = va + bv[ta][r]; vb = vb + bv[ta][r];
#if RELU
            va = (va > 0.f) ? va : 0.f; vb = (vb > 0.f) ? vb : 0.f;
#endif
            auto const sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, va), __builtin_bit_cast(unsigned, vb), false, false);
            int const rc = rowc(ta, r);
            if (!(edge && (ibl + rc >= p.Mi))) __builtin_amdgcn_raw_buffer_store_b32((int)sw[0], rD, (int)jpart, (int)((unsigned)rc * S4), 0);
            if (!(edge && (ibl + rc + 4 >= p.Mi))) __builtin_amdgcn_raw_buffer_store_b32((int)sw[1], rD, (int)jpart, (int)((unsigned)(rc + 4) * S4), 0);
          }
      }
    };
#else
    auto store_all = [&](bool const edge) {
#pragma unroll
      for (int tb = 0; tb < kTJ; ++tb) {
        int const jg = j0 + wj * (kTJ * 32) + tb * 32 + (lane & 31);
        if (jg >= p.Nj) continue;
        int const OHW = COH * COW;
        int const img = jg / OHW, pel = jg - img * OHW;
        unsigned const jpart = (((unsigned)img * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel) * 4u;
#pragma unroll
        for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (edge && (ib + rowc(ta, r) >= p.Mi)) continue;
            float v = acc[ta][tb][r] + bv[ta][r];
#if RELU
            v = (v > 0.f) ? v : 0.f;
#endif
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rD, (int)(jpart + ipart), (int)((unsigned)rowc(ta, r) * S4), 0);
          }
      }
    };
#endif
    if (i0 + BI <= p.Mi) store_all(false); else store_all(true); // workgroup-uniform
  }
}
#endif // FILT_ONLY

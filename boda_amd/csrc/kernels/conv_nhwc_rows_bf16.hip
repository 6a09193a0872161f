// conv_nhwc_rows_bf16.hip -- ROLLING-ROWS form of the channels-last bf16 convolution for gfx950: short-K, narrow-out_chan layers whose time is their OUTPUT (the 7x7/2
// stems of GoogLeNet / ResNet-50 after space-to-depth: K = 2 x 16 taps x 8 channels, 64 out_chans, 112 x 112 outputs -- 103 MB written at 64 images against 27 MB read).
// Same contract and operand layouts as bodahip_conv_nhwc_patch_bf16 (kernels/conv_nhwc_patch_bf16.hip: in img:y:x:chan bf16, F'[g][ky][kx][oc][8], biases float, bias +
// ReLU, optional channel slice of a wider output), same MFMA chain per output -- k-iterations in (K step, k-slot pair) order, the slots of a step ordered (group, ky, kx) --
// hence the same bits.  What differs is who does what:
//   * a workgroup (WJ waves) owns a run of consecutive output ROWS of one image and walks down it, TR rows (TR x COW <= BJ pels) per tile; every wave multiplies ALL
//     out_chans (two 32-row MFMA blocks) by its own BJ / WJ pels;
//   * the FILTERS stay in registers for the whole walk: kNIT k-iterations x 2 blocks x 16 bytes per lane (128 VGPRs for the stems) -- no filter traffic after the prologue,
//     where the patch kernel's 3136 two-step workgroups each pulled the whole 32-KB filter set through their CU's L1 again;
//   * the input rows of tile t + 1 are loaded (registers) under the MFMAs of tile t and stored into the other LDS patch image;
//   * the outputs of a tile land in an LDS ring of output rows ([row % kNR][col][oc], bf16, bias and ReLU applied) and leave from there as whole 128-byte lines --
//   * -- or DO NOT LEAVE AT ALL (PKH > 0): the max pooling that follows the convolution (test/rtc/pool.cucl in the reference: a function of its own, src/rtc_fwd.cc:545-549)
//     reads its windows from that ring as soon as their last row is there, and the across-channel LRN behind it (LRN_N > 0; the reference's lrn function) is applied to the
//     pooled row in LDS: the convolution's output tensor -- four times the pooled one -- is never written or read.  Pooled values are maxima of the bf16-rounded ReLU outputs
//     (taken on the bit patterns as signed 16-bit integers: non-negative floats order like their bits), rounded where the pooling kernel rounds; the LRN is the LRN kernels'
//     own expression on them (boda_amd/nhwc.py POOL_LRN_SPEC_SRC): bit-identical to the three launches run apart.
// Pooled rows [y0, y1) of a workgroup need output rows first(y0) .. last(y1 - 1); a row shared with the neighbouring workgroup (overlapping windows) is computed by both.
//
// The LRN variant is compiled with -ffast-math like the generated functions it replaces (src/rtc_func_gen / be=hip compile them that way: the reference's --use_fast_math),
// so that the compiler takes the same liberties with the same expression; alpha / beta / k are run-time arguments as they are there.
//
// -D parameters: KNAME CIN CG KH KW PY PX CH CW COH COW RELU WJ TR [PKH PKW PSY PSX PPY PPX POH POW] [LRN_N]   (SY == SX == 1, out_chan <= 64)

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifndef RELU
#define RELU 0
#endif
#ifndef PKH
#define PKH 0      // > 0: max pooling PKH x PKW / stride PSY x PSX / padding PPY x PPX behind the convolution; the output is the POOLED tensor img:POH:POW:chan
#define PKW 0
#define PSY 1
#define PSX 1
#define PPY 0
#define PPX 0
#define POH 0
#define POW 0
#endif
#ifndef LRN_N
#define LRN_N 0    // > 0 (odd): across-channel LRN of that local size behind the pooling
#endif

struct rows_args_t {       // (host: native_kernels.cc conv_nhwc_rows)
  void const *filts; void const *in; void *out; float const *bias;
  int n_img, oc;           // images; out_chans (<= 64)
  unsigned filts_bytes, in_bytes, out_bytes;
  int out_ctot, out_coff;  // channels of the output tensor's rows, first channel this convolution writes
  int n_chunks, rows_per_chunk;   // workgroups per image; output rows (pooled rows if PKH) each of them owns
  float lrn_alpha, lrn_beta, lrn_k;
};

namespace {
constexpr int BI = 64, BJ = 256;
constexpr int kNT = WJ * 64, kTI = 2, kTJ = BJ / (WJ * 32);
static_assert(BJ % (WJ * 32) == 0 && kTJ >= 1, "wave layout");
static_assert(CIN % 8 == 0, "channels-last tensors carry whole 16-byte chunks per position");
static_assert(TR >= 1 && TR * COW <= BJ, "a tile is TR whole output rows");
static_assert(PKH == 0 || RELU, "the fused pooling orders bit patterns: non-negative values only");
static_assert(LRN_N == 0 || (PKH > 0 && (LRN_N & 1)), "the LRN follows the pooling");
constexpr int kTaps = KH * KW, kNCG = CIN / 8, kNKT = (kNCG + CG - 1) / CG;
constexpr int kNPr = CG * kTaps, kNP = kNPr + (kNPr & 1), kN = kNP / 2, kNIT = kNKT * kN;      // k-slots per K step (padded to whole MFMAs), k-iterations per step / in all
static_assert(kNIT * kTI * 4 <= 160, "the filters must fit the registers");
constexpr int kWr = CW + 2 * PX;
constexpr int wpitch() { for (int p = kWr; p < kWr + 16; ++p) if ((p - COW) % 16 == 0) return p; return kWr; }   // (as the patch kernel: conflict-free fragment reads across row ends)
constexpr int kWp = wpitch();
constexpr int kSlots = TR + KH - 1;                               // input rows of a tile
constexpr int kCS = kSlots * kWp, kCSp = kCS + ((2 - kCS % 16) + 16) % 16;   // chunks per channel group; group pitch 2 (mod 16)
constexpr int kPatchC = kNCG * kCSp;                              // chunks of one patch image (ALL channel groups: a tile's whole K)
constexpr int kLoadN = kSlots * kWr * kNCG, kPE = (kLoadN + kNT - 1) / kNT;
constexpr int kNR = (PKH > 0) ? TR + PKH - 1 : TR;                // rows of the output ring: what the next pooling window may still need + the tile
constexpr int kEP = BI * 2 + 16;                                  // bytes per position of the ring (rows de-phased by 4 banks, as the patch kernel's epilogue tile)
constexpr int kERow = COW * kEP;
constexpr int kC8 = BI / 8;
constexpr int kPP = (kC8 + 2) * 16;                               // bytes per position of the pooled row: 10 chunks -- eight positions x two chunks land in 16 different bank groups
constexpr int kPatchB = 2 * 16 * kPatchC, kRingB = kNR * kERow, kPoolB = (PKH > 0) ? POW * kPP : 0, kBiasB = BI * 4;
constexpr int kSmem = kPatchB + kRingB + kPoolB + kBiasB;
static_assert(kSmem <= 160 * 1024, "LDS");
constexpr int kOOB = (int)0x80000000;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(void const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ u32x4 bload4(rsrc_t r, int voff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0); }
// chunk offset of k-slot q of K step kt inside a patch image, relative to a lane's output position: group (kt CG + g), tap (ky, kx)
constexpr int slot_off(int kt, int q) {
  return (q >= kNPr || kt * CG + q / kTaps >= kNCG) ? 0 : ((kt * CG + q / kTaps) * kCSp + ((q % kTaps) / KW) * kWp + (q % KW));   // (pad slots read tap 0: their filter rows are zero)
}
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
} // namespace

extern "C" __global__ __launch_bounds__(WJ * 64, 1) void KNAME(rows_args_t const p) {
  __shared__ __attribute__((aligned(16))) char smem[kSmem];
  u32x4 *const Js0 = reinterpret_cast<u32x4 *>(smem);
  char *const E = smem + kPatchB;
  char *const P = E + kRingB;
  float *const Bs = reinterpret_cast<float *>(P + kPoolB);
  int const tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
  int const wj = __builtin_amdgcn_readfirstlane(tid >> 6);

  int img, chunk;     // XCD-aware workgroup -> (image, chunk): the chunks of an image (shared halo rows) run on one XCD
  {
    int const bid = (int)blockIdx.x, nb = p.n_img * p.n_chunks;
    int const q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    img = nid / p.n_chunks; chunk = nid - img * p.n_chunks;
  }
  // rows of this workgroup
#if PKH > 0
  int const y0 = chunk * p.rows_per_chunk, y1 = min(POH, y0 + p.rows_per_chunk);
  if (y0 >= y1) return;
  int const r_lo = max(0, y0 * PSY - PPY), r_hi = min(COH - 1, (y1 - 1) * PSY - PPY + PKH - 1);
  int y_next = y0;
#else
  int const r_lo = chunk * p.rows_per_chunk, r_hi = min(COH, r_lo + p.rows_per_chunk) - 1;
  if (r_lo > r_hi) return;
#endif
  int const n_tiles = (r_hi - r_lo + TR) / TR;

  rsrc_t const rI = make_rsrc(p.filts, p.filts_bytes), rJ = make_rsrc(p.in, p.in_bytes), rD = make_rsrc(p.out, p.out_bytes), rB = make_rsrc(p.bias, (unsigned)p.oc * 4u);
  if (tid < BI) Bs[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, (tid < p.oc) ? tid * 4 : kOOB, 0, 0));

  // ---- the filters: this lane's MFMA A fragments of every k-iteration (k-slot row kt * kNPr + 2 s + h of the [k-slot][out_chan] matrix; rows past its end, the zero
  //      pad slot of an odd step and out_chans past the layer's read as zero)
  u32x4 a[kNIT][kTI];
#pragma unroll
  for (int kt = 0; kt < kNKT; ++kt)
#pragma unroll
    for (int s = 0; s < kN; ++s)
#pragma unroll
      for (int t = 0; t < kTI; ++t) {
        int const oc = t * 32 + (lane & 31), row = kt * kNPr + 2 * s + h;
        bool const ok = (oc < p.oc) && !((kNPr & 1) && s == kN - 1 && h);
        a[kt * kN + s][t] = bload4(rI, ok ? (row * p.oc + oc) * 16 : kOOB);
      }

  // ---- this thread's patch chunks: element el = (position, channel group), position-major (kNCG consecutive lanes read kNCG x 16 contiguous bytes); slot s = input row
  //      r0 - PY + s of the tile that starts at output row r0
  int pgoff[kPE], pdst[kPE], prow[kPE];
#pragma unroll
  for (int e = 0; e < kPE; ++e) {
    int const el = tid + e * kNT, pos = el / kNCG, g = el - pos * kNCG, s = pos / kWr, col = pos - s * kWr, ix = col - PX;
    bool const ok = (el < kLoadN) && ((unsigned)ix < (unsigned)CW);
    prow[e] = ok ? (s - PY) : -0x10000;                                          // (+ r0 = the input row; far negative: never in range)
    pgoff[e] = ((img * CH + (s - PY)) * CW + ix) * (CIN * 2) + g * 16;
    pdst[e] = g * kCSp + s * kWp + col;
  }
  u32x4 rp[kPE];
  auto load_patch = [&](int r0) {
#pragma unroll
    for (int e = 0; e < kPE; ++e) rp[e] = bload4(rJ, ((unsigned)(prow[e] + r0) < (unsigned)CH) ? (pgoff[e] + r0 * (CW * CIN * 2)) : kOOB);
  };
  auto store_patch = [&](int buf) {
    u32x4 *const Js = Js0 + buf * kPatchC;
#pragma unroll
    for (int e = 0; e < kPE; ++e) if (((e + 1) * kNT <= kLoadN) || (tid + e * kNT < kLoadN)) Js[pdst[e]] = rp[e];
  };
  // MFMA B operand: this lane's output position inside a tile (row rr of the tile, column ox); positions past the tile's TR rows compute on position 0 and are dropped
  int bj[kTJ], e_rr[kTJ], e_ox[kTJ];
#pragma unroll
  for (int t = 0; t < kTJ; ++t) {
    int const j = wj * (kTJ * 32) + t * 32 + (lane & 31), rr = j / COW, ox = j - rr * COW;
    bool const live = j < TR * COW;
    bj[t] = live ? (rr * kWp + ox) : 0; e_rr[t] = live ? rr : -1; e_ox[t] = ox;
  }

  f32x16 acc[kTI][kTJ];
#pragma unroll
  for (int x = 0; x < kTI; ++x)
#pragma unroll
    for (int y = 0; y < kTJ; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  load_patch(r_lo);
  store_patch(0);
  __syncthreads();

  bool const vec_ok = ((p.out_ctot | p.out_coff) & 7) == 0;
  for (int t = 0; t < n_tiles; ++t) {
    int const r0 = r_lo + t * TR;
    u32x4 const *const Js = Js0 + (t & 1) * kPatchC;
    if (t + 1 < n_tiles) load_patch(r0 + TR);
    // ---- the tile's whole K: kNIT k-iterations, fragments of the next one read under the MFMAs of this one
    {
      bf16x8 b[2][kTJ];
#pragma unroll
      for (int tb = 0; tb < kTJ; ++tb) b[0][tb] = __builtin_bit_cast(bf16x8, Js[bj[tb] + (h ? slot_off(0, 1) : slot_off(0, 0))]);
#pragma unroll
      for (int it = 0; it < kNIT; ++it) {
        if (it + 1 < kNIT) {
          int const kt = (it + 1) / kN, s = (it + 1) - kt * kN;
          int const jo = h ? slot_off(kt, 2 * s + 1) : slot_off(kt, 2 * s);
#pragma unroll
          for (int tb = 0; tb < kTJ; ++tb) b[(it + 1) & 1][tb] = __builtin_bit_cast(bf16x8, Js[bj[tb] + jo]);
        }
#pragma unroll
        for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
          for (int tb = 0; tb < kTJ; ++tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[it][ta]), b[it & 1][tb], acc[ta][tb], 0, 0, 0);
      }
    }
    __syncthreads();       // (the ring rows this tile overwrites have been pooled / stored by every wave)
    // ---- bias, ReLU, bf16 -> the ring.  C/D layout of the 32x32 MFMA family: column j = lane & 31, rows i = 8 g + 4 (lane >> 5) + e for register 4 g + e
#pragma unroll
    for (int tb = 0; tb < kTJ; ++tb) {
      int const r = r0 + e_rr[tb];
      bool const keep = (e_rr[tb] >= 0) && (r < COH);
      char *const at = E + (r % kNR) * kERow + e_ox[tb] * kEP;
#pragma unroll
      for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 const bv = *reinterpret_cast<f32x4 const *>(Bs + ta * 32 + 8 * g + 4 * h);
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { float x = acc[ta][tb][4 * g + e] + bv[e]; if (RELU) x = (x > 0.f) ? x : 0.f; v[e] = (__bf16)x; acc[ta][tb][4 * g + e] = 0.f; }
          if (keep) *reinterpret_cast<bf16x4 *>(at + (ta * 32 + 8 * g + 4 * h) * 2) = v;
        }
    }
    if (t + 1 < n_tiles) store_patch((t & 1) ^ 1);
    __syncthreads();       // (the tile's rows are in the ring; the next tile's patch is complete)
#if PKH > 0
    // ---- every pooled row whose last output row is there now
    int const r_done = r0 + TR - 1;
    while (y_next < y1 && min(COH - 1, y_next * PSY - PPY + PKH - 1) <= r_done) {
      int const y = y_next++;
      int const ra = max(0, y * PSY - PPY), rb = min(COH, y * PSY - PPY + PKH);          // the window's rows, clipped to the plane (duplicates are harmless to a maximum)
      // (16 consecutive lanes = 8 positions x 2 chunks: with a 144-byte position pitch and windows 2 positions apart that is 16 different 16-byte bank groups)
      for (int idx = tid; idx < ((POW + 7) / 8) * (kC8 / 2) * 16; idx += kNT) {
        int const rest = idx >> 4, cp = rest % (kC8 / 2), c = cp * 2 + ((idx >> 3) & 1), x = (rest / (kC8 / 2)) * 8 + (idx & 7);
        if (POW % 8 != 0 && x >= POW) continue;
        int const xa = max(0, x * PSX - PPX), xb = min(COW, x * PSX - PPX + PKW);
        s16x8 m = {0, 0, 0, 0, 0, 0, 0, 0};                                             // (values are non-negative: 0 is the identity)
#pragma unroll
        for (int ky = 0; ky < PKH; ++ky) {
          int const r = min(ra + ky, rb - 1);
          char const *const row = E + (r % kNR) * kERow + c * 16;
#pragma unroll
          for (int kx = 0; kx < PKW; ++kx) { int const cx = min(xa + kx, xb - 1); m = __builtin_elementwise_max(m, *reinterpret_cast<s16x8 const *>(row + cx * kEP)); }
        }
        *reinterpret_cast<s16x8 *>(P + x * kPP + c * 16) = m;
      }
      __syncthreads();
      for (int idx = tid; idx < POW * kC8; idx += kNT) {                                 // (chunk fastest: eight lanes store one position's 128 bytes)
        int const x = idx / kC8, c = idx - x * kC8;
        int const oc = c * 8;
        if (oc >= p.oc) continue;
        bf16x8 const mid = *reinterpret_cast<bf16x8 const *>(P + x * kPP + c * 16);
        bf16x8 r = mid;
#if LRN_N > 0
        {
          constexpr int HALF = LRN_N / 2;
          float const alpha = p.lrn_alpha, beta = p.lrn_beta, k = p.lrn_k;
          float const per_elem = alpha / (float)LRN_N;
          int const nc8 = (p.oc + 7) / 8;
          bool const has_lo = c > 0, has_hi = c + 1 < nc8;
          bf16x8 const lo = *reinterpret_cast<bf16x8 const *>(P + x * kPP + (has_lo ? c - 1 : c) * 16), hi = *reinterpret_cast<bf16x8 const *>(P + x * kPP + (has_hi ? c + 1 : c) * 16);
          float pv[8 + 2 * HALF], sq[8 + 2 * HALF];
          for (int e = 0; e < HALF; ++e) { pv[e] = has_lo ? (float)lo[8 - HALF + e] : 0.0f; pv[HALF + 8 + e] = has_hi ? (float)hi[e] : 0.0f; }
          for (int e = 0; e < 8; ++e) pv[HALF + e] = (float)mid[e];
          for (int e = 0; e < 8 + 2 * HALF; ++e) sq[e] = pv[e] * pv[e];
          for (int e = 0; e < 8; ++e) {
            float sumsq = 0.0f;
            for (int d = 0; d < 2 * HALF + 1; ++d) sumsq += sq[e + d];                  // ascending channel order, as the LRN kernels
            r[e] = (__bf16)(pv[HALF + e] * __builtin_amdgcn_exp2f(-beta * __builtin_amdgcn_logf(k + sumsq * per_elem)));
          }
        }
#endif
        unsigned const off = ((unsigned)((img * POH + y) * POW + x) * (unsigned)p.out_ctot + (unsigned)p.out_coff + (unsigned)oc) * 2u;
        u32x4 const v = __builtin_bit_cast(u32x4, r);
        if (vec_ok && oc + 8 <= p.oc) __builtin_amdgcn_raw_buffer_store_b128(v, rD, (int)off, 0, 0);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (oc + e < p.oc) __builtin_amdgcn_raw_buffer_store_b16((short)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu), rD, (int)(off + 2u * e), 0, 0);
        }
      }
      if (y_next < y1 && min(COH - 1, y_next * PSY - PPY + PKH - 1) <= r_done) __syncthreads();   // (another row this tile: the pooled-row image is reused)
    }
#else
    // ---- the tile's rows leave as whole 128-byte lines
    for (int idx = tid; idx < TR * COW * kC8; idx += kNT) {
      int const pos = idx / kC8, c = idx - pos * kC8, rr = pos / COW, col = pos - rr * COW, r = r0 + rr, oc = c * 8;
      if (r > r_hi || oc >= p.oc) continue;
      u32x4 const v = *reinterpret_cast<u32x4 const *>(E + (r % kNR) * kERow + col * kEP + c * 16);
      unsigned const off = ((unsigned)((img * COH + r) * COW + col) * (unsigned)p.out_ctot + (unsigned)p.out_coff + (unsigned)oc) * 2u;
      if (vec_ok && oc + 8 <= p.oc) __builtin_amdgcn_raw_buffer_store_b128(v, rD, (int)off, 0, 0);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (oc + e < p.oc) __builtin_amdgcn_raw_buffer_store_b16((short)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu), rD, (int)(off + 2u * e), 0, 0);
      }
    }
#endif
  }
}

// conv_nhwc_multi_bf16.hip -- SEVERAL INDEPENDENT channels-last bf16 convolutions (+ bias, ReLU) in ONE launch, for gfx950 (BASELINE config 5).
//
// A per-layer op list at 64 images is 54-64 launches of which most are tile-starved and latency-bound: an inception module's 1x1 convolutions on 14x14 / 7x7 maps are
// 50-200 tiles for 256 CUs, a launch costs ~4-6 us whatever it computes, and the roofline time of such a layer is 0.3-2 us.  The reference has
// no counterpart (one launch per op, src/rtc_fwd.cc:545-549); on this chip the remedy is to give the hardware dispatcher ALL the tiles of many small problems at once:
//
//   * the grid is the concatenation of the members' tile lists, longest tiles (most K steps) first -- workgroups are handed to CUs as CUs free up, which makes the
//     dispatcher itself a longest-processing-time-first scheduler; within a member the order is XCD-aware (the eight consecutive workgroups that land on the eight XCDs
//     are eight different pel tiles; one XCD walks the out_chan tiles of "its" pel tiles, so an input tile is fetched into one L2 only);
//   * everything that kernels/conv_nhwc_bf16.hip takes as a -D constant (channels, window, stride, padding, plane sizes) is RUN-TIME data here: a 128-byte descriptor per
//     member in device memory, read with scalar loads (the member index is workgroup-uniform);
//   * the per-output arithmetic is that kernel's, chunk for chunk: D[oc][pel] accumulates v_mfma_f32_32x32x16_bf16 over k = (ky, kx, c) in ascending 16-k groups, zero
//     beyond K -- so a member's result is BIT-IDENTICAL to its own hip_conv_nhwc launch on the implicit-GEMM kernel without K slices (tests/test_gpu_nhwc.py).
//
// Operands (per member):  in img:y:x:chan bf16 (chan % 8 == 0) | filts out_chan:y:x:in_chan bf16 | biases float | out img:y:x:chan bf16 (OUT_F32: float), optionally a
// channel slice [out_coff, out_coff + OC) of a wider tensor.  Data path as conv_nhwc_bf16.hip: both operand images of a K step (BK consecutive k = BK/8 16-byte chunks)
// are filled by `buffer_load_dwordx4 ... lds` (no staging registers, out-of-range lanes read zero), row-major [row][BK] with the chunk position XOR-swizzled through the
// source address, NBUF-deep ring with counted vmcnt and the bare s_barrier; bf16 epilogue through an LDS tile so that every global store writes 16 contiguous bytes.
// The k position of a lane's chunk (tap row / column / channel group) is carried incrementally from step to step: no division in the K loop.
//
// -D parameters: KNAME BI BJ BK(32|64) WI WJ MINW NBUF(2..4) OUT_F32

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef OUT_F32
#define OUT_F32 0
#endif
#ifndef NBUF
#define NBUF 3
#endif

struct prob_t {   // one member convolution (128 bytes; host mirror: native_kernels.cc)
  void const *I; void const *J; void *D; float const *bias;     // filts, in, out, biases
  unsigned I_bytes, J_bytes, D_bytes; int Mi;                    // buffer extents; out_chans
  int Nj, CIN, KH, KW;                                           // output positions (img * oy * ox); stored in_chans (% 8 == 0); window
  int SY, SX, PY, PX;
  int CH, CW, COH, COW;                                          // input plane, output plane
  int kCG, kKC, nK, relu;                                        // chunks per tap (CIN / 8), chunks along k (taps * kCG), K steps, ReLU
  int out_ctot, out_coff, tiles_i, tiles_j;
};
struct tile_t { int prob, tile_i, tile_j, pad; };
struct multi_args_t { prob_t const *probs; tile_t const *tiles; int n_tiles; int n_probs; };

namespace {
constexpr int kNW = WI * WJ, kNT = kNW * 64;
constexpr int kTI = BI / (WI * 32), kTJ = BJ / (WJ * 32);
static_assert(BI % (WI * 32) == 0 && BJ % (WJ * 32) == 0, "tile must be a multiple of the 32x32 MFMA tile per wave");
static_assert(BK == 32 || BK == 64, "BK: 32 | 64");
constexpr int kRowB = BK * 2;                 // bytes per LDS row
constexpr int kCPR = kRowB / 16;              // 16-byte chunks per LDS row
constexpr int kRP = 256 / kRowB;              // LDS rows per 256 bytes (one pass over the 64 banks)
constexpr int kIInst = BI * kCPR / 64, kJInst = BJ * kCPR / 64;   // 1-KB wave instructions per image
static_assert((BI * kCPR) % 64 == 0 && (BJ * kCPR) % 64 == 0, "an image must be a whole number of 1-KB wave loads");
static_assert(kIInst % kNW == 0 && kJInst % kNW == 0, "loads are counted per wave: every wave must issue the same number");
constexpr int kISlots = kIInst / kNW, kJSlots = kJInst / kNW;
static_assert(NBUF >= 2 && NBUF <= 4, "ring depth 2..4");
constexpr int kLoadsPerStep = kISlots + kJSlots;
static_assert((NBUF - 2) * kLoadsPerStep <= 56, "loads kept in flight must fit the 6-bit vmcnt");
constexpr int kIImg = BI * kRowB, kJImg = BJ * kRowB;            // bytes
constexpr int kEPitch = BI * 2 + 16;                               // epilogue tile [pel][oc] bf16, rows de-phased by 4 banks
constexpr int kStage = NBUF * (kIImg + kJImg), kEpi = OUT_F32 ? 0 : BJ * kEPitch;
constexpr int kSmem = kStage > kEpi ? kStage : kEpi;
constexpr int kOOB = (int)0x80000000;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void *lds_t;
__device__ __forceinline__ rsrc_t make_rsrc(void const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ constexpr int swz(int row) { return (row / kRP) & (kCPR - 1); }
} // namespace

extern "C" __global__ __launch_bounds__(WI * WJ * 64, MINW) void KNAME(multi_args_t const a) {
  __shared__ __attribute__((aligned(1024))) char smem[kSmem];
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave / WJ, wj = wave % WJ;

  tile_t const *__restrict__ const tp = a.tiles + blockIdx.x;     // workgroup-uniform: scalar loads
  int const t_prob = __builtin_amdgcn_readfirstlane(tp->prob), t_i = __builtin_amdgcn_readfirstlane(tp->tile_i), t_j = __builtin_amdgcn_readfirstlane(tp->tile_j);
  prob_t const &q = *(prob_t const *__restrict__)(a.probs + t_prob);
#define U(x) __builtin_amdgcn_readfirstlane(x)
  int const Mi = U(q.Mi), Nj = U(q.Nj), CIN = U(q.CIN), KW = U(q.KW), KH = U(q.KH), SY = U(q.SY), SX = U(q.SX), PY = U(q.PY), PX = U(q.PX), CH = U(q.CH), CW = U(q.CW),
            COH = U(q.COH), COW = U(q.COW);
  int const kCG = U(q.kCG), kKC = U(q.kKC), nk = U(q.nK);
  int const i0 = t_i * BI, j0 = t_j * BJ;
  rsrc_t const rI = make_rsrc(q.I, q.I_bytes), rJ = make_rsrc(q.J, q.J_bytes);

  // ---- this thread's LDS slots: slot s of an image = chunk ((s*kNW + wave)*64 + lane) of the row-major [row][kCPR] image
  int ibase[kISlots], ichunk[kISlots];        // filters: byte offset of row oc (k = 0) or OOB; logical chunk of the slot
  int jbase[kJSlots], jyx[kJSlots];           // input: byte offset of (img, oy*SY-PY, ox*SX-PX, c = 0); (iy0 << 16) | (ix0 & 0xffff)
  int jcg[kJSlots], jky[kJSlots], jkx[kJSlots];   // the k position of this slot's chunk in the step being staged: channel group, tap row, tap column
#pragma unroll
  for (int s = 0; s < kISlots; ++s) {
    int const ci = (s * kNW + wave) * 64 + lane, row = ci / kCPR, pos = ci % kCPR;
    ichunk[s] = pos ^ swz(row);
    ibase[s] = (i0 + row < Mi) ? (int)((unsigned)(i0 + row) * (unsigned)(kKC * 16)) : kOOB;
  }
  int const ohw = COH * COW;
#pragma unroll
  for (int s = 0; s < kJSlots; ++s) {
    int const ci = (s * kNW + wave) * 64 + lane, row = ci / kCPR, pos = ci % kCPR;
    int const kc0 = pos ^ swz(row);
    int const pel = j0 + row;
    int const img = pel / ohw, rem = pel - img * ohw, oy = rem / COW, ox = rem - oy * COW;
    int const iy0 = oy * SY - PY, ix0 = ox * SX - PX;
    jbase[s] = ((img * CH + iy0) * CW + ix0) * (CIN * 2);
    jyx[s] = (pel < Nj) ? ((iy0 << 16) | (ix0 & 0xffff)) : (int)0x80008000;   // (a row of no image: every tap fails the range test)
    int const t0 = kc0 / kCG;                                                   // (kc0 < kCPR <= 8)
    jcg[s] = kc0 - t0 * kCG; jky[s] = t0 / KW; jkx[s] = t0 - jky[s] * KW;
  }

  // One K step's operand loads are kISlots + kJSlots 1-KB pieces per wave.  `stage` issues step `step` into ring slot `buf`; steps are staged in ascending order,
  // so the input side's (channel group, tap) position advances by kCPR chunks after every staged step.
  auto stage = [&](int step, int buf) {
    char *const Ib = smem + buf * (kIImg + kJImg), *const Jb = Ib + kIImg;
#pragma unroll
    for (int s = 0; s < kISlots; ++s) {
      int const kc = step * kCPR + ichunk[s];
      int off = ibase[s] + kc * 16;
      if (kc >= kKC || ibase[s] == kOOB) off = kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rI, (lds_t)(Ib + (s * kNW + wave) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < kJSlots; ++s) {
      int const iy = (jyx[s] >> 16) + jky[s], ix = (int)(short)(jyx[s] & 0xffff) + jkx[s];
      bool const ok = (jky[s] < KH) && ((unsigned)iy < (unsigned)CH) && ((unsigned)ix < (unsigned)CW);
      int const off = jbase[s] + ((jky[s] * CW + jkx[s]) * CIN + jcg[s] * 8) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rJ, (lds_t)(Jb + (s * kNW + wave) * 1024), 16, ok ? off : kOOB, 0, 0, 0);
      jcg[s] += kCPR;
      while (jcg[s] >= kCG) { jcg[s] -= kCG; if (++jkx[s] == KW) { jkx[s] = 0; ++jky[s]; } }
    }
  };

  f32x16 acc[kTI][kTJ];
#pragma unroll
  for (int x = 0; x < kTI; ++x)
#pragma unroll
    for (int b = 0; b < kTJ; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][b][r] = 0.f;

  // fragment reads: lane l holds row (l & 31), k-chunk (2*kk + (l >> 5)) of each 16-deep MFMA step, at the swizzled chunk position
  int const h = lane >> 5, fsw = swz(lane & 31);
  constexpr int kKK = BK / 16;
  int xo[kKK];
#pragma unroll
  for (int kk = 0; kk < kKK; ++kk) xo[kk] = ((2 * kk + h) ^ fsw) * 16;
  int const arow = (wi * (kTI * 32) + (lane & 31)) * kRowB, brow = (wj * (kTJ * 32) + (lane & 31)) * kRowB;

  auto wait_loads = [&](int in_flight_steps) {   // (the count is an immediate: one case per depth)
    switch (in_flight_steps < 0 ? 0 : in_flight_steps) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerStep) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF > 3 ? 2 * kLoadsPerStep : 0) : "memory"); break;
    }
  };
  auto barrier = [&]() { asm volatile("s_barrier" ::: "memory"); };
#pragma unroll
  for (int s0 = 0; s0 < NBUF - 1; ++s0) if (s0 < nk) stage(s0, s0);
  wait_loads((nk < NBUF - 1 ? nk : NBUF - 1) - 1);
  barrier();
  int cur = 0;
  for (int step = 0; step < nk; ++step) {
    int const nxt = (cur == 0) ? NBUF - 1 : cur - 1;   // == (step + NBUF - 1) % NBUF
    if (step + NBUF - 1 < nk) stage(step + NBUF - 1, nxt);
    char const *const Ib = smem + cur * (kIImg + kJImg), *const Jb = Ib + kIImg;
#pragma unroll
    for (int kk = 0; kk < kKK; ++kk) {
      bf16x8 av[kTI], bw[kTJ];
#pragma unroll
      for (int x = 0; x < kTI; ++x) av[x] = *reinterpret_cast<bf16x8 const *>(Ib + arow + x * (32 * kRowB) + xo[kk]);
#pragma unroll
      for (int x = 0; x < kTJ; ++x) bw[x] = *reinterpret_cast<bf16x8 const *>(Jb + brow + x * (32 * kRowB) + xo[kk]);
#pragma unroll
      for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
        for (int tb = 0; tb < kTJ; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ta], bw[tb], acc[ta][tb], 0, 0, 0);
    }
    // loads still wanted in flight after this wait: those of steps step + 2 .. step + NBUF - 1 that exist
    int const newest = (nk - 1 < step + NBUF - 1) ? nk - 1 : step + NBUF - 1;
    wait_loads(NBUF == 2 ? 0 : newest - (step + 1));
    // this wave's fragment reads of the step must have RETURNED before it releases the slot: an LDS-DMA write of another wave does NOT queue behind ds_reads that
    // were merely issued (measured, tools/multi_stress.py: 32x128 tiles / 1x4 waves / ring of two without this wait: 17 of 40 launches with 64-1300 wrong outputs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();
    cur = (cur + 1 == NBUF) ? 0 : cur + 1;
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA family: column j = lane & 31, rows i = 8*g + 4*(lane >> 5) + e for register 4*g + e
  rsrc_t const rD = make_rsrc(q.D, q.D_bytes), rB = make_rsrc(q.bias, (unsigned)Mi * 4u);
  int const o_ctot = U(q.out_ctot), o_coff = U(q.out_coff);
  bool const relu = U(q.relu) != 0;
#undef U
  f32x4 bv[kTI][4];
#pragma unroll
  for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int const oc = i0 + wi * (kTI * 32) + ta * 32 + 8 * g + 4 * h;
      if (oc + 4 <= Mi) bv[ta][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, oc * 4, 0, 0));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[ta][g][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, (oc + e < Mi) ? (oc + e) * 4 : kOOB, 0, 0));
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
        float x[4];   // (scalars, not elements of a vector: see conv_nhwc_bf16.hip)
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = acc[ta][tb][4 * g + e] + bv[ta][g][e]; if (relu) x[e] = (x[e] > 0.f) ? x[e] : 0.f; }
        if (pel < Nj) {
          if (oc + 4 <= Mi && ((o_ctot | o_coff) & 3) == 0) {
            f32x4 v; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rD, (int)(rowoff + (unsigned)oc * 4u), 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (oc + e < Mi) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x[e]), rD, (int)(rowoff + (unsigned)(oc + e) * 4u), 0, 0);
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
          for (int e = 0; e < 4; ++e) { float x = acc[ta][tb][4 * g + e] + bv[ta][g][e]; if (relu) x = (x > 0.f) ? x : 0.f; v[e] = (__bf16)x; }
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
      int const pel = j0 + prow, oc = i0 + cc * 8;
      if (pel >= Nj || oc >= Mi) continue;
      u32x4 const v = *reinterpret_cast<u32x4 const *>(E + prow * kEPitch + cc * 16);
      unsigned const off = ((unsigned)pel * (unsigned)o_ctot + (unsigned)o_coff + (unsigned)oc) * 2u;
      if (vec_ok && oc + 8 <= Mi) __builtin_amdgcn_raw_buffer_store_b128(v, rD, (int)off, 0, 0);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (oc + e < Mi) __builtin_amdgcn_raw_buffer_store_b16((short)((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu), rD, (int)(off + 2u * e), 0, 0);
      }
    }
  }
#endif
}

// fc_f32.hip -- fp32 MFMA kernel for convolutions whose window is the whole input ("ipconv": fully-connected layers) on gfx950 (MI355X).
//
//   out[img][oc] = act( sum_k filts[oc][k] * in[img][k] + bias[oc] ),   k = (in_chan, y, x) flattened: BOTH operands are k-contiguous rows
//   (the reference's ipconv case: src/cnn_op.cc:104-117, test/rtc/ipconv.cucl; epilogue src/cnn_codegen.cc:35-42)
//
// Why a kernel of its own.  AlexNet fc6 / fc7 at 256 images are 4096 x 256 outputs with K = 9216 / 4096: 256 tiles of 64 x 64, one workgroup per CU, one 32 x 32
// accumulator per wave -- nothing hides anything.  gemm_conv_f32.hip runs them at 235 / 108 us (82 / 79 TF/s).  What was measured on the way here (tools/r4o.sh,
// tools/pmc_fc.sh; fc6, isolated launches):
//   * a lone chain of dependent v_mfma_f32_32x32x2_f32 issues every 71 cycles, not 64 (no LDS traffic at all: 132 us = the floor of ANY one-chain-per-wave layout),
//     and every ~100 cycles once operand reads and lane-half selects sit between the links (PMC: 61 % MFMA-busy, no LDS bank conflicts, LDS 23 % busy);
//   * the global loads are hidden (replacing them by constants changes nothing); the reference's random data costs 13 % against zeros through the clock (2.2 vs 2.47 GHz);
//   * LDS stores issued by the multiplying waves stall them whatever their width, pattern or position in the step.
// Hence:
//   * FOUR independent chains per wave: the wave's 32 x 32 block is 2 x 2 sub-blocks of v_mfma_f32_16x16x4_f32 (same flops per cycle).  Lane l supplies row l % 16 and
//     k = 4 q + l / 16 of a k quad -- the k of a lane is part of its LDS address: one ds_read_b32 per operand half and quad, no selects;
//   * x-major LDS images ([row][BKF + 4] floats, like the tensors): a staged float4 is ONE ds_write_b128;
//   * two roles: waves 0-3 only read operands and multiply, waves 4-7 only stage (global -> registers, PF K tiles in flight -> LDS).  Three LDS stages: tile kt + 2 is
//     written during step kt and the barrier that ends a step waits only for the stores of the step before;
//   * MFMA rows are images and columns are out_chans (the output's contiguous dim): plain row stores;
//   * workgroup ids are permuted so that the image tiles of one out_chan tile (they share the filter rows) sit next to each other in one XCD's L2.
// Result: fc6 235 -> 187 us (103 TF/s), fc7 108 -> 88 (98 TF/s).  Same ascending-k chain of exact fp32 fmas per output as every other fp32 kernel here (the 16x16x4
// MFMA adds its four k in order): bit-identical to the oracle.
// Needs K % 4 == 0 (16-byte-aligned rows).  Compile-time parameters (-D): KNAME BKF (k per step: 32 | 64) PF (K tiles in flight in registers: 2 | 4 | 6 | 8) RELU;
// variants kept for the record: SPEC (0: four waves do both) NS3 (0: two LDS stages, full barrier) M16 (0: one 32x32x2 chain per wave) RA LDP SELA

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef ABLATE
#define ABLATE 0 // experiment hook (BODAHIP_EXTRA_DEFS): 2 no global loads | 4 no in-loop LDS stores
#endif

struct gemm_args_t { // same layout as gemm_conv_f32.hip (one host-side struct serves all fp32 kernels)
  float const *I; float const *J; float *D; float const *bias;   // I = filts [Mi][K], J = in [Nj][K]
  int Mi, Nj, K;
  int ldI, ldJ, ldD;
  int C, H, W, OH, OW;
  int tiles_i, tiles_j;       // out_chan tiles | image tiles (64 each)
  int splitk, kt_per;
  float *ws; long ws_slab;
  unsigned I_bytes, J_bytes;
  unsigned D_bytes;
  int out_ctot, out_coff;
  int const *ktab; int ktab_n;
  long bsI, bsJ, bsD;
};

#ifndef M16
#define M16 1 // 1: four 16x16x4 MFMA chains per wave | 0: one 32x32x2 chain
#endif
namespace {
#ifndef SPEC
#define SPEC 1 // 1: eight waves -- waves 0-3 multiply, waves 4-7 stage (global -> registers -> LDS); 0: four waves that do both
#endif
constexpr int kNT = 256;                      // staging threads (= multiplying threads)
constexpr int kThreads = SPEC ? 512 : 256;
#ifndef NS3
#define NS3 1 // SPEC only: 1 = three LDS stages -- tile kt + 2 is written during step kt and the barrier waits only for the PREVIOUS step's stores
#endif
constexpr int kNS = (SPEC * NS3) ? 3 : 2;
#ifndef LDP
#define LDP 4 // (2: conflict-free ds_read_b32 for the 16x16x4 path at the price of 8-byte stores -- measured level with 4)
#endif
// floats per LDS row.  32x32x2 path (ds_read_b128, 16 lanes per service group): BKF + 4 = an odd number of sixteen-byte units.  16x16x4 path (ds_read_b32: lane l reads
// row l % 16, dword 4 q + l / 16; 32 lanes per service group): BKF + 2 -- the pitch is 2 mod 32 banks, so the group's 16 rows x 2 k land on 32 different banks (BKF + 4
// is 2-way conflicted); a staged float4 is then two 8-byte stores
constexpr int kLDK = BKF + LDP;
constexpr int kQ = BKF / 4;                   // k quads per step = ds_read_b128 per operand, lane and step
constexpr int kNL = BKF / 16;                 // float4s per thread, operand and step (64 rows x BKF / 4 units over 256 threads)
constexpr int kStage = 64 * kLDK;             // floats per operand image
constexpr int kOOB = (int)0x80000000;
static_assert(BKF == 32 || BKF == 64, "BKF: 32 | 64");
static_assert(PF % 2 == 0 && PF >= 2 && PF <= 8, "PF: 2 | 4 | 6 | 8");
#ifndef SELA
#define SELA 1 // 1: operand selects one quad ahead of their MFMAs
#endif
#ifndef RA
#define RA 2 // operand quads read ahead of the MFMAs (measured, 16x16x4 path: 2 best, 1 / 3 / 5 slower)
#endif
static_assert(RA >= 1 && RA < kQ, "RA");
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(float const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
__device__ __forceinline__ float bload1(rsrc_t r, int voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0)); }
} // namespace

extern "C" __global__ __launch_bounds__(kThreads, 1) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) float sm[kNS * 2 * kStage];   // [stage][operand: 0 = in rows (MFMA A), 1 = filter rows (MFMA B)][row][kLDK]
  int const lane = threadIdx.x & 63;
  int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bool const loader = SPEC ? (wave >= 4) : true, multiplier = SPEC ? (wave < 4) : true;
  int const tid = threadIdx.x & 255;           // id among the staging threads / among the multiplying threads
  int const wr = (wave & 3) >> 1, wc = wave & 1;     // 2 x 2 multiplying waves: image half, out_chan half of the 64 x 64 tile

  // workgroup b runs on XCD b % 8; with a multiple of 8 workgroups the ids are permuted so that one XCD owns a contiguous range of tiles, image tiles
  // fastest: the workgroups that share a filter tile (all image tiles of an out_chan tile) sit next to each other in one L2
  int const G = p.tiles_i * p.tiles_j;
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (v & 7) * (G >> 3) + (v >> 3);
  int const ti = v / p.tiles_j, tj = v - ti * p.tiles_j;
  int const oc0 = ti * 64, img0 = tj * 64;

  rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);
  // staging: unit c = tid + n * 256 of an operand tile is (row c / kQ, k quad c % kQ); byte offsets of this thread's units at k = 0 (out of range past the tensor's rows)
  int goff[2][kNL], loff[kNL];
#pragma unroll
  for (int n = 0; n < kNL; ++n) {
    int const c = tid + n * kNT, row = c / kQ, qd = c - row * kQ;
    goff[0][n] = (img0 + row < p.Nj) ? (((img0 + row) * p.K + 4 * qd) * 4) : kOOB;
    goff[1][n] = (oc0 + row < p.Mi) ? (((oc0 + row) * p.K + 4 * qd) * 4) : kOOB;
#if ABLATE & 32
    loff[n] = c * 4;            // (timing experiment: linear, fully contiguous LDS stores -- wrong results)
#else
    loff[n] = row * kLDK + 4 * qd;
#endif
  }
  int const nkt = (p.K + BKF - 1) / BKF;
  // (a K tail: k quads past K must read zeros -- the buffer's range check only covers the END of the tensor, so the quad index is tested)
  auto gload = [&](int op, int n, int kt) -> f32x4 {
#if ABLATE & 2
    float const f = (float)(kt + n); return f32x4{f, f, f, f};
#else
    int const c = tid + n * kNT, qd = c % kQ;
    bool const in_k = (kt * BKF + 4 * qd) < p.K;
    return bload4(op ? rI : rJ, in_k ? goff[op][n] : kOOB, kt * (BKF * 4));
#endif
  };
  auto lstore = [&](int op, int n, int stage, f32x4 const &val) {
#if !(ABLATE & 4)
#if ABLATE & 64
    float *const d = sm + (stage * 2 + op) * kStage + loff[n];   // (timing experiment: two 8-byte stores)
    *reinterpret_cast<float2 *>(d) = float2{val[0], val[1]}; *reinterpret_cast<float2 *>(d + 2) = float2{val[2], val[3]};
#elif ABLATE & 128
    float *const d = sm + (stage * 2 + op) * kStage + loff[n];   // (timing experiment: four 4-byte stores)
    d[0] = val[0]; d[1] = val[1]; d[2] = val[2]; d[3] = val[3];
#elif (LDP % 4) != 0
    float *const d = sm + (stage * 2 + op) * kStage + loff[n];   // rows are only 8-byte aligned
    *reinterpret_cast<float2 *>(d) = float2{val[0], val[1]}; *reinterpret_cast<float2 *>(d + 2) = float2{val[2], val[3]};
#else
    *reinterpret_cast<f32x4 *>(sm + (stage * 2 + op) * kStage + loff[n]) = val;
#endif
#endif
  };

  f32x4 ring[PF][2][kNL];
  bool const hi = (lane >> 5) != 0;
  float const *const a_base = sm + (wr * 32 + (lane & 31)) * kLDK;               // + stage * 2 * kStage
  float const *const b_base = sm + kStage + (wc * 32 + (lane & 31)) * kLDK;
#if M16
  // 2 x 2 sub-blocks of 16 x 16 per wave, v_mfma_f32_16x16x4_f32: FOUR independent accumulation chains per wave.  One dependent chain of 32x32x2 MFMAs runs at 71 cycles
  // per instruction by itself (64 of arithmetic) and at ~100 with operand traffic around it; four interleaved chains keep the pipe issuing every 32 cycles.  Lane l supplies
  // row l % 16 and k = 4 q + l / 16 of a quad: the k of a lane is part of its LDS address, one ds_read_b32 per operand half and quad, no selects.
  f32x4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float const *const a16 = sm + (wr * 32 + (lane & 15)) * kLDK + (lane >> 4);
  float const *const b16 = sm + kStage + (wc * 32 + (lane & 15)) * kLDK + (lane >> 4);
#else
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#endif

  // one K step of the multiplying role: kQ operand quads out of LDS stage `st`, two MFMAs each; the reads run RA quads ahead
#if M16
  auto multiply = [&](int st) {
    float const *const A = a16 + st * 2 * kStage, *const B = b16 + st * 2 * kStage;
    float a0[RA + 1], a1[RA + 1], b0[RA + 1], b1[RA + 1];
#pragma unroll
    for (int i = 0; i < RA; ++i) { a0[i] = A[4 * i]; a1[i] = A[16 * kLDK + 4 * i]; b0[i] = B[4 * i]; b1[i] = B[16 * kLDK + 4 * i]; }
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      if (q + RA < kQ) { int const j = (q + RA) % (RA + 1); a0[j] = A[4 * (q + RA)]; a1[j] = A[16 * kLDK + 4 * (q + RA)]; b0[j] = B[4 * (q + RA)]; b1[j] = B[16 * kLDK + 4 * (q + RA)]; }
      int const c = q % (RA + 1);
      __builtin_amdgcn_sched_barrier(0);   // (keeps the reads RA quads ahead: left alone, the scheduler sinks them to one MFMA before their use)
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b0[c], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b1[c], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b0[c], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b1[c], acc[3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#else
  auto multiply = [&](int st) {
    float const *const A = a_base + st * 2 * kStage, *const B = b_base + st * 2 * kStage;
    f32x4 va[RA + 1], vb[RA + 1];
#pragma unroll
    for (int i = 0; i < RA; ++i) { va[i] = *reinterpret_cast<f32x4 const *>(A + 4 * i); vb[i] = *reinterpret_cast<f32x4 const *>(B + 4 * i); }
    // the lane-half selects of quad q + 1 are made BEFORE the MFMAs of quad q are issued (SELA): an MFMA whose operand registers were written by the VALU a few cycles
    // earlier does not start back to back with its predecessor
    float sa[2][2], sb[2][2];
    sa[0][0] = hi ? va[0][1] : va[0][0]; sa[0][1] = hi ? va[0][3] : va[0][2]; sb[0][0] = hi ? vb[0][1] : vb[0][0]; sb[0][1] = hi ? vb[0][3] : vb[0][2];
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      if (q + RA < kQ) { va[(q + RA) % (RA + 1)] = *reinterpret_cast<f32x4 const *>(A + 4 * (q + RA)); vb[(q + RA) % (RA + 1)] = *reinterpret_cast<f32x4 const *>(B + 4 * (q + RA)); }
#if SELA
      if (q + 1 < kQ) {
        f32x4 const ca = va[(q + 1) % (RA + 1)], cb = vb[(q + 1) % (RA + 1)];
        sa[(q + 1) & 1][0] = hi ? ca[1] : ca[0]; sa[(q + 1) & 1][1] = hi ? ca[3] : ca[2]; sb[(q + 1) & 1][0] = hi ? cb[1] : cb[0]; sb[(q + 1) & 1][1] = hi ? cb[3] : cb[2];
      }
      __builtin_amdgcn_sched_barrier(0);
#else
      { f32x4 const ca = va[q % (RA + 1)], cb = vb[q % (RA + 1)];
        sa[q & 1][0] = hi ? ca[1] : ca[0]; sa[q & 1][1] = hi ? ca[3] : ca[2]; sb[q & 1][0] = hi ? cb[1] : cb[0]; sb[q & 1][1] = hi ? cb[3] : cb[2]; }
#endif
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[q & 1][0], sb[q & 1][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[q & 1][1], sb[q & 1][1], acc, 0, 0, 0);
#if SELA
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
#endif
  // Every loop below runs over whole rounds of steps with NO conditions inside: a tile past the last one reads zeros (its k quads fail the range test), is stored to a
  // stage nobody reads afterwards, and multiplies as +0 -- branch-free code keeps the compiler's vmcnt bookkeeping exact (with a branch around a store it waits for
  // EVERY load in flight and the register ring is gone).

#if SPEC && NS3
  // Eight waves, two roles, three LDS stages: tile t lives in stage t % 3.  During step kt the staging waves write tile kt + 2 (stage (kt + 2) % 3, last read in step
  // kt - 1) and refill its registers with tile kt + 2 + PF; the barrier that ends the step makes them wait only for the stores of step kt - 1 (lgkmcnt(2 kNL)).
  // Rounds of 3 PF steps keep the stage and ring indices compile-time.  Both roles execute the same number of s_barrier instructions.
  constexpr int kU = 3 * PF;
  int const nkt_pad = (nkt + kU - 1) / kU * kU;
  if (!multiplier) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < kNL; ++n) ring[u][op][n] = gload(op, n, u);
#pragma unroll
    for (int t = 0; t < 2; ++t)      // tiles 0 and 1 go to their stages before the first step; their registers take tiles PF and PF + 1
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < kNL; ++n) { lstore(op, n, t, ring[t % PF][op][n]); ring[t % PF][op][n] = gload(op, n, t + PF); }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
          for (int n = 0; n < kNL; ++n) { lstore(op, n, (u + 2) % 3, ring[(u + 2) % PF][op][n]); ring[(u + 2) % PF][op][n] = gload(op, n, kb + u + 2 + PF); }
        asm volatile("s_waitcnt lgkmcnt(%0)\n s_barrier" :: "n"(2 * kNL) : "memory");
      }
    }
    return;
  }
  asm volatile("s_barrier" ::: "memory");
  for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
    for (int u = 0; u < kU; ++u) { multiply(u % 3); asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
  }
#else
  int const nkt_pad = (nkt + PF - 1) / PF * PF;
  if (loader) {
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int n = 0; n < kNL; ++n) ring[0][op][n] = gload(op, n, 0);
#pragma unroll
    for (int u = 1; u <= PF; ++u) {
      if (u == PF) {   // tile 0 leaves its register set before tile PF takes it
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
          for (int n = 0; n < kNL; ++n) lstore(op, n, 0, ring[0][op][n]);
      }
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < kNL; ++n) ring[u % PF][op][n] = gload(op, n, u);   // (tiles past the last one read zeros)
    }
  }
  __syncthreads();
#if SPEC
  // Eight waves, two roles, two stages: the staging waves write tile kt + 1 to the other stage and refill its registers with tile kt + 1 + PF while the multiplying waves
  // work on tile kt; a full barrier per step.
  if (!multiplier) {
    for (int kb = 0; kb < nkt_pad; kb += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
          for (int n = 0; n < kNL; ++n) { lstore(op, n, (u & 1) ^ 1, ring[(u + 1) % PF][op][n]); ring[(u + 1) % PF][op][n] = gload(op, n, kb + u + 1 + PF); }
        __syncthreads();
      }
    }
    return;
  }
  for (int kb = 0; kb < nkt_pad; kb += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) { multiply(u & 1); __syncthreads(); }
  }
#else
  // Four waves that do both: the stores of tile kt + 1 and the loads of tile kt + 1 + PF are issued ahead of the step's MFMAs.
  for (int kb = 0; kb < nkt_pad; kb += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int op = 0; op < 2; ++op)
#pragma unroll
        for (int n = 0; n < kNL; ++n) { lstore(op, n, (u & 1) ^ 1, ring[(u + 1) % PF][op][n]); ring[(u + 1) % PF][op][n] = gload(op, n, kb + u + 1 + PF); }
      multiply(u & 1);
      __syncthreads();
    }
  }
#endif
#endif

  // ---- epilogue.  32x32: row (image) = 8 * (r / 4) + r % 4 + 4 * (lane / 32), column (out_chan) = lane % 32;  16x16: row = 4 * (lane / 16) + r, column = lane % 16
  {
    rsrc_t const rD = make_rsrc(p.D, p.D_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
#if M16
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      int const oc = oc0 + wc * 32 + (b & 1) * 16 + (lane & 15);
      float const bias = bload1(rB, oc * 4);             // columns past out_chan read 0 and are not stored
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int const img = img0 + wr * 32 + (b >> 1) * 16 + 4 * (lane >> 4) + r;
        float val = acc[b][r] + bias;
#if RELU
        val = (val > 0.f) ? val : 0.f;
#endif
        int const off = ((oc < p.Mi) && (img < p.Nj)) ? (int)(((unsigned)img * (unsigned)p.out_ctot + (unsigned)(p.out_coff + oc)) * 4u) : kOOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rD, off, 0, 0);
      }
    }
#else
    int const oc = oc0 + wc * 32 + (lane & 31);
    float const bias = bload1(rB, oc * 4);               // columns past out_chan read 0 and are not stored
    int const row0 = img0 + wr * 32 + (hi ? 4 : 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int const img = row0 + (r & 3) + 8 * (r >> 2);
      float val = acc[r] + bias;
#if RELU
      val = (val > 0.f) ? val : 0.f;
#endif
      int const off = ((oc < p.Mi) && (img < p.Nj)) ? (int)(((unsigned)img * (unsigned)p.out_ctot + (unsigned)(p.out_coff + oc)) * 4u) : kOOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rD, off, 0, 0);
    }
#endif
  }
}

// fc_f32.hip -- fp32 MFMA kernel for convolutions whose window is the whole input ("ipconv": fully-connected layers) on gfx950 (MI355X).
//
//   out[img][oc] = act( sum_k filts[oc][k] * in[img][k] + bias[oc] ),   k = (in_chan, y, x) flattened: BOTH operands are k-contiguous rows
//   (the reference's ipconv case: src/cnn_op.cc:104-117, test/rtc/ipconv.cucl; epilogue src/cnn_codegen.cc:35-42)
//
// Why a kernel of its own.  AlexNet fc6 / fc7 at 256 images are 4096 x 256 outputs with K = 9216 / 4096: 256 tiles of 64 x 64, one workgroup per CU, one 32 x 32
// accumulator per wave -- nothing hides anything.  gemm_conv_f32.hip runs them at 235 / 108 us (82 / 79 TF/s).  What was measured on the way here (profiles/r04_probe_fc_chain.txt,
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
// The workgroup tile is TM images x TN out_chans (64 | 32 each): 64 x 64 where that gives every CU a workgroup, 32-wide tiles for the tile-starved layers (AlexNet fc8:
// 1000 x 256 outputs = 256 tiles of 32 x 32, one 16 x 16 chain per wave).  The variants that lost on the way (four waves doing both jobs, two LDS stages with a full
// barrier, one 32x32x2 chain per wave with lane-half selects) are in the history of this file (round 4) and in DESIGN.md section 3.1e.
// Needs K % 4 == 0 (16-byte-aligned rows).  Compile-time parameters (-D): KNAME TM TN BKF (k per step: 32 | 64) PF (K tiles in flight in registers: 2 | 4) RELU RA

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef TM
#define TM 64
#endif
#ifndef TN
#define TN 64
#endif
#ifndef RA
#define RA 2 // operand quads read ahead of the MFMAs (measured: 2 best, 1 / 3 / 5 slower)
#endif

struct gemm_args_t { // same layout as gemm_conv_f32.hip (one host-side struct serves all fp32 kernels)
  float const *I; float const *J; float *D; float const *bias;   // I = filts [Mi][K], J = in [Nj][K]
  int Mi, Nj, K;
  int ldI, ldJ, ldD;
  int C, H, W, OH, OW;
  int tiles_i, tiles_j;       // out_chan tiles (TN each) | image tiles (TM each)
  int splitk, kt_per;
  float *ws; long ws_slab;
  unsigned I_bytes, J_bytes;
  unsigned D_bytes;
  int out_ctot, out_coff;
  int const *ktab; int ktab_n;
  long bsI, bsJ, bsD;
};

namespace {
constexpr int kLDK = BKF + 4;                 // floats per LDS row (16-byte aligned rows: a staged float4 is one ds_write_b128)
constexpr int kQ = BKF / 4;                   // k quads per step
constexpr int kNLA = TM * kQ / 256, kNLB = TN * kQ / 256;   // float4s per staging thread and step: image rows, filter rows
constexpr int kImgA = TM * kLDK, kImgB = TN * kLDK, kStage = kImgA + kImgB;   // floats per operand image / per stage
constexpr int kSBM = TM / 32, kSBN = TN / 32; // 16 x 16 sub-blocks per wave: rows x columns (2 x 2 multiplying waves share the tile)
constexpr int kU = 3 * PF;                    // steps per unrolled round (stage = step % 3, register set = step % PF: compile-time)
constexpr int kOOB = (int)0x80000000;
static_assert((TM == 32 || TM == 64) && (TN == 32 || TN == 64) && (BKF == 32 || BKF == 64) && (PF == 2 || PF == 4), "TM, TN: 32 | 64; BKF: 32 | 64; PF: 2 | 4");
static_assert(kNLA >= 1 && kNLB >= 1 && RA >= 1 && RA < kQ, "staging units per thread, read-ahead");
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(float const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
__device__ __forceinline__ float bload1(rsrc_t r, int voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0)); }
} // namespace

#ifndef STGPRIO
#define STGPRIO 0 // experiment hooks: issue priority of the staging / the multiplying waves (s_setprio)
#endif
#ifndef MULPRIO
#define MULPRIO 0
#endif
extern "C" __global__ __launch_bounds__(512, 1) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) float sm[3 * kStage];   // [stage][image rows TM | filter rows TN][kLDK]
  int const lane = threadIdx.x & 63;
  int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int const tid = threadIdx.x & 255;           // id among the staging threads / among the multiplying threads

  // workgroup b runs on XCD b % 8; with a multiple of 8 workgroups the ids are permuted so that one XCD owns a contiguous range of tiles, image tiles
  // fastest: the workgroups that share a filter tile (all image tiles of an out_chan tile) sit next to each other in one L2
  int const G = p.tiles_i * p.tiles_j;
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (v & 7) * (G >> 3) + (v >> 3);
  int const ti = v / p.tiles_j, tj = v - ti * p.tiles_j;
  int const oc0 = ti * TN, img0 = tj * TM;
  int const nkt = (p.K + BKF - 1) / BKF;
  // Whole rounds of kU steps with NO conditions inside the loops: a tile past the last one reads zeros (its k quads fail the range test), is stored to a stage nobody reads
  // afterwards, and multiplies as +0 -- branch-free code keeps the compiler's vmcnt bookkeeping exact (with a branch around a store it waits for EVERY load in flight).
  int const nkt_pad = (nkt + kU - 1) / kU * kU;

  if (wave >= 4) {
#if STGPRIO
    __builtin_amdgcn_s_setprio(STGPRIO);
#endif
    // ---- staging waves: tile t lives in stage t % 3.  During step kt they write tile kt + 2 (stage (kt + 2) % 3, last read in step kt - 1) and refill its registers
    // with tile kt + 2 + PF; the barrier that ends the step makes them wait only for the stores of step kt - 1.
    rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);
    // unit c = tid + n * 256 of an operand image is (row c / kQ, k quad c % kQ); byte offsets at k = 0 (out of range past the tensor's rows)
    int goffA[kNLA], loffA[kNLA], goffB[kNLB], loffB[kNLB];
#pragma unroll
    for (int n = 0; n < kNLA; ++n) { int const c = tid + n * 256, row = c / kQ, qd = c - row * kQ;
      goffA[n] = (img0 + row < p.Nj) ? (((img0 + row) * p.K + 4 * qd) * 4) : kOOB; loffA[n] = row * kLDK + 4 * qd; }
#pragma unroll
    for (int n = 0; n < kNLB; ++n) { int const c = tid + n * 256, row = c / kQ, qd = c - row * kQ;
      goffB[n] = (oc0 + row < p.Mi) ? (((oc0 + row) * p.K + 4 * qd) * 4) : kOOB; loffB[n] = kImgA + row * kLDK + 4 * qd; }
    // (a K tail: k quads past K must read zeros -- the buffer's range check only covers the END of the tensor, so the quad index is tested)
    auto gloadA = [&](int n, int kt) -> f32x4 { int const qd = (tid + n * 256) % kQ; return bload4(rJ, ((kt * BKF + 4 * qd) < p.K) ? goffA[n] : kOOB, kt * (BKF * 4)); };
    auto gloadB = [&](int n, int kt) -> f32x4 { int const qd = (tid + n * 256) % kQ; return bload4(rI, ((kt * BKF + 4 * qd) < p.K) ? goffB[n] : kOOB, kt * (BKF * 4)); };
    f32x4 ringA[PF][kNLA], ringB[PF][kNLB];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int n = 0; n < kNLA; ++n) ringA[u][n] = gloadA(n, u);
#pragma unroll
      for (int n = 0; n < kNLB; ++n) ringB[u][n] = gloadB(n, u);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {    // tiles 0 and 1 go to their stages before the first step; their registers take tiles PF and PF + 1
#pragma unroll
      for (int n = 0; n < kNLA; ++n) { *reinterpret_cast<f32x4 *>(sm + t * kStage + loffA[n]) = ringA[t % PF][n]; ringA[t % PF][n] = gloadA(n, t + PF); }
#pragma unroll
      for (int n = 0; n < kNLB; ++n) { *reinterpret_cast<f32x4 *>(sm + t * kStage + loffB[n]) = ringB[t % PF][n]; ringB[t % PF][n] = gloadB(n, t + PF); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
#pragma unroll
        for (int n = 0; n < kNLA; ++n) { *reinterpret_cast<f32x4 *>(sm + ((u + 2) % 3) * kStage + loffA[n]) = ringA[(u + 2) % PF][n]; ringA[(u + 2) % PF][n] = gloadA(n, kb + u + 2 + PF); }
#pragma unroll
        for (int n = 0; n < kNLB; ++n) { *reinterpret_cast<f32x4 *>(sm + ((u + 2) % 3) * kStage + loffB[n]) = ringB[(u + 2) % PF][n]; ringB[(u + 2) % PF][n] = gloadB(n, kb + u + 2 + PF); }
        asm volatile("s_waitcnt lgkmcnt(%0)\n s_barrier" :: "n"(kNLA + kNLB) : "memory");
      }
    }
    return;
  }

#if MULPRIO
  __builtin_amdgcn_s_setprio(MULPRIO);
#endif
  // ---- multiplying waves (2 x 2): wave (wr, wc) owns rows [wr * TM / 2, +TM / 2) x columns [wc * TN / 2, +TN / 2) as kSBM x kSBN sub-blocks of 16 x 16, one
  // v_mfma_f32_16x16x4_f32 chain each.  Lane l supplies row l % 16 and k = 4 q + l / 16 of a k quad: one ds_read_b32 per operand sub-block and quad.
  int const wr = wave >> 1, wc = wave & 1;
  float const *const a16 = sm + (wr * (TM / 2) + (lane & 15)) * kLDK + (lane >> 4);
  float const *const b16 = sm + kImgA + (wc * (TN / 2) + (lane & 15)) * kLDK + (lane >> 4);
  f32x4 acc[kSBM][kSBN];
#pragma unroll
  for (int i = 0; i < kSBM; ++i)
#pragma unroll
    for (int j = 0; j < kSBN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_barrier" ::: "memory");
  for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      float const *const A = a16 + (u % 3) * kStage, *const B = b16 + (u % 3) * kStage;
      float a[RA + 1][kSBM], b[RA + 1][kSBN];
#pragma unroll
      for (int r = 0; r < RA; ++r) {
#pragma unroll
        for (int i = 0; i < kSBM; ++i) a[r][i] = A[i * 16 * kLDK + 4 * r];
#pragma unroll
        for (int j = 0; j < kSBN; ++j) b[r][j] = B[j * 16 * kLDK + 4 * r];
      }
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        if (q + RA < kQ) {
#pragma unroll
          for (int i = 0; i < kSBM; ++i) a[(q + RA) % (RA + 1)][i] = A[i * 16 * kLDK + 4 * (q + RA)];
#pragma unroll
          for (int j = 0; j < kSBN; ++j) b[(q + RA) % (RA + 1)][j] = B[j * 16 * kLDK + 4 * (q + RA)];
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the reads RA quads ahead: left alone, the scheduler sinks them to one MFMA before their use)
#pragma unroll
        for (int i = 0; i < kSBM; ++i)
#pragma unroll
          for (int j = 0; j < kSBN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q % (RA + 1)][i], b[q % (RA + 1)][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    }
  }

  // ---- epilogue.  16x16 accumulator: row = 4 * (lane / 16) + r, column = lane % 16
  {
    rsrc_t const rD = make_rsrc(p.D, p.D_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
#pragma unroll
    for (int j = 0; j < kSBN; ++j) {
      int const oc = oc0 + wc * (TN / 2) + j * 16 + (lane & 15);
      float const bias = bload1(rB, oc * 4);             // columns past out_chan read 0 and are not stored
#pragma unroll
      for (int i = 0; i < kSBM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int const img = img0 + wr * (TM / 2) + i * 16 + 4 * (lane >> 4) + r;
          float val = acc[i][j][r] + bias;
#if RELU
          val = (val > 0.f) ? val : 0.f;
#endif
          int const off = ((oc < p.Mi) && (img < p.Nj)) ? (int)(((unsigned)img * (unsigned)p.out_ctot + (unsigned)(p.out_coff + oc)) * 4u) : kOOB;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rD, off, 0, 0);
        }
    }
  }
}

// sgemm_big_f32.hip -- fp32 MFMA SGEMM for large k-major operands on gfx950 (MI355X): 256 x 256 tiles, multiplying waves and staging waves.
//
//   c[m][n] = sum_k a[k][m] * b[k][n]      a: K x M, b: K x N, c: M x N, all row-major (the reference's sgemm: test/rtc/sgemm.cucl:17-43, src/cnn_op.cc:338-378)
//
// gemm_conv_f32.hip runs the big sizes of test/sgemm-ops-full.txt on this tile at 90 % MFMA-busy (8192^3: 140 TF/s): its eight waves multiply a K tile, then all of
// them write the next tile from registers to LDS and meet at a barrier -- the matrix pipe drains once per K step.  Here (the structure that fc_f32.hip measured out)
//   * waves 0-7 (two per SIMD, 128 x 64 outputs = 4 x 2 accumulators of v_mfma_f32_32x32x2_f32 each) ONLY read operands from LDS and multiply;
//   * waves 8-11 (one per SIMD) ONLY stage: float4 global loads of tile t + 2 + PF into registers, ds_write_b128 of tile t + 2 into the k-major LDS images
//     ([BKS][256 + 4] floats per operand: MFMA operand fetches are conflict-free ds_read_b32, lane l -> row 2 kk + l / 32, column l % 32);
//   * three LDS stages: tile t lives in stage t % 3, is written during step t - 2, and the barrier that ends a step makes the staging waves wait only for the stores
//     of the step before (s_waitcnt lgkmcnt(N)) -- a store issued while eight waves stream operand reads completes late;
//   * XCD-aware tile map as in gemm_conv_f32.hip (each XCD walks a contiguous band of tiles in groups of GROUP_I i-tiles: neighbours share panels in their L2).
// Every output is one ascending-k chain of exact fp32 fmas in one thread: bit-identical to gemm_conv_f32.hip, the oracle and the reference's golden digests.
// M % 4 == 0 and N % 4 == 0 (16-byte row segments); tile edges, K tail: out-of-range buffer offsets read 0, stores past the edges are dropped.
// Round 5: the tile is a parameter.  TBI x TBJ = 256 x 256 (above), 128 x 128 or 256 x 128; always eight multiplying waves as WI x WJ (each (TBI / WI) x (TBJ / WJ)
// = TI x TJ blocks of 32 x 32) and four staging waves.  The smaller tiles serve the sizes where 256 x 256 leaves CUs idle (2048^3 = 64 tiles; 3072^3 = 144) and the LAST
// round of the large ones (10240^3 = 1600 tiles = 6.25 rounds): same data path, same ascending-k chain per output -- bit-identical whatever the tile.  With
// 128 x 128 tiles two workgroups share a CU (MINW 2: 67 KB of LDS and 24 waves).
// Round 6: FOUR multiplying waves as well (WI x WJ = 2 x 2 | 1 x 4 | 4 x 1; 512 threads) and wave tiles of one row block -- 128 x 128 (64 x 64 per wave), 64 x 128,
// 128 x 64, 64 x 64: the tiles of the sizes that give 256 CUs less than a 128 x 128 tile each (1024^3) or whose last round is ragged (the rest launch of the two-level
// tiling).  Several such workgroups share a CU (MINW = waves per SIMD the register budget is set for).
// Compile-time parameters (-D): KNAME BKS (k per step: 16) PF (K tiles in flight in registers per staging thread: 2 | 4) GROUP_I [TBI TBJ WI WJ MINW]

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifndef GROUP_I
#define GROUP_I 8
#endif
#ifndef BKS
#define BKS 8
#endif
#ifndef PF
#define PF 2
#endif
#ifndef TBI
#define TBI 256
#endif
#ifndef TBJ
#define TBJ 256
#endif
#ifndef WI
#define WI 2     // multiplying waves along i x along j: always eight
#endif
#ifndef WJ
#define WJ 4
#endif
#ifndef MINW
#define MINW 1
#endif
#ifndef MULPRIO
#define MULPRIO 1 // round 6: the multiplying waves at priority 1, the staging waves at 0 -- in the list (tools/stgprio_ab.sh; TF/s) 144.5 -> 145.3: 4096^3 143.0 -> 144.5, 8192^3 145.9 -> 147.0, 12288^3
#endif          // 146.6 -> 147.5, 2048^3 127.5 -> 129.3; the other way round (STGPRIO) the 64 x 64 form loses 15 %
#ifndef STGPRIO
#define STGPRIO 0
#endif
#ifndef NSTG
#define NSTG 4 // LDS stages: tile t lives in stage t % NSTG and is written during step t - (NSTG - 1).  4: a tile is complete one barrier before its step, so the
#endif         // multiplying waves fetch its first operands BEFORE the barrier that ends the previous step (nothing but the barrier itself between two steps)

struct gemm_args_t { // same layout as gemm_conv_f32.hip (one host-side struct serves all fp32 kernels)
  float const *I; float const *J; float *D; float const *bias;   // I = a (K x Mi), J = b (K x Nj), D = c (Mi x Nj)
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

namespace {
constexpr int kNMW = WI * WJ;                 // multiplying waves
static_assert(kNMW == 8 || kNMW == 4, "eight (or four) multiplying waves");
constexpr int kTI = TBI / (WI * 32), kTJ = TBJ / (WJ * 32);   // 32 x 32 blocks per wave
static_assert(TBI % (WI * 32) == 0 && TBJ % (WJ * 32) == 0 && (kTI == 4 || kTI == 2 || kTI == 1) && (kTJ == 2 || kTJ == 1), "wave tile: 4 | 2 | 1 row blocks x 2 | 1 column blocks");
constexpr int kLDI = TBI + 4, kLDJ = TBJ + 4; // floats per k row of the a / b image
constexpr int kImgI = BKS * kLDI, kImgJ = BKS * kLDJ, kImg2 = kImgI + kImgJ;   // floats per operand image; per stage
constexpr int kNLI = BKS * (TBI / 4) / 256, kNLJ = BKS * (TBJ / 4) / 256;      // float4s per staging thread, operand and step (BKS rows x TB / 4 units over 256 threads)
static_assert((BKS * (TBI / 4)) % 256 == 0 && (BKS * (TBJ / 4)) % 256 == 0, "whole float4 units per staging thread");
constexpr int kKK = BKS / 2;                  // MFMA k pairs per step
constexpr int kU = (NSTG == 3) ? 3 * PF : ((PF > NSTG) ? PF : NSTG);   // steps per unrolled round: a multiple of NSTG and PF (stage = step % NSTG, register set = step % PF: compile-time)
constexpr int kD = NSTG - 1;                  // a tile is written kD steps ahead of its step
static_assert(NSTG == 3 || NSTG == 4, "NSTG: 3 | 4");
constexpr int kOOB = (int)0x80000000;
static_assert(BKS % 4 == 0 && (PF == 2 || PF == 4), "BKS % 4 == 0, PF: 2 | 4");
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(float const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
} // namespace

#if (TBI / (WI * 32)) == 4
typedef f32x4 avec_t;
#elif (TBI / (WI * 32)) == 2
typedef float2 avec_t;
#else
typedef float avec_t;
#endif
#if (TBJ / (WJ * 32)) == 2
typedef float2 bvec_t;
#else
typedef float bvec_t;
#endif
namespace {
__device__ __forceinline__ float vget(f32x4 const &v, int i) { return v[i]; }
__device__ __forceinline__ float vget(float2 const &v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ float vget(float const &v, int) { return v; }
} // namespace

extern "C" __global__ __launch_bounds__((kNMW + 4) * 64, MINW) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) float sm[NSTG * kImg2];   // [stage][operand: a (MFMA A, rows i: kImgI floats), then b (MFMA B, columns j)][k][kLDI | kLDJ]
  int const lane = threadIdx.x & 63;
  int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bool const stager = wave >= kNMW;

  int tile_i, tile_j;
  {
    int const bid = blockIdx.x, nb = p.tiles_i * p.tiles_j;
    int const q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int const group_sz = GROUP_I * p.tiles_j, gid = nid / group_sz, first_i = gid * GROUP_I;
    int const gsz = min(p.tiles_i - first_i, GROUP_I), in_g = nid - gid * group_sz;
    tile_i = first_i + in_g % gsz; tile_j = in_g / gsz;
  }
  int const i0 = tile_i * TBI, j0 = tile_j * TBJ;
  int const nkt = (p.K + BKS - 1) / BKS;
  int const nkt_pad = (nkt + kU - 1) / kU * kU;   // whole rounds, no conditions inside the loops: tiles past the last one read zeros and multiply as +0

  if (stager) {
#if STGPRIO
    __builtin_amdgcn_s_setprio(STGPRIO);   // the staging waves ahead of the multiplying waves in the SIMD's issue arbitration (priority, then age: a co-resident younger workgroup's staging waves otherwise get the leftover slots)
#endif
    int const tid = threadIdx.x - kNMW * 64;
    rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);
    // unit c = tid + n * 256 of an operand tile: k row c / (TB / 4), columns 4 (c % (TB / 4)) .. +3
    int goffI[kNLI], goffJ[kNLJ], loffI[kNLI], loffJ[kNLJ], krowI[kNLI], krowJ[kNLJ];
#pragma unroll
    for (int n = 0; n < kNLI; ++n) {
      int const c = tid + n * 256, kr = c / (TBI / 4), x = 4 * (c % (TBI / 4));
      krowI[n] = kr; loffI[n] = kr * kLDI + x;
      goffI[n] = (i0 + x < p.Mi) ? ((kr * p.ldI + i0 + x) * 4) : kOOB;
    }
#pragma unroll
    for (int n = 0; n < kNLJ; ++n) {
      int const c = tid + n * 256, kr = c / (TBJ / 4), x = 4 * (c % (TBJ / 4));
      krowJ[n] = kr; loffJ[n] = kImgI + kr * kLDJ + x;
      goffJ[n] = (j0 + x < p.Nj) ? ((kr * p.ldJ + j0 + x) * 4) : kOOB;
    }
    // (K tail: rows past K must read zeros -- the range check only covers the end of the tensor, so the row index is tested; the per-tile part of the offset,
    //  kt * BKS rows, goes through the scalar offset operand -- clamped to the last real tile, so that it stays inside the tensor (< 2^31) for the padded steps too)
    auto gloadI = [&](int n, int kt) -> f32x4 { return bload4(rI, (kt * BKS + krowI[n] < p.K) ? goffI[n] : kOOB, min(kt, nkt - 1) * (BKS * 4) * p.ldI); };
    auto gloadJ = [&](int n, int kt) -> f32x4 { return bload4(rJ, (kt * BKS + krowJ[n] < p.K) ? goffJ[n] : kOOB, min(kt, nkt - 1) * (BKS * 4) * p.ldJ); };
    auto lstoreI = [&](int n, int stage, f32x4 const &v) { *reinterpret_cast<f32x4 *>(sm + stage * kImg2 + loffI[n]) = v; };
    auto lstoreJ = [&](int n, int stage, f32x4 const &v) { *reinterpret_cast<f32x4 *>(sm + stage * kImg2 + loffJ[n]) = v; };
    f32x4 ringI[PF][kNLI], ringJ[PF][kNLJ];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int n = 0; n < kNLI; ++n) ringI[u][n] = gloadI(n, u);
#pragma unroll
      for (int n = 0; n < kNLJ; ++n) ringJ[u][n] = gloadJ(n, u);
    }
#pragma unroll
    for (int t = 0; t < kD; ++t) {   // tiles 0 .. kD - 1 go to their stages before the first step; their register sets take tiles PF ..
#pragma unroll
      for (int n = 0; n < kNLI; ++n) { lstoreI(n, t, ringI[t % PF][n]); ringI[t % PF][n] = gloadI(n, t + PF); }
#pragma unroll
      for (int n = 0; n < kNLJ; ++n) { lstoreJ(n, t, ringJ[t % PF][n]); ringJ[t % PF][n] = gloadJ(n, t + PF); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
#pragma unroll
        for (int n = 0; n < kNLI; ++n) { lstoreI(n, (u + kD) % NSTG, ringI[(u + kD) % PF][n]); ringI[(u + kD) % PF][n] = gloadI(n, kb + u + kD + PF); }
#pragma unroll
        for (int n = 0; n < kNLJ; ++n) { lstoreJ(n, (u + kD) % NSTG, ringJ[(u + kD) % PF][n]); ringJ[(u + kD) % PF][n] = gloadJ(n, kb + u + kD + PF); }
        asm volatile("s_waitcnt lgkmcnt(%0)\n s_barrier" :: "n"(kNLI + kNLJ) : "memory");
      }
    }
    return;
  }

  // ---- multiplying waves: wave w owns rows [wi * kTI * 32, +kTI * 32) x columns [wj * kTJ * 32, +kTJ * 32) of the tile (256 x 256: 128 x 64).  Which row of the tile an
  // MFMA row stands for is free: row rho of row block t is tile row kTI rho + t, column kappa of column block u is tile column kTJ kappa + u.  Then a lane's kTI A
  // operands of a k (and its kTJ B operands) are CONTIGUOUS in the k-major LDS image: one ds_read_b128 + one ds_read_b64 per k pair (256 x 256) instead of six
  // ds_read_b32 (which the compiler pairs into ds_read2_b32 with 8-bit offsets and a base register per (stage, k pair): 48 address registers, spills), and the
  // column blocks of a row leave as one store.
#if MULPRIO
  __builtin_amdgcn_s_setprio(MULPRIO);
#endif
  int const wi = wave / WJ, wj = wave % WJ;
  float const *const a_base = sm + (lane >> 5) * kLDI + wi * (kTI * 32) + kTI * (lane & 31);        // + stage * kImg2 + kk * 2 * kLDI
  float const *const b_base = sm + kImgI + (lane >> 5) * kLDJ + wj * (kTJ * 32) + kTJ * (lane & 31);
  f32x16 acc[kTI][kTJ];
#pragma unroll
  for (int t = 0; t < kTI; ++t)
#pragma unroll
    for (int u = 0; u < kTJ; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  asm volatile("s_barrier" ::: "memory");
  avec_t a[2]; bvec_t b[2];                               // operands of k pair kk and kk + 1: the reads run one pair (kTI x kTJ MFMAs) ahead
  a[0] = *reinterpret_cast<avec_t const *>(a_base); b[0] = *reinterpret_cast<bvec_t const *>(b_base);
  static_assert(kKK % 2 == 0, "an even number of k pairs per step (the operand double buffer carries over from step to step)");
  for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
    for (int s = 0; s < kU; ++s) {
      float const *const A = a_base + (s % NSTG) * kImg2, *const B = b_base + (s % NSTG) * kImg2;
      float const *const An = a_base + ((s + 1) % NSTG) * kImg2, *const Bn = b_base + ((s + 1) % NSTG) * kImg2;
#pragma unroll
      for (int kk = 0; kk < kKK; ++kk) {
        if (kk + 1 < kKK) { a[(kk + 1) & 1] = *reinterpret_cast<avec_t const *>(A + (kk + 1) * 2 * kLDI); b[(kk + 1) & 1] = *reinterpret_cast<bvec_t const *>(B + (kk + 1) * 2 * kLDJ); }
#if NSTG == 4
        else { a[0] = *reinterpret_cast<avec_t const *>(An); b[0] = *reinterpret_cast<bvec_t const *>(Bn); }   // the next tile's first pair, before the barrier
#endif
        __builtin_amdgcn_sched_barrier(0);
        avec_t const ca = a[kk & 1]; bvec_t const cb = b[kk & 1];
#pragma unroll
        for (int t = 0; t < kTI; ++t)
#pragma unroll
          for (int u = 0; u < kTJ; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget(ca, t), vget(cb, u), acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#if NSTG == 4
      asm volatile("s_barrier" ::: "memory");
#else
      asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
      a[0] = *reinterpret_cast<avec_t const *>(An); b[0] = *reinterpret_cast<bvec_t const *>(Bn);
#endif
    }
  }

  // ---- epilogue: MFMA row rho = 8 * (r / 4) + r % 4 + 4 * (lane / 32) of row block t is tile row kTI rho + t; column kappa = lane % 32 of column block u is tile
  // column kTJ kappa + u: one store of kTJ floats per (t, r) -- 256 (kTJ = 2) or 128 contiguous bytes per half wave
  {
    rsrc_t const rD = make_rsrc(p.D, p.D_bytes);
    int const j = j0 + wj * (kTJ * 32) + kTJ * (lane & 31);
    int const il = i0 + wi * (kTI * 32) + kTI * 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < kTI; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int const i = il + kTI * ((r & 3) + 8 * (r >> 2)) + t;
        int const off = ((i < p.Mi) && (j < p.Nj)) ? ((i * p.ldD + j) * 4) : kOOB;   // (N % 4 == 0: a column pair is inside or outside as a whole)
#if (TBJ / (WJ * 32)) == 2
        float2 const v = float2{acc[t][0][r], acc[t][1][r]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rD, off, 0, 0);
#else
        float const v = acc[t][0][r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rD, off, 0, 0);
#endif
      }
  }
}

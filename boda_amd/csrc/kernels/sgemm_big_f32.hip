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
// Compile-time parameters (-D): KNAME BKS (k per step: 16) PF (K tiles in flight in registers per staging thread: 2 | 4) GROUP_I

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
constexpr int kLD = 256 + 4;                  // floats per k row of an LDS image
constexpr int kImg = BKS * kLD;               // floats per operand image
constexpr int kNL = BKS * 64 / 256;           // float4s per staging thread, operand and step (BKS rows x 64 units over 256 threads)
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

extern "C" __global__ __launch_bounds__(768, 1) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) float sm[NSTG * 2 * kImg];   // [stage][operand: 0 = a (MFMA A, rows i), 1 = b (MFMA B, columns j)][k][kLD]
  int const lane = threadIdx.x & 63;
  int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bool const stager = wave >= 8;

  int tile_i, tile_j;
  {
    int const bid = blockIdx.x, nb = p.tiles_i * p.tiles_j;
    int const q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int const group_sz = GROUP_I * p.tiles_j, gid = nid / group_sz, first_i = gid * GROUP_I;
    int const gsz = min(p.tiles_i - first_i, GROUP_I), in_g = nid - gid * group_sz;
    tile_i = first_i + in_g % gsz; tile_j = in_g / gsz;
  }
  int const i0 = tile_i * 256, j0 = tile_j * 256;
  int const nkt = (p.K + BKS - 1) / BKS;
  int const nkt_pad = (nkt + kU - 1) / kU * kU;   // whole rounds, no conditions inside the loops: tiles past the last one read zeros and multiply as +0

  if (stager) {
    int const tid = threadIdx.x - 512;
    rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);
    // unit c = tid + n * 256 of an operand tile: k row c / 64, columns 4 (c % 64) .. +3
    int goffI[kNL], goffJ[kNL], loff[kNL], krow[kNL];
#pragma unroll
    for (int n = 0; n < kNL; ++n) {
      int const c = tid + n * 256, kr = c >> 6, x = 4 * (c & 63);
      krow[n] = kr; loff[n] = kr * kLD + x;
      goffI[n] = (i0 + x < p.Mi) ? ((kr * p.ldI + i0 + x) * 4) : kOOB;
      goffJ[n] = (j0 + x < p.Nj) ? ((kr * p.ldJ + j0 + x) * 4) : kOOB;
    }
    // (K tail: rows past K must read zeros -- the range check only covers the end of the tensor, so the row index is tested; the per-tile part of the offset,
    //  kt * BKS rows, goes through the scalar offset operand -- clamped to the last real tile, so that it stays inside the tensor (< 2^31) for the padded steps too)
    auto gloadI = [&](int n, int kt) -> f32x4 { return bload4(rI, (kt * BKS + krow[n] < p.K) ? goffI[n] : kOOB, min(kt, nkt - 1) * (BKS * 4) * p.ldI); };
    auto gloadJ = [&](int n, int kt) -> f32x4 { return bload4(rJ, (kt * BKS + krow[n] < p.K) ? goffJ[n] : kOOB, min(kt, nkt - 1) * (BKS * 4) * p.ldJ); };
    auto lstore = [&](int op, int n, int stage, f32x4 const &v) { *reinterpret_cast<f32x4 *>(sm + (stage * 2 + op) * kImg + loff[n]) = v; };
    f32x4 ringI[PF][kNL], ringJ[PF][kNL];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int n = 0; n < kNL; ++n) { ringI[u][n] = gloadI(n, u); ringJ[u][n] = gloadJ(n, u); }
#pragma unroll
    for (int t = 0; t < kD; ++t)     // tiles 0 .. kD - 1 go to their stages before the first step; their register sets take tiles PF ..
#pragma unroll
      for (int n = 0; n < kNL; ++n) {
        lstore(0, n, t, ringI[t % PF][n]); ringI[t % PF][n] = gloadI(n, t + PF);
        lstore(1, n, t, ringJ[t % PF][n]); ringJ[t % PF][n] = gloadJ(n, t + PF);
      }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
    for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
#pragma unroll
        for (int n = 0; n < kNL; ++n) {
          lstore(0, n, (u + kD) % NSTG, ringI[(u + kD) % PF][n]); ringI[(u + kD) % PF][n] = gloadI(n, kb + u + kD + PF);
          lstore(1, n, (u + kD) % NSTG, ringJ[(u + kD) % PF][n]); ringJ[(u + kD) % PF][n] = gloadJ(n, kb + u + kD + PF);
        }
        asm volatile("s_waitcnt lgkmcnt(%0)\n s_barrier" :: "n"(2 * kNL) : "memory");
      }
    }
    return;
  }

  // ---- multiplying waves: wave w owns rows [wi * 128, +128) x columns [wj * 64, +64) of the tile.  Which row of the tile an MFMA row stands for is free: row rho of
  // row block t is tile row 4 rho + t, column kappa of column block u is tile column 2 kappa + u.  Then a lane's four A operands of a k (and its two B operands) are
  // CONTIGUOUS in the k-major LDS image: one ds_read_b128 + one ds_read_b64 per k pair instead of six ds_read_b32 (which the compiler pairs into ds_read2_b32 with
  // 8-bit offsets and a base register per (stage, k pair): 48 address registers, spills), and the two column blocks of a row leave as one 8-byte store.
  int const wi = wave >> 2, wj = wave & 3;
  float const *const a_base = sm + (lane >> 5) * kLD + wi * 128 + 4 * (lane & 31);        // + stage * 2 * kImg + kk * 2 * kLD
  float const *const b_base = sm + kImg + (lane >> 5) * kLD + wj * 64 + 2 * (lane & 31);
  f32x16 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  asm volatile("s_barrier" ::: "memory");
  f32x4 a[2]; float2 b[2];                               // operands of k pair kk and kk + 1: the reads run one pair (eight MFMAs) ahead
  a[0] = *reinterpret_cast<f32x4 const *>(a_base); b[0] = *reinterpret_cast<float2 const *>(b_base);
  static_assert(kKK % 2 == 0, "an even number of k pairs per step (the operand double buffer carries over from step to step)");
  for (int kb = 0; kb < nkt_pad; kb += kU) {
#pragma unroll
    for (int s = 0; s < kU; ++s) {
      float const *const A = a_base + (s % NSTG) * 2 * kImg, *const B = b_base + (s % NSTG) * 2 * kImg;
      float const *const An = a_base + ((s + 1) % NSTG) * 2 * kImg, *const Bn = b_base + ((s + 1) % NSTG) * 2 * kImg;
#pragma unroll
      for (int kk = 0; kk < kKK; ++kk) {
        if (kk + 1 < kKK) { a[(kk + 1) & 1] = *reinterpret_cast<f32x4 const *>(A + (kk + 1) * 2 * kLD); b[(kk + 1) & 1] = *reinterpret_cast<float2 const *>(B + (kk + 1) * 2 * kLD); }
#if NSTG == 4
        else { a[0] = *reinterpret_cast<f32x4 const *>(An); b[0] = *reinterpret_cast<float2 const *>(Bn); }   // the next tile's first pair, before the barrier
#endif
        __builtin_amdgcn_sched_barrier(0);
        f32x4 const ca = a[kk & 1]; float2 const cb = b[kk & 1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[t], cb.x, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[t], cb.y, acc[t][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#if NSTG == 4
      asm volatile("s_barrier" ::: "memory");
#else
      asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
      a[0] = *reinterpret_cast<f32x4 const *>(An); b[0] = *reinterpret_cast<float2 const *>(Bn);
#endif
    }
  }

  // ---- epilogue: MFMA row rho = 8 * (r / 4) + r % 4 + 4 * (lane / 32) of row block t is tile row 4 rho + t; column kappa = lane % 32 of column block u is tile
  // column 2 kappa + u: one 8-byte store per (t, r), 256 contiguous bytes per half wave
  {
    rsrc_t const rD = make_rsrc(p.D, p.D_bytes);
    int const j = j0 + wj * 64 + 2 * (lane & 31);
    int const il = i0 + wi * 128 + 16 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int const i = il + 4 * ((r & 3) + 8 * (r >> 2)) + t;
        int const off = ((i < p.Mi) && (j < p.Nj)) ? ((i * p.ldD + j) * 4) : kOOB;   // (N % 4 == 0: a column pair is inside or outside as a whole)
        float2 const v = float2{acc[t][0][r], acc[t][1][r]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rD, off, 0, 0);
      }
  }
}

// k1_stream_f32.hip -- streaming fp32 MFMA kernel for short-K 1x1 convolutions on gfx950 (MI355X), specialised by hiprtc.
//
//   out[img][out_chan][pel] = act( sum_c filts[out_chan][c] * in[img][c][pel] + bias[out_chan] ),   1x1 kernel, stride 1, no padding
//   (the reference's k1conv case: src/cnn_op.cc:51-60, test/rtc/k1conv.cucl; epilogue src/cnn_codegen.cc:35-42)
//
// Why a second kernel: with in_chan <= ~128 these layers sit at the fp32 ridge point (NiN cccp1/2: 24 flop/B, ResNet-50 res2
// 64<->256: 26 flop/B) -- every input and output element crosses HBM exactly once and there are only 2-4 K steps per output tile,
// so the tiled kernel (gemm_conv_f32.hip: stage -> barrier -> multiply -> store per workgroup) spends its time in prologues and
// epilogues: 3.3 TB/s of algorithmic traffic measured.  Here nothing is tiled in K at all:
//   * the whole filter block (all in_chans x the workgroup's out_chans, <= 64 KB) is put into LDS once, k-major, with the biases;
//     after that single barrier the waves never synchronise again;
//   * workgroups are persistent: each wave walks over blocks of CB*32 pels (all images form one flat pel axis; a lane derives
//     (img, pel) itself, so blocks may straddle images) with a fixed stride;
//   * the input is never staged: lane l's MFMA B operand for K step s is in[img][2s + l/32][pel(l%32)] -- a plain dword buffer load
//     (32 lanes = 128 contiguous bytes of one channel plane).  A block's KC*CB/2 operand registers are loaded in one burst; as soon
//     as step s of the current block has been multiplied its register is re-loaded with step s of the wave's NEXT block, so the HBM
//     stream of block n+1 runs under the MFMAs and the stores of block n;
//   * A operands are conflict-free ds_read_b32 of the resident filter image; accumulators leave through the same coalesced
//     buffer-store epilogue as the tiled kernel (bias add, ReLU, optional channel slice of a wider output).
// Numerics: identical to the tiled kernel -- one v_mfma_f32_32x32x2_f32 chain per output in ascending in_chan, then + bias, then
// ReLU: bit-identical to the oracle.  An odd in_chan count is padded with a zero filter row and a zero (out-of-range) load.
//
// Compile-time parameters (-D): KNAME KC (in_chans) HW (pels per image plane) WI WJ (waves along out_chan / pel) OCB (32-row blocks
// per wave) CB (32-pel blocks per wave) MINW RELU EDGE_OC

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef ABLATE
#define ABLATE 0 // experiment hook (BODAHIP_EXTRA_DEFS): 1 no stores | 2 no input loads | 4 no MFMAs
#endif

struct gemm_args_t { // same layout as gemm_conv_f32.hip (one host-side struct serves both)
  float const *I; float const *J; float *D; float const *bias;
  int Mi, Nj, K;
  int ldI, ldJ, ldD;
  int C, H, W, OH, OW;
  int tiles_i, tiles_j;       // out_chan tiles | pel super-blocks (WJ*CB*32 pels each)
  int splitk, kt_per;         // kt_per: workgroups per out_chan tile == the super-block stride of a workgroup
  float *ws; long ws_slab;
  unsigned I_bytes, J_bytes;
  unsigned D_bytes;
  int out_ctot, out_coff;
  int const *ktab; int ktab_n;
  long bsI, bsJ, bsD; // (batched sgemm launches of gemm_conv_f32.hip only)
};

namespace {
constexpr int kNT = WI * WJ * 64;
constexpr int kOCT = WI * OCB * 32;          // out_chans per workgroup
constexpr int kKP = (KC + 1) / 2 * 2;        // in_chans padded to whole MFMA K steps
constexpr int kSteps = kKP / 2;
constexpr int kLD = kOCT | 1;                // LDS pitch of one k row (odd: the transposing stores of the staging pass spread over the banks)
constexpr int kSB = WJ * CB * 32;            // pels per super-block
constexpr int kOOB = (int)0x80000000;
#ifndef EDGE_OC
#define EDGE_OC 1 // 0: out_chan is a multiple of the workgroup's out_chan tile (no per-row range test in the stores)
#endif
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(float const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float bload1(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
// Resident filter image Fs[k][oc] = filts[row0 + oc][k] (ROWS x KCC, k-major with pitch LD; rows past the tensor and the pad row of an odd KCC are zero).  The filter
// block is one contiguous run of floats: every thread issues ALL its 16-byte loads first, branch-free, and transposes into LDS afterwards -- one memory round trip for
// the whole image.  (The first form -- one dword load, wait, one LDS store per element -- was a serial chain of 36-72 round trips per thread: 25-50 us at the head of
// an 85-150 us kernel; a branch per vector serialises the same way, the compiler drains vmcnt at every join.)  A vector is wholly inside the valid run or not loaded;
// the up to three floats behind the last whole vector go dword by dword: nothing relies on per-dword range checks.
template <int ROWS, int KCC, int LD, int NT>
__device__ __forceinline__ void stage_filters(float *Fs, rsrc_t const rI, int const row0, int const rows_valid, int const tid) {
  constexpr int kN4 = (ROWS * KCC + 3) / 4, kNB = (kN4 + NT - 1) / NT;
  int const nvalid = ((rows_valid < 0) ? 0 : ((rows_valid > ROWS) ? ROWS : rows_valid)) * KCC, t0 = nvalid & ~3;
  f32x4 fv[kNB];
#pragma unroll
  for (int b = 0; b < kNB; ++b) {
    int const e0 = 4 * (tid + b * NT);
    fv[b] = bload4(rI, (e0 + 4 <= nvalid) ? ((row0 * KCC + e0) * 4) : kOOB, 0);
  }
  float const tail = bload1(rI, (t0 + tid < nvalid) ? ((row0 * KCC + t0 + tid) * 4) : kOOB, 0);
  if (KCC & 1) for (int e = tid; e < ROWS; e += NT) Fs[KCC * LD + e] = 0.f;
#pragma unroll
  for (int b = 0; b < kNB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int const e = 4 * (tid + b * NT) + j, oc = e / KCC, k = e - oc * KCC;
      if ((e < ROWS * KCC) && !((e >= t0) && (e < nvalid))) Fs[k * LD + oc] = fv[b][j];
    }
  if (t0 + tid < nvalid) { int const e = t0 + tid, oc = e / KCC, k = e - oc * KCC; Fs[k * LD + oc] = tail; }
}
} // namespace

extern "C" __global__ __launch_bounds__(WI * WJ * 64, MINW) void KNAME(gemm_args_t const p) {
  __shared__ float Fs[kKP * kLD + kOCT];
  float *const Bs = Fs + kKP * kLD;
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave / WJ, wj = wave % WJ;
  int const tile_i = blockIdx.x % p.tiles_i, wg_j = blockIdx.x / p.tiles_i;
  int const oc0 = tile_i * kOCT;

  // ---- resident filter image: Fs[k][oc] = filts[oc0 + oc][k] (zero rows / columns past the tensor), Bs[oc] = bias
  {
    rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
    stage_filters<kOCT, KC, kLD, kNT>(Fs, rI, oc0, p.Mi - oc0, tid);
    for (int e = tid; e < kOCT; e += kNT) Bs[e] = bload1(rB, (oc0 + e) * 4, 0); // rows past out_chan read 0 (range-checked)
  }
  __syncthreads();

  rsrc_t const rJ = make_rsrc(p.J, p.J_bytes), rD = make_rsrc(p.D, p.D_bytes);
  bool const hi = (lane >> 5) != 0;                    // lanes 32-63 carry the odd in_chan of a K step
  float const *const a_base = Fs + (hi ? kLD : 0) + wi * (OCB * 32) + (lane & 31);

  // byte offset of in[img][hi ? 1 : 0][pel] for this lane's pel of column block cb of super-block sb (kOOB past the last pel)
  auto in_off = [&](int sb, int cb) -> int {
    int const jg = (sb * WJ + wj) * (CB * 32) + cb * 32 + (lane & 31);
    int const img = jg / HW, pel = jg - img * HW;
    return (sb < p.tiles_j && jg < p.Nj) ? (((img * KC + (hi ? 1 : 0)) * HW + pel) * 4) : kOOB;
  };
  auto load_step = [&](int s, int off) -> float { // B operand of K step s: in_chan 2s (+1 for the upper lanes); the padding chan of an odd KC reads 0
    bool const pad_k = (KC & 1) && (s == kSteps - 1);
#if ABLATE & 2
    return (float)(off + s);
#else
    return bload1(rJ, (pad_k && hi) ? kOOB : off, s * (2 * HW * 4));
#endif
  };

  // One block = one pass over all K steps into one accumulator set.  Two sets alternate: while block n is multiplied into one, the
  // finished block n-1 leaves from the other, a few stores per K step slotted between the MFMAs -- so a wave keeps the matrix pipe
  // busy through its own epilogues (with one set the waves of a SIMD, which run in step, all stop multiplying at the same time).
  constexpr int kT = OCB * CB * 16;                         // accumulator registers (= stores) per block
  constexpr int kEPS = (kT + kSteps - 1) / kSteps;          // stores slotted into one K step
  unsigned const S4 = (unsigned)HW * 4u;
  int const row0 = wi * (OCB * 32) + 4 * (lane >> 5);
  // byte offset of out[img][out_coff + oc0 + row0][pel] for this lane's pel of column block cb of super-block sb; kOOB (the store is
  // dropped by the range check) past the last pel or when there is no such block
  auto out_off = [&](int sb, int cb) -> int {
    int const jg = (sb * WJ + wj) * (CB * 32) + cb * 32 + (lane & 31);
    int const img = jg / HW, pel = jg - img * HW;
    return (sb >= 0 && jg < p.Nj) ? (int)((((unsigned)img * (unsigned)p.out_ctot + (unsigned)(p.out_coff + oc0 + row0)) * (unsigned)HW + (unsigned)pel) * 4u) : kOOB;
  };
  auto elem_rc = [](int e) { int const r = e % 16, rb = (e / 16) % OCB; return rb * 32 + (r & 3) + 8 * (r >> 2); }; // wave-uniform part of the row of store e
  auto store_elem = [&](f32x16 const (&acc)[OCB][CB], int const (&jpart)[CB], int e, float bias) { // e = (cb*OCB + rb)*16 + r, compile-time after unrolling
    int const r = e % 16, rb = (e / 16) % OCB, cb = e / (16 * OCB), rc = elem_rc(e);
    if (EDGE_OC && (oc0 + row0 + rc >= p.Mi)) return;
    float v = acc[rb][cb][r] + bias;
#if RELU
    v = (v > 0.f) ? v : 0.f;
#endif
#if ABLATE & 1
    if (v == 123.456f)
#endif
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rD, jpart[cb], (int)((unsigned)rc * S4), 0);
  };

  float bq[kSteps][CB];
  {
    int off[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) off[cb] = in_off(wg_j, cb);
#pragma unroll
    for (int s = 0; s < kSteps; ++s)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bq[s][cb] = load_step(s, off[cb]);
  }
  // multiply super-block sb into `acc` (refilling bq from super-block sb + stride), draining `prev` (super-block sb_prev, -1: none)
  auto block = [&](f32x16 (&acc)[OCB][CB], f32x16 const (&prev)[OCB][CB], int sb, int sb_prev) {
    int offn[CB], jprev[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) { offn[cb] = in_off(sb + p.kt_per, cb); jprev[cb] = out_off(sb_prev, cb); }
#pragma unroll
    for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    // Per K step, in issue order: (1) LDS reads for LATER use -- the A operands of step s+1 and the biases of this step's stores;
    // (2) this step's MFMAs (A operands fetched a step ago); (3) the refill load and this step's share of the previous block's
    // stores.  Every LDS / memory wait then falls behind a full step of MFMAs.
    float a[OCB], an[OCB];
#pragma unroll
    for (int rb = 0; rb < OCB; ++rb) a[rb] = a_base[rb * 32];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      float bv[kEPS];
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb) an[rb] = a_base[((s + 1 < kSteps) ? (s + 1) : 0) * 2 * kLD + rb * 32];
#pragma unroll
      for (int i = 0; i < kEPS; ++i) bv[i] = Bs[row0 + elem_rc((s * kEPS + i < kT) ? (s * kEPS + i) : 0)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
#if ABLATE & 8
          acc[rb][cb][s & 15] += bq[s][cb];   // memory pattern only: loads consumed by one add, no LDS operand, no MFMA
#elif ABLATE & 4
          acc[rb][cb][s & 15] += a[rb] * bq[s][cb];
#else
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb], bq[s][cb], acc[rb][cb], 0, 0, 0);
#endif
        }
      __builtin_amdgcn_sched_barrier(0); // keep the refill behind this step's MFMAs (hoisted, it would need a second register set)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bq[s][cb] = load_step(s, offn[cb]); // refill with the next block's step s (zeros when there is none)
#pragma unroll
      for (int i = 0; i < kEPS; ++i) if (s * kEPS + i < kT) store_elem(prev, jprev, s * kEPS + i, bv[i]);
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb) a[rb] = an[rb];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto drain = [&](f32x16 const (&acc)[OCB][CB], int sb) {
    int jl[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) jl[cb] = out_off(sb, cb);
#pragma unroll
    for (int e = 0; e < kT; ++e) store_elem(acc, jl, e, Bs[row0 + elem_rc(e)]);
  };

  f32x16 accA[OCB][CB], accB[OCB][CB];
  int sb = wg_j, sb_prev = -1;
  if (sb < p.tiles_j) {
    for (;;) {
      block(accA, accB, sb, sb_prev);
      sb_prev = sb; sb += p.kt_per;
      if (sb >= p.tiles_j) { drain(accA, sb_prev); break; }
      block(accB, accA, sb, sb_prev);
      sb_prev = sb; sb += p.kt_per;
      if (sb >= p.tiles_j) { drain(accB, sb_prev); break; }
    }
  }
}

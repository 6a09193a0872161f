// k1_quad_f32.hip -- streaming fp32 MFMA kernel for short-K 1x1 convolutions on NCHW planes, 16 bytes per lane both ways (gfx950 / MI355X).
//
//   out[img][out_chan][pel] = act( sum_c filts[out_chan][c] * in[img][c][pel] + bias[out_chan] ),   1x1 kernel, stride 1, no padding
//   (the reference's k1conv case: src/cnn_op.cc:51-60, test/rtc/k1conv.cucl; epilogue src/cnn_codegen.cc:35-42)
//
// NiN cccp1/2 (96 -> 96 on 55x55 planes, 24 flop/B) are the HBM-bound layers of the fp32 lists.  On planes whose byte size is not a
// multiple of 128 the access pattern of the other two kernels -- 128-byte (256-byte when paired) row segments per instruction -- is
// what holds them at 3.4-3.8 TB/s: tools/mem_pattern_probe.py, reads + writes of that layout with no arithmetic: 4 bytes per lane
// 2.9-3.0 TB/s, 8 bytes 3.9-4.0, 16 bytes 4.1-4.3.  This kernel moves 16 bytes per lane in both directions by choosing which pel an
// MFMA column stands for: a wave's tile is OCB*32 out_chans x 128 consecutive pels of ONE image, computed as four 32-column blocks, and
// column c of block cb is pel 4c + cb.  Then
//   * lane l's B operands of K step s for all four blocks are ONE dwordx4 load: in[img][2s + l/32][p0 .. p0+3], p0 = 4 (l % 32) -- 32
//     lanes read 512 contiguous bytes of a channel plane;
//   * lane l's accumulators [cb = 0..3][r] are four consecutive pels of one output row: ONE dwordx4 store per (row block, r), 512
//     contiguous bytes per half wave;
//   * the last block of an image clamps p0 to HW - 4: its surplus columns recompute (and re-store, with identical values) the image's last
//     four pels, so there is no tail path, no partial vector and nothing is ever read or written outside a plane.
// Everything else is the streaming design of k1_stream_f32.hip: the filter block (all in_chans x the workgroup's out_chans) goes k-major
// into LDS once, after that single barrier the waves never synchronise; workgroups are persistent; the input is never staged -- a ring
// of RING K steps of operand loads per wave runs ahead of the MFMAs, across block boundaries.  One accumulator set per wave (OCB*64
// registers): two waves per SIMD cover each other's epilogues.  Workgroup ids are mapped so that one XCD walks neighbouring blocks (the
// partial cache lines at block boundaries meet in one L2).
// Numerics: one v_mfma_f32_32x32x2_f32 chain per output in ascending in_chan, then + bias, then ReLU -- bit-identical to the tiled
// kernel and the oracle (which pel a column stands for does not enter the arithmetic).  An odd in_chan count is padded with a zero
// filter row and a zero (out-of-range) load.
//
// Compile-time parameters (-D): KNAME KC (in_chans) HW (pels per image plane, >= 4) WJ (waves per workgroup) OCB (32-row blocks per wave)
// RING (K steps in flight, divides the K step count) MINW RELU EDGE_OC
//
// CHAIN=1 (with MID, OCB2, RELU2): TWO 1x1 convolutions back to back, the intermediate tensor never leaves the registers (NiN's cccp1 -> cccp2;
// the reference chains them through memory, one launch per op: src/rtc_fwd.cc:495-503, 545-549):
//   out[img][oc2][pel] = act2( sum_m filts2[oc2][m] * act1( sum_c filts[m][c] * in[img][c][pel] + bias[m] ) + bias2[oc2] ),   m < MID = the first conv's out_chans
// A wave's finished first tile IS the second convolution's B operand, up to a half exchange: accumulator register r of a 32-row block holds row 8(r/4) + r%4 in
// lanes 0-31 and that row + 4 in lanes 32-63; the MFMA wants rows 2s | 2s+1.  v_permlane32_swap on the register pair (4g, 4g+1) yields rows (8g | 8g+1) and
// (8g+4 | 8g+5), on (4g+2, 4g+3) rows (8g+2 | 8g+3) and (8g+6 | 8g+7): 24 swaps per 32 x 32 block, in place, no LDS, no extra registers.  The second K loop runs
// one 32-row block of its out_chans at a time (64 more accumulator registers) and stores it; both filter images are resident in LDS.  Per output the arithmetic is
// that of the two launches (same ascending MFMA chains, bias, ReLU, the intermediate rounded to fp32 in between): bit-identical.  The intermediate tensor is
// stored as well when p.Dmid is set (tests, and nodes somebody asks for).

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#ifndef NOPV
#define NOPV 0
#endif
#ifndef CHAIN
#define CHAIN 0
#endif
#ifndef ABLATE
#define ABLATE 0 // experiment hook (BODAHIP_EXTRA_DEFS): 1 no stores | 2 no input loads | 4 no MFMAs | 8 (CHAIN) nothing between the two convolutions
#endif

struct gemm_args_t { // same layout as gemm_conv_f32.hip (one host-side struct serves all fp32 kernels)
  float const *I; float const *J; float *D; float const *bias;
  int Mi, Nj, K;
  int ldI, ldJ, ldD;
  int C, H, W, OH, OW;
  int tiles_i, tiles_j;       // out_chan tiles | super-blocks (WJ blocks of 128 pels each)
  int splitk, kt_per;         // kt_per: workgroups per out_chan tile == the super-block stride of a workgroup
  float *ws; long ws_slab;
  unsigned I_bytes, J_bytes;
  unsigned D_bytes;
  int out_ctot, out_coff;
  int const *ktab; int ktab_n;
  long bsI, bsJ, bsD;
  // CHAIN: the second convolution's filters (M2 x MID) / biases, its out_chan count, and the optional copy of the intermediate tensor (img:MID:pel)
  float const *I2; float const *bias2; float *Dmid;
  int M2; unsigned I2_bytes, Dmid_bytes;
};

namespace {
constexpr int kNT = WJ * 64;
constexpr int kOCT = OCB * 32;               // out_chans per workgroup (every wave computes all of them)
constexpr int kKP = (KC + 1) / 2 * 2;        // in_chans padded to whole MFMA K steps
constexpr int kSteps = kKP / 2;
constexpr int kLD = kOCT | 1;                // LDS pitch of one k row (odd: the transposing stores of the staging pass spread over the banks)
constexpr int kNBLK = (HW + 127) / 128;      // blocks per image
constexpr int kOOB = (int)0x80000000;
static_assert(kSteps % RING == 0, "RING must divide the number of K steps");
#if CHAIN
constexpr int kKP2 = (MID + 1) / 2 * 2;      // the second convolution's in_chans (= the first one's out_chans) padded to whole K steps
constexpr int kSteps2 = kKP2 / 2;
constexpr int kOCT2 = OCB2 * 32;
constexpr int kLD2 = kOCT2 | 1;
static_assert(kKP2 <= kOCT, "the intermediate channels are the rows of one accumulator set");
#endif
static_assert(HW >= 4, "planes of fewer than four pels are not covered");
#ifndef EDGE_OC
#define EDGE_OC 1 // 0: out_chan is a multiple of the workgroup's out_chan tile (no per-row range test in the stores)
#endif
#ifndef EDGE_OC2
#define EDGE_OC2 1
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

extern "C" __global__ __launch_bounds__(WJ * 64, MINW) void KNAME(gemm_args_t const p) {
#if CHAIN
  __shared__ float Fs[kKP * kLD + kOCT + kKP2 * kLD2 + kOCT2];
  float *const Fs2 = Fs + kKP * kLD + kOCT, *const Bs2 = Fs2 + kKP2 * kLD2;
#else
  __shared__ float Fs[kKP * kLD + kOCT];
#endif
  float *const Bs = Fs + kKP * kLD;
  int const tid = threadIdx.x, lane = tid & 63;
  int const wj = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const tile_i = blockIdx.x % p.tiles_i;
  int const oc0 = tile_i * kOCT;
  // workgroup g of an out_chan tile runs on XCD (blockIdx.x % 8); with a workgroup count that is a multiple of 8 (the host's choice) the
  // ids are permuted so that XCD x owns the contiguous range [x, x+1) * kt_per/8 of super-block slots
  int wg_j = blockIdx.x / p.tiles_i;
  if ((p.tiles_i == 1) && (p.kt_per % 8 == 0)) wg_j = (wg_j % 8) * (p.kt_per / 8) + wg_j / 8;

  // ---- resident filter image: Fs[k][oc] = filts[oc0 + oc][k] (zero rows / columns past the tensor), Bs[oc] = bias
  {
    rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
    stage_filters<kOCT, KC, kLD, kNT>(Fs, rI, oc0, p.Mi - oc0, tid);
    for (int e = tid; e < kOCT; e += kNT) Bs[e] = bload1(rB, (oc0 + e) * 4, 0); // rows past out_chan read 0 (range-checked)
#if CHAIN
    rsrc_t const rI2 = make_rsrc(p.I2, p.I2_bytes), rB2 = make_rsrc(p.bias2, (unsigned)p.M2 * 4u);
    stage_filters<kOCT2, MID, kLD2, kNT>(Fs2, rI2, 0, p.M2, tid);
    for (int e = tid; e < kOCT2; e += kNT) Bs2[e] = bload1(rB2, e * 4, 0);
#endif
  }
  __syncthreads();

  rsrc_t const rJ = make_rsrc(p.J, p.J_bytes), rD = make_rsrc(p.D, p.D_bytes);
  bool const hi = (lane >> 5) != 0;                    // lanes 32-63 carry the odd in_chan of a K step (and rows +4 of the result)
  float const *const a_base = Fs + (hi ? kLD : 0) + (lane & 31);
  int const n_units = (p.Nj / HW) * kNBLK;

  // unit u = block (u % kNBLK) of image (u / kNBLK); this lane's first pel of it
  auto pel0 = [&](int u, int &img) -> int { img = u / kNBLK; int const b = u - img * kNBLK; int const q = b * 128 + 4 * (lane & 31); return (q > HW - 4) ? (HW - 4) : q; };
  auto in_off = [&](int u) -> int { int img; int const q = pel0(u, img); return (u < n_units) ? (((img * KC + (hi ? 1 : 0)) * HW + q) * 4) : kOOB; };
  auto out_off = [&](int u) -> int {
    int img; int const q = pel0(u, img);
    return (u < n_units) ? (int)((((unsigned)img * (unsigned)p.out_ctot + (unsigned)(p.out_coff + oc0 + (hi ? 4 : 0))) * (unsigned)HW + (unsigned)q) * 4u) : kOOB;
  };
  auto load_step = [&](int s, int off) -> f32x4 { // B operands of K step s, four column blocks: in_chan 2s (+1 for the upper lanes); the padding chan of an odd KC reads 0
    bool const pad_k = (KC & 1) && (s == kSteps - 1);
#if ABLATE & 2
    float const f = (float)(off + s); return f32x4{f, f, f, f};
#else
    return bload4(rJ, (pad_k && hi) ? kOOB : off, s * (2 * HW * 4));
#endif
  };
  unsigned const S4 = (unsigned)HW * 4u;

  int u = wg_j * WJ + wj;
  int const u_stride = p.kt_per * WJ;
  int off_cur = in_off(u);
  f32x4 bq[RING];
#pragma unroll
  for (int s = 0; s < RING; ++s) bq[s] = load_step(s, off_cur);

  while (u < n_units) {
    int const off_next = in_off(u + u_stride);
    f32x16 acc[OCB][4];
#pragma unroll
    for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    float a[OCB], an[OCB];
#pragma unroll
    for (int rb = 0; rb < OCB; ++rb) a[rb] = a_base[rb * 32];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      // in issue order: the A operands of step s+1 (LDS), this step's MFMAs, the refill of this step's ring slot with step s+RING (of this
      // block, or the first steps of the wave's next block)
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb) an[rb] = a_base[((s + 1 < kSteps) ? (s + 1) : 0) * 2 * kLD + rb * 32];
      f32x4 const b = bq[s % RING];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
#if ABLATE & 4
          acc[rb][cb][s & 15] += a[rb] * b[cb];
#else
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb], b[cb], acc[rb][cb], 0, 0, 0);
#endif
        }
      __builtin_amdgcn_sched_barrier(0);
      bq[s % RING] = (s + RING < kSteps) ? load_step(s + RING, off_cur) : load_step(s + RING - kSteps, off_next);
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb) a[rb] = an[rb];
    }
#if CHAIN
    // ---- first convolution's bias / ReLU in place; the intermediate rows leave only when somebody wants them
#if !(ABLATE & 8)   // (8: measurement only -- no bias / ReLU / half exchange between the two convolutions)
    {
      int img; int const q = pel0(u, img);
      int const moff = (p.Dmid && (u < n_units)) ? (int)((((unsigned)img * (unsigned)MID + (hi ? 4u : 0u)) * (unsigned)HW + (unsigned)q) * 4u) : kOOB;
      rsrc_t const rM = make_rsrc(p.Dmid, p.Dmid_bytes);
#pragma unroll
      for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int const rc = rb * 32 + (r & 3) + 8 * (r >> 2);
          float const bias = Bs[rc + (hi ? 4 : 0)];   // (rows past MID: zero filters, zero bias -> 0)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            float v = acc[rb][cb][r] + bias;
#if RELU
            v = (v > 0.f) ? v : 0.f;
#endif
            acc[rb][cb][r] = v;
          }
          if (p.Dmid && (rc + (hi ? 4 : 0) < MID)) {
            f32x4 const v = f32x4{acc[rb][0][r], acc[rb][1][r], acc[rb][2][r], acc[rb][3][r]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rM, moff, (int)((unsigned)rc * S4), 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3");   // (the wide-store hazard below)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    // ---- half exchange: register pair (4g, 4g+1) -> rows (8g | 8g+1), (8g+4 | 8g+5); pair (4g+2, 4g+3) -> rows (8g+2 | 8g+3), (8g+6 | 8g+7)
#pragma unroll
    for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          float const va = acc[rb][cb][2 * t], vb = acc[rb][cb][2 * t + 1];   // (__builtin_bit_cast applied to a vector ELEMENT reads element 0, whatever the index: scalar copies on both sides of the swap)
          auto const sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, va), __builtin_bit_cast(unsigned, vb), false, false);
          unsigned const s0 = sw[0], s1 = sw[1];
          acc[rb][cb][2 * t] = __builtin_bit_cast(float, s0); acc[rb][cb][2 * t + 1] = __builtin_bit_cast(float, s1);
        }
#endif
    // ---- second convolution: K step s2 = 16 rb + 4 g + j reads the (swapped) register 4g + {0, 2, 1, 3}[j] of row block rb
    int const ooff = out_off(u);
    float const *const a2_base = Fs2 + (hi ? kLD2 : 0) + (lane & 31);
#pragma unroll
    for (int rb2 = 0; rb2 < OCB2; ++rb2) {
      f32x16 acc2[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[cb][r] = 0.f;
      float a2 = a2_base[rb2 * 32], a2n;
#pragma unroll
      for (int s2 = 0; s2 < kSteps2; ++s2) {
        a2n = a2_base[((s2 + 1 < kSteps2) ? (s2 + 1) : 0) * 2 * kLD2 + rb2 * 32];
        int const rb = s2 >> 4, g = (s2 >> 2) & 3, j = s2 & 3, r = 4 * g + (((j & 1) << 1) | (j >> 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, acc[rb][cb][r], acc2[cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a2 = a2n;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int const rc = rb2 * 32 + (r & 3) + 8 * (r >> 2);
        if (EDGE_OC2 && (rc + (hi ? 4 : 0) >= p.M2)) continue;
        float const bias = Bs2[rc + (hi ? 4 : 0)];
        f32x4 v = f32x4{acc2[0][r], acc2[1][r], acc2[2][r], acc2[3][r]} + bias;
#if RELU2
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (v[i] > 0.f) ? v[i] : 0.f;
#endif
#if ABLATE & 1
        if (v[0] == 123.456f)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rD, ooff, (int)((unsigned)rc * S4), 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 3");   // (the wide-store hazard: see the plain epilogue)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#else
    // epilogue: row rb*32 + 8*(r/4) + r%4 (+4 for the upper lanes), this lane's four pels
    int const ooff = out_off(u);
#pragma unroll
    for (int rb = 0; rb < OCB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int const rc = rb * 32 + (r & 3) + 8 * (r >> 2);
        if (EDGE_OC && (oc0 + rc + (hi ? 4 : 0) >= p.Mi)) continue;
        float const bias = Bs[rc + (hi ? 4 : 0)];
        f32x4 v = f32x4{acc[rb][0][r], acc[rb][1][r], acc[rb][2][r], acc[rb][3][r]} + bias;
#if RELU
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (v[i] > 0.f) ? v[i] : 0.f;
#endif
#if ABLATE & 1
        if (v[0] == 123.456f)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), rD, ooff, (int)((unsigned)rc * S4), 0);
        // A 16-byte store still reads its data registers for a few cycles after it issues.  The compiler's hazard model assumes that a store with an SGPR soffset
        // is exempt, packs the next row's v_pk_add_f32 into the same registers right behind the store -- and on gfx950 the last dword of the last lanes of a
        // 16-lane group then leaves with the NEXT row's value (found by the parity test: rows 8, 12, 16, ... of 16-byte-aligned planes).  Keep the wait states.
        __builtin_amdgcn_sched_barrier(0);
#if NOPV == 1
        asm volatile("s_nop 0");
#elif NOPV == 2
        asm volatile("s_nop 1");
#elif NOPV == 3
        ;
#else
        asm volatile("s_nop 3");
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    off_cur = off_next; u += u_stride;
  }
}

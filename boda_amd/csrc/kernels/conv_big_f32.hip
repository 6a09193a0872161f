// conv_big_f32.hip -- fp32 MFMA convolution (+ bias, ReLU) for gfx950 (MI355X): multiplying waves and staging waves (round 6).
//
//   out[img][oc][oy][ox] = act( biases[oc] + sum_{k = (ic, ky, kx) ascending} filts[oc][k] * in[img][ic][oy*SY - PY + ky][ox*SX - PX + kx] )
//   (contract of test/rtc/cudnn_conv.cucl:1-7; semantics test/rtc/conv.cucl:24-44 + the bias / ReLU epilogue of src/cnn_codegen.cc:35-42; this is the kernel that stands
//    in for the reference's k1conv / tconv / conv variants, src/cnn_codegen.cc:625-823, on NCHW / OIHW tensors read in place)
//
// gemm_conv_f32.hip gives every wave both jobs: gather + LDS stores of K tile t + 1, then the MFMAs of tile t, one barrier per step.  A workgroup that is alone on its CU
// then runs at ~0.72 of the matrix peak (two co-resident ones hide each other's staging: ~0.88), and tile-starved layers get exactly that lone workgroup.  Here the two
// jobs belong to different waves, the structure sgemm_big_f32.hip measured out at 0.92 for plain matrices:
//   * WI x WJ multiplying waves (eight: two per SIMD) ONLY read operands from LDS and issue v_mfma_f32_32x32x2_f32: wave (wi, wj) owns kTI x kTJ blocks of 32 x 32 outputs.
//     Which tile row an MFMA row stands for is free: MFMA row rho of row block t is tile row kTI rho + t (and column kappa of column block u is tile column kTJ kappa + u),
//     so a lane's kTI A operands (kTJ B operands) of a k are CONTIGUOUS in the k-major LDS image: one ds_read_b128 / b64 / b32 each per k pair, and in the epilogue a lane
//     holds kTJ consecutive pels of a row: 8 / 12 / 16-byte stores straight into NCHW planes.  (Three blocks are laid out at pitch four: every operand read stays one
//     aligned power-of-two read.)
//   * four staging waves (one per SIMD) ONLY stage: the filter block (out_chan-major in memory, k contiguous: float4 / float2 / scalar loads along k, transposed on the way
//     into the k-major LDS image) and the im2col image of the tile's pels, gathered element by element from NCHW (table gather: any kernel size / stride / padding;
//     J_MODE 5: 1 x 1 without padding, no table) -- PF K tiles in flight in registers, so a gathered element's ~8 VALU operations and its load never stand in front of an MFMA;
//   * NSTG LDS stages: tile t lives in stage t % NSTG and is written during step t - (NSTG - 1); the multiplying waves fetch a tile's first operands before the barrier
//     that ends the previous step;
//   * XCD-aware tile map as in gemm_conv_f32.hip.
// Numerics: every output is ONE ascending-k chain of exact fp32 fmas inside one MFMA accumulator register -- bit-identical to gemm_conv_f32.hip, to the oracle's per-output
// fmaf loop and to the reference's golden digests, whatever the tile.  Zero padding / halo / K tail: out-of-range buffer offsets read 0 (fma(x, 0, acc) == acc).
// What the staging waves ISSUE matters as much as what they wait for: they share their SIMD's issue port with two multiplying waves, and every VALU instruction of theirs
// pushes an MFMA back by a few cycles (measured, round 6: the per-element table gather -- ~7 VALU per gathered element -- costs the multiplying waves 15 % of their rate,
// whatever PF).  So the pel side has three forms, cheapest first:
//   J_MODE 7  stride 1 in x, KH x KW >= 2 taps: the K step is kCB whole input channels and the LDS holds, per channel, the zero-padded INPUT ROWS the tile's pels touch
//             ("slots" of CW + 2 PX floats, as gemm_conv_f32.hip's patch mode) -- staged by plain coalesced loads (per-thread offsets fixed for the whole K loop, the
//             channel in the scalar offset operand: no VALU per element, KH KW x fewer elements than an im2col image) and read IN PLACE by the multiplying waves:
//             B(k = (c, ky, kx), pel) = patch[c][slot(pel) + ky][ox(pel) + kx] = lds[bj(pel) + koff(k)], per-lane base + compile-time offset;
//   J_MODE 5  1 x 1 without padding: one load per element, k in the scalar offset operand (no VALU per element);
//   J_MODE 2  table gather (strided / anything else).
// Compile-time parameters (-D): KNAME TBI TBJ WI WJ BKS PF NSTG MINW I_VW (filter loads: 4 | 2 | 1 floats along k; K % I_VW == 0, BKS % I_VW == 0)
//                               J_MODE (2 | 5 | 7; 7 needs CH CW COH COW = input / output plane sizes and BKS = channels per step x KH KW) KH KW SY SX PY PX RELU GROUP_I

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef GROUP_I
#define GROUP_I 8
#endif
#ifndef BKS
#define BKS 16
#endif
#ifndef PF
#define PF 2
#endif
#ifndef MULPRIO
#define MULPRIO 0
#endif
#ifndef STGPRIO
#define STGPRIO 0
#endif
#ifndef NSTG
#define NSTG 4
#endif
#ifndef MINW
#define MINW 1
#endif
#ifndef I_VW
#define I_VW 1
#endif
#ifndef J_MODE
#define J_MODE 2
#endif
#ifndef KH
#define KH 1
#define KW 1
#define SY 1
#define SX 1
#define PY 0
#define PX 0
#endif
#ifndef RELU
#define RELU 0
#endif
#ifndef MINWAVES
#define MINWAVES MINW // HIP's second launch bound is WAVES PER SIMD, not workgroups per CU.  MINW workgroups of WI WJ + 4 waves would need MINW (WI WJ + 4) / 4: for the twelve-wave tiles that caps
#endif                // the kernel at 80 registers, and the spills that follow cost more than the second resident workgroup brings (round 6 A/B, AlexNet conv3 528 -> 583 us): the lax bound stays
#ifndef PERMJ
#define PERMJ (J_MODE != 7) // 1: column kappa of column block u is tile column kTJ kappa + u (a lane's B operands of a k contiguous in the im2col image, kTJ consecutive pels per lane in the
#endif                      // epilogue); 0 (patch form): column block u is tile columns 32 u .. 32 u + 31 -- consecutive lanes read consecutive patch elements (no bank conflicts)
#ifndef TSTAMP
#define TSTAMP 0 // experiment hook (BODAHIP_CBIG_TSTAMP, tools/cbig_timeline.py): 1 = every workgroup leaves clock stamps in p.ws: [0] start [1] XCC id [2] stager at the first barrier
#endif           // [3] first multiplying wave leaves the K loop [4] ... has issued its stores [5] ... has its stores acknowledged (s_memrealtime: 100 MHz, chip-wide)
#ifndef ABLATE
#define ABLATE 0 // experiment hook: 1 no output stores | 2 no pel loads | 4 no LDS stores in the staging waves | 8 no filter loads | 16 (with 2 | 8) pseudo-random values made in registers instead
#endif

struct gemm_args_t { // same layout as gemm_conv_f32.hip (one host-side struct serves all fp32 kernels)
  float const *I; float const *J; float *D; float const *bias;   // I = filts (Mi x K), J = in (NCHW), D = out (NCHW)
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

#ifndef XPOSE_ONLY
namespace {
constexpr int kNMW = WI * WJ;                 // multiplying waves
constexpr int kNST = 256;                     // staging threads (four waves)
constexpr int kTI = TBI / (WI * 32), kTJ = TBJ / (WJ * 32);   // 32 x 32 blocks per wave
static_assert(kNMW == 8 || kNMW == 4, "eight (or four) multiplying waves");
static_assert(TBI % (WI * 32) == 0 && TBJ % (WJ * 32) == 0 && kTI >= 1 && kTI <= 4 && kTJ >= 1 && kTJ <= 4 && kTI * kTJ <= 8, "wave tile: up to 4 x 2 | 2 x 4 blocks");
constexpr int kTIp = (kTI == 3) ? 4 : kTI, kTJp = (kTJ == 3) ? 4 : kTJ;   // pitch of a lane's operand group in the LDS image
constexpr int kLDI = (TBI / kTI) * kTIp + 4, kLDJ = (PERMJ ? (TBJ / kTJ) * kTJp : TBJ) + 4; // floats per k row of the filter / pel image
#if J_MODE == 7
#if !defined(CH) || !defined(CW) || !defined(COH) || !defined(COW)
#error "J_MODE 7 needs -DCH -DCW -DCOH -DCOW (input / output plane sizes are compile-time)"
#endif
// patch geometry (the arithmetic of gemm_conv_f32.hip's J_MODE 7): output rows of one image share slots (row r + 1 starts SY slots after row r); a tile that crosses into the
// next image starts a new slot group
#ifndef RDEC
#define RDEC 0 // 1: ROW-DECIMATED patch for strided convolutions without padding (conv1 layers: 11x11 / 4; the form gemm_conv_f32.hip introduced in round 4).  The convolution is
#endif         // presented as C0 * KH0 "channels" -- row set (in_chan, kernel row) -- of 1 x KW kernels with stride 1 in y over a plane of COH rows: row r of row set (c, ky) is
               // input row r * SY0 + ky.  k = (c, ky, kx) keeps its order, a K step is kCB whole row sets, the LDS holds ONE input row per output row and row set (coalesced
               // row loads).  Here the row is stored PHASE-MAJOR -- input column x at (x % SX) * kWq + x / SX -- so that tap kx of consecutive pels is consecutive LDS words
               // (no bank conflicts at stride SX).  Needs -DC0 -DH0 -DKH0 -DSY0 and KH = SY = 1, PY = PX = 0, CH = COH; p.C = C0 * KH0.
constexpr int kTaps = KH * KW, kCB = BKS / kTaps, kWq = (CW + SX - 1) / SX, kWp = RDEC ? SX * kWq : CW + 2 * PX;
static_assert(BKS % kTaps == 0 && kTaps >= 2 && (RDEC ? (KH == 1 && SY == 1 && PY == 0 && PX == 0 && CH == COH) : (SX == 1 && KH >= SY)), "patch mode: whole channels per K step; stride 1 in x or the row-decimated form");
constexpr int kRowsMax = (TBJ - 2) / COW + 2;                    // output rows a TBJ-pel tile can touch
constexpr int kSegFull = (COH - 1) * SY + KH;                    // slots of a whole image
constexpr int kSegMax0 = (COH - 1 + kRowsMax - 1) / COH + 1;     // images a tile can touch
constexpr int kSegMax = kSegMax0 < kRowsMax ? kSegMax0 : kRowsMax;
constexpr int kSlots = (kRowsMax - kSegMax) * SY + kSegMax * KH; // slots per channel (upper bound over tile positions)
constexpr int kCS = kSlots * kWp;                                // floats per channel in LDS
constexpr int kCSL = RDEC ? kSlots * CW : kCS;                   // elements per channel the staging waves load (row-decimated: the rows themselves, CW wide)
constexpr int kEPT = (kCSL + kNST - 1) / kNST;                   // patch elements per staging thread and channel
constexpr int kImgJ = kCB * kCS;
constexpr int koffin(int kx) { return RDEC ? (kx % SX) * kWq + kx / SX : kx; }   // tap kx inside a stored row
constexpr int koff(int k) { return (k / kTaps) * kCS + ((k % kTaps) / KW) * kWp + koffin(k % KW); }
// k = 2 kk + 1 sits a fixed distance after k = 2 kk: next tap of the row (row-decimated: next phase | first column of the next phase group) | first tap of the next row |
// first tap of the next channel.  Lanes 32-63 (odd k) fold that distance into their base address once; the even-k offset koff(2 kk) is then an immediate.
constexpr int kD0 = RDEC ? kWq : 1, kD1 = RDEC ? 1 - (SX - 1) * kWq : kWp - KW + 1, kD2 = kCS - (KH - 1) * kWp - koffin(KW - 1);
constexpr int kdelta_class(int kk) {
  if (RDEC) return ((2 * kk) % KW == KW - 1) ? 2 : ((((2 * kk) % KW) % SX == SX - 1) ? 1 : 0);
  return ((2 * kk) % KW != KW - 1) ? 0 : ((((2 * kk) % kTaps) != kTaps - 1) ? 1 : 2);
}
#else
constexpr int kImgJ = BKS * kLDJ;
#endif
constexpr int kImgI = BKS * kLDI, kImg2 = kImgI + (kImgJ + 3) / 4 * 4;
static_assert(NSTG * kImg2 * 4 <= 160 * 1024, "LDS stages exceed 160 KB");
constexpr int kKK = BKS / 2;                  // MFMA k pairs per step
static_assert(BKS % 2 == 0, "BKS: even (two k per MFMA)");
static_assert(NSTG == 3 || NSTG == 4, "NSTG: 3 | 4");
static_assert(PF == 1 || PF == 2 || PF == 4, "PF: 1 | 2 | 4");
constexpr int lcm(int a, int b) { int x = a; while (x % b) x += a; return x; }
constexpr int kU = lcm(lcm(NSTG, PF), 2);     // steps per unrolled round (stage = step % NSTG, register set = step % PF, operand buffer = (step * kKK + kk) % 2: compile-time)
constexpr int kD = NSTG - 1;                  // a tile is written kD steps ahead of its step
constexpr int kOOB = (int)0x80000000;
// filter staging.  I_VW 0 (the default plan): the filters come K-MAJOR -- [k][out_chan padded to a multiple of 4], rows k >= K zero up to a whole number of K steps -- from the
// call's scratch, where bodahip_conv_big_xpose (below) puts them first; a unit = four out_chans of one k: a 16-byte load whose lanes run along out_chan (whole cache lines,
// like the sgemm kernel's operands) and one ds_write_b128.  Why: reading OIHW rows directly (I_VW 4 | 2 | 1: a unit = I_VW consecutive k of one out_chan row, transposed on
// the way into LDS) gives every wave-load 8-16 rows K floats apart -- measured with clock stamps (round 6, AlexNet conv3 on 128 x 512 tiles): the staging waves then sit 94 %
// of the K loop waiting for those loads (9,600 cycles per step for five loads per thread; 290 for the six pel loads), and the multiplying waves wait for them at the barriers.
#if I_VW == 0
constexpr int kUPR = TBI / 4;                 // units per k row
constexpr int kUnitsI = BKS * kUPR;
constexpr int kIW = 4;
static_assert(TBI % 4 == 0, "k-major filter units: four out_chans");
#else
constexpr int kUPR = BKS / I_VW;              // units per row and step
constexpr int kUnitsI = TBI * kUPR;
constexpr int kIW = I_VW;
static_assert(BKS % I_VW == 0 && (I_VW == 1 || I_VW == 2 || I_VW == 4), "I_VW: 0 | 1 | 2 | 4");
#endif
constexpr int kNUI = (kUnitsI + kNST - 1) / kNST;
// pel staging: a thread owns kCPT columns (pels) of the tile and kRPT k rows of every step
constexpr int kCPT = (TBJ + 255) / 256;       // columns per thread
constexpr int kTW = TBJ / kCPT;               // threads across the tile's pels
constexpr int kRG = kNST / kTW;               // row groups
constexpr int kRPT = BKS / kRG;               // k rows per thread and step
constexpr int kNJ = kCPT * kRPT;
static_assert(J_MODE == 7 || (TBJ % kCPT == 0 && kTW % 64 == 0 && kRG >= 1 && BKS % kRG == 0 && (J_MODE == 5 || kRPT % 4 == 0)), "pel staging: whole waves per k row, whole table quads per thread");
static_assert(J_MODE == 2 || J_MODE == 5 || J_MODE == 7, "J_MODE: 2 | 5 | 7");
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(void const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float bload1(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ f32x2 bload2(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
constexpr int colI(int x) { return (x / kTI) * kTIp + x % kTI; }   // LDS column of tile row x (out_chan) ...
__device__ __forceinline__ int colJ(int x) { return PERMJ ? (x / kTJ) * kTJp + x % kTJ : x; }   // ... and of tile column x (pel; natural column blocks: itself)
} // namespace

#if (TBI / (WI * 32)) >= 3
typedef f32x4 avec_t;
#elif (TBI / (WI * 32)) == 2
typedef f32x2 avec_t;
#else
typedef float avec_t;
#endif
#if (TBJ / (WJ * 32)) >= 3
typedef f32x4 bvec_t;
#elif (TBJ / (WJ * 32)) == 2
typedef f32x2 bvec_t;
#else
typedef float bvec_t;
#endif
namespace {
__device__ __forceinline__ float vget(f32x4 const &v, int i) { return v[i]; }
__device__ __forceinline__ float vget(f32x2 const &v, int i) { return v[i]; }
__device__ __forceinline__ float vget(float const &v, int) { return v; }
struct ivec_t { float v[kIW]; };
} // namespace

extern "C" __global__ __launch_bounds__((kNMW + 4) * 64, MINWAVES) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) float sm[NSTG * kImg2];   // [stage][filter image (MFMA A, out_chans) | pel image (MFMA B)][k][kLDI | kLDJ]
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
  int const i0 = tile_i * TBI, j0 = tile_j * TBJ + (int)p.bsJ;   // (p.bsJ: first pel of this launch -- the tail launch of a two-level tiling starts behind the main launch's whole rounds)
  int const nkt = (p.K + BKS - 1) / BKS;
#if TSTAMP
  unsigned long long *const ts = reinterpret_cast<unsigned long long *>(p.ws) + (size_t)blockIdx.x * 16;
  auto stamp = [&](int e) { if ((threadIdx.x & 63) == 0) ts[e] = __builtin_amdgcn_s_memrealtime(); };
  if (threadIdx.x == 0) { ts[0] = __builtin_amdgcn_s_memrealtime(); ts[1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); }   // HW_REG_XCC_ID, bits 0-3
#endif

  if (stager) {
#if STGPRIO
    __builtin_amdgcn_s_setprio(STGPRIO);   // the staging waves ahead of the multiplying waves in the SIMD's issue arbitration (priority, then age: a co-resident younger workgroup's staging waves otherwise get the leftover slots)
#endif
    int const tid = threadIdx.x - kNMW * 64;
    rsrc_t const rI = make_rsrc(p.I, p.I_bytes), rJ = make_rsrc(p.J, p.J_bytes);
#if I_VW == 0
    // ---- filter units (k-major scratch, p.ldI = padded out_chans per k row): unit c = tid + n * 256 -> k row c / kUPR, out_chans 4 (c % kUPR) .. + 3
    int goffI[kNUI], loffI[kNUI];
#pragma unroll
    for (int n = 0; n < kNUI; ++n) {
      int const c = tid + n * kNST, kr = c / kUPR, x = 4 * (c % kUPR);
      goffI[n] = (c < kUnitsI && i0 + x < p.ldI) ? ((kr * p.ldI + i0 + x) * 4) : kOOB;   // (out_chans past the padded end: 0)
      loffI[n] = (c < kUnitsI) ? (kr * kLDI + colI(x)) : -1;
    }
    auto gloadI = [&](int n, int kt) -> ivec_t {   // (tiles fetched past the end re-read the last one: the scratch ends with it)
      ivec_t r;
      f32x4 const v = bload4(rI, (ABLATE & 8) ? kOOB : goffI[n], min(kt, nkt - 1) * (BKS * 4) * p.ldI);
      r.v[0] = v[0]; r.v[1] = v[1]; r.v[2] = v[2]; r.v[3] = v[3];
      if ((ABLATE & 24) == 24) { unsigned h = ((unsigned)tid * 2654435761u + (unsigned)(kt * 40503 + n * 977)) * 2246822519u; for (int e = 0; e < 4; ++e) { h = h * 1664525u + 1013904223u; r.v[e] = (float)(int)(h >> 8) * (1.f / 1677721.6f) - 5.f; } }
      return r;
    };
    auto lstoreI = [&](int n, int stage, ivec_t const &v) {
      if ((loffI[n] >= 0) && !((ABLATE & 4) && v.v[0] != 123.456f)) {
        if constexpr (kTI == 3) {   // (three row blocks at pitch four: the four out_chans of a unit are not contiguous in the image)
          int const x = 4 * ((tid + n * kNST) % kUPR), kr = (tid + n * kNST) / kUPR;
#pragma unroll
          for (int e = 0; e < 4; ++e) sm[stage * kImg2 + kr * kLDI + colI(x + e)] = v.v[e];
        } else {
          f32x4 const w = {v.v[0], v.v[1], v.v[2], v.v[3]};
          *reinterpret_cast<f32x4 *>(sm + stage * kImg2 + loffI[n]) = w;
        }
      }
    };
#else
    // ---- filter units: unit c = tid + n * 256 -> out_chan row c / kUPR, k offset I_VW * (c % kUPR).  Rows past Mi of an edge tile read whatever lies there (or 0 past
    // the tensor): their outputs are never stored and no other output sees them.
    int goffI[kNUI], loffI[kNUI];
#pragma unroll
    for (int n = 0; n < kNUI; ++n) {
      int const c = tid + n * kNST, ir = c / kUPR, kq = (c % kUPR) * I_VW;
      goffI[n] = ((i0 + ir) * p.K + kq) * 4;
      loffI[n] = (c < kUnitsI) ? (kq * kLDI + colI(ir)) : -1;
    }
    // (the per-step part of the address, kt * BKS floats, goes through the scalar offset operand -- clamped to the last real tile for the tiles fetched past the end.
    //  K tail: k >= K must read ZEROS -- only the last tile can hold such k, a wave-uniform case: there, and only there, the lanes test their k)
    auto gloadI = [&](int n, int kt) -> ivec_t {
      ivec_t r;
      int const ktc = min(kt, nkt - 1), soff = ktc * (BKS * 4);
      int voff = goffI[n];
      if ((ktc + 1) * BKS > p.K) { int const kq = ((tid + n * kNST) % kUPR) * I_VW; voff = (ktc * BKS + kq < p.K) ? voff : kOOB; }
      if (ABLATE & 8) voff = kOOB;
      if constexpr (I_VW == 4) { f32x4 const v = bload4(rI, voff, soff); r.v[0] = v[0]; r.v[1] = v[1]; r.v[2] = v[2]; r.v[3] = v[3]; }
      else if constexpr (I_VW == 2) { f32x2 const v = bload2(rI, voff, soff); r.v[0] = v[0]; r.v[1] = v[1]; }
      else r.v[0] = bload1(rI, voff, soff);
      return r;
    };
    auto lstoreI = [&](int n, int stage, ivec_t const &v) {
      if ((loffI[n] >= 0) && !((ABLATE & 4) && v.v[0] != 123.456f)) {
#pragma unroll
        for (int e = 0; e < I_VW; ++e) sm[stage * kImg2 + loffI[n] + e * kLDI] = v.v[e];
      }
    };
#endif
#if J_MODE == 7
    // ---- input patch: element el = tid + e * 256 of a channel's kCSL elements = (slot s, column x); its byte offset inside channel 0 is fixed for the whole K loop
    int goffJ[kEPT], lofsJ[kEPT];
    {
      int const R0 = j0 / COW, img0 = R0 / COH, oy0 = R0 - img0 * COH;       // first output row of the tile (workgroup-uniform)
      int const seg0 = (COH - 1 - oy0) * SY + KH;                            // slots of the first image's part
      int const n_img = p.Nj / (COH * COW);
      constexpr int kRowW = RDEC ? CW : kWp;
#pragma unroll
      for (int e = 0; e < kEPT; ++e) {
        int const el = tid + e * kNST, sl = el / kRowW, xc = el - sl * kRowW, ix = xc - PX;
        int const s2 = sl - seg0, im2 = s2 / kSegFull;
        int const img = (sl < seg0) ? img0 : (img0 + 1 + im2);
        int const iy = (sl < seg0) ? (oy0 * SY - PY + sl) : (s2 - im2 * kSegFull - PY);
        bool const ok = (el < kCSL) && (img < n_img) && ((unsigned)iy < (unsigned)CH) && ((unsigned)ix < (unsigned)CW);
#if RDEC
        goffJ[e] = ok ? (((img * (C0 * H0) + iy * SY0) * CW + ix) * 4) : kOOB;   // row iy of the decimated plane is input row iy * SY0 (+ the row set's kernel row: in coff)
        lofsJ[e] = (el < kCSL) ? (sl * kWp + (xc % SX) * kWq + xc / SX) : -1;    // stored phase-major
#else
        goffJ[e] = ok ? (((img * p.C * CH + iy) * CW + ix) * 4) : kOOB;          // padding / past the end: out of range -> 0
        lofsJ[e] = (el < kCSL) ? el : -1;
#endif
      }
    }
    struct jset_t { float v[kCB * kEPT]; };
    auto gloadJ = [&](int kt) -> jset_t {
      jset_t r;
      int const c0 = kt * kCB;
#pragma unroll
      for (int cc = 0; cc < kCB; ++cc) {
        // (scalar.  Channels past the end -- K tail, tiles fetched past the end -- meet zero filter values: any finite data will do -> the image's last channel again)
        int const cq = min(c0 + cc, p.C - 1);
#if RDEC
        int const coff = ((cq / KH0) * H0 + cq % KH0) * (CW * 4);
#else
        int const coff = cq * (CH * CW * 4);
#endif
#pragma unroll
        for (int e = 0; e < kEPT; ++e) {
          r.v[cc * kEPT + e] = bload1(rJ, (ABLATE & 2) ? kOOB : goffJ[e], coff);
          if ((ABLATE & 18) == 18) { unsigned h = ((unsigned)tid * 2654435761u + (unsigned)(kt * 7919 + cc * 31 + e)) * 2246822519u; h = h * 1664525u + 1013904223u; r.v[cc * kEPT + e] = (float)(int)(h >> 8) * (1.f / 1677721.6f) - 5.f; }
        }
      }
      return r;
    };
    auto lstoreJ = [&](int stage, jset_t const &v) {
#pragma unroll
      for (int cc = 0; cc < kCB; ++cc)
#pragma unroll
        for (int e = 0; e < kEPT; ++e)
          if ((lofsJ[e] >= 0) && !((ABLATE & 4) && v.v[0] != 123.456f)) sm[stage * kImg2 + kImgI + cc * kCS + lofsJ[e]] = v.v[cc * kEPT + e];
    };
#else
    // ---- pel columns of this thread
    bool const jact = tid < kTW * kRG;                           // (192-wide tiles leave the fourth staging wave idle)
    int const rg = __builtin_amdgcn_readfirstlane(tid / kTW), tcol = tid - rg * kTW;   // row group (wave-uniform: kTW % 64 == 0)
    int loffJ[kCPT];
#if J_MODE == 5
    int base4[kCPT];
#else
    int gbase[kCPT], giy0[kCPT], gix0[kCPT];
#endif
    {
      int const OHW = p.OH * p.OW;
#pragma unroll
      for (int c = 0; c < kCPT; ++c) {
        int const jl = tcol + c * kTW, jg = j0 + jl;
        int const img = jg / OHW, pel = jg - img * OHW, oy = pel / p.OW, ox = pel - oy * p.OW;
        loffJ[c] = kImgI + (rg * kRPT) * kLDJ + colJ(jl);
        bool const ok = jact && (jg < p.Nj);
#if J_MODE == 5
        base4[c] = ok ? (((img * p.C * p.H + oy * SY) * p.W + ox * SX) * 4) : kOOB;
#else
        giy0[c] = ok ? (oy * SY - PY) : (1 << 29);               // columns past the end fail the row-range test for every k
        gix0[c] = ox * SX - PX;
        gbase[c] = (img * p.C * p.H + (oy * SY - PY)) * p.W + gix0[c];
#endif
      }
    }
    struct jset_t { float v[kNJ]; };
    auto gloadJ = [&](int kt) -> jset_t {
      jset_t r;
      int const kb = kt * BKS + rg * kRPT;                       // first k row of this wave in tile kt (scalar)
#if J_MODE == 5
      int const hw4 = p.H * p.W * 4;
#pragma unroll
      for (int q = 0; q < kRPT; ++q) {
        int const soff = min(kb + q, p.K - 1) * hw4;             // scalar: no VALU per element.  k >= K meets zero filter values: any finite data will do -> the last channel again
#pragma unroll
        for (int c = 0; c < kCPT; ++c) r.v[q * kCPT + c] = bload1(rJ, (ABLATE & 2) ? kOOB : base4[c], soff);
      }
#else
      // (in_chan, ky, kx) decode and element offset of a k: host-built table read through the scalar cache (s_load_dwordx4); rows k >= K carry ky = 2^30
      typedef i32x4 const __attribute__((address_space(4))) *ctab_t;
      ctab_t const t_off = (ctab_t)(p.ktab + kb), t_ky = (ctab_t)(p.ktab + p.ktab_n + kb), t_kx = (ctab_t)(p.ktab + 2 * p.ktab_n + kb);
#pragma unroll
      for (int q4 = 0; q4 < kRPT / 4; ++q4) {
        i32x4 const ko = t_off[q4], ky = t_ky[q4], kx = t_kx[q4];
        int const kov[4] = {ko.x, ko.y, ko.z, ko.w}, kyv[4] = {ky.x, ky.y, ky.z, ky.w}, kxv[4] = {kx.x, kx.y, kx.z, kx.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < kCPT; ++c) {
            int const iy = giy0[c] + kyv[e], ix = gix0[c] + kxv[e];
            int off = (gbase[c] + kov[e]) * 4;
            asm volatile("" : "+v"(off));                        // keep the offset unconditional: the select below must stay a v_cndmask, not a branch
            bool const ok = ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
            r.v[(q4 * 4 + e) * kCPT + c] = bload1(rJ, ((ABLATE & 2) || !ok) ? kOOB : off, 0);
          }
      }
#endif
      return r;
    };
    auto lstoreJ = [&](int stage, jset_t const &v) {
      if (jact) {
#pragma unroll
        for (int q = 0; q < kRPT; ++q)
#pragma unroll
          for (int c = 0; c < kCPT; ++c) sm[stage * kImg2 + loffJ[c] + q * kLDJ] = v.v[q * kCPT + c];
      }
    };
#endif

    ivec_t ringI[PF][kNUI]; jset_t ringJ[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int n = 0; n < kNUI; ++n) ringI[u][n] = gloadI(n, u);
      ringJ[u] = gloadJ(u);
    }
#pragma unroll
    for (int t = 0; t < kD; ++t) {   // tiles 0 .. kD - 1 go to their stages before the first step; their register sets take tiles PF ..
#pragma unroll
      for (int n = 0; n < kNUI; ++n) { lstoreI(n, t, ringI[t % PF][n]); ringI[t % PF][n] = gloadI(n, t + PF); }
      lstoreJ(t, ringJ[t % PF]); ringJ[t % PF] = gloadJ(t + PF);
    }
#if TSTAMP
    if (wave == kNMW) stamp(2);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");
#if TSTAMP
    unsigned long long st_a = 0, st_b = 0, st_l = 0, st_w = 0;   // shader cycles of the first staging wave: filter part | pel part | LDS drain | at the barrier
#define ST_T0 unsigned long long t_ = __builtin_amdgcn_s_memtime(), t2_;
#define ST_ADD(ACC) t2_ = __builtin_amdgcn_s_memtime(); ACC += t2_ - t_; t_ = t2_;
#else
#define ST_T0
#define ST_ADD(ACC)
#endif
    for (int kb = 0; kb < nkt; kb += kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (kb + u >= nkt) break;                                // (wave-uniform; the multiplying waves leave at the same step)
        ST_T0
#pragma unroll
        for (int n = 0; n < kNUI; ++n) { lstoreI(n, (u + kD) % NSTG, ringI[(u + kD) % PF][n]); ringI[(u + kD) % PF][n] = gloadI(n, kb + u + kD + PF); }
        ST_ADD(st_a)
        lstoreJ((u + kD) % NSTG, ringJ[(u + kD) % PF]); ringJ[(u + kD) % PF] = gloadJ(kb + u + kD + PF);
        ST_ADD(st_b)
#if TSTAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ST_ADD(st_l)
        asm volatile("s_barrier" ::: "memory");
        ST_ADD(st_w)
#else
        asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory");   // (scalar loads share the counter and return out of order: only 0 is a safe count)
#endif
      }
    }
#if TSTAMP
    if (wave == kNMW && lane == 0) { ts[8] = st_a; ts[9] = st_b; ts[10] = st_l; ts[11] = st_w; }
#endif
    return;
  }

  // ---- multiplying waves
#if MULPRIO
  __builtin_amdgcn_s_setprio(MULPRIO);   // the multiplying waves ahead of whatever else shares the SIMD at priority 0 (STGPRIO, if set, should be higher)
#endif
  int const wi = wave / WJ, wj = wave % WJ;
  float const *const a_base = sm + (lane >> 5) * kLDI + (wi * 32 + (lane & 31)) * kTIp;            // + stage * kImg2 + kk * 2 * kLDI
#if J_MODE == 7
  // this lane's pels (column block u: tile column kTJ (lane % 32) + u) inside the patch; lanes 32-63 hold the odd k of a pair: + the distance to the next tap (three classes)
  int bj[kTJ][3];
  {
    int const R0 = j0 / COW, img0 = R0 / COH, oy0 = R0 - img0 * COH, seg0 = (COH - 1 - oy0) * SY + KH;
    bool const odd = (lane >> 5) != 0;
#pragma unroll
    for (int u = 0; u < kTJ; ++u) {
      int const jg = min(j0 + (PERMJ ? ((wj * 32 + (lane & 31)) * kTJ + u) : ((wj * kTJ + u) * 32 + (lane & 31))), p.Nj - 1);
      int const R = jg / COW, ox = jg - R * COW, img = R / COH, oy = R - img * COH;
      int const slot = (img == img0) ? ((oy - oy0) * SY) : (seg0 + (img - img0 - 1) * kSegFull + oy * SY);
      int const b0 = kImgI + slot * kWp + ox;
      bj[u][0] = b0 + (odd ? kD0 : 0); bj[u][1] = b0 + (odd ? kD1 : 0); bj[u][2] = b0 + (odd ? kD2 : 0);
    }
  }
  struct bop_t { float v[kTJ]; };
  auto readB = [&](int stage, int kk) -> bop_t {   // (stage, kk: compile-time after unrolling)
    bop_t r;
#pragma unroll
    for (int u = 0; u < kTJ; ++u) r.v[u] = sm[stage * kImg2 + bj[u][kdelta_class(kk)] + koff(2 * kk)];
    return r;
  };
#define BGET(B, U) ((B).v[U])
#else
#if PERMJ
  float const *const b_base = sm + kImgI + (lane >> 5) * kLDJ + (wj * 32 + (lane & 31)) * kTJp;
  typedef bvec_t bop_t;
  auto readB = [&](int stage, int kk) -> bop_t { return *reinterpret_cast<bvec_t const *>(b_base + stage * kImg2 + kk * 2 * kLDJ); };
#define BGET(B, U) vget(B, U)
#else   // natural column blocks on an im2col / 1x1 image (-DPERMJ=0): kTJ single reads per k pair, consecutive lanes consecutive words; the epilogue's paired row stores
  float const *const b_base = sm + kImgI + (lane >> 5) * kLDJ + wj * (kTJ * 32) + (lane & 31);
  struct bop_t { float v[kTJ]; };
  auto readB = [&](int stage, int kk) -> bop_t { bop_t r;
#pragma unroll
    for (int u = 0; u < kTJ; ++u) r.v[u] = b_base[stage * kImg2 + kk * 2 * kLDJ + u * 32];
    return r; };
#define BGET(B, U) ((B).v[U])
#endif
#endif
  auto readA = [&](int stage, int kk) -> avec_t { return *reinterpret_cast<avec_t const *>(a_base + stage * kImg2 + kk * 2 * kLDI); };
  f32x16 acc[kTI][kTJ];
#pragma unroll
  for (int t = 0; t < kTI; ++t)
#pragma unroll
    for (int u = 0; u < kTJ; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

 asm volatile("s_barrier" ::: "memory");
#if TSTAMP
  if (wave == 0 && lane == 0) ts[6] = __builtin_amdgcn_s_memtime();   // shader clock at the start ...
#endif
#if TSTAMP
  unsigned long long bar_wait = 0;
#endif
  avec_t a[2]; bop_t b[2];                                // operands of k pair n and n + 1 (n counted across steps: buffer n % 2): the reads run one pair (kTI x kTJ MFMAs) ahead
  a[0] = readA(0, 0); b[0] = readB(0, 0);
  for (int kb = 0; kb < nkt; kb += kU) {
#pragma unroll
    for (int s = 0; s < kU; ++s) {
      if (kb + s >= nkt) break;
#pragma unroll
      for (int kk = 0; kk < kKK; ++kk) {
        int const cur = (s * kKK + kk) & 1;               // (kU * kKK is even: a round starts on buffer 0)
        if (kk + 1 < kKK) { a[cur ^ 1] = readA(s % NSTG, kk + 1); b[cur ^ 1] = readB(s % NSTG, kk + 1); }
        else { a[cur ^ 1] = readA((s + 1) % NSTG, 0); b[cur ^ 1] = readB((s + 1) % NSTG, 0); }   // the next tile's first pair, before the barrier (complete: written a step earlier)
        __builtin_amdgcn_sched_barrier(0);
        avec_t const ca = a[cur]; bop_t const cb = b[cur];
#pragma unroll
        for (int t = 0; t < kTI; ++t)
#pragma unroll
          for (int u = 0; u < kTJ; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget(ca, t), BGET(cb, u), acc[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#if TSTAMP
      unsigned long long const tb0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_barrier" ::: "memory");
      bar_wait += __builtin_amdgcn_s_memtime() - tb0;
#else
      asm volatile("s_barrier" ::: "memory");
#endif
    }
  }
#if TSTAMP
  if (wave == 0 && lane == 0) ts[1] = (ts[1] & 15) | (bar_wait << 4);   // shader cycles the first multiplying wave spent at the K loop's barriers
#endif

#if TSTAMP
  if (wave == 0) { stamp(3); if (lane == 0) ts[7] = __builtin_amdgcn_s_memtime(); }   // ... and at the end of the K loop: cycles / microseconds = the clock the loop ran at
#endif
  // ---- epilogue: MFMA row rho = 8 * (r / 4) + r % 4 + 4 * (lane / 32) of row block t is tile row kTI rho + t; column kappa = lane % 32 of column block u is tile column
  // kTJ kappa + u: a lane holds kTJ CONSECUTIVE pels of out_chan row (t, r) -- one 4 kTJ-byte store where they lie in one image (NCHW planes are contiguous in pel; the
  // address is 4-byte aligned only: odd planes), element stores for the few lanes whose pels straddle two images or the end.
  {
    rsrc_t const rD = make_rsrc(p.D, p.D_bytes), rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
    int const OHW = p.OH * p.OW;
    unsigned const S4 = (unsigned)OHW * 4u;
    int const ib = i0 + wi * (kTI * 32) + kTI * 4 * (lane >> 5);                 // this lane's first out_chan; + kTI * ((r & 3) + 8 * (r >> 2)) + t: wave-uniform
    auto rowc = [](int t, int r) { return kTI * ((r & 3) + 8 * (r >> 2)) + t; };
    // biases and ReLU first, IN PLACE, row block by row block (16 loads in flight, one wait): a bias load issued between the stores would wait for every store before it
    // (one counter, in order) -- measured with the clock stamps: 27.6 us to issue 48 stores that way (AlexNet conv1, 96 x 256 tile), 4.6 us for 64 with the loads batched
    constexpr int kBB = (kTI * kTJ >= 8) ? 4 : ((kTI * kTJ >= 6) ? 8 : 16);   // biases in flight at a time (what the registers beside the accumulators allow)
    constexpr bool kBiasFirst = !(PERMJ && kTI * kTJ >= 8);                   // (the 128-accumulator tiles of the permuted-column forms have no registers to spare: bias per row, as before)
#pragma unroll
    for (int t = 0; t < (kBiasFirst ? kTI : 0); ++t)
#pragma unroll
      for (int rb = 0; rb < 16; rb += kBB) {
        float bv[kBB];
#pragma unroll
        for (int r = 0; r < kBB; ++r) bv[r] = bload1(rB, ib * 4, rowc(t, rb + r) * 4);   // (rows past Mi read 0)
#pragma unroll
        for (int u = 0; u < kTJ; ++u)
#pragma unroll
          for (int r = 0; r < kBB; ++r) { float x = acc[t][u][rb + r] + bv[r]; if (RELU) x = (x > 0.f) ? x : 0.f; acc[t][u][rb + r] = x; }
        __builtin_amdgcn_sched_barrier(0);
      }
#if PERMJ
    int const jg0 = j0 + (wj * 32 + (lane & 31)) * kTJ;
    int const img0 = jg0 / OHW, pel0 = jg0 - img0 * OHW;
    bool const whole = (pel0 + kTJ <= OHW) && (jg0 + kTJ <= p.Nj);              // all kTJ pels in one image (and inside the tensor)
    unsigned const jpart0 = (((unsigned)img0 * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel0) * 4u;
    unsigned jpart[kTJ]; bool jok[kTJ];
#pragma unroll
    for (int u = 0; u < kTJ; ++u) {
      int const jg = jg0 + u, img = (pel0 + u >= OHW) ? img0 + (pel0 + u) / OHW : img0, pel = jg - img * OHW;
      jok[u] = !whole && (jg < p.Nj);
      jpart[u] = (((unsigned)img * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel) * 4u;
    }
    bool const any_split = __builtin_amdgcn_ballot_w64(!whole && (jg0 < p.Nj)) != 0;   // wave-uniform
    unsigned const ipart = (unsigned)ib * S4;
    auto store_all = [&](bool const edge) {
#pragma unroll
      for (int t = 0; t < kTI; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int const rc = rowc(t, r);
          bool const row_ok = !edge || (ib + rc < p.Mi);
          float v[kTJ];
#pragma unroll
          for (int u = 0; u < kTJ; ++u) v[u] = acc[t][u][r];
          if constexpr (!kBiasFirst) { float const bvr = bload1(rB, ib * 4, rc * 4); for (int u = 0; u < kTJ; ++u) { float x = v[u] + bvr; if (RELU) x = (x > 0.f) ? x : 0.f; v[u] = x; } }
#if ABLATE & 1
          if (v[0] == 123.456f)
#endif
          {
          int const off = (whole && row_ok) ? (int)(jpart0 + ipart) : kOOB, soff = (int)((unsigned)rc * S4);
          if constexpr (kTJ == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), rD, off, soff, 0);
          else if constexpr (kTJ == 2) { f32x2 const w = {v[0], v[1]}; __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, w), rD, off, soff, 0); }
          else if constexpr (kTJ == 3) { f32x3 const w = {v[0], v[1], v[2]}; __builtin_amdgcn_raw_buffer_store_b96(__builtin_bit_cast(u32x3, w), rD, off, soff, 0); }
          else { f32x4 const w = {v[0], v[1], v[2], v[3]}; __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rD, off, soff, 0); }
          if (kTJ > 1 && any_split) {
#pragma unroll
            for (int u = 0; u < kTJ; ++u) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[u]), rD, (jok[u] && row_ok) ? (int)(jpart[u] + ipart) : kOOB, soff, 0);
          }
          }
        }
      }
    };
#else
    // natural column blocks: a lane holds ONE pel per column block (32 u + lane % 32), lanes 32-63 the rows kTI * 4 further down.  Pairs of blocks leave through
    // v_permlane32_swap (as gemm_conv_f32.hip's paired stores): one register then holds 64 CONSECUTIVE pels of one row, the other of the row kTI * 4 below -- 256
    // contiguous bytes per store instruction instead of two 128-byte pieces in two rows; an odd last block leaves as it is.
    unsigned const ipart = (unsigned)(i0 + wi * (kTI * 32)) * S4;                 // (no per-lane row part in the paired stores)
    auto jpart_of = [&](int jg) -> unsigned {
      int const img = jg / OHW, pel = jg - img * OHW;
      return (jg < p.Nj) ? ((((unsigned)img * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel) * 4u) : 0x80000000u;
    };
    auto store_all = [&](bool const edge) {
      int const ibl = i0 + wi * (kTI * 32);
#pragma unroll
      for (int tp = 0; tp < kTJ / 2; ++tp) {
        unsigned const jp = jpart_of(j0 + (wj * kTJ + 2 * tp) * 32 + lane);     // 64 consecutive pels: lanes 0-31 block 2 tp, lanes 32-63 block 2 tp + 1
#pragma unroll
        for (int t = 0; t < kTI; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int const rc = rowc(t, r);
            float const va = acc[t][2 * tp][r], vb = acc[t][2 * tp + 1][r];
            auto const sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, va), __builtin_bit_cast(unsigned, vb), false, false);
#if ABLATE & 1
            if (va == 123.456f)
#endif
            {
            bool const ok0 = (jp != 0x80000000u) && (!edge || (ibl + rc < p.Mi)), ok1 = (jp != 0x80000000u) && (!edge || (ibl + rc + kTI * 4 < p.Mi));
            __builtin_amdgcn_raw_buffer_store_b32(sw[0], rD, ok0 ? (int)(jp + ipart) : kOOB, (int)((unsigned)rc * S4), 0);
            __builtin_amdgcn_raw_buffer_store_b32(sw[1], rD, ok1 ? (int)(jp + ipart) : kOOB, (int)((unsigned)(rc + kTI * 4) * S4), 0);
            }
          }
      }
      if constexpr (kTJ % 2) {
        unsigned const jp = jpart_of(j0 + (wj * kTJ + kTJ - 1) * 32 + (lane & 31));
        unsigned const ipl = (unsigned)ib * S4;
#pragma unroll
        for (int t = 0; t < kTI; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int const rc = rowc(t, r);
            float const v = acc[t][kTJ - 1][r];
#if ABLATE & 1
            if (v == 123.456f)
#endif
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rD, ((jp != 0x80000000u) && (!edge || (ib + rc < p.Mi))) ? (int)(jp + ipl) : kOOB, (int)((unsigned)rc * S4), 0);
          }
      }
    };
#endif
    if (i0 + TBI <= p.Mi) store_all(false); else store_all(true);   // workgroup-uniform
  }
#if TSTAMP
  if (wave == 0) { stamp(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(5); }
#endif
}
#else  // XPOSE_ONLY
// bodahip_conv_big_xpose: filts[out_chan][k] (OIHW, k = (in_chan, ky, kx) contiguous) -> dst[k][out_chan], out_chans padded to Mi4 (a multiple of 4) and k to Kp (whole K
// steps) with zeros: the layout the staging waves read with whole-line loads.  32 x 32 tiles through LDS: both sides coalesced.  (The reference runs the same kind of pass in
// front of its k1conv / tconv variants: xpose_filts, src/cnn_codegen.cc; here it is part of the call, a few microseconds for the megabytes of a layer's filters.)
extern "C" __global__ __launch_bounds__(256) void KNAME(float const *__restrict__ filts, float *__restrict__ dst, int Mi, int Mi4, int K, int Kp) {
  __shared__ float t[32][33];
  int const k0 = blockIdx.x * 32, i0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 4; ++r) { int const i = i0 + ty + 8 * r, k = k0 + tx; t[ty + 8 * r][tx] = (i < Mi && k < K) ? filts[(long)i * K + k] : 0.f; }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) { int const k = k0 + ty + 8 * r, i = i0 + tx; if (k < Kp && i < Mi4) dst[(long)k * Mi4 + i] = t[tx][ty + 8 * r]; }
}
#endif

// gemm_conv_f32.hip -- fp32 MFMA contraction kernel for gfx950 (MI355X), specialised at run time by hiprtc.
//
//   D[i][j] = sum_k I(k,i) * J(k,j)          (+ bias[i], ReLU for convolutions)
//
// One kernel body serves both ops on Boda's hot path; the host picks the operand roles so that the *j* index is the
// contiguous dimension of the output (coalesced 128-B row stores straight out of the MFMA accumulator layout):
//   sgemm        i = M, j = N, k = K      I = a[K][M], J = b[K][N], D = c[M][N]
//                (contract of test/rtc/cublas_sgemm.cucl:1-4; semantics test/rtc/sgemm.cucl:17-43)
//   Convolution  i = out_chan, j = pel=(img,oy,ox), k = (in_chan,ky,kx)
//                I = filts[out_chan][k] (OIHW), J = im2col view of in (NCHW) gathered on the fly -- never materialised,
//                D = out[img][out_chan][oy][ox]   (contract of test/rtc/cudnn_conv.cucl:1-7;
//                semantics test/rtc/conv.cucl:24-44 + bias/ReLU epilogue src/cnn_codegen.cc:35-42)
// This replaces the reference's sgemm / conv / k1conv / tconv / ipconv CUCL variants (src/cnn_codegen.cc:165-823) and
// their separate layout-transform passes (xpose_filts, k1conv_xpose_in, tconv_xpose_in): operands are read in
// reference layout and re-laid-out only inside LDS.
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain in ascending k, so (without SPLITK) every output is
// bit-identical to one thread of the reference accumulating fmaf() over k = 0..K-1 (zero-padded k / halo terms add
// fma(x,0,acc) == acc).  With SPLITK the K range is cut into slices that are summed in ascending order afterwards
// (fp32 re-association; inside the reference's 2e-4 tolerance, src/rtc_prof.cc:161).
//
// Structure per workgroup (WI x WJ waves of 64 lanes):
//   * BI x BJ output tile, BK-deep K steps, LDS double-buffered as k-major [BK][BI+4] / [BK][BJ+4] float images
//     (the +4 pad keeps 16-B alignment for ds_write_b128 and de-phases rows); register-staged prefetch of K-tile t+1
//     is issued before the MFMAs of tile t and written to the other LDS buffer after them: one barrier per K step.
//     All global loads are buffer loads through a wave-uniform resource descriptor: out-of-range lanes (tile edges, conv
//     halo / zero padding, K tail) are given an offset beyond num_records and the hardware returns 0 -- no branches, no
//     selects on the loaded value, 32-bit offsets only, so the loads stay in flight until the LDS write after the MFMAs.
//   * each wave owns a (TI*32) x (TJ*32) sub-tile = TI*TJ accumulators of 16 VGPRs; A/B operands are single
//     conflict-free ds_read_b32 (lane l reads row 2*kk+(l>>5), column (l&31)).
//   * the im2col gather assigns each thread one fixed output position (column j) and wave-uniform k rows; the
//     (in_chan,ky,kx) decode is a host-built table read through the scalar cache, so an element costs ~6 VALU ops.
//   * workgroup ids are remapped XCD-aware (block b runs on XCD b%8): each XCD gets a contiguous band of
//     tiles, walked in groups of GROUP_I tiles along i so neighbouring workgroups share I / J panels in their L2.
//
// Compile-time parameters (-D):  KNAME BI BJ BK WI WJ MINW I_MODE J_MODE EPI [KH KW SY SX PY PX RELU SPLITK MT]
//   I_MODE 0 k-major float4 | 1 k-major scalar | 2 i-major (k contiguous) float4 | 3 i-major scalar | 4 i-major float2
//   J_MODE 0 k-major float4 | 1 k-major scalar | 2 convolution gather from NCHW, per element | 3 j-major (k contiguous) float4 | 4 j-major scalar
//          6 convolution gather from NCHW, per (in_chan,ky) ROW of KW taps (-DJROWS rows per K step, BK = JROWS*KW): the default for KW >= 6 (wide kernels)
//          7 convolution from an LDS input patch (SX == 1, KH*KW >= 2; -DCH -DCW -DCOH -DCOW, BK = whole channels): see kCB below
//          5 1x1 convolution without padding, any stride (the reference's k1conv case, src/cnn_op.cc:51-60): k = in_chan, so
//            J(k,j) = in[base_j + k*H*W] -- one add per gathered element, no table, no halo tests
//          (3/4: convolutions whose output is 1x1 with no padding -- the reference's "ipconv" case, src/cnn_op.cc:49-50 --
//           where the im2col row of image j is simply the contiguous image: J(k,j) = in[j*K + k])
//   EPI    0 plain store    | 1 + bias[i], optional ReLU, NCHW scatter of j=(img,pel)
//   SPLITK 0|1: with 1 the grid is tiles x p.splitk; slice s accumulates K-tiles [s*kt_per, (s+1)*kt_per) and stores its raw
//          partial tile to slab s of p.ws (same indexing as D); bodahip_splitk_reduce then sums the slabs in ascending s and
//          applies the epilogue.  Used when a problem has too few output tiles to occupy 256 CUs (e.g. AlexNet fc6-fc8).

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h> // offline (hipcc) builds only; hiprtc provides the device runtime implicitly
#endif

#ifndef ST_AUX
#define ST_AUX 0   // cache policy of the paired output stores (experiments: bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef KNAME
#define KNAME bodahip_gemm_f32
#endif
#ifndef GROUP_I
#define GROUP_I 8
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
#ifndef SPLITK
#define SPLITK 0
#endif
#ifndef SETPRIO
#define SETPRIO 0
#endif
#ifndef ABLATE
#define ABLATE 0 // experiment hook (tools/tile_sweep.py via BODAHIP_EXTRA_DEFS): 1 no epilogue stores | 2 no J loads | 4 no in-loop LDS stores | 8 no I loads
#endif
#ifndef HALF
#define HALF 0 // 1 (sgemm, I_MODE / J_MODE 0 | 1, EPI 0): a / b / c are STORED as IEEE half (2-byte elements; ld* and the element offsets unchanged), converted to
#endif         // float on the way into LDS, accumulated by the same fp32 MFMA chain, rounded to half (RNE) when stored -- the reference's 16-bit-storage /
               // fp32-math sgemm (vload_half / vstore_half, src/cnn_codegen.cc:440-449, test/sgemm-ops-debug-half.txt).  half -> float is exact, so the
               // result is bit-identical to the reference's per-thread fmaf loop over the converted values.
#ifndef SWAPST
#define SWAPST (((BJ / (WJ * 32)) % 2 == 0) && !HALF) // paired 256-byte row stores in the epilogue (see there); 0 forces the plain per-block stores
#endif
#ifndef SPECW
#define SPECW 0 // 1: as many STAGING waves as multiplying waves (threads WI*WJ*64 .. 2*WI*WJ*64 - 1): they do all global loads, gathers and LDS stores (two K tiles in
#endif          // flight in their registers), the multiplying waves only read operands and issue MFMAs.  Same barriers, same k order: bit-identical results.
#ifndef STF
#define STF 0 // PF >= 3 only: 1 = the LDS stores of tile kt + 1 are issued before the MFMAs of tile kt (they drain under the MFMA phase)
#endif
#ifndef PF
#define PF 1 // K-tiles prefetched ahead in registers: 1 | 2 (two register sets; for workgroups that run alone on their CU) | 4 | 6 | 8 (a ring of register sets: tile-starved long-K shapes)
#endif
#ifndef KHO
#define KHO 0 // 1: SEQUENTIAL K HAND-OFF (round 5; opt-in: eleventh field of a tile string = segments per tile) -- an EXACT way of evening out launches whose tiles do not deal
#endif        // out evenly over the CUs.  A tile's K range is cut into p.splitk SEGMENTS; segment s CONTINUES the ascending-k fma chains of segment s - 1 from its stored
              // accumulators (unlike K slices nothing is re-associated: every output is still one fmaf chain over k = 0..K-1, bit-identical to the unsegmented kernel, to the
              // reference and to the oracle).  Jobs (tile, segment) are numbered segment-major and PULLED by persistent workgroups from one counter, so the producer of any
              // accumulator tile a workgroup waits for is already running: no assumption on dispatch order, placement or co-residency
              // (/opt/skills/guides/cdna_hip_programming.md, Guideline 16).  Hand-off = the guide's write-through recipe: 16-byte sc1 slab stores, every wave drains vmcnt,
              // barrier, one sc1 flag store; the consumer's first wave polls the flag (sc1 load, s_sleep, bounded), barrier, 16-byte sc1 slab loads.  Workspace p.ws: word 0
              // the job counter, word 1 the exit counter, word 2 "a wait timed out", words 16.. one flag per tile, slabs from p.ws + p.ws_slab on -- counters and flags are
              // zero between launches (the last segment of a tile clears its flag, the last workgroup to leave clears the counters): a captured graph replays as it is.
              // What it is worth (profiles/r05_probe_k_hand_off.txt): a tile's segments run one after the other, so never more chains are active than there are tiles --
              // the gain is only that chains migrate between CUs at segment boundaries.  With one workgroup per CU (no job ever waits) NiN conv4 at 256 images 104.9 ->
              // 111.2 TF/s on 128x256 tiles in eight segments, AlexNet conv5 100.7 -> 103.5; everything else level or slower (a lone workgroup per CU runs these
              // kernels at ~0.72 of peak, two co-resident ones at ~0.88 -- and with two per CU half of the persistent workgroups wait on a predecessor).  Not a default plan.
#ifndef KHO_DBG
#define KHO_DBG 0 // experiment hook (BODAHIP_EXTRA_DEFS): 1 no slab stores / loads | 2 no flags (no poll, no flag stores) | 4 no exit protocol | 8 static job assignment (no counter)
#endif
#ifndef MT
#define MT 32 // MFMA tile: 32 -> v_mfma_f32_32x32x2_f32 (default), 16 -> v_mfma_f32_16x16x4_f32 (4x more, smaller wave tiles for
#endif        // shapes with too few 32x32 tiles to give every SIMD a wave; same fp32 rate, same ascending-k fma chain)

struct gemm_args_t {
  float const *I; float const *J; float *D; float const *bias;
  int Mi, Nj, K;          // extents of i, j, k
  int ldI, ldJ, ldD;      // row pitches (elements) of I, J (k-major: per k row; i/j-major: per i/j row) and of D per i
  int C, H, W, OH, OW;    // convolution geometry (J_MODE 2 / EPI 1)
  int tiles_i, tiles_j;
  int splitk, kt_per;     // SPLITK: number of K slices, K-tiles per slice
  float *ws; long ws_slab; // SPLITK: partial-sum slabs, ws_slab elements apart
  unsigned I_bytes, J_bytes; // sizes of the I / J tensors (buffer-descriptor num_records; host guarantees <= 2^31)
  unsigned D_bytes;             // size of the output tensor (host guarantees < 2^32: 32-bit store offsets)
  int out_ctot, out_coff;      // EPI 1: channels of the output tensor and first channel written (== Mi, 0 unless the conv writes a slice
                               // of a wider tensor: Concat elimination)
  int const *ktab; int ktab_n; // J_MODE 2: per-k gather tables, three arrays of ktab_n ints: element offset of (in_chan,ky,kx)
                               // inside one image | ky | kx ; rows k >= K carry ky = 2^30 (fail the row-range test)
  long bsI, bsJ, bsD;          // batched launches (gridDim.y problems of one shape, e.g. the 16 Winograd-domain sgemms): element strides of I / J / D per blockIdx.y
};

#ifndef REDUCE_ONLY
namespace {
constexpr int kNT = WI * WJ * 64;
constexpr int kTI = BI / (WI * MT);
constexpr int kTJ = BJ / (WJ * MT);
constexpr int kKS = (MT == 32) ? 2 : 4;   // k consumed per MFMA
constexpr int kNA = (MT == 32) ? 16 : 4;  // accumulator registers per MFMA tile
static_assert(MT == 32 || MT == 16, "MT must be 32 or 16");
#if MT == 32
typedef f32x16 acc_t;
#else
typedef f32x4 acc_t;
#endif
constexpr int kPAD = 4;
constexpr int kLDI = BI + kPAD;
constexpr int kLDJ = BJ + kPAD;
constexpr int kITile = BK * kLDI;
static_assert(BI % (WI * MT) == 0 && BJ % (WJ * MT) == 0, "tile must be a multiple of the MFMA tile per wave");
static_assert(MT == 32 || BK % 4 == 0, "16x16x4 MFMA consumes four k per step");
static_assert(BK % 2 == 0, "BK must be even (two k per MFMA)");
// staged floats per thread: whole passes of kNT threads, the last pass may be partial (lanes past the tile read nothing and
// store nothing); vector modes stage 4 (2) floats per lane per pass
constexpr int vec_w(int mode) { return (mode == 0 || mode == 2) ? 4 : (mode == 4 ? 2 : 1); }
constexpr int staged(int elems, int w) { return w * ((elems / w + kNT - 1) / kNT); }
constexpr int kNI = staged(BK * BI, vec_w(I_MODE));
static_assert(vec_w(I_MODE) == 1 || BK % vec_w(I_MODE) == 0, "vector staging of I needs BK % width == 0");
#if J_MODE == 7
// Patch mode (stride-1-in-x KH x KW convolutions): the K step is kCB whole input channels; instead of an im2col image the LDS
// holds, per channel, the zero-padded input rows the tile's output positions touch ("slots" of kWp floats).  Output rows of one
// image share slots (row r+1 starts SY slots after row r); a tile that crosses into the next image starts a new slot group.
//   element J(k=(c,ky,kx), j) = patch[c][ slot(j) + ky ][ ox(j)*SX + kx ]  =  lds[ bj(j) + koff(k) ]
// -> the MFMA B operand is read straight from the patch (per-lane base + per-k constant); staging a channel costs
//    kCS/kNT coalesced loads per thread instead of KH*KW*BJ/kNT gathered ones, and needs no per-element index arithmetic.
#ifndef RDEC
#define RDEC 0 // 1: ROW-DECIMATED patch for strided convolutions without padding (conv1 layers: 11x11 / 4).  The convolution is presented as C0*KH0 "channels" -- row set
#endif         // (in_chan, kernel row) -- of KH = 1 x KW kernels with stride 1 in y over a plane of COH rows: row r of row set (c, ky) is input row r * SY0 + ky.  k = (c, ky, kx)
               // keeps its order, a K step is kCB whole row sets, the LDS holds ONE input row per output row and row set (coalesced row loads instead of a gather of
               // KW-tap windows that overlap (KW - SX) / KW), the MFMA B operand is read in place at lane stride SX.  Needs -DC0 -DH0 -DKH0 -DSY0; p.C = C0 * KH0.
#ifndef PKH
#define PKH 1  // PKH x PKW > 1 (round 5): a MAX POOLING fused in front of the convolution (the reference runs two functions: test/rtc/pool.cucl, then the conv -- src/rtc_fwd.cc:545-549).
#define PKW 1  // The convolution's "input plane" CH x CW is then the POOLED plane; the tensor behind p.J is the pooling's input, planes of UH x UW, and patch element
#define PSY 1  // (c, iy, ix) is max_{dy < PKH, dx < PKW} in[img][c][iy * PSY + dy][ix * PSX + dx] -- formed in registers while the patch is staged (PKH * PKW loads per element instead
#define PSX 1  // of one: a K step of the patch is a few hundred elements per workgroup against ~100 MFMAs of 64 cycles per wave).  Only windows that never leave the plane
#define UH CH  // ((CH - 1) * PSY + PKH <= UH, no pooling pad: NiN's and AlexNet's 3x3 / 2 pools); the maximum is exact, the convolution's fma chains are its own: bit-identical to
#define UW CW  // pooling and convolution run apart.  The loads of window row r fly under a third of the K step's MFMAs (PF == 1 only).
#endif
#if !defined(CH) || !defined(CW) || !defined(COH) || !defined(COW)
#error "J_MODE 7 needs -DCH -DCW -DCOH -DCOW (input / output plane sizes are compile-time)"
#endif
static_assert(PKH >= 1 && PKH <= 3 && PKW >= 1 && PKW <= 3 && (PKH * PKW == 1 || (!RDEC && !SPECW && (PF == 1) && (CH - 1) * PSY + PKH <= UH && (CW - 1) * PSX + PKW <= UW)), "fused pooling: windows of up to 3 x 3 that stay inside the plane, plain patch mode, one K tile in flight");
constexpr int kTaps = KH * KW, kCB = BK / kTaps, kWp = CW + 2 * PX;
static_assert(BK % kTaps == 0 && KH >= SY && MT == 32, "patch mode geometry");
constexpr int kRowsMax = (BJ - 2) / COW + 2;                     // output rows a BJ-pel tile can touch
constexpr int kSegFull = (COH - 1) * SY + KH;                    // slots of a whole image
constexpr int kSegMax0 = (COH - 1 + kRowsMax - 1) / COH + 1;     // images a tile can touch
constexpr int kSegMax = kSegMax0 < kRowsMax ? kSegMax0 : kRowsMax;
constexpr int kSlots = (kRowsMax - kSegMax) * SY + kSegMax * KH; // slots per channel (upper bound over tile positions)
constexpr int kCS = kSlots * kWp;                                // floats per channel
constexpr int kEPT = (kCS + kNT - 1) / kNT;                      // patch elements per thread per channel
constexpr int kNJ = kCB * kEPT;
constexpr int kJTile = kCB * kCS;
constexpr int koff(int k) { return (k / kTaps) * kCS + ((k % kTaps) / KW) * kWp + (k % KW); }
// k = 2kk+1 sits a fixed distance after k = 2kk: next tap of the row | first tap of the next row | first tap of the next channel.
// Lanes 32-63 (odd k) fold that distance into their base address once; the even-k offset koff(2kk) is then an immediate.
constexpr int kD0 = 1, kD1 = kWp - KW + 1, kD2 = kCS - (KH - 1) * kWp - (KW - 1);
constexpr int kdelta_class(int kk) { return ((2 * kk) % KW != KW - 1) ? 0 : ((((2 * kk) % kTaps) != kTaps - 1) ? 1 : 2); }
#else
constexpr int kNJ = (J_MODE == 2 || J_MODE == 5 || J_MODE == 6) ? (BK * BJ / kNT) : staged(BK * BJ, (J_MODE == 0 || J_MODE == 3) ? 4 : 1);
constexpr int kJTile = BK * kLDJ;
#endif

// ---------------------------------------------------------------------------------------------------------------
// global -> registers.  MODE 0/1: k-major rows of BX floats (x contiguous); MODE 2/3: x-major rows, k contiguous.
// ---------------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOOB = (int)0x80000000; // byte offset beyond any num_records (tensors are <= 2^31 bytes): hardware returns 0
__device__ __forceinline__ rsrc_t make_rsrc(float const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ f32x4 bload4(rsrc_t r, int byte_off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0)); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
// (whole-vector bit_casts: element-wise bit_casts of the b64/b96 results get mis-shrunk to a 1-dword load by this compiler)
__device__ __forceinline__ f32x2 bload2(rsrc_t r, int byte_off) { return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0)); }
__device__ __forceinline__ f32x3 bload3(rsrc_t r, int byte_off) { return __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, byte_off, 0, 0)); }
__device__ __forceinline__ float bload1(rsrc_t r, int byte_off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0)); }

#if HALF
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
static_assert((I_MODE == 0 || I_MODE == 1) && (J_MODE == 0 || J_MODE == 1) && EPI == 0 && !SPLITK, "half storage: plain k-major sgemm operands only");
#endif
template <int MODE, int BX, int NR>
__device__ __forceinline__ void load_tile(float (&r)[NR], rsrc_t P, int ld, int x0, int X, int k0, int K, int tid) {
#if HALF
  if constexpr (MODE == 0) {   // four consecutive halves = one 8-byte load (rows start 8-byte aligned: ld % 4 == 0)
    constexpr int VPR = BX / 4, TOT = BK * VPR;
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      int const v = tid + p * kNT, row = v / VPR, c4 = v % VPR;
      int const k = k0 + row, x = x0 + 4 * c4;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (v < TOT);
      h16x4 const val = __builtin_bit_cast(h16x4, __builtin_amdgcn_raw_buffer_load_b64(P, (in_tile && (k < K) && (x < X)) ? ((k * ld + x) * 2) : kOOB, 0, 0));
      r[4 * p + 0] = (float)val[0]; r[4 * p + 1] = (float)val[1]; r[4 * p + 2] = (float)val[2]; r[4 * p + 3] = (float)val[3];
    }
  } else {
    constexpr int TOT = BK * BX;
#pragma unroll
    for (int p = 0; p < NR; ++p) {
      int const e = tid + p * kNT, row = e / BX, c = e % BX;
      int const k = k0 + row, x = x0 + c;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (e < TOT);
      r[p] = (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(P, (in_tile && (k < K) && (x < X)) ? ((k * ld + x) * 2) : kOOB, 0, 0));
    }
  }
#else
  if constexpr (MODE == 0) {
    constexpr int VPR = BX / 4, TOT = BK * VPR;
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      int const v = tid + p * kNT, row = v / VPR, c4 = v % VPR;
      int const k = k0 + row, x = x0 + 4 * c4;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (v < TOT);
      f32x4 const val = bload4(P, (in_tile && (k < K) && (x < X)) ? ((k * ld + x) * 4) : kOOB);
      r[4 * p + 0] = val[0]; r[4 * p + 1] = val[1]; r[4 * p + 2] = val[2]; r[4 * p + 3] = val[3];
    }
  } else if constexpr (MODE == 1) {
    constexpr int TOT = BK * BX;
#pragma unroll
    for (int p = 0; p < NR; ++p) {
      int const e = tid + p * kNT, row = e / BX, c = e % BX;
      int const k = k0 + row, x = x0 + c;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (e < TOT);
      r[p] = bload1(P, (in_tile && (k < K) && (x < X)) ? ((k * ld + x) * 4) : kOOB);
    }
  } else if constexpr (MODE == 2) {
    constexpr int VPR = BK / 4, TOT = BX * VPR;
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      int const v = tid + p * kNT, xr = v / VPR, k4 = v % VPR;
      int const x = x0 + xr, k = k0 + 4 * k4;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (v < TOT);
      f32x4 const val = bload4(P, (in_tile && (x < X) && (k < K)) ? ((x * ld + k) * 4) : kOOB);
      r[4 * p + 0] = val[0]; r[4 * p + 1] = val[1]; r[4 * p + 2] = val[2]; r[4 * p + 3] = val[3];
    }
  } else if constexpr (MODE == 4) {
    constexpr int VPR = BK / 2, TOT = BX * VPR;
#pragma unroll
    for (int p = 0; p < NR / 2; ++p) {
      int const v = tid + p * kNT, xr = v / VPR, k2 = v % VPR;
      int const x = x0 + xr, k = k0 + 2 * k2;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (v < TOT);
      f32x2 const val = bload2(P, (in_tile && (x < X) && (k < K)) ? ((x * ld + k) * 4) : kOOB);
      r[2 * p + 0] = val[0]; r[2 * p + 1] = val[1];
    }
  } else {
    constexpr int TOT = BK * BX;
#pragma unroll
    for (int p = 0; p < NR; ++p) {
      int const e = tid + p * kNT, xr = e / BK, kk = e % BK;
      int const x = x0 + xr, k = k0 + kk;
      bool const in_tile = ((p + 1) * kNT <= TOT) || (e < TOT);
      r[p] = bload1(P, (in_tile && (x < X) && (k < K)) ? ((x * ld + k) * 4) : kOOB);
    }
  }
#endif
}

// registers -> LDS image [BK][LD] (k-major)
template <int MODE, int BX, int LD, int NR>
__device__ __forceinline__ void store_tile(float const (&r)[NR], float *__restrict__ S, int tid) {
  if constexpr (MODE == 0) {
    constexpr int VPR = BX / 4, TOT = BK * VPR;
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      int const v = tid + p * kNT, row = v / VPR, c4 = v % VPR;
      f32x4 val = {r[4 * p + 0], r[4 * p + 1], r[4 * p + 2], r[4 * p + 3]};
      if (((p + 1) * kNT <= TOT) || (v < TOT)) *reinterpret_cast<f32x4 *>(S + row * LD + 4 * c4) = val;
    }
  } else if constexpr (MODE == 1) {
    constexpr int TOT = BK * BX;
#pragma unroll
    for (int p = 0; p < NR; ++p) { int const e = tid + p * kNT; if (((p + 1) * kNT <= TOT) || (e < TOT)) S[(e / BX) * LD + (e % BX)] = r[p]; }
  } else if constexpr (MODE == 2) {
    constexpr int VPR = BK / 4, TOT = BX * VPR;
#pragma unroll
    for (int p = 0; p < NR / 4; ++p) {
      int const v = tid + p * kNT, xr = v / VPR, k4 = v % VPR;
      if (((p + 1) * kNT <= TOT) || (v < TOT)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) S[(4 * k4 + e) * LD + xr] = r[4 * p + e];
      }
    }
  } else if constexpr (MODE == 4) {
    constexpr int VPR = BK / 2, TOT = BX * VPR;
#pragma unroll
    for (int p = 0; p < NR / 2; ++p) {
      int const v = tid + p * kNT, xr = v / VPR, k2 = v % VPR;
      if (((p + 1) * kNT <= TOT) || (v < TOT)) { S[(2 * k2) * LD + xr] = r[2 * p]; S[(2 * k2 + 1) * LD + xr] = r[2 * p + 1]; }
    }
  } else {
    constexpr int TOT = BK * BX;
#pragma unroll
    for (int p = 0; p < NR; ++p) { int const e = tid + p * kNT; if (((p + 1) * kNT <= TOT) || (e < TOT)) S[(e % BK) * LD + (e / BK)] = r[p]; }
  }
}

#if J_MODE == 6
// Row gather for KH x KW convolutions (KW >= 2; the host picks it for KW >= 6, where it measures faster).  The KW taps of one (in_chan, ky) row are contiguous in memory for a fixed
// output position, whatever the stride: a K step is JROWS whole rows (BK = JROWS*KW), each thread owns one output position and
// kRPT rows, and fetches a row with ONE address computation and ceil(KW/4) wide buffer loads.  Row validity (zero padding
// above/below, K tail) makes the whole row's offset out of range; column validity (left/right padding) depends only on the
// thread's ix0 and kx, is computed once per thread, and is applied as a select when the row is written to LDS (after the
// MFMAs, so the loads stay in flight).  Per gathered element: ~1.3 VALU + 1/KW address + 1/min(KW,4) loads, vs ~6 + 1 + 1
// of the per-element gather (J_MODE 2).  k order is unchanged (ascending (in_chan,ky,kx)): results stay bit-identical.
#ifndef JROWS
#error "J_MODE 6 needs -DJROWS"
#endif
constexpr int kRPP = kNT / BJ;      // row groups per K step (a wave belongs to exactly one)
constexpr int kRPT = JROWS / kRPP;  // rows per thread per K step
static_assert(BJ % 64 == 0 && kNT % BJ == 0 && JROWS % kRPP == 0 && BK == JROWS * KW && KW >= 2, "row gather geometry");
struct gather_t { int base; int iy0; bool mx[KW]; bool edge_tile; };
template <int N> __device__ __forceinline__ void bload_n(float *dst, rsrc_t r, int off) {
  if constexpr (N >= 4) { f32x4 const v = bload4(r, off); dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; if constexpr (N > 4) bload_n<N - 4>(dst + 4, r, off + 16); }
  else if constexpr (N == 3) { f32x3 const v = bload3(r, off); dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; }
  else if constexpr (N == 2) { f32x2 const v = bload2(r, off); dst[0] = v[0]; dst[1] = v[1]; }
  else { dst[0] = bload1(r, off); }
}
__device__ __forceinline__ void load_gather(float (&r)[kNJ], rsrc_t in, gather_t const &g, gemm_args_t const &p, int k0, int tid) {
  int const row0 = __builtin_amdgcn_readfirstlane(tid / BJ);
  int const rr = k0 / KW + row0 * kRPT; // first (in_chan,ky) row of this wave (scalar)
  typedef int const __attribute__((address_space(4))) *ctab_t;
  ctab_t const t_off = (ctab_t)(p.ktab + rr), t_ky = (ctab_t)(p.ktab + p.ktab_n + rr);
#pragma unroll
  for (int q = 0; q < kRPT; ++q) {
    int const iy = g.iy0 + t_ky[q];
    int off = (g.base + t_off[q]) * 4;
    asm volatile("" : "+v"(off));
    bool const row_ok = (unsigned)iy < (unsigned)p.H;
    if (PX > 0 && g.edge_tile) {
      // With left/right padding a row window can start before the tensor (very first input row: negative = out-of-range
      // offset) or end past it (very last input row: the hardware range-checks a wide load as a whole, so the in-range taps
      // of a window that crosses num_records would read 0 too).  Only tiles at the very start / end of the pel range can contain
      // such a window: they gather tap by tap.
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) r[q * KW + kx] = bload1(in, (row_ok && g.mx[kx]) ? (off + 4 * kx) : kOOB);
    } else {
      bload_n<KW>(&r[q * KW], in, row_ok ? off : kOOB);
    }
  }
}
__device__ __forceinline__ void store_gather(float const (&r)[kNJ], float *__restrict__ S, gather_t const &g, int tid) {
  int const row0 = tid / BJ, jj = tid % BJ;
#pragma unroll
  for (int q = 0; q < kRPT; ++q)
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) S[((row0 * kRPT + q) * KW + kx) * kLDJ + jj] = g.mx[kx] ? r[q * KW + kx] : 0.f;
}
#define GATHER_ARG , g
#define GATHER_PARM , gather_t const &g
#elif J_MODE == 7
struct gather_t { int goff[kEPT]; }; // byte offset of (img, chan 0, iy, ix) of this thread's patch elements; kOOB: zero (padding / past the end)
__device__ __forceinline__ void load_gather(float (&r)[kNJ], rsrc_t in, gather_t const &g, gemm_args_t const &p, int k0, int tid) {
  int const c0 = k0 / kTaps; // scalar
#pragma unroll
  for (int cc = 0; cc < kCB; ++cc) {
    // channels past the end (K tail) meet zero filter values: any finite data will do -> re-read the last channel of the image
#if RDEC
    int const cq = min(c0 + cc, p.C - 1);                           // "channel" = row set (in_chan cq / KH0, kernel row cq % KH0) of the strided convolution
    int const coff = ((cq / KH0) * H0 + cq % KH0) * (CW * 4);
#else
    int const coff = min(c0 + cc, p.C - 1) * (UH * UW * 4); // scalar; goff + coff < 2^32 and stays >= 2^31 for kOOB entries
#endif
#pragma unroll
    for (int e = 0; e < kEPT; ++e) {
      float v = bload1(in, g.goff[e] + coff);
#if PKH * PKW > 1
#pragma unroll
      for (int t = 1; t < PKH * PKW; ++t) v = fmaxf(v, bload1(in, g.goff[e] + coff + ((t / PKW) * UW + (t % PKW)) * 4));   // (a padding element's offsets all stay out of range: max of zeros)
#endif
      r[cc * kEPT + e] = v;
    }
  }
}
#if PKH * PKW > 1
// fused pooling, one window ROW at a time: the PKW loads of row `wr` of every patch element of the K step (issued before a third of the step's MFMAs) ...
__device__ __forceinline__ void load_pool_row(float (&raw)[PKW * kNJ], rsrc_t in, gather_t const &g, gemm_args_t const &p, int k0, int const wr) {
  int const c0 = k0 / kTaps;
#pragma unroll
  for (int cc = 0; cc < kCB; ++cc) {
    int const coff = min(c0 + cc, p.C - 1) * (UH * UW * 4) + wr * (UW * 4);
#pragma unroll
    for (int e = 0; e < kEPT; ++e)
#pragma unroll
      for (int x = 0; x < PKW; ++x) raw[(cc * kEPT + e) * PKW + x] = bload1(in, g.goff[e] + coff + x * 4);
  }
}
// ... and their maximum folded into the element's running maximum (after those MFMAs)
__device__ __forceinline__ void fold_pool_row(float (&r)[kNJ], float const (&raw)[PKW * kNJ], bool const first) {
#pragma unroll
  for (int n = 0; n < kNJ; ++n) {
    float v = raw[n * PKW];
#pragma unroll
    for (int x = 1; x < PKW; ++x) v = fmaxf(v, raw[n * PKW + x]);
    r[n] = first ? v : fmaxf(r[n], v);
  }
}
#endif
__device__ __forceinline__ void store_gather(float const (&r)[kNJ], float *__restrict__ S, int tid) {
#pragma unroll
  for (int cc = 0; cc < kCB; ++cc)
#pragma unroll
    for (int e = 0; e < kEPT; ++e)
      if (((e + 1) * kNT <= kCS) || (tid + e * kNT < kCS)) S[cc * kCS + e * kNT + tid] = r[cc * kEPT + e];
}
#define GATHER_ARG , g
#define GATHER_PARM , gather_t const &g
#elif J_MODE == 5
// 1x1 / pad 0 convolution: thread = one output position, wave = kNJ consecutive input channels per K step.
struct gather_t { int base4; }; // byte offset of (img, chan 0, oy*SY, ox*SX); columns past the end: 0x80000000 (out of range for every k)
static_assert(BJ % 64 == 0 && kNT % BJ == 0, "gather: a wave must sit inside one k row (BJ multiple of 64, BJ <= threads)");
constexpr int kKTail = 0x7ffffff0; // scalar offset for k >= K: base4 + kKTail >= num_records (tensors are < 2^31 - 16 bytes), no 32-bit wrap
__device__ __forceinline__ void load_gather(float (&r)[kNJ], rsrc_t in, gather_t const &g, gemm_args_t const &p, int k0, int tid) {
  int const row0 = __builtin_amdgcn_readfirstlane(tid / BJ);
  int const kb = k0 + row0 * kNJ, hw4 = p.H * p.W * 4;
#pragma unroll
  for (int q = 0; q < kNJ; ++q) {
    int const kg = kb + q;
    int const soff = (kg < p.K) ? kg * hw4 : kKTail; // scalar unit
    r[q] = bload1(in, g.base4 + soff);
  }
}
__device__ __forceinline__ void store_gather(float const (&r)[kNJ], float *__restrict__ S, int tid) {
  int const row0 = tid / BJ, jj = tid % BJ;
#pragma unroll
  for (int q = 0; q < kNJ; ++q) S[(row0 * kNJ + q) * kLDJ + jj] = r[q];
}
#define GATHER_ARG , g
#define GATHER_PARM , gather_t const &g
#elif J_MODE == 2
// im2col gather.  Each thread serves one fixed output position (column jj of the tile); each wave serves kNJ consecutive
// k rows per K step.  Everything that depends only on k -- the (in_chan,ky,kx) decode and the element offset inside an
// image -- comes from a small host-built table read through the scalar cache (one s_load per wave per K step), so a
// gathered element costs ~6 VALU ops: 2 adds + 2 unsigned compares (halo / zero padding), offset add, select.
// Offsets are 32-bit (the host guarantees the input tensor is <= 2^31 bytes).
struct gather_t { int base; int iy0, ix0; };
static_assert(BJ % 64 == 0 && kNT % BJ == 0, "gather: a wave must sit inside one k row (BJ multiple of 64, BJ <= threads)");
static_assert(kNJ % 4 == 0, "gather: rows per wave must be a multiple of 4 (vector scalar loads)");
__device__ __forceinline__ void load_gather(float (&r)[kNJ], rsrc_t in, gather_t const &g, gemm_args_t const &p, int k0, int tid) {
  int const row0 = __builtin_amdgcn_readfirstlane(tid / BJ); // wave-uniform
  int const kb = k0 + row0 * kNJ;                           // multiple of 4: 16-B aligned table rows
  // constant address space + wave-uniform address => s_load_dwordx4 through the scalar cache (the table is never written by a kernel)
  typedef int4 const __attribute__((address_space(4))) *ctab_t;
  ctab_t const t_off = (ctab_t)(p.ktab + kb), t_ky = (ctab_t)(p.ktab + p.ktab_n + kb), t_kx = (ctab_t)(p.ktab + 2 * p.ktab_n + kb);
#pragma unroll
  for (int q4 = 0; q4 < kNJ / 4; ++q4) {
    int4 const ko = t_off[q4], ky = t_ky[q4], kx = t_kx[q4]; // s_load_dwordx4 each (scalar cache)
    int const kov[4] = {ko.x, ko.y, ko.z, ko.w}, kyv[4] = {ky.x, ky.y, ky.z, ky.w}, kxv[4] = {kx.x, kx.y, kx.z, kx.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int const iy = g.iy0 + kyv[e], ix = g.ix0 + kxv[e];
      int off = (g.base + kov[e]) * 4;
      asm volatile("" : "+v"(off)); // keep the offset unconditional: the select below must stay a v_cndmask, not a branch
      bool const ok = ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
      r[q4 * 4 + e] = bload1(in, ok ? off : kOOB); // masked lanes: out-of-range offset -> hardware returns 0
    }
  }
}
__device__ __forceinline__ void store_gather(float const (&r)[kNJ], float *__restrict__ S, int tid) {
  int const row0 = tid / BJ, jj = tid % BJ;
#pragma unroll
  for (int q = 0; q < kNJ; ++q) S[(row0 * kNJ + q) * kLDJ + jj] = r[q];
}
#define GATHER_ARG , g
#define GATHER_PARM , gather_t const &g
#else
#define GATHER_ARG
#define GATHER_PARM
#endif

__device__ __forceinline__ void load_J(float (&rj)[kNJ], rsrc_t J, gemm_args_t const &p, int j0, int k0, int tid GATHER_PARM) {
#if J_MODE == 2 || J_MODE == 5 || J_MODE == 6 || J_MODE == 7
  load_gather(rj, J, g, p, k0, tid);
#elif J_MODE == 3 || J_MODE == 4
  load_tile<J_MODE - 1, BJ, kNJ>(rj, J, p.ldJ, j0, p.Nj, k0, p.K, tid);
#else
  load_tile<J_MODE, BJ, kNJ>(rj, J, p.ldJ, j0, p.Nj, k0, p.K, tid);
#endif
}
__device__ __forceinline__ void store_J(float const (&rj)[kNJ], float *__restrict__ S, int tid GATHER_PARM) {
#if J_MODE == 6
  store_gather(rj, S, g, tid);
#elif J_MODE == 2 || J_MODE == 5 || J_MODE == 7
  store_gather(rj, S, tid);
#elif J_MODE == 3 || J_MODE == 4
  store_tile<J_MODE - 1, BJ, kLDJ, kNJ>(rj, S, tid);
#else
  store_tile<J_MODE, BJ, kLDJ, kNJ>(rj, S, tid);
#endif
}
// one K-tile of MFMAs out of LDS (or its k pairs [KK0, KK1)): Ic / Jc point at this lane's first A / B element of the tile
template <int KK0 = 0, int KK1 = -1>
__device__ __forceinline__ void mma_ktile(acc_t (&acc)[kTI][kTJ], float const *__restrict__ Ic, float const *__restrict__ Jc
#if J_MODE == 7
                                          , int const (&bj)[kTJ][3]
#endif
) {
#if SETPRIO
  __builtin_amdgcn_s_setprio(1); // co-resident waves of other workgroups are in their load phase: favour the MFMA issuer
#endif
#pragma unroll
  for (int kk = KK0; kk < ((KK1 < 0) ? (BK / kKS) : KK1); ++kk) {
    float a[kTI], b[kTJ];
#pragma unroll
    for (int t = 0; t < kTI; ++t) a[t] = Ic[kk * kKS * kLDI + t * MT];
#pragma unroll
#if J_MODE == 7
    for (int t = 0; t < kTJ; ++t) b[t] = Jc[bj[t][kdelta_class(kk)] + koff(2 * kk)];
#else
    for (int t = 0; t < kTJ; ++t) b[t] = Jc[kk * kKS * kLDJ + t * MT];
#endif
#pragma unroll
    for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
      for (int tb = 0; tb < kTJ; ++tb) {
#if MT == 32
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
#else
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
#endif
      }
  }
#if SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}
} // namespace

extern "C" __global__ __launch_bounds__(WI * WJ * 64 * (SPECW ? 2 : 1), MINW) void KNAME(gemm_args_t const p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (kITile + kJTile)];
#if SPECW
  bool const stager = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= WI * WJ;
  int const tid = threadIdx.x - (stager ? kNT : 0);   // id among the staging threads / among the multiplying threads
#else
  int const tid = threadIdx.x;
#endif
  int const lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave / WJ, wj = wave % WJ;

#if KHO
  static_assert(!SPLITK && !SPECW, "K hand-off: an exact form -- no K slices; the staging-wave form returns early");
  int const n_tiles_all = p.tiles_i * p.tiles_j, n_jobs = n_tiles_all * p.splitk;
  // The counters and flags are reached through a buffer descriptor, by the first wave only, with every lane but lane 0 given an out-of-range offset (dropped by the
  // hardware): no `if (threadIdx.x == 0)` anywhere in the job loop.  (With such divergent branches -- and a poll loop inside one -- this compiler turned the job loop
  // and the K loop into exec-masked divergent loops and lost the flag address on the way: memory faults.)  Every branch the hand-off adds is wave-uniform.
  rsrc_t const rQ = make_rsrc(p.ws, (unsigned)(16 + n_tiles_all) * 4u);
  int const q_l0 = (lane == 0) ? 0 : kOOB;
  int kho_pass = 0; (void)kho_pass;
  for (;;) {   // persistent workgroup: one (tile, segment) job per pass
#if KHO_DBG & 8
  int const job = (int)blockIdx.x + (int)gridDim.x * kho_pass; ++kho_pass;
#else
  if (wave == 0) reinterpret_cast<int *>(smem)[0] = __builtin_amdgcn_readfirstlane((lane == 0) ? __hip_atomic_fetch_add(reinterpret_cast<int *>(p.ws), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
  __syncthreads();   // (every wave is past the previous job's last LDS read: each K step ends with a barrier, the epilogue does not touch the LDS)
  int const job = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<unsigned *>(smem)[0]);
  __syncthreads();
#endif
  if (job >= n_jobs) break;
  int const seg = job / n_tiles_all, bid = job - seg * n_tiles_all;
  int const flag_off = (16 + bid) * 4;   // (wave-uniform: job comes out of a readfirstlane)
#elif SPLITK
  // ---- XCD-aware workgroup -> tile map (bijective for any grid size) ------------------------------------------
  int const bid = blockIdx.x / p.splitk, slice = blockIdx.x % p.splitk;
#else
  int const bid = blockIdx.x;
#endif
  int tile_i, tile_j;
  {
    int const nb = p.tiles_i * p.tiles_j;
    int const q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    int const nid = ((xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    int const group_sz = GROUP_I * p.tiles_j, gid = nid / group_sz, first_i = gid * GROUP_I;
    int const gsz = min(p.tiles_i - first_i, GROUP_I), in_g = nid - gid * group_sz;
    tile_i = first_i + in_g % gsz; tile_j = in_g / gsz;
  }
  int const i0 = tile_i * BI, j0 = tile_j * BJ;

  float *const Is0 = smem, *const Is1 = smem + kITile;
  float *const Js0 = smem + 2 * kITile, *const Js1 = smem + 2 * kITile + kJTile;

#if J_MODE == 6
  gather_t g;
  {
    int const OHW = p.OH * p.OW;
    int const jg = j0 + (tid % BJ);
    int const img = jg / OHW, pel = jg - img * OHW, oy = pel / p.OW, ox = pel - oy * p.OW;
    int const ix0 = ox * SX - PX;
    g.iy0 = (jg < p.Nj) ? (oy * SY - PY) : (1 << 29); // columns past the end fail the row test for every row
    g.base = (img * p.C * p.H + (oy * SY - PY)) * p.W + ix0;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) g.mx[kx] = (unsigned)(ix0 + kx) < (unsigned)p.W;
    // tiles holding output rows whose window touches the first input row of image 0 (oy <= PY/SY) or the last input row of the
    // last image (at most KH/SY+1 output rows): workgroup-uniform
    g.edge_tile = (j0 < (PY / SY + 1) * p.OW) || (j0 + BJ > p.Nj - (KH / SY + 1) * p.OW);
  }
#elif J_MODE == 7
  gather_t g;
  int bj[kTJ][3];
  {
    int const R0 = j0 / COW, img0 = R0 / COH, oy0 = R0 - img0 * COH;       // first output row of the tile (workgroup-uniform)
    int const seg0 = (COH - 1 - oy0) * SY + KH;                            // slots of the first image's part
    int const n_img = p.Nj / (COH * COW);
#pragma unroll
    for (int e = 0; e < kEPT; ++e) { // this thread's patch elements: slot s, padded column x
      int const el = tid + e * kNT, s = el / kWp, ix = el - s * kWp - PX;
      int const s2 = s - seg0, im2 = s2 / kSegFull;
      int const img = (s < seg0) ? img0 : (img0 + 1 + im2);
      int const iy = (s < seg0) ? (oy0 * SY - PY + s) : (s2 - im2 * kSegFull - PY);
      bool const ok = (el < kCS) && (img < n_img) && ((unsigned)iy < (unsigned)CH) && ((unsigned)ix < (unsigned)CW);
#if RDEC
      g.goff[e] = ok ? (((img * (C0 * H0) + iy * SY0) * CW + ix) * 4) : kOOB;   // row iy of the decimated plane is input row iy * SY0 (+ the row set's kernel row, in coff)
#else
      g.goff[e] = ok ? (((img * p.C * UH + iy * PSY) * UW + ix * PSX) * 4) : kOOB;   // (fused pooling: the window's first element in the pooling's input)
#endif
    }
#pragma unroll
    for (int t = 0; t < kTJ; ++t) { // MFMA B operand: this lane's output positions
      int const jg = min(j0 + wj * (kTJ * MT) + t * MT + (lane % MT), p.Nj - 1);
      int const R = jg / COW, ox = jg - R * COW, img = R / COH, oy = R - img * COH;
      int const slot = (img == img0) ? ((oy - oy0) * SY) : (seg0 + (img - img0 - 1) * kSegFull + oy * SY);
      int const b0 = slot * kWp + ox * SX;
      bool const odd = (lane / MT) != 0;
      bj[t][0] = b0 + (odd ? kD0 : 0); bj[t][1] = b0 + (odd ? kD1 : 0); bj[t][2] = b0 + (odd ? kD2 : 0);
    }
  }
#elif J_MODE == 5
  gather_t g;
  {
    int const OHW = p.OH * p.OW;
    int const jg = j0 + (tid % BJ);
    int const img = jg / OHW, pel = jg - img * OHW, oy = pel / p.OW, ox = pel - oy * p.OW;
    g.base4 = (jg < p.Nj) ? (((img * p.C * p.H + oy * SY) * p.W + ox * SX) * 4) : kOOB;
  }
#elif J_MODE == 2
  gather_t g;
  {
    int const OHW = p.OH * p.OW;
    int const jg = j0 + (tid % BJ);
    int const img = jg / OHW, pel = jg - img * OHW, oy = pel / p.OW, ox = pel - oy * p.OW;
    g.iy0 = (jg < p.Nj) ? (oy * SY - PY) : (1 << 29); // columns past the end fail the row-range test for every k
    g.ix0 = ox * SX - PX;
    g.base = (img * p.C * p.H + (oy * SY - PY)) * p.W + g.ix0;
  }
#endif

  acc_t acc[kTI][kTJ];
#if KHO
  constexpr int kHQ = kNA / 4, kSlabB = kTI * kTJ * kHQ * kNT * 16;   // accumulator quads per MFMA tile; bytes of a tile's slab: the accumulator registers, thread by thread (coalesced 16-byte accesses)
  rsrc_t const rW = make_rsrc(p.ws + p.ws_slab + (long)bid * (long)(kSlabB / 4), (unsigned)kSlabB);
  if (seg > 0 && !(KHO_DBG & 1)) {   // continue the chains of segment seg - 1: wait for its accumulators (its workgroup pulled its job before this one was pulled: it is running or done)
    if (wave == 0 && !(KHO_DBG & 2)) {   // (a scalar branch: `wave` comes out of a readfirstlane)
      // The first wave polls as a whole, on a wave-uniform value (every lane loads the same word; readfirstlane): a WAVE-UNIFORM loop.  (With the loop inside an
      // `if (threadIdx.x == 0)` the compiler turned the job loop and the K loop around it into exec-masked divergent loops -- and mis-placed the flag address.)
      // Bounded: 4 s of the 100 MHz clock; a hand-off that never arrives -- a bug -- then computes on whatever the slab holds and says so in word 2 of the
      // workspace instead of hanging the device.
      unsigned const t0 = (unsigned)__builtin_amdgcn_s_memrealtime();   // (32 bits of the 100 MHz clock: differences are good for 42 s)
      bool timed_out = false;
      for (;;) {
        unsigned const have = (unsigned)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_raw_buffer_load_b32(rQ, flag_off, 0, 16));   // (sc1: past the L1)
        if (have >= (unsigned)seg) break;
        __builtin_amdgcn_s_sleep(32);
        if ((unsigned)__builtin_amdgcn_s_memrealtime() - t0 > 400000000u) { timed_out = true; break; }
      }
      if (timed_out) { __builtin_amdgcn_raw_buffer_store_b32(1, rQ, q_l0 + 8, 0, 16); __builtin_trap(); }   // (round 6: a lost hand-off fails the launch loudly -- nobody read word 2, and computing on would have given wrong output silently)
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < kTI; ++a)
#pragma unroll
      for (int b = 0; b < kTJ; ++b)
#pragma unroll
        for (int q = 0; q < kHQ; ++q) {
          f32x4 const v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (((a * kTJ + b) * kHQ + q) * kNT + tid) * 16, 0, 16));
          float const y0 = v[0], y1 = v[1], y2 = v[2], y3 = v[3];
          acc[a][b][4 * q] = y0; acc[a][b][4 * q + 1] = y1; acc[a][b][4 * q + 2] = y2; acc[a][b][4 * q + 3] = y3;
        }
  } else
#endif
  {
#pragma unroll
  for (int a = 0; a < kTI; ++a)
#pragma unroll
    for (int b = 0; b < kTJ; ++b)
#pragma unroll
      for (int r = 0; r < kNA; ++r) acc[a][b][r] = 0.f;
  }

  float ri[kNI], rj[kNJ];
  int const nkt_all = (p.K + BK - 1) / BK;
#if KHO
  int const kt_begin = seg * p.kt_per;
  int const nkt = max(0, min(nkt_all, kt_begin + p.kt_per) - kt_begin);
#elif SPLITK
  int const kt_begin = slice * p.kt_per;
  int const nkt = max(0, min(nkt_all, kt_begin + p.kt_per) - kt_begin);
#else
  int const kt_begin = 0;
  int const nkt = nkt_all;
#endif

  rsrc_t const rI = make_rsrc(p.I + (long)blockIdx.y * p.bsI, p.I_bytes), rJ = make_rsrc(p.J + (long)blockIdx.y * p.bsJ, p.J_bytes); // kernel args and block ids only: provably wave-uniform
#if SPECW
  if (stager)
#endif
  {
    load_tile<I_MODE, BI, kNI>(ri, rI, p.ldI, i0, p.Mi, kt_begin * BK, p.K, tid);
    load_J(rj, rJ, p, j0, kt_begin * BK, tid GATHER_ARG);
    store_tile<I_MODE, BI, kLDI, kNI>(ri, Is0, tid);
    store_J(rj, Js0, tid GATHER_ARG);
  }
  __syncthreads();

  // MFMA operand fetch: lane l holds A[i = l % MT][k = l / MT] and B[k = l / MT][j = l % MT]
  int const a_off = wi * (kTI * MT) + (lane % MT) + (lane / MT) * kLDI;
#if J_MODE == 7
  int const b_off = 0;
#else
  int const b_off = wj * (kTJ * MT) + (lane % MT) + (lane / MT) * kLDJ;
#endif

#if J_MODE == 7
#define MMA_KTILE(IS, JS) mma_ktile(acc, (IS) + a_off, (JS) + b_off, bj)
#else
#define MMA_KTILE(IS, JS) mma_ktile(acc, (IS) + a_off, (JS) + b_off)
#endif
#if ABLATE & 8
#define LOAD_I(R, KT)
#else
#define LOAD_I(R, KT) load_tile<I_MODE, BI, kNI>(R, rI, p.ldI, i0, p.Mi, (kt_begin + (KT)) * BK, p.K, tid)
#endif
#if ABLATE & 2
#define LOAD_J(R, KT)
#else
#define LOAD_J(R, KT) load_J(R, rJ, p, j0, (kt_begin + (KT)) * BK, tid GATHER_ARG)
#endif
#if ABLATE & 4
#define STORE_IJ(RI, RJ, IS, JS)
#else
#define STORE_IJ(RI, RJ, IS, JS) do { store_tile<I_MODE, BI, kLDI, kNI>(RI, IS, tid); store_J(RJ, JS, tid GATHER_ARG); } while (0)
#endif

#if SPECW
  // Two roles, the same sequence of barriers.  The staging waves keep two K tiles in flight in two register sets: in step kt they write tile kt + 1 (loaded two steps
  // ago) to the other LDS stage and refill its registers with tile kt + 3.
  if (stager) {
    float ri2[kNI], rj2[kNJ];
    if (nkt > 1) { LOAD_I(ri, 1); LOAD_J(rj, 1); }
    if (nkt > 2) { LOAD_I(ri2, 2); LOAD_J(rj2, 2); }
    for (int kt = 0; kt < nkt; kt += 2) {
      if (kt + 1 < nkt) STORE_IJ(ri, rj, Is1, Js1);
      if (kt + 3 < nkt) { LOAD_I(ri, kt + 3); LOAD_J(rj, kt + 3); }
      __syncthreads();
      if (kt + 1 >= nkt) break;
      if (kt + 2 < nkt) STORE_IJ(ri2, rj2, Is0, Js0);
      if (kt + 4 < nkt) { LOAD_I(ri2, kt + 4); LOAD_J(rj2, kt + 4); }
      __syncthreads();
    }
    return;
  }
  for (int kt = 0; kt < nkt; kt += 2) {
    MMA_KTILE(Is0, Js0);
    __syncthreads();
    if (kt + 1 >= nkt) break;
    MMA_KTILE(Is1, Js1);
    __syncthreads();
  }
#elif PF == 2
  // Two K-tiles in flight: while tile t is multiplied out of LDS, tile t+1 sits in one register set (its loads were issued a
  // whole step earlier) and the loads of tile t+2 are issued into the other.  For workgroups that are alone on their CU
  // (tile-starved shapes: one ~0.5 us MFMA phase per K step against ~1 us of HBM latency) this roughly doubles the bytes in flight.
  float ri2[kNI], rj2[kNJ];
  if (nkt > 1) { LOAD_I(ri, 1); LOAD_J(rj, 1); }
  for (int kt = 0; kt < nkt; kt += 2) {
    if (kt + 2 < nkt) { LOAD_I(ri2, kt + 2); LOAD_J(rj2, kt + 2); }
    MMA_KTILE(Is0, Js0);
    if (kt + 1 < nkt) STORE_IJ(ri, rj, Is1, Js1);
    __syncthreads();
    if (kt + 1 >= nkt) break;
    if (kt + 3 < nkt) { LOAD_I(ri, kt + 3); LOAD_J(rj, kt + 3); }
    MMA_KTILE(Is1, Js1);
    if (kt + 2 < nkt) STORE_IJ(ri2, rj2, Is0, Js0);
    __syncthreads();
  }
#elif PF >= 3
  // PF K-tiles in flight (PF even: 4 | 6 | 8), for the tile-starved long-K shapes whose workgroups run one wave per SIMD and have registers to spare
  // (fully-connected layers: K = 4096 / 9216 on 64-256 tiles): a K step's MFMA phase is ~0.2-0.5 us, an HBM fetch under load 1-2 us, so two tiles ahead
  // leave the matrix pipe idle half of the time.  Register set s = tile mod PF; the LDS stays double-buffered.  In step kt: multiply tile kt out of LDS
  // stage kt & 1, store tile kt + 1 (whose loads were issued PF steps ago) to the other stage, then refill its register set with tile kt + 1 + PF.
  // Same ascending-k MFMA chain per output: bit-identical results.
  static_assert(PF % 2 == 0 && PF <= 8, "PF: 1 | 2 | 4 | 6 | 8");
  float rri[PF][kNI], rrj[PF][kNJ];
#pragma unroll
  for (int u = 1; u <= PF; ++u) if (u < nkt) { LOAD_I(rri[u % PF], u); LOAD_J(rrj[u % PF], u); }
  for (int kb = 0; kb < nkt; kb += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      int const kt = kb + u;
      if (kt < nkt) {
#if STF
        // store first: tile kt + 1 goes to the other LDS stage BEFORE this step's MFMAs are issued (that stage was last read in step kt - 1, behind the barrier), so the
        // transposing ds_writes and their completion wait drain under the MFMA phase instead of standing between it and the barrier
        if (kt + 1 < nkt) STORE_IJ(rri[(u + 1) % PF], rrj[(u + 1) % PF], (u & 1) ? Is0 : Is1, (u & 1) ? Js0 : Js1);
        if (kt + 1 + PF < nkt) { LOAD_I(rri[(u + 1) % PF], kt + 1 + PF); LOAD_J(rrj[(u + 1) % PF], kt + 1 + PF); }
        MMA_KTILE((u & 1) ? Is1 : Is0, (u & 1) ? Js1 : Js0);
#else
        MMA_KTILE((u & 1) ? Is1 : Is0, (u & 1) ? Js1 : Js0);
        if (kt + 1 < nkt) STORE_IJ(rri[(u + 1) % PF], rrj[(u + 1) % PF], (u & 1) ? Is0 : Is1, (u & 1) ? Js0 : Js1);
        if (kt + 1 + PF < nkt) { LOAD_I(rri[(u + 1) % PF], kt + 1 + PF); LOAD_J(rrj[(u + 1) % PF], kt + 1 + PF); }
#endif
        __syncthreads();
      }
    }
  }
#elif J_MODE == 7 && (PKH * PKW > 1)
  // Fused pooling: the next K tile's patch elements are window maxima.  Window row r's loads are issued before the r-th part of this step's MFMAs and folded into the running
  // maxima after it: PKW x kNJ loads in flight at a time instead of PKH x PKW x kNJ, nothing waits for memory in front of an MFMA.
  constexpr int kKKn = BK / kKS, kP1 = (PKH >= 2) ? (kKKn / PKH) : kKKn, kP2 = (PKH >= 3) ? (2 * kKKn / PKH) : kKKn;
  for (int kt = 0; kt < nkt; ++kt) {
    bool const more = (kt + 1) < nkt;
    float const *const Ic = ((kt & 1) ? Is1 : Is0) + a_off, *const Jc = ((kt & 1) ? Js1 : Js0) + b_off;
    int const k0n = (kt_begin + kt + 1) * BK;
    float raw[PKW * kNJ];
    if (more) { LOAD_I(ri, kt + 1); load_pool_row(raw, rJ, g, p, k0n, 0); }
    mma_ktile<0, kP1>(acc, Ic, Jc, bj);
    if (more) { fold_pool_row(rj, raw, true); if (PKH >= 2) load_pool_row(raw, rJ, g, p, k0n, 1); }
    if (PKH >= 2) mma_ktile<kP1, kP2>(acc, Ic, Jc, bj);
    if (more && PKH >= 2) { fold_pool_row(rj, raw, false); if (PKH >= 3) load_pool_row(raw, rJ, g, p, k0n, 2); }
    if (PKH >= 3) mma_ktile<kP2, kKKn>(acc, Ic, Jc, bj);
    if (more && PKH >= 3) fold_pool_row(rj, raw, false);
    if (more) STORE_IJ(ri, rj, (kt & 1) ? Is0 : Is1, (kt & 1) ? Js0 : Js1);
    __syncthreads();
  }
#else
  for (int kt = 0; kt < nkt; ++kt) {
    bool const more = (kt + 1) < nkt;
    if (more) { LOAD_I(ri, kt + 1); LOAD_J(rj, kt + 1); } // prefetch K-tile kt+1 into registers; the loads fly under the MFMAs below
    MMA_KTILE((kt & 1) ? Is1 : Is0, (kt & 1) ? Js1 : Js0);
    if (more) STORE_IJ(ri, rj, (kt & 1) ? Is0 : Is1, (kt & 1) ? Js0 : Js1);
    __syncthreads();
  }
#endif

#if KHO
  if (seg + 1 < p.splitk) {   // not the tile's last segment: hand the raw accumulators on
#pragma unroll
    for (int a = 0; a < ((KHO_DBG & 1) ? 0 : kTI); ++a)
#pragma unroll
      for (int b = 0; b < kTJ; ++b)
#pragma unroll
        for (int q = 0; q < kHQ; ++q) {
          float const x0 = acc[a][b][4 * q], x1 = acc[a][b][4 * q + 1], x2 = acc[a][b][4 * q + 2], x3 = acc[a][b][4 * q + 3];
          f32x4 v; v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rW, (((a * kTJ + b) * kHQ + q) * kNT + tid) * 16, 0, 16);
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier" ::: "memory");   // every wave's share of the slab has left (write-through: acknowledged by the fabric)
    if (wave == 0 && !(KHO_DBG & 2)) __builtin_amdgcn_raw_buffer_store_b32(seg + 1, rQ, q_l0 + flag_off, 0, 16);   // (sc1: write-through, like the slab)
  } else {
  if (p.splitk > 1 && wave == 0 && !(KHO_DBG & 2)) __builtin_amdgcn_raw_buffer_store_b32(0, rQ, q_l0 + flag_off, 0, 16);   // the tile is complete: its flag is ready for the next launch / graph replay
#endif
  // ---- epilogue: MFMA C/D layout.  32x32: column j = lane&31, row i = (r&3) + 8*(r>>2) + 4*(lane>>5), r < 16
  //                                16x16: column j = lane&15, row i = 4*(lane>>4) + r,               r < 4
#if !SPLITK
  // The biases of this lane's kTI*kNA rows are fetched up front (one batch of loads in flight instead of a load -> wait -> add ->
  // store chain per output; rows past the end read 0), and the stores are buffer stores with 32-bit offsets (per-lane column part
  // + a wave-uniform per-row part in the scalar offset operand).  For layers with a short K loop (1x1 convolutions, K = 96..1024) the old per-output chain cost about as much as the K loop itself.
  {
    rsrc_t const rD = make_rsrc(p.D + (long)blockIdx.y * p.bsD, p.D_bytes);
#if EPI == 1
    rsrc_t const rB = make_rsrc(p.bias, (unsigned)p.Mi * 4u);
    unsigned const S4 = (unsigned)(p.OH * p.OW) * 4u;
#elif HALF
    unsigned const S4 = (unsigned)p.ldD * 2u;   // (row pitch in bytes of the half-typed c)
#else
    unsigned const S4 = (unsigned)p.ldD * 4u;
#endif
    int const ib = i0 + wi * (kTI * MT) + 4 * (lane / MT);
    auto rowc = [](int ta, int r) { return ta * MT + ((MT == 32) ? ((r & 3) + 8 * (r >> 2)) : r); }; // wave-uniform part of the row index
    unsigned const ipart = (unsigned)ib * S4;   // this lane's first row; the (ta, r) part of the row offset is wave-uniform -> scalar offset operand
#if EPI == 1
    float bv[kTI][kNA];
#pragma unroll
    for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
      for (int r = 0; r < kNA; ++r) bv[ta][r] = bload1(rB, (ib + rowc(ta, r)) * 4);
#endif
#if MT == 32 && SWAPST
    // Paired stores (even number of 32-pel blocks per wave): the MFMA layout gives a lane row i (+4 for lanes 32-63) of ONE 32-pel block,
    // so a plain store instruction writes two 128-byte segments in two different rows.  After bias / ReLU (done in that layout: each
    // lane has its own row's bias) v_permlane32_swap exchanges the upper half of block 2t's register with the lower half of block
    // 2t+1's: one register then holds 64 CONSECUTIVE pels of row i, the other of row i+4, and each store writes 256 contiguous bytes.
    // On planes that are not a multiple of 128 bytes (55x55, 27x27, 13x13 ...) that is one whole cache line + two partial ones per
    // instruction instead of four partial ones (partial-line writes are what caps the NCHW R+W stream, tools/mem_pattern_probe.py).
    auto store_all = [&](bool const edge) {
      int const ibl = i0 + wi * (kTI * MT);                                  // first row of this wave (no per-lane part any more)
#pragma unroll
      for (int tp = 0; tp < kTJ / 2; ++tp) {
        int const jg = j0 + wj * (kTJ * MT) + tp * 64 + lane;              // 64 consecutive pels: lanes 0-31 block 2tp, lanes 32-63 block 2tp+1
#if EPI == 1
        int const OHW = p.OH * p.OW;
        int const img = jg / OHW, pel = jg - img * OHW;
        unsigned const jpart = (jg < p.Nj) ? ((((unsigned)img * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel) * 4u + (unsigned)ibl * S4) : 0x80000000u;
#else
        unsigned const jpart = (jg < p.Nj) ? ((unsigned)jg * 4u + (unsigned)ibl * S4) : 0x80000000u;   // (past the last column: dropped by the range check)
#endif
#pragma unroll
        for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
          for (int r = 0; r < kNA; ++r) {
            float va = acc[ta][2 * tp][r], vb = acc[ta][2 * tp + 1][r];
#if EPI == 1
            va = va + bv[ta][r]; vb = vb + bv[ta][r];
#if RELU
            va = (va > 0.f) ? va : 0.f; vb = (vb > 0.f) ? vb : 0.f;
#endif
#endif
            auto const sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, va), __builtin_bit_cast(unsigned, vb), false, false);
            int const rc = rowc(ta, r);
            if (!(edge && (ibl + rc >= p.Mi)))
              __builtin_amdgcn_raw_buffer_store_b32((int)sw[0], rD, (int)jpart, (int)((unsigned)rc * S4), ST_AUX);
            if (!(edge && (ibl + rc + 4 >= p.Mi)))
              __builtin_amdgcn_raw_buffer_store_b32((int)sw[1], rD, (int)jpart, (int)((unsigned)(rc + 4) * S4), ST_AUX);
          }
      }
    };
#else
    // the store loop twice: branch-free for tiles inside the row range (all of them when out_chan / M is a multiple of BI), with a
    // per-row range test for the last tile row
    auto store_all = [&](bool const edge) {
#pragma unroll
      for (int tb = 0; tb < kTJ; ++tb) {
        int const jg = j0 + wj * (kTJ * MT) + tb * MT + (lane % MT);
        if (jg >= p.Nj) continue;
#if EPI == 1
        int const OHW = p.OH * p.OW;
        int const img = jg / OHW, pel = jg - img * OHW;
        unsigned const jpart = (((unsigned)img * (unsigned)p.out_ctot + (unsigned)p.out_coff) * (unsigned)OHW + (unsigned)pel) * 4u;
#elif HALF
        unsigned const jpart = (unsigned)jg * 2u;
#else
        unsigned const jpart = (unsigned)jg * 4u;
#endif
#pragma unroll
        for (int ta = 0; ta < kTI; ++ta)
#pragma unroll
          for (int r = 0; r < kNA; ++r) {
            if (edge && (ib + rowc(ta, r) >= p.Mi)) continue;
            float v = acc[ta][tb][r];
#if EPI == 1
            v = v + bv[ta][r];
#if RELU
            v = (v > 0.f) ? v : 0.f;
#endif
#endif
#if ABLATE & 1
            if (v == 123.456f)
#endif
#if HALF
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v), rD, (int)(jpart + ipart), (int)((unsigned)rowc(ta, r) * S4), 0);   // (RNE, as vstore_half)
#else
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rD, (int)(jpart + ipart), (int)((unsigned)rowc(ta, r) * S4), 0);
#endif
          }
      }
    };
#endif
    if (i0 + BI <= p.Mi) store_all(false); else store_all(true); // workgroup-uniform
  }
#else  // SPLITK: raw partial tiles into this slice's slab (64-bit addressing; bias / ReLU happen in the reduce kernel)
  float *const Dp = p.ws + (long)slice * p.ws_slab;
#pragma unroll
  for (int tb = 0; tb < kTJ; ++tb) {
    int const jg = j0 + wj * (kTJ * MT) + tb * MT + (lane % MT);
    if (jg >= p.Nj) continue;
#if EPI == 1
    int const OHW = p.OH * p.OW;
    int const img = jg / OHW, pel = jg - img * OHW;
    long const joff = (long)img * p.Mi * OHW + pel;
    long const istride = OHW;
#else
    long const joff = jg;
    long const istride = p.ldD;
#endif
#pragma unroll
    for (int ta = 0; ta < kTI; ++ta) {
#pragma unroll
      for (int r = 0; r < kNA; ++r) {
#if MT == 32
        int const ig = i0 + wi * (kTI * 32) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#else
        int const ig = i0 + wi * (kTI * 16) + ta * 16 + 4 * (lane >> 4) + r;
#endif
        if (ig < p.Mi) {
          float v = acc[ta][tb][r];
#if EPI == 1 && !SPLITK
          v = v + p.bias[ig];
#if RELU
          v = (v > 0.f) ? v : 0.f;
#endif
#endif
#if ABLATE & 1
          if (v == 123.456f) // keeps the accumulators live without storing
#endif
          Dp[joff + (long)ig * istride] = v;
        }
      }
    }
  }
#endif
#if KHO
  }   // the tile's last segment
  }   // job loop
  if (wave == 0 && !(KHO_DBG & 4)) {   // the last workgroup to leave clears both counters (every other one has drawn its final, out-of-range job before it counted itself out)
    int const n = __builtin_amdgcn_readfirstlane((lane == 0) ? __hip_atomic_fetch_add(reinterpret_cast<int *>(p.ws) + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
    if (n == (int)gridDim.x - 1) { __builtin_amdgcn_raw_buffer_store_b32(0, rQ, q_l0 + 4, 0, 16); __builtin_amdgcn_raw_buffer_store_b32(0, rQ, q_l0, 0, 16); }
  }
#endif
}
#endif // !REDUCE_ONLY

#ifdef REDUCE_ONLY
// Second pass of SPLITK: D[e] = epilogue( sum_{s ascending} ws[s][e] ).  Memory-bound: float4 per lane, grid-stride.
//   RED_EPI 0: plain   1: + bias[(e / chan_stride) % n_chan], optional ReLU (RED_RELU)
extern "C" __global__ __launch_bounds__(256) void KNAME(float const *__restrict__ ws, long ws_slab, int splitk, float *__restrict__ D, long n,
                                                        float const *__restrict__ bias, int chan_stride, int n_chan) {
  long const stride = (long)gridDim.x * blockDim.x * 4;
  for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < n; e += stride) {
    if (e + 3 < n) {
      f32x4 s = *reinterpret_cast<f32x4 const *>(ws + e);
      for (int k = 1; k < splitk; ++k) { f32x4 const t = *reinterpret_cast<f32x4 const *>(ws + (long)k * ws_slab + e); s += t; }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = s[c];
#if RED_EPI == 1
        v += bias[((e + c) / chan_stride) % n_chan];
#if RED_RELU
        v = (v > 0.f) ? v : 0.f;
#endif
#endif
        s[c] = v;
      }
      *reinterpret_cast<f32x4 *>(D + e) = s;
    } else {
      for (long ee = e; ee < n; ++ee) {
        float v = ws[ee];
        for (int k = 1; k < splitk; ++k) v += ws[(long)k * ws_slab + ee];
#if RED_EPI == 1
        v += bias[(ee / chan_stride) % n_chan];
#if RED_RELU
        v = (v > 0.f) ? v : 0.f;
#endif
#endif
        D[ee] = v;
      }
    }
  }
}
#endif

// winograd_f32.hip -- transform kernels of the F(2x2, 3x3) Winograd path for 3x3 / stride-1 convolutions (gfx950).
//
// Opt-in (tune key conv_algo = "winograd"): the reference credits any fast algorithm with the full 2*M*N*K ("effective" flops,
// src/latex-util.H:116-133) and compares Winograd results (cuDNN's, func cudnn_conv on 3x3 kernels) at mrd < 2e-3
// (src/rtc_prof.cc:317-319,436); the default path stays the bit-exact direct kernel.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        per 2x2 output tile, d = the 4x4 input patch it reads, g = 3x3 filter
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// Pipeline per chunk of images (chunks keep the transformed tensors inside the 256 MB Infinity Cache):
//   bodahip_wino_filt : U[xn][c][oc]   = (G g G^T)[xn]                 once per call          (xn = 4*xi + nu, 16 planes)
//   bodahip_wino_in   : V[xn][c][t]    = (B^T d B)[xn]                 t = (img, ty, tx) tile index within the chunk
//   16 x sgemm        : M[xn][oc][t]   = sum_c U[xn][c][oc] * V[xn][c][t]      -- ONE batched launch of bodahip_sgemm_f32
//                       (U[xn] is a K:M operand, V[xn] a K:N operand, M[xn] an M:N result: exactly the hip_sgemm contract)
//   bodahip_wino_out  : out[img][oc][2ty+y][2tx+x] = act( (A^T M A)[y][x] + bias[oc] )
// 16 multiplies per 4 outputs and input channel instead of 36: 2.25x fewer MFMA flops; the transforms are streaming kernels.

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

struct wino_args_t {
  float const *in; float const *filts; float const *bias; float *out;
  float *U; float *V; float *M;
  int B0, Bc;                 // first image and number of images of this chunk
  int C, H, W, OC, OH, OW;
  int TH, TW, Tc;             // tiles per image (rows, cols), tiles in the chunk = Bc*TH*TW
  int PY, PX, relu;
  int out_ctot, out_coff;     // channels of the output tensor, first channel written
};

// U[(xn*C + c)*OC + oc]; one thread per (c, oc), oc fastest (coalesced stores; the 9 filter taps are strided reads of a small tensor)
extern "C" __global__ __launch_bounds__(256) void bodahip_wino_filt(wino_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.C * p.OC) return;
  int const oc = (int)(idx % p.OC), c = (int)(idx / p.OC);
  float const *g = p.filts + ((long)oc * p.C + c) * 9;
  float r[4][3]; // G g
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float const g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    r[0][j] = g0; r[1][j] = 0.5f * (g0 + g1 + g2); r[2][j] = 0.5f * (g0 - g1 + g2); r[3][j] = g2;
  }
  long const plane = (long)p.C * p.OC;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float const a = r[i][0], b = r[i][1], d = r[i][2];
    float *u = p.U + (long)(4 * i) * plane + idx;
    u[0] = a; u[plane] = 0.5f * (a + b + d); u[2 * plane] = 0.5f * (a - b + d); u[3 * plane] = d;
  }
}

// V[(xn*C + c)*Tc + t]; one thread per (c, t), t fastest
extern "C" __global__ __launch_bounds__(256) void bodahip_wino_in(wino_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.C * p.Tc) return;
  int const t = (int)(idx % p.Tc), c = (int)(idx / p.Tc);
  int const tpi = p.TH * p.TW, il = t / tpi, r = t - il * tpi, ty = r / p.TW, tx = r - ty * p.TW;
  float const *src = p.in + ((long)(p.B0 + il) * p.C + c) * p.H * p.W;
  int const y0 = 2 * ty - p.PY, x0 = 2 * tx - p.PX;
  float q[4][4]; // (d B) rows: q[a][nu]
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int const y = y0 + a;
    bool const yok = (unsigned)y < (unsigned)p.H;
    float d[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { int const x = x0 + b; d[b] = (yok && (unsigned)x < (unsigned)p.W) ? src[(long)y * p.W + x] : 0.f; }
    q[a][0] = d[0] - d[2]; q[a][1] = d[1] + d[2]; q[a][2] = d[2] - d[1]; q[a][3] = d[1] - d[3];
  }
  long const plane = (long)p.C * p.Tc;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    float *v = p.V + (long)nu * plane + idx;
    v[0] = q[0][nu] - q[2][nu]; v[4 * plane] = q[1][nu] + q[2][nu]; v[8 * plane] = q[2][nu] - q[1][nu]; v[12 * plane] = q[1][nu] - q[3][nu];
  }
}

// out tile from M[(xn*OC + oc)*Tc + t]; one thread per (oc, t), t fastest
extern "C" __global__ __launch_bounds__(256) void bodahip_wino_out(wino_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.OC * p.Tc) return;
  int const t = (int)(idx % p.Tc), oc = (int)(idx / p.Tc);
  int const tpi = p.TH * p.TW, il = t / tpi, r = t - il * tpi, ty = r / p.TW, tx = r - ty * p.TW;
  long const plane = (long)p.OC * p.Tc;
  float s[4][2]; // (M A) rows: s[xi][x]
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    float const *m = p.M + (long)(4 * xi) * plane + idx;
    float const m0 = m[0], m1 = m[plane], m2 = m[2 * plane], m3 = m[3 * plane];
    s[xi][0] = m0 + m1 + m2; s[xi][1] = m1 - m2 - m3;
  }
  float const bias = p.bias[oc];
  float *dst = p.out + ((long)(p.B0 + il) * p.out_ctot + p.out_coff + oc) * p.OH * p.OW;
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    int const oy = 2 * ty + y;
    if (oy >= p.OH) continue;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      int const ox = 2 * tx + x;
      if (ox >= p.OW) continue;
      float v = (y == 0) ? (s[0][x] + s[1][x] + s[2][x]) : (s[1][x] - s[2][x] - s[3][x]);
      v += bias;
      if (p.relu) v = (v > 0.f) ? v : 0.f;
      dst[(long)oy * p.OW + ox] = v;
    }
  }
}

// U in the layout of the fused kernel: Ut[(c*OC + oc)*16 + xn] -- the 16 transform positions of one (in_chan, out_chan) are one 64-byte
// record (four coalesced 16-byte stores per thread here, four 16-byte loads per record there).  Same expressions as bodahip_wino_filt.
typedef float wf32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void bodahip_wino_filt_t(wino_args_t const p) {
  long const idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.C * p.OC) return;
  int const oc = (int)(idx % p.OC), c = (int)(idx / p.OC);
  float const *g = p.filts + ((long)oc * p.C + c) * 9;
  float r[4][3]; // G g
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float const g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    r[0][j] = g0; r[1][j] = 0.5f * (g0 + g1 + g2); r[2][j] = 0.5f * (g0 - g1 + g2); r[3][j] = g2;
  }
  wf32x4 *u = reinterpret_cast<wf32x4 *>(p.U + idx * 16);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float const a = r[i][0], b = r[i][1], d = r[i][2];
    u[i] = wf32x4{a, 0.5f * (a + b + d), 0.5f * (a - b + d), d};
  }
}

// ---- fused F(2x2,3x3): input transform -> 16 transform-domain MFMA chains -> output transform in ONE kernel; no V / M tensors -------------
//
// A workgroup of eight waves owns 64 out_chans x 64 tiles for ALL 16 transform positions xn.  Wave (wi, wj, xh) owns the 32 x 32 block
// (wi, wj) for the eight positions xn = 8 xh .. 8 xh + 7 (transform rows xi = 2 xh, 2 xh + 1) as eight 32x32 accumulators: 128 registers,
// TWO waves per SIMD: issue is in order, so whatever a wave issues between two of its MFMAs beyond the 64 cycles the first one runs is pipe time lost;
// a second wave's MFMAs fill those gaps (a first version -- sixteen accumulators, one wave per SIMD -- measured the same bare-loop floor but hid
// less of its staging work: 769 vs 645 us on AlexNet conv4).
// Output transform: A^T M A = rows of (M A) combined over xi.  A wave forms s[xi][x] = (M A)[xi][x] for its two xi lane-locally; the xh = 1
// wave hands its s[2], s[3] to its xh = 0 partner through LDS (the operand buffers are free by then), which finishes
// (s0 + s1) + s2 and (s1 - s2) - s3 -- the association of the three-kernel pipeline -- adds the bias and stores.  No M tensor.
// Per stage of 8 input channels:
//   Ut[c][oc][xn] (pre-transformed once per call by bodahip_wino_filt_t) is copied to LDS with 16-byte loads;
//   every thread loads the 4x4 input patch of one (tile, channel) pair, applies B^T d B in registers and writes the 16 V values to LDS;
//   both LDS images are [k][64 rows][16 xn]: a row's 16 transform positions are one 64-byte record, so a lane fetches the operands of
//   eight MFMAs with two ds_read_b128 per operand; the 16-byte chunk j of row e sits at chunk position j ^ ((e >> 1) & 3), which spreads
//   eight consecutive rows over all banks;
//   loads of stage n+1 are issued under the first MFMAs of stage n, transformed / stored under its last ones: one barrier per stage.
// Each accumulator is one ascending-in_chan chain of v_mfma_f32_32x32x2_f32, the transforms use the expressions of the three-kernel
// pipeline above: results are bit-identical to it (tests/test_gpu_parity.py), at 1/16 of its HBM traffic.
#ifndef WABLATE
#define WABLATE 0   // measurement only (wrong results): 1 no input loads | 2 no U loads | 4 no MFMAs | 8 no operand reads after k-step 0 | 16 no transform / LDS writes
#endif
typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
namespace {
constexpr int kWBK = 8;                    // input channels per stage
constexpr int kWImg = kWBK * 64 * 16;      // floats of one operand image of a stage
constexpr int kWOOB = (int)0x80000000;
constexpr int kWFar = 0x40000000;
__device__ __forceinline__ wrsrc_t wmake_rsrc(float const *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00020000); }
} // namespace

extern "C" __global__ __launch_bounds__(512, 1) void bodahip_wino_fused(wino_args_t const p) {
  __shared__ __attribute__((aligned(16))) float smem[4 * kWImg];   // U0 V0 U1 V1 (128 KB)
  int const tid = threadIdx.x, lane = tid & 63;
  int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int const wi = wave >> 2, wj = (wave >> 1) & 1, xh = wave & 1, h = lane >> 5;
  int const n_ocb = (p.OC + 63) / 64;
  int const bi = blockIdx.x % n_ocb, bj = blockIdx.x / n_ocb;   // out_chan blocks of one tile block are neighbours: they share its input in L2
  int const oc0 = bi * 64, t0 = bj * 64;
  int const tpi = p.TH * p.TW;

  // ---- U staging: 4 x 16-byte loads per thread and stage: chunk tid & 3 of the record of out_chan (tid >> 2) & 63, channels (tid >> 8) + 2 i
  wrsrc_t const rU = wmake_rsrc(p.U, (unsigned)((long)16 * p.C * p.OC * 4));
  int const u_j = tid & 3, u_oc = (tid >> 2) & 63, u_k0 = tid >> 8;
  int const u_off = (oc0 + u_oc < p.OC) ? (((oc0 + u_oc) * 16 + 4 * u_j) * 4) : kWOOB;
  int const u_lds = u_oc * 16 + 4 * (u_j ^ ((u_oc >> 1) & 3));

  // ---- V staging: tile t0 + (tid & 63), channel tid >> 6 of the stage
  // Patch element (a, b) of channel c is read at byte offset cbase(c) + poff[4a + b]; elements outside the image (zero padding), tiles past
  // the end and channels past in_chan get kWFar added once or twice: the sum stays >= 2^30 as an unsigned offset, beyond the (< 2^30 byte)
  // tensor, so the hardware range check returns 0 -- no branch, no select per load.
  wrsrc_t const rIn = wmake_rsrc(p.in, (unsigned)((long)p.Bc * p.C * p.H * p.W * 4));
  int const tl = tid & 63, v_kc = tid >> 6;
  int v_base, poff[16];
  {
    int const t = t0 + tl, img = t / tpi, r = t - img * tpi, ty = r / p.TW, tx = r - ty * p.TW;
    int const y0 = 2 * ty - p.PY, x0 = 2 * tx - p.PX;
    v_base = ((img * p.C * p.H + y0) * p.W + x0) * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        poff[4 * a + b] = (t < p.Tc && (unsigned)(y0 + a) < (unsigned)p.H && (unsigned)(x0 + b) < (unsigned)p.W) ? ((a * p.W + b) * 4) : kWFar;
  }
  int const v_sw = (tl >> 1) & 3;
  int const v_lds = v_kc * 1024 + tl * 16;

  // The staging work of the next stage is cut into slices, one behind each of the stage's 32 MFMAs (sched_barrier keeps them there):
  //   MFMAs  0..15: the 20 global loads (the patch elements first, then the 4 U chunks);
  //   MFMAs 16..31: one slice for d B (in place), four slices of one transform row + one 16-byte LDS write each, the 4 U copies;
  //   every fourth MFMA of k-steps 0..2 is preceded by two ds_read_b128: the operands of four transform positions of the next k-step.
  // Past the last stage the loads fall outside the tensors (zeros) and the writes go to the buffer nobody reads: no branch, the whole stage
  // is one basic block and every wait the compiler inserts counts exactly the loads it has to.
  wf32x4 ureg[4]; float dreg[16];
  int cbase = 0;
  auto load_one = [&](int c0, int l) {   // l = 0..19
    if (l == 0) { int const c = c0 + v_kc; cbase = (c < p.C) ? (v_base + c * (p.H * p.W * 4)) : kWFar; }
    if (l < 16) {
#if WABLATE & 1
      dreg[l] = (float)(cbase + poff[l]);
#else
      dreg[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rIn, cbase + poff[l], 0, 0));
#endif
    } else {
      int const i = l - 16, c = c0 + u_k0 + 2 * i;
      int const uo = (c < p.C) ? (u_off + c * (p.OC * 64)) : kWOOB;
#if WABLATE & 2
      ureg[i] = wf32x4{(float)uo, 1.f, 2.f, (float)i};
#else
      ureg[i] = __builtin_bit_cast(wf32x4, __builtin_amdgcn_raw_buffer_load_b128(rU, uo, 0, 0));
#endif
    }
  };
  auto store_one = [&](int buf, int k) {   // k = 0..15: 0: d B; 1, 4, 7, 10: transform rows xi = 0..3; 2, 5, 8, 11: U copies 0..3
    float *const Us = smem + buf * (2 * kWImg), *const Vs = Us + kWImg;
    if (k == 0) {                               // d B, row by row, in place: dreg[4a + nu] = q[a][nu]
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float const d0 = dreg[4 * a], d1 = dreg[4 * a + 1], d2 = dreg[4 * a + 2], d3 = dreg[4 * a + 3];
        dreg[4 * a] = d0 - d2; dreg[4 * a + 1] = d1 + d2; dreg[4 * a + 2] = d2 - d1; dreg[4 * a + 3] = d1 - d3;
      }
    } else if (k < 12 && k % 3 == 1) {
      int const xi = k / 3;
      wf32x4 v;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        float const q0 = dreg[nu], q1 = dreg[4 + nu], q2 = dreg[8 + nu], q3 = dreg[12 + nu];
        v[nu] = (xi == 0) ? (q0 - q2) : (xi == 1) ? (q1 + q2) : (xi == 2) ? (q2 - q1) : (q1 - q3);
      }
      *reinterpret_cast<wf32x4 *>(Vs + v_lds + 4 * (xi ^ v_sw)) = v;
    } else if (k < 12 && k % 3 == 2) {
      int const i = k / 3;
      *reinterpret_cast<wf32x4 *>(Us + (u_k0 + 2 * i) * 1024 + u_lds) = ureg[i];
    }
  };

  wf32x16 acc[8];
#pragma unroll
  for (int xn = 0; xn < 8; ++xn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xn][r] = 0.f;

  int const nst = (p.C + kWBK - 1) / kWBK;
#pragma unroll
  for (int l = 0; l < 20; ++l) load_one(0, l);
#pragma unroll
  for (int k = 0; k < 16; ++k) store_one(0, k);
  __syncthreads();
  // operand records of this lane: row wi*32 + (lane & 31) of U / wj*32 + (lane & 31) of V, k = 2s + h; chunk 2 xh + j at position (2 xh + j) ^ swizzle(row)
  int a_off[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int const ra = wi * 32 + (lane & 31), rb = wj * 32 + (lane & 31);
    a_off[j] = (h * 64 + ra) * 16 + 4 * ((2 * xh + j) ^ ((ra >> 1) & 3)); b_off[j] = (h * 64 + rb) * 16 + 4 * ((2 * xh + j) ^ ((rb >> 1) & 3));
  }
  for (int st = 0; st < nst; ++st) {
    float const *const Us = smem + (st & 1) * (2 * kWImg), *const Vs = Us + kWImg;
    wf32x4 fa[2][2], fb[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { fa[0][j] = *reinterpret_cast<wf32x4 const *>(Us + a_off[j]); fb[0][j] = *reinterpret_cast<wf32x4 const *>(Vs + b_off[j]); }
#pragma unroll
    for (int s = 0; s < kWBK / 2; ++s) {
#pragma unroll
      for (int xn = 0; xn < 8; ++xn) {
        int const m = s * 8 + xn;
#if WABLATE & 8
        constexpr int fs = 0;
#else
        int const fs = s & 1;
        if (s + 1 < kWBK / 2 && (xn & 3) == 0) {
          int const j = xn >> 2;
          fa[(s + 1) & 1][j] = *reinterpret_cast<wf32x4 const *>(Us + (s + 1) * 2048 + a_off[j]);
          fb[(s + 1) & 1][j] = *reinterpret_cast<wf32x4 const *>(Vs + (s + 1) * 2048 + b_off[j]);
        }
#endif
#if WABLATE & 4
        acc[xn][s] += fa[fs][xn >> 2][xn & 3] * fb[fs][xn >> 2][xn & 3];
#else
        acc[xn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fs][xn >> 2][xn & 3], fb[fs][xn >> 2][xn & 3], acc[xn], 0, 0, 0);
#endif
        if (m < 16) {
#pragma unroll
          for (int l = m * 20 / 16; l < (m + 1) * 20 / 16; ++l) load_one((st + 1) * kWBK, l);
        } else {
#if !(WABLATE & 16)
          store_one((st + 1) & 1, m - 16);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout: column = lane & 31 (tile), row = (r & 3) + 8 (r >> 2) + 4 h (out_chan).  s[i][x] = (M A)[2 xh + i][x];
  // exchange buffer: X[pair][r][i][x][lane] (the loop's last barrier has passed: the operand buffers are free)
  float sv[16][2][2];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float const m0 = acc[4 * i][r], m1 = acc[4 * i + 1][r], m2 = acc[4 * i + 2][r], m3 = acc[4 * i + 3][r];
      sv[r][i][0] = m0 + m1 + m2; sv[r][i][1] = m1 - m2 - m3;
    }
  float *const X = smem + (wave >> 1) * (16 * 4 * 64) + lane;
  if (xh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int x = 0; x < 2; ++x) X[((r * 2 + i) * 2 + x) * 64] = sv[r][i][x];
  }
  __syncthreads();
  if (xh == 1) return;
  int const t = t0 + wj * 32 + (lane & 31);
  if (t >= p.Tc) return;
  int const img = t / tpi, rr = t - img * tpi, ty = rr / p.TW, tx = rr - ty * p.TW;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int const oc = oc0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (oc >= p.OC) continue;
    float const bias = p.bias[oc];
    float *dst = p.out + ((long)(p.B0 + img) * p.out_ctot + p.out_coff + oc) * p.OH * p.OW;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      int const ox = 2 * tx + x;
      if (ox >= p.OW) continue;
      float const s2 = X[((r * 2 + 0) * 2 + x) * 64], s3 = X[((r * 2 + 1) * 2 + x) * 64];
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        int const oy = 2 * ty + y;
        if (oy >= p.OH) continue;
        float v = (y == 0) ? (sv[r][0][x] + sv[r][1][x] + s2) : (sv[r][1][x] - s2 - s3);
        v += bias;
        if (p.relu) v = (v > 0.f) ? v : 0.f;
        dst[(long)oy * p.OW + ox] = v;
      }
    }
  }
}

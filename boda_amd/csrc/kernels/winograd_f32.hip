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

#!/usr/bin/env python3
"""What the matrix pipe SUSTAINS on this chip: a register-only loop of v_mfma_f32_32x32x2_f32 (eight independent accumulators per wave, two waves per SIMD, every CU; no LDS,
no memory traffic) timed for tens of milliseconds with (a) all-zero operands, (b) random operands (new values every MFMA from a per-lane ring of 8 x 8 registers), and the
clock it ran at (s_memtime cycles / s_memrealtime).  Round 6 found the fp32 convolution / sgemm kernels at 94-98 % matrix-pipe duty in CYCLES and still at 0.80-0.88 of the
2.4 GHz peak: the chip lowers its clock under load, and by how much depends on the operand data (tools/cbig_timeline.py: the same launch 2.29-2.32 GHz on zeros,
2.03-2.05 GHz on the reference's random data).  This probe gives the ceiling that follows for a kernel that does nothing but multiply.   usage: python tools/mfma_power_probe.py [ms]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.op import Dims, Op
from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc

SRC = """
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern "C" __global__ __launch_bounds__(512) void mfma_power( float const * in, float * out, int n ) {
  f32x16 acc[8];
  for( int a = 0; a < 8; ++a ) for( int e = 0; e < 16; ++e ) acc[a][e] = 0.f;
  float x[8], y[8];
  for( int i = 0; i < 8; ++i ) { x[i] = in[(threadIdx.x * 16 + i) & 16383]; y[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 16383]; }
  unsigned long long const t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for( int i = 0; i < n; ++i ) {
#pragma unroll
    for( int u = 0; u < 8; ++u ) {
#pragma unroll
      for( int a = 0; a < 8; ++a ) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32( x[(a + u) & 7], y[(a + 3 * u) & 7], acc[a], 0, 0, 0 );
    }
  }
  unsigned long long const t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for( int a = 0; a < 8; ++a ) for( int e = 0; e < 16; ++e ) s += acc[a][e];
  if( s == 123.456f ) out[0] = s;
  if( threadIdx.x == 0 ) { out[16 + 2 * blockIdx.x] = (float)( t1 - t0 ); out[17 + 2 * blockIdx.x] = (float)( c1 - c0 ); }
}

// the same loop with its operands READ FROM LDS every k pair, as the kernels do: MODE 0 one ds_read_b128 + one ds_read_b64 (4 x 2 blocks per wave), 1 six ds_read_b32,
// 2 as 0 with a third of the reads (2 x 2 x ... no: 8 x 1 -- the operands of eight MFMAs from ONE b128 + the same b64 reused), i.e. less LDS traffic per MFMA
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int MEM> __device__ void lds_body( float const * in, float * out, int n, float const * big = 0 ) {
  __shared__ __attribute__((aligned(16))) float sm[16 * 1024];
  for( int i = threadIdx.x; i < 16 * 1024; i += 512 ) sm[i] = in[i & 16383];
  __syncthreads();
  f32x16 acc[8];
  for( int a = 0; a < 8; ++a ) for( int e = 0; e < 16; ++e ) acc[a][e] = 0.f;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float const * const ab = sm + ( lane >> 5 ) * 260 + ( lane & 31 ) * 4 + ( wave & 1 ) * 128, * const bb = sm + 8192 + ( lane >> 5 ) * 260 + ( lane & 31 ) * 2 + ( wave >> 1 ) * 64;
  unsigned long long const t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  f32x4 const * const bigv = reinterpret_cast<f32x4 const *>( big ) + (size_t)blockIdx.x * 65536 + threadIdx.x;   // MEM: one 16-byte load per lane every MEM-th iteration (1 KB per wave
  for( int i = 0; i < n; ++i ) {                                                                                   // per 64 MFMAs at MEM 1 = ~1.2 TB/s chip-wide), streaming through a 2 GB buffer
    if( MODE == 2 && ( i % 144 ) == 143 ) { for( int a = 0; a < 8; ++a ) for( int e = 0; e < 16; ++e ) acc[a][e] *= 1e-30f; }   // MODE 2: the accumulators start over every 2304 k (a layer's K), as a tile's do
    if( MEM && ( i % MEM ) == 0 ) { f32x4 const v = bigv[ (size_t)( ( i / MEM ) & 127 ) * 512 ]; asm volatile( "" :: "v"(v) ); }
#pragma unroll
    for( int u = 0; u < 8; ++u ) {   // k pair u of a 16-deep K step: rows 2 u, 2 u + 1 of the k-major images
      float av[4], bv[2];
      if( MODE == 1 ) { for( int e = 0; e < 4; ++e ) av[e] = ab[u * 520 + e]; for( int e = 0; e < 2; ++e ) bv[e] = bb[u * 520 + e]; asm volatile( "" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(bv[0]), "+v"(bv[1]) ); }
      else { f32x4 const a4 = *reinterpret_cast<f32x4 const *>( ab + u * 520 ); f32x2 const b2 = *reinterpret_cast<f32x2 const *>( bb + u * 520 );
             av[0] = a4[0]; av[1] = a4[1]; av[2] = a4[2]; av[3] = a4[3]; bv[0] = b2[0]; bv[1] = b2[1]; }
#pragma unroll
      for( int a = 0; a < 8; ++a ) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32( av[a >> 1], bv[a & 1], acc[a], 0, 0, 0 );
    }
  }
  unsigned long long const t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for( int a = 0; a < 8; ++a ) for( int e = 0; e < 16; ++e ) s += acc[a][e];
  if( s == 123.456f ) out[0] = s;
  if( threadIdx.x == 0 ) { out[16 + 2 * blockIdx.x] = (float)( t1 - t0 ); out[17 + 2 * blockIdx.x] = (float)( c1 - c0 ); }
}
extern "C" __global__ __launch_bounds__(512) void mfma_lds0( float const * in, float * out, int n ) { lds_body<0, 0>( in, out, n ); }
extern "C" __global__ __launch_bounds__(512) void mfma_lds1( float const * in, float * out, int n ) { lds_body<1, 0>( in, out, n ); }
extern "C" __global__ __launch_bounds__(512) void mfma_rst( float const * in, float * out, int n ) { lds_body<2, 0>( in, out, n ); }
extern "C" __global__ __launch_bounds__(512) void mfma_mem1( float const * in, float * out, int n, float const * big ) { lds_body<0, 1>( in, out, n, big ); }
extern "C" __global__ __launch_bounds__(512) void mfma_mem4( float const * in, float * out, int n, float const * big ) { lds_body<0, 4>( in, out, n, big ); }
"""
want_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rtc = make_rtc("(be=hip)", 0); rtc.init()
FNS = ["mfma_lds0", "mfma_rst"]
rtc.compile([RtcFuncInfo(f, SRC if i == 0 else "", ["in", "out", "n"] + (["big"] if "mem" in f else []), Op({"type": "probe", "func_name": f}, {})) for i, f in enumerate(FNS)])
rtc.create_var_with_dims("big", Dims.make("float", x=256 * 65536 * 4 + 128 * 512 * 4 + 4096))   # (0.27 GB: every workgroup its own 1 MB window, 128 x 8 KB apart: past the L2)
rtc.create_var_with_dims("o", Dims.make("float", x=1024)); rtc.create_var_with_dims("i", Dims.make("float", x=16384)); di = Dims.make("float", x=16384)
rng = np.random.default_rng(7)
N = int(want_ms * 1e-3 * 2.1e9 / (64 * 64 * 2))   # 64 MFMAs per iteration per wave, two waves per SIMD, 64 cycles each
for fn, label, data in [(f, l, d) for f in FNS for l, d in (("zeros", np.zeros(16384, np.float32)), ("U(-5,5)", rng.uniform(-5, 5, 16384).astype(np.float32)), ("N(0,1)", rng.standard_normal(16384).astype(np.float32)))]:
    label = fn + " " + label
    rtc.copy_nda_to_var("i", data, di)
    am = {"in": RtcArg.var("i"), "out": RtcArg.var("o"), "n": RtcArg.scalar(N, "int32_t")}
    if "mem" in fn: am["big"] = RtcArg.var("big")
    ids = [rtc.run(RtcFuncCall(fn, am, tpb=512, blks=256)) for _ in range(int(os.environ.get('NLAUNCH', '4')))]   # back to back: the clock settles in the first
    rtc.finish_and_sync()
    ms = [rtc.get_dur(i, i) for i in ids]
    o = rtc.copy_var_to_nda("o")
    rt, cy = o[16:16 + 512:2], o[17:17 + 512:2]
    ghz = float(np.median(cy / np.maximum(rt, 1))) * 0.1   # cycles per 10 ns
    mf = 64.0 * N * 2                                      # MFMAs per SIMD
    tf = [256 * 4 * mf * 4096 / (m * 1e-3) / 1e12 for m in ms]
    print(f"{label:20s}: {ms[-1]:8.2f} ms per launch  {tf[-1]:6.1f} TF/s (launches: {' '.join('%.1f' % t for t in tf[:12])})  duty {100 * mf * 64 / float(np.median(cy)):5.1f} % of cycles  clock {ghz:.3f} GHz", flush=True)
rtc.release_var("o"); rtc.release_var("i"); rtc.release_var("big"); rtc.close()

#!/usr/bin/env python3
"""Issue rate of v_mfma_f32_32x32x2_f32 from ONE wave per SIMD: a chain of dependent MFMAs (one accumulator) against 2 / 4 / 8 independent chains,
and the same with two waves per SIMD.  Answers what a tile-starved exact layer (one 32x32 output tile per SIMD, every output one ascending-k chain)
can reach at best.  (MI355X)  usage: python tools/mfma_chain_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.op import Dims, Nda, Op
from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo, make_rtc

SRC = """
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC> __device__ void body( float * out, int n ) {
  f32x16 acc[NACC];
  for( int a = 0; a < NACC; ++a ) for( int e = 0; e < 16; ++e ) acc[a][e] = 0.f;
  float x = (float)threadIdx.x * 1e-3f, y = 1.0f + (float)blockIdx.x * 1e-6f;
  for( int i = 0; i < n; ++i ) {
#pragma unroll
    for( int u = 0; u < 8 / NACC; ++u ) {
#pragma unroll
      for( int a = 0; a < NACC; ++a ) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32( x, y, acc[a], 0, 0, 0 );
    }
  }
  float s = 0.f;
  for( int a = 0; a < NACC; ++a ) for( int e = 0; e < 16; ++e ) s += acc[a][e];
  if( s == 123.456f ) out[0] = s;
}
extern "C" __global__ void chain1( float * out, int n ) { body<1>( out, n ); }
extern "C" __global__ void chain2( float * out, int n ) { body<2>( out, n ); }
extern "C" __global__ void chain4( float * out, int n ) { body<4>( out, n ); }
extern "C" __global__ void chain8( float * out, int n ) { body<8>( out, n ); }
"""
rtc = make_rtc("(be=hip)", 0); rtc.init()
names = ["chain1", "chain2", "chain4", "chain8"]
rtc.compile([RtcFuncInfo(n, SRC if i == 0 else "", ["out", "n"], Op({"type": "probe", "func_name": n}, {})) for i, n in enumerate(names)])
rtc.create_var_with_dims("o", Dims.make("float", x=64))
N = 20000   # iterations of 8 MFMAs
for tpb, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD"), (1024, "4 waves/SIMD")):
    for fn in names:
        am = {"out": RtcArg.var("o"), "n": RtcArg.scalar(N, "int32_t")}
        ids = [rtc.run(RtcFuncCall(fn, am, tpb=tpb, blks=256)) for _ in range(3)]
        rtc.finish_and_sync()
        ms = min(rtc.get_dur(i, i) for i in ids)
        mf = 8 * N * (tpb // 64) / 4          # MFMAs per SIMD
        tf = 256 * 4 * mf * 4096 / (ms * 1e-3) / 1e12
        print(f"{label:14s} {fn}: {ms:8.3f} ms  {ms * 1e-3 / mf * 2.4e9:6.1f} cycles@2.4GHz per MFMA per SIMD  -> {tf:6.1f} TF/s chip-wide")
rtc.release_var("o"); rtc.close()

#!/usr/bin/env python3
"""HBM ceiling of the NCHW 1x1-conv access pattern, no arithmetic: every wave reads C channel planes of a block of pels and writes OC
planes of it, VW floats per lane (128*VW contiguous bytes per half wave and plane).  Answers: how much of the copy rate (5.3 TB/s) can
ANY 1x1 kernel reach with this layout, and does the per-lane width matter?   SHAPE=B:C:HW:OC"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.op import Op, Dims, Nda
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo
SRC = r"""
typedef float vf __attribute__((ext_vector_type(VW)));
CUCL_GLOBAL_KERNEL void %(name)( GASQ float const * const in, GASQ float * const out, uint32_t const n_units, uint32_t const stride ) {
  int const lane = LOC_ID_1D & 63, hi = lane >> 5, j = lane & 31;
  int const wave = GLOB_ID_1D >> 6;
#if CHUNK
  uint32_t const per = ( n_units + stride - 1 ) / stride;
  for( uint32_t u = wave*per; u < n_units && u < (wave+1)*per; ++u ) {
#else
  for( uint32_t u = wave; u < n_units; u += stride ) {
#endif
    uint32_t const img = u / NBLK, b = u - img*NBLK;
    int p0 = b*(32*VW) + VW*j; if( p0 > HW - VW ) { p0 = HW - VW; }
    vf acc = 0;
#if MODE & 1
#pragma unroll
    for( int c = 0; c < C; c += 2 ) { acc += *(GASQ vf const *)( in + ( (size_t)img*C + c + hi )*HW + p0 ); }
#endif
#if MODE & 2
#pragma unroll
    for( int oc = 0; oc < OC; oc += 2 ) {
#if ALIGNW   // ceiling probe: every row's store starts on a 128-byte line (what an LDS-shifted epilogue would write); coverage is approximate
      size_t e = ( (size_t)img*OC + oc + hi )*HW + b*(32*VW); e &= ~(size_t)31; e += VW*j; if( e > (size_t)TOT - VW ) { e = (size_t)TOT - VW; }
      *(GASQ vf *)( out + e ) = acc + (float)oc;
#else
      *(GASQ vf *)( out + ( (size_t)img*OC + oc + hi )*HW + p0 ) = acc + (float)oc;
#endif
    }
#else
    if( acc[0] == 123.456f ) { out[u] = acc[0]; }
#endif
  }
}
"""
B, C, HW, OC = [int(x) for x in os.environ.get("SHAPE", "256:96:3025:96").split(":")]
rtc = make_rtc(); rtc.init()
rtc.create_var_with_dims("in", Dims(("n",), (B * C * HW,), "float")); rtc.create_var_with_dims("out", Dims(("n",), (B * OC * HW,), "float"))
u32 = lambda v: RtcArg.scalar(int(v), "uint32_t")
CHUNK = int(os.environ.get("CHUNK", "0")); ALIGNW = int(os.environ.get("ALIGNW", "0"))
for vw in (1, 2, 4):
    nblk = -(-HW // (32 * vw))
    for mode in (3,):
        for wpc in (8, 16, 32):   # waves per CU
            name = f"memp_{vw}_{mode}"
            src = f"#define ALIGNW {ALIGNW}\n#define TOT {B*OC*HW}\n#define CHUNK {CHUNK}\n#define VW {vw}\n#define MODE {mode}\n#define C {C}\n#define OC {OC}\n#define HW {HW}\n#define NBLK {nblk}\n" + SRC.replace("%(name)", name)
            if wpc == 8: rtc.compile([RtcFuncInfo(name, src, ["in", "out", "n_units", "stride"], Op({"type": "memp", "func_name": name}, {}))])
            n_units = B * nblk; waves = min(n_units, 256 * wpc); per = -(-n_units // waves); waves = -(-n_units // per)
            call = RtcFuncCall(name, {"in": RtcArg.var("in"), "out": RtcArg.var("out"), "n_units": u32(n_units), "stride": u32(waves)}, tpb=256, blks=-(-waves // 4))
            for _ in range(100): rtc.run(call)
            rtc.finish_and_sync(); rtc.release_per_call_id_data()
            ids = [rtc.run(call) for _ in range(30)]; rtc.finish_and_sync()
            ms = np.array([rtc.get_dur(c, c) for c in ids]); rtc.release_per_call_id_data()
            by = 4.0 * B * HW * ((C if mode & 1 else 0) + (OC if mode & 2 else 0))
            print(f"VW={vw} mode={'R' if mode==1 else 'W' if mode==2 else 'R+W'} waves/CU={wpc:2d}: {ms.mean()*1e3:7.1f} us  {by/ms.mean()/1e6:6.0f} GB/s", flush=True)

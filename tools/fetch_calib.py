#!/usr/bin/env python3
"""Calibration of rocprofv3 FETCH_SIZE on this box: a streaming kernel that reads a known number of bytes with the same
access shape the GEMM staging uses (16 B per lane, wave-contiguous 1 KiB), run under `rocprofv3 --pmc FETCH_SIZE`.
The guide (MI355X_MICROARCH.md, HBM section) says gfx950 FETCH_SIZE reports 1/2 of the bytes of such a stream; this
measures the factor here instead of assuming it.   usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python fetch_calib.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from boda_amd.op import Dims, Op
from boda_amd.rtc import make_rtc, RtcArg, RtcFuncCall, RtcFuncInfo

SRC = """
typedef float f4_t __attribute__((ext_vector_type(4)));
CUCL_GLOBAL_KERNEL void calib_stream_read( GASQ float const * const a, GASQ float * const out, uint32_t const n4 ) {
  f4_t acc = {0.f,0.f,0.f,0.f};
  for( uint32_t i = GLOB_ID_1D; i < n4; i += LOC_SZ_1D * 2048 ) { f4_t const v = ((GASQ f4_t const *)a)[i]; acc += v; }
  if( acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f ) { out[0] = acc[0]; }
}
"""
rtc = make_rtc(); rtc.init()
n = 1 << 28  # 2^28 floats = 1 GiB (> 256 MiB Infinity Cache)
rtc.create_var_with_dims("a", Dims.make("float", n=n)); rtc.create_var_with_dims("o", Dims.make("float", n=4))
rtc.compile([RtcFuncInfo("calib_stream_read", SRC, ["a", "out", "n4"], Op({"func_name": "calib_stream_read"}, {}))])
rfc = RtcFuncCall("calib_stream_read", {"a": RtcArg.var("a"), "out": RtcArg.var("o"), "n4": RtcArg.scalar(n // 4, "uint32_t")}, tpb=256, blks=2048)
ids = [rtc.run(rfc) for _ in range(3)]
rtc.finish_and_sync()
ms = rtc.get_dur(ids[-1], ids[-1])
print(f"calib_stream_read: {n*4} bytes in {ms:.3f} ms = {n*4/ms/1e6:.1f} GB/s (known bytes per launch = {n*4})")

"""boda_amd -- MI355X-native `rtc_compute_t` backend (be=hip) for Boda's conv_fwd / SGEMM hot path.

Layout (only what the path needs):
  csrc/        C++17 host library + C ABI (libbodahip.so) and the HIP kernel templates it
               specialises at run time with hiprtc (kernels/*.hip)
  rtc.py       ctypes mirror of the reference's rtc_compute_t interface (src/rtc_compute.H:35-97)
  op.py        op_base_t / dims_t / lexp text forms (src/op_base.H, src/lexp.cc, src/nesi.cc:661-785)
  digest.py    nda_digest_t (src/boda_base.cc:210-384) and wisdom files (src/op-tuner.cc:42-126)
  cnn_op.py    variant selection for the native side door (src/cnn_op.cc:46-68,338-378)
  ops_prof.py  per-op profile + parity harness (src/rtc_prof.cc:44-371)

The HIP extension is mandatory: importing `boda_amd.rtc` raises if libbodahip.so is missing, and
nothing in this package falls back to a CPU implementation.
"""
__version__ = "0.1.0"

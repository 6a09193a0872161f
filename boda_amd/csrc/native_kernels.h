// native_kernels.h -- the native side door of the hip backend: hand-written MFMA kernels bound by op func_name.
//
// Precedent in the reference: nvrtc_compute_t::run() routes functions whose op.func_name starts with cublas_ / cudnn_
// to a vendor library (src/nvrtc_util.cc:369-372, src/culibs-wrap.cc:66-247) while every tensor stays in reference
// layout (src/cnn_op.cc:47-48,142,339-340).  This backend does the same for
//     hip_sgemm  (alias cublas_sgemm)   args a:K:M  b:K:N  c:M:N                      test/rtc/cublas_sgemm.cucl:1-4
//     hip_conv   (alias cudnn_conv)     args filts biases in stride(REF) in_pad(REF) out   test/rtc/cudnn_conv.cucl:1-7
//     hip_sgemm_bf16 / hip_conv_bf16    same contracts; bf16 operands (converted while staging), fp32 accumulate (config 5)
//     hip_conv_nhwc                     Convolution on channels-last bf16 tensors (filts out_chan:y:x:in_chan, in / out img:y:x:chan, type bfloat16;
//                                       out may be float): the layout the bf16 matrix cores want -- reached through xpose functions like the
//                                       reference's k1conv / tconv (src/rtc_prof.cc:92-121)
//     hip_conv_nhwc_grp                 up to four hip_conv_nhwc that read the same `in` with the same kernel geometry, as ONE launch (stacked filts / biases, outputs
//                                       out_0 .. out_3): an inception module's same-input 1x1 convs, a ResNet stage's branch1 + branch2a
//     hip_conv_nhwc_multi               up to 256 INDEPENDENT hip_conv_nhwc convolutions (own tensors, any geometries) as ONE launch: the tile lists of a per-layer op list's
//                                       small members handed to the hardware dispatcher together, longest first (kernels/conv_nhwc_multi_bf16.hip)
//     hip_conv_nhwc_set                 up to 16 independent hip_conv_nhwc convolutions, each on ITS OWN specialised kernel code, as one launch: an inception module's
//                                       3x3 / 5x5 / pool-projection convolutions (the kernel sources instantiated per member inside one wrapper kernel, built at run time)
//     hip_conv_winograd                 same contract as hip_conv; 3x3 / stride-1 layers through F(2x2,3x3) Winograd (mrd <= ~2e-3)
// and lands them on kernels/gemm_conv_f32.hip (and, for short-K 1x1 convs with a long pel axis, kernels/k1_stream_f32.hip),
// specialised with hiprtc per shape class at first use.
#pragma once
#include "rtc_types.h"
#include <hip/hip_runtime.h>

namespace bodahip {

void hip_err_chk(hipError_t e, char const *what);
std::vector<char> hiprtc_compile(string const &src, string const &name, string const &arch, vect_string const &opts, string *log_out, bool use_cache);
void hiprtc_compile_stats(uint64_t *hits, uint64_t *misses, double *compile_ms);   // process-wide: code objects from the on-disk cache / compiled, time spent compiling
string cucl_prelude();
string default_cache_dir();

// what the native kernels need from their backend
struct native_host_t {
  virtual ~native_host_t() {}
  virtual hipStream_t nh_stream() = 0;
  // launch a kernel of the current call on the backend's stream (grid gx x gy workgroups of `block` threads).  The backend decides how the call is TIMED: with marker
  // events around the call, or -- timing mode "kernel" -- with start / stop events bound to the call's own dispatches (hipExtModuleLaunchKernel)
  virtual hipError_t nh_launch(hipFunction_t f, uint32_t gx, uint32_t gy, uint32_t block, void **params) = 0;
  virtual bool nh_capturing() = 0; // stream capture (hipGraph) in progress: nothing may be compiled / allocated / synchronised
  virtual int nh_live_graphs() = 0; // captured graphs not yet destroyed (their kernel arguments may point into the kernel scratch)
  virtual string const &nh_arch() = 0;
  virtual int nh_num_cus() = 0;
  virtual int nh_device() = 0;      // HIP device ordinal this backend runs on
  virtual void *nh_var_ptr(string const &vn) = 0;
  virtual dims_t nh_var_dims(string const &vn) = 0;
  virtual rtc_compute_t &nh_rtc() = 0;
};

// blocking of one specialisation (the op_tune_t analogue for the native kernels: MNb/MNt/Kb of src/cnn_op.H:18-20
// become workgroup tile / waves / K step)
struct tile_cfg_t {
  int BI = 128, BJ = 128, BK = 16, WI = 2, WJ = 2, MINW = 2, SPLITK = 1, MT = 32, PF = 1, SW = 0;   // SW: 1 = as many staging waves as multiplying waves (gemm_conv_f32.hip -DSPECW)
  int KHO = 0;   // > 1: sequential K hand-off in KHO segments per tile (gemm_conv_f32.hip -DKHO=1: exact, persistent workgroups pulling (tile, segment) jobs)
  int threads() const { return (SW == 2 || SW == 3) ? (WI * WJ * 64 + 256) : (WI * WJ * 64 * (SW ? 2 : 1)); }   // SW 2: kernels/conv_big_f32.hip -- WI x WJ multiplying waves + four staging waves; 3: kernels/sgemm_big_f32.hip likewise
  string str() const {
    return std::to_string(BI) + "x" + std::to_string(BJ) + "x" + std::to_string(BK) + "_w" + std::to_string(WI) + "x" + std::to_string(WJ) +
           (MT != 32 ? ("_m" + std::to_string(MT)) : string()) + (SPLITK > 1 ? ("_s" + std::to_string(SPLITK)) : string()) + (PF != 1 ? ("_p" + std::to_string(PF)) : string()) + (SW == 2 ? "_big" : (SW == 3 ? "_stg" : (SW ? "_sw" : ""))) + (KHO > 1 ? ("_h" + std::to_string(KHO)) : string()); }
};

struct conv_geom_t { int B, C, H, W, OC, KH, KW, SY, SX, PY, PX, OH, OW; bool relu;
  // round 5, fp32 hip_conv only: a max pooling fused in FRONT of the convolution (kernels/gemm_conv_f32.hip, PKH).  H x W are then the POOLED plane -- the convolution's
  // input --, UH x UW the planes of the tensor actually read (the pooling's input); PKH == PKW == 1: no pooling (UH / UW unused).
  int PKH = 1, PKW = 1, PSY = 1, PSX = 1, UH = 0, UW = 0;
  bool pooled() const { return PKH * PKW > 1; } };

// channels-last bf16, rolling-rows kernel (kernels/conv_nhwc_rows_bf16.hip): what follows the convolution INSIDE the launch -- a max pooling of its output (PKH > 0:
// window, stride, padding; POH x POW = the pooled planes, the tensor the launch writes) and an across-channel LRN of the pooled values (LRN_N > 0: local size, alpha, beta, k)
struct post_ops_t { int PKH = 0, PKW = 0, PSY = 1, PSX = 1, PPY = 0, PPX = 0, POH = 0, POW = 0, LRN_N = 0; float alpha = 0.f, beta = 0.f, k = 0.f; bool pooled() const { return PKH > 0; } };

struct launch_info_t { string kernel; tile_cfg_t cfg; uint32_t grid = 0, block = 0; double flops = 0, algo_bytes = 0; };

struct native_kernels_t {
  explicit native_kernels_t(native_host_t *host_);
  ~native_kernels_t();
  static bool is_native_func_name(string const &fn);
  void check_compile_time(rtc_func_info_t const &fi);
  void run(rtc_func_info_t const &fi, map_str_rtc_arg_t const &arg_map);

  // direct entry points on raw device pointers (used by run() and by the C ABI's fast paths)
  void sgemm(float const *a, float const *b, float *c, uint32_t M, uint32_t N, uint32_t K, bool bf16 = false, bool half = false);   // half: 2-byte IEEE half elements, fp32 math
  void conv(float const *filts, float const *biases, float const *in, float *out, conv_geom_t const &g, bool bf16 = false, int out_ctot = 0, int out_coff = 0,
            char const *algo = nullptr, float const *filts_km = nullptr);   // filts_km: the caller's k-major copy of filts (hip_conv_filts_kmajor), or null

  // channels-last bf16 tensors (kernels/conv_nhwc_bf16.hip): filts out_chan:y:x:in_chan, in / out img:y:x:chan; g.C = stored channels (multiple of 8)
  void conv_nhwc_rows(void const *filts, float const *biases, void const *in, void *out, conv_geom_t const &g, post_ops_t const &post, int out_ctot = 0, int out_coff = 0);   // F' filts; g.OH x g.OW = the convolution's own output planes
  void conv_nhwc(void const *filts, float const *biases, void const *in, void *out, conv_geom_t const &g, bool out_f32, int out_ctot = 0, int out_coff = 0, bool patch_filts = false, bool pool = false);   // pool: g.KH x g.KW / g.PY, g.PX are a max-pooling window fused in front of 1x1 filters (patch form); patch_filts: filts are F'[in_grp][ky][kx][out_chan][8] -> the LDS input-patch kernel
  // horizontally fused channels-last convolutions (same `in`, same kernel geometry; filts / biases stacked along out_chan, members padded to `pad` rows)
  void conv_nhwc_grp(void const *filts, float const *biases, void const *in, conv_geom_t const &g, bool out_f32, int n, int const *noc, void *const *outs,
                     int const *ctot, int const *coff, int pad);
  // several independent channels-last convolutions as ONE launch (kernels/conv_nhwc_multi_bf16.hip); members: raw device pointers + geometry (g.C = stored channels)
  struct multi_member_t { void const *filts; float const *biases; void const *in; void *out; conv_geom_t g; int out_ctot, out_coff; bool pool = false;
                          // a horizontally fused member (hip_conv_nhwc_grp as a member of a set): filts / biases stacked, g.OC = the padded total, `out` unused
                          int grp_n = 0; int grp_noc[4] = {0, 0, 0, 0}; void *grp_out[4] = {nullptr, nullptr, nullptr, nullptr}; int grp_ctot[4] = {0, 0, 0, 0}, grp_coff[4] = {0, 0, 0, 0}; int grp_pad = 0; };   // pool: g's window / padding are a max pooling fused in front of 1x1 filters
  void conv_nhwc_multi(int n, multi_member_t const *members, bool out_f32);
  // a few independent channels-last convolutions, each on its own specialised kernel code (implicit-GEMM or input-patch form), as one launch (wrapper kernel built at run time)
  void conv_nhwc_set(int n, multi_member_t const *members, bool const *patch_filts, bool out_f32);
  // two 1x1 / stride-1 / unpadded fp32 convolutions back to back as one launch, the intermediate tensor in registers (kernels/k1_quad_f32.hip -DCHAIN=1); g = the first
  // convolution's geometry, mid (optional) also receives its output
  void conv_k1_chain(float const *filts, float const *biases, float const *filts2, float const *biases2, float const *in, float *out, float *mid, conv_geom_t const &g, int oc2,
                     bool relu2, int out_ctot, int out_coff);
  void conv_winograd(float const *filts, float const *biases, float const *in, float *out, conv_geom_t const &g, int out_ctot, int out_coff);

  // tuning overrides ("" clears): key "sgemm_tile" / "conv_tile" -> "BIxBJxBKxWIxWJ[xMINW[xSPLITK[xMT]]]"; key "k1_stream" -> "off" | "WIxWJxOCBxCB[xMINW]";
  // key "conv_algo" -> "winograd": 3x3 / stride-1 convs go through the F(2x2,3x3) path (kernels/winograd_f32.hip; not bit-exact)
  // key "exact" -> "0": tolerance mode -- fp32 results within the reference's bound for re-associating kernels (mrd < 2e-3, src/rtc_prof.cc:317-319) instead of bit-identical to
  // its fma chain: deterministic K slices on tile-starved long-K layers, Winograd where it is faster.  Default ("" / "1"): bit-exact plans only
  void set_tune(string const &key, string const &val);
  static size_t prebuild(op_base_t const &op, string const &arch, int num_cus, string const &tile, string *plan_out = nullptr);
  launch_info_t last_launch;
  uint32_t num_specialisations() const;
  struct impl_t;

 private:
  impl_t *impl;
  native_host_t *host;
};

} // namespace bodahip

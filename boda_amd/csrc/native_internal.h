// native_internal.h -- what the translation units behind native_kernels.h share: kernel argument structs, the plan record, the planners' and the launch helpers' prototypes.
// native_plan.cc: tile / variant selection (host logic only); native_kernels.cc: specialisation, launch, the fp32 entry points; native_nhwc.cc: the channels-last bf16 entry
// points; native_run.cc: the rtc function interface (run, prebuild / explain_plan).
#pragma once
#include "native_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <sstream>

namespace bodahip {

struct gemm_args_t { // must match kernels/gemm_conv_f32.hip
  float const *I; float const *J; float *D; float const *bias;
  int Mi, Nj, K;
  int ldI, ldJ, ldD;
  int C, H, W, OH, OW;
  int tiles_i, tiles_j;
  int splitk, kt_per;
  float *ws; long ws_slab;
  unsigned I_bytes, J_bytes;
  unsigned D_bytes;
  int out_ctot, out_coff;
  void const *ktab; int ktab_n;
  long bsI, bsJ, bsD;
};
// kernels/k1_quad_f32.hip -DCHAIN=1: gemm_args_t followed by the second convolution of a 1x1 chain.  (A struct of its own: gemm_args_t is also the element type of the member
// tables of hip_conv_nhwc_set, whose device-side declaration must keep the size.)
struct chain_args_t : gemm_args_t {
  float const *I2; float const *bias2; float *Dmid;
  int M2; unsigned I2_bytes, Dmid_bytes;
};
static_assert(sizeof(gemm_args_t) == 176, "gemm_args_t is declared with this size by every kernel source (and by set_kernel_source's text)");

struct kernel_t { hipModule_t mod = nullptr; hipFunction_t func = nullptr; int occ = 0; };   // occ: resident workgroups per CU (queried on first use by the persistent forms)

struct native_kernels_t::impl_t {
  std::map<string, kernel_t> kernels; // key = option string
  std::map<string, string> tune;
  void *ws = nullptr; size_t ws_bytes = 0; // split-K partial-sum slabs (grow-only scratch, like the reference's cudnn scratch var)
  std::vector<void *> ws_retired;          // outgrown scratch buffers that captured graphs may still point into (freed with the backend)
  std::map<string, void *> ktabs;           // im2col gather tables, one per (C,H,W,KH,KW) (device memory)
  int call_ws_hold = 0;                       // > 0: a launch is collecting several of them (hip_conv_nhwc_set): none may be dropped
  size_t call_ws_bytes = 0;                   // sum of the per-call workspaces ("ksl:" / "kho:" entries of ktabs): bounded, see call_ws_make_room
  size_t ts_off = 0, ts_bytes = 0; string ts_hdr;   // experiment hook BODAHIP_CBIG_TSTAMP=<file>:late -- the clock stamps of the LAST staging-wave launch, written out when the backend goes
  hipModule_t wino_mod = nullptr; hipFunction_t wino_filt = nullptr, wino_in = nullptr, wino_out = nullptr, wino_fused = nullptr, wino_filt_t = nullptr; // kernels/winograd_f32.hip
};


struct plan_t { tile_cfg_t cfg; vect_string defs; string kname; long split_pels = 0; tile_cfg_t tail_cfg; vect_string tail_defs;   /* split_pels > 0 (staging-wave convolution, round 6): two-level tiling along the pels -- this plan's tiles over the first split_pels pels (whole rounds of the CUs), tail_cfg's over the rest */ bool ipconv = false, k1 = false, bf16 = false, patch = false, stream = false, quad = false, fc = false, big = false, cbig = false, rdec = false, patch16 = false, nhwc = false, nhwc_patch = false, nhwc_multi = false, nhwc_rows = false, ksl = false; int rows = 0, cg = 0; };

struct rows_args_t { // must match kernels/conv_nhwc_rows_bf16.hip
  void const *filts; void const *in; void *out; float const *bias;
  int n_img, oc; unsigned filts_bytes, in_bytes, out_bytes; int out_ctot, out_coff; int n_chunks, rows_per_chunk; float lrn_alpha, lrn_beta, lrn_k;
};

struct sgemm_split_t { uint32_t m_main = 0; string tail_tile; double t_single = 0, t_split = 0; };
static char const *const kBigTile = "256x256x16x2x4x1x1x32x2";
struct sgemm_part_t { uint32_t m0 = 0, rows = 0, n0 = 0, cols = 0; string tile; };
struct set_member_in_t { conv_geom_t g; bool patch_filts, pool; int grp_pad; };
struct set_layout_t { std::vector<int> in_set, alone, variant_of; std::vector<plan_t const *> variants; std::vector<string> vkeys; int minw = 8; string skey; };

// embedded kernel sources (kernels_embed.inc, included by native_kernels.cc)
extern char const *const k_src_conv_nhwc_bf16_ptr, *const k_src_conv_nhwc_patch_bf16_ptr, *const k_src_winograd_f32_ptr;
bool parse_tile(string const &s, tile_cfg_t &c);
void launch(native_host_t *host, kernel_t &k, gemm_args_t &a, tile_cfg_t const &c);
plan_t plan_sgemm(uint32_t M, uint32_t N, uint32_t K, int num_cus, string const &tile, bool bf16 = false, int batch = 1, bool allow_big = true);
bool plan_patch_bf16(conv_geom_t const &g, int num_cus, plan_t &p);
plan_t plan_conv_nhwc(conv_geom_t const &g, int num_cus, string const &tile, bool out_f32, int grp_pad = 0, bool allow_split = true);
plan_t plan_conv_nhwc_patch(conv_geom_t const &g, int num_cus, string const &tile_arg, bool out_f32, bool pool = false);
bool plan_conv_nhwc_rows(conv_geom_t const &g, post_ops_t const &post, int num_cus, plan_t &p, string *why = nullptr);
bool rows_auto(conv_geom_t const &g, int num_cus, string const &tile);
bool apply_post_ops(op_base_t const &op, conv_geom_t &g, post_ops_t &post, char const *what);
bool plan_ipconv_dma(conv_geom_t const &g, int num_cus, plan_t &p);
plan_t plan_conv(conv_geom_t const &g, int num_cus, string const &tile, bool bf16 = false, string const &k1s = string(), bool allow_splitk = true, bool exact = true);
std::vector<char> compile_plan(plan_t const &p, string const &arch, string *log);
void ensure_ws(native_kernels_t::impl_t *impl, native_host_t *host, size_t need);
void call_ws_make_room(native_kernels_t::impl_t *impl, native_host_t *host, size_t need);
void setup_ksl(native_kernels_t::impl_t *impl, native_host_t *host, gemm_args_t &ga, tile_cfg_t const &cfg, long nk, void const *key_ptr, char const *what);
string tune_of(native_kernels_t::impl_t *impl, char const *key);
sgemm_split_t plan_sgemm_split(uint32_t M, uint32_t N, uint32_t K, int num_cus);
string sgemm_wide_tile(uint32_t M, uint32_t N, uint32_t K, long cus);
bool parse_parts_env(uint32_t M, uint32_t N, uint32_t K, std::vector<sgemm_part_t> &out);
std::vector<sgemm_part_t> plan_sgemm_parts(uint32_t M, uint32_t N, uint32_t K, int cus);
bool s2d_geom(conv_geom_t const &g, conv_geom_t &g2, int &pry, int &prx);
bool winograd_applies(conv_geom_t const &g, string const &algo);
long wino_chunk_imgs(conv_geom_t const &g);
bool plan_k1_chain(conv_geom_t const &g, int oc2, bool relu2, plan_t &p);
plan_t plan_conv_nhwc_multi(std::vector<conv_geom_t> const &gs, string const &tile, bool out_f32);
string set_kernel_source(std::vector<plan_t const *> const &variants, int threads, int minw);
plan_t plan_set_member(set_member_in_t const &mi, int num_cus, bool out_f32);
double set_tile_cost(set_member_in_t const &mi, plan_t const &p);
set_layout_t layout_set(std::vector<plan_t> const &plans, std::vector<double> const &tile_cost);
kernel_t &get_kernel(native_kernels_t::impl_t *impl, native_host_t *host, plan_t const &p);

} // namespace bodahip

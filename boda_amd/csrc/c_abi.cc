// c_abi.cc -- flat C ABI (include/bodahip.h) over hip_compute_t.  Exceptions -> {0 ok, 1 unsupported, 2 fatal} + message.
#include "../../include/bodahip.h"
#include "native_kernels.h"
#include "rtc_types.h"
#include <cstdio>
#include <sstream>

using namespace bodahip;
namespace bodahip {
void *hip_compute_stream(rtc_compute_t *rtc);
void hip_compute_set_timing(rtc_compute_t *rtc, int mode);
void hip_any_graph_begin(rtc_compute_t *rtc);      // (hip_multi.cc: one device or all the devices of a multi-device backend)
uint32_t hip_any_graph_end(rtc_compute_t *rtc);
uint32_t hip_any_graph_launch(rtc_compute_t *rtc, uint32_t id);
uint32_t hip_any_graph_num_calls(rtc_compute_t *rtc, uint32_t id);
void hip_any_graph_destroy(rtc_compute_t *rtc, uint32_t id);
uint32_t hip_any_graph_end_deps(rtc_compute_t *rtc, uint32_t n, uint32_t const *ptr, uint32_t const *idx);
native_kernels_t *hip_compute_native(rtc_compute_t *rtc);
void hip_compute_compile_code_object(rtc_compute_t *rtc, void const *code, size_t code_sz, vect_rtc_func_info_t const &fis);
p_rtc_compute_t make_hip_multi_compute(std::vector<int> const &device_ordinals);
rtc_compute_t *hip_multi_sub(rtc_compute_t *rtc, uint32_t i);
uint32_t hip_multi_num_devices(rtc_compute_t *rtc);
extern char const *const k_src_gemm_conv_f32_ptr;
}

struct bodahip_ctx { p_rtc_compute_t rtc; };
static thread_local std::string g_last_error;

#define ABI_TRY try {
#define ABI_CATCH                                                                                                           \
  return BODAHIP_OK; }                                                                                                      \
  catch (unsup_exception const &e) { g_last_error = e.what(); return BODAHIP_UNSUPPORTED; }                                  \
  catch (std::exception const &e) { g_last_error = e.what(); return BODAHIP_ERROR; }                                         \
  catch (...) { g_last_error = "unknown exception"; return BODAHIP_ERROR; }

static rtc_compute_t &R(bodahip_ctx *ctx) { if (!ctx || !ctx->rtc) rt_err("null bodahip_ctx"); return *ctx->rtc; }
static string S(char const *s, char const *what) { if (!s) rt_err(string("null string for ") + what); return s; }
static dims_t to_dims(bodahip_dims const *d) {
  if (!d) rt_err("null dims");
  dims_t r; r.tn = S(d->tn, "dims.tn"); tn_size(r.tn);
  for (uint32_t i = 0; i < d->ndims; ++i) r.add_dims(d->names ? S(d->names[i], "dim name") : string(), d->sizes[i]);
  r.calc_strides();
  return r;
}
static void put_str(char *buf, size_t n, string const &s, char const *what) {
  if (!buf || n < s.size() + 1) rt_err(string("buffer too small for ") + what);
  memcpy(buf, s.c_str(), s.size() + 1);
}

extern "C" {
int bodahip_abi_version(void) { return BODAHIP_ABI_VERSION; }
const char *bodahip_last_error(void) { return g_last_error.c_str(); }

int bodahip_create(bodahip_ctx **out, int device_ordinal) {
  ABI_TRY if (!out) rt_err("null out"); *out = new bodahip_ctx{make_hip_compute(device_ordinal)}; ABI_CATCH }
int bodahip_create_multi(bodahip_ctx **out, uint32_t n_devices, const int *device_ordinals) {
  ABI_TRY if (!out || !device_ordinals || !n_devices) rt_err("null / empty argument");
  *out = new bodahip_ctx{make_hip_multi_compute(std::vector<int>(device_ordinals, device_ordinals + n_devices))}; ABI_CATCH }
int bodahip_num_devices(bodahip_ctx *ctx, uint32_t *n) { ABI_TRY if (!n) rt_err("null out"); *n = hip_multi_num_devices(&R(ctx)); ABI_CATCH }
int bodahip_create_be(bodahip_ctx **out, const char *be, int device_ordinal) {
  ABI_TRY if (!out) rt_err("null out"); string const b = S(be, "be");
  if (b == "hip") *out = new bodahip_ctx{make_hip_compute(device_ordinal)};
  else if (b == "cpu") *out = new bodahip_ctx{make_cpu_compute()};
  else rt_err("unknown rtc back-end '" + b + "' (this library provides be=hip and be=cpu)");
  ABI_CATCH }
void bodahip_destroy(bodahip_ctx *ctx) { try { delete ctx; } catch (...) {} }
int bodahip_set_gen_src(bodahip_ctx *ctx, uint32_t gen_src, const char *dir) {
  ABI_TRY R(ctx).gen_src = gen_src; if (dir) R(ctx).gen_src_output_dir = dir; ABI_CATCH }
int bodahip_init(bodahip_ctx *ctx) { ABI_TRY R(ctx).init(); ABI_CATCH }
int bodahip_get_plat_tag(bodahip_ctx *ctx, char *buf, size_t n) { ABI_TRY put_str(buf, n, R(ctx).get_plat_tag(), "plat_tag"); ABI_CATCH }
int bodahip_create_var(bodahip_ctx *ctx, const char *vn, const bodahip_dims *dims) { ABI_TRY R(ctx).create_var_with_dims(S(vn, "vn"), to_dims(dims)); ABI_CATCH }
int bodahip_create_view(bodahip_ctx *ctx, const char *vn, const bodahip_dims *dims, const char *src_vn) {
  ABI_TRY R(ctx).create_var_with_dims_as_reshaped_view_of_var(S(vn, "vn"), to_dims(dims), S(src_vn, "src_vn")); ABI_CATCH }
int bodahip_release_var(bodahip_ctx *ctx, const char *vn) { ABI_TRY R(ctx).release_var(S(vn, "vn")); ABI_CATCH }
int bodahip_get_var_dims(bodahip_ctx *ctx, const char *vn, char *tn_buf, size_t tn_n, uint32_t *ndims_inout, uint32_t *sizes, char *names_buf, size_t names_n) {
  ABI_TRY
  dims_t d = R(ctx).get_var_dims(S(vn, "vn"));
  put_str(tn_buf, tn_n, d.tn, "tn");
  if (!ndims_inout || *ndims_inout < d.sz()) rt_err("get_var_dims: sizes capacity too small");
  string names;
  for (uint32_t i = 0; i < d.sz(); ++i) { if (sizes) sizes[i] = d.dims(i); names += d.names(i); names.push_back('\0'); }
  if (!names_buf || names_n < names.size() + 1) rt_err("get_var_dims: names buffer too small");
  memcpy(names_buf, names.data(), names.size()); names_buf[names.size()] = 0;
  *ndims_inout = d.sz();
  ABI_CATCH }
int bodahip_set_var_to_zero(bodahip_ctx *ctx, const char *vn) { ABI_TRY R(ctx).set_var_to_zero(S(vn, "vn")); ABI_CATCH }
int bodahip_compile(bodahip_ctx *ctx, uint32_t n, const bodahip_func_info *funcs, const bodahip_compile_opts *opts) {
  ABI_TRY
  if (n && !funcs) rt_err("compile: null func_infos with n > 0");
  vect_rtc_func_info_t fis;
  for (uint32_t i = 0; i < n; ++i) {
    rtc_func_info_t fi; fi.func_name = S(funcs[i].func_name, "func_name"); fi.func_src = funcs[i].func_src ? funcs[i].func_src : "";
    for (uint32_t a = 0; a < funcs[i].n_args; ++a) fi.arg_names.push_back(S(funcs[i].arg_names[a], "arg name"));
    if (funcs[i].op && funcs[i].op[0]) fi.op = parse_op_lexp(funcs[i].op);
    fis.push_back(std::move(fi));
  }
  rtc_compile_opts_t o; if (opts) { o.show_compile_log = opts->show_compile_log; o.enable_lineinfo = opts->enable_lineinfo; o.show_func_attrs = opts->show_func_attrs; o.show_rtc_calls = opts->show_rtc_calls; }
  R(ctx).compile(fis, o);
  ABI_CATCH }
int bodahip_compile_code_object(bodahip_ctx *ctx, const void *code, size_t code_sz, uint32_t n, const bodahip_func_info *funcs) {
  ABI_TRY
  if (n && !funcs) rt_err("compile_code_object: null func_infos with n > 0");
  vect_rtc_func_info_t fis;
  for (uint32_t i = 0; i < n; ++i) {
    rtc_func_info_t fi; fi.func_name = S(funcs[i].func_name, "func_name");
    for (uint32_t a = 0; a < funcs[i].n_args; ++a) fi.arg_names.push_back(S(funcs[i].arg_names[a], "arg name"));
    if (funcs[i].op && funcs[i].op[0]) fi.op = parse_op_lexp(funcs[i].op);
    fis.push_back(std::move(fi));
  }
  if (hip_multi_num_devices(&R(ctx)) != 1) unsup_err("compile_code_object: single-device be=hip contexts only");
  hip_compute_compile_code_object(&R(ctx), code, code_sz, fis);
  ABI_CATCH }
int bodahip_compile_to_file(const char *src, const char *arch, int add_prelude, const char *out_path) {
  ABI_TRY
  string log;
  std::vector<char> const code = hiprtc_compile((add_prelude ? cucl_prelude() : string()) + S(src, "src"), "to_file", S(arch, "arch"), {"-ffast-math"}, &log, false);
  FILE *f = fopen(S(out_path, "out_path").c_str(), "wb");
  if (!f) rt_err(string("cannot write ") + out_path);
  size_t const w = fwrite(code.data(), 1, code.size(), f); fclose(f);
  if (w != code.size()) rt_err(string("short write to ") + out_path);
  ABI_CATCH }
int bodahip_release_func(bodahip_ctx *ctx, const char *fn) { ABI_TRY R(ctx).release_func(S(fn, "func_name")); ABI_CATCH }
int bodahip_release_all_funcs(bodahip_ctx *ctx) { ABI_TRY R(ctx).release_all_funcs(); ABI_CATCH }
int bodahip_run(bodahip_ctx *ctx, const char *fn, uint32_t n_args, const bodahip_arg *args, uint32_t tpb, uint32_t blks, uint32_t *call_id_out) {
  ABI_TRY
  rtc_func_call_t rfc; rfc.rtc_func_name = S(fn, "rtc_func_name"); rfc.tpb = tpb; rfc.blks = blks;
  for (uint32_t i = 0; i < n_args; ++i) {
    bodahip_arg const &a = args[i];
    string const an = S(a.name, "arg name");
    if (a.kind == 0) { rfc.arg_map[an] = rtc_arg_t(S(a.var, "arg var")); }
    else if (a.kind == 1) { rfc.arg_map[an] = rtc_arg_t(std::make_shared<nda_t>(to_dims(&a.dims), const_cast<void *>(a.data))); }
    else rt_err("run: bad arg kind for '" + an + "'");
  }
  uint32_t const id = R(ctx).run(rfc);
  if (call_id_out) *call_id_out = id;
  ABI_CATCH }
int bodahip_finish_and_sync(bodahip_ctx *ctx) { ABI_TRY R(ctx).finish_and_sync(); ABI_CATCH }
int bodahip_release_per_call_id_data(bodahip_ctx *ctx) { ABI_TRY R(ctx).release_per_call_id_data(); ABI_CATCH }
int bodahip_get_dur(bodahip_ctx *ctx, uint32_t b, uint32_t e, float *ms) { ABI_TRY if (!ms) rt_err("null ms_out"); *ms = R(ctx).get_dur(b, e); ABI_CATCH }
int bodahip_profile_start(bodahip_ctx *ctx) { ABI_TRY R(ctx).profile_start(); ABI_CATCH }
int bodahip_profile_stop(bodahip_ctx *ctx) { ABI_TRY R(ctx).profile_stop(); ABI_CATCH }
int bodahip_copy_to_var(bodahip_ctx *ctx, const char *vn, const bodahip_dims *dims, const void *host) {
  ABI_TRY if (!host) rt_err("null host pointer"); R(ctx).copy_nda_to_var(S(vn, "vn"), std::make_shared<nda_t>(to_dims(dims), const_cast<void *>(host))); ABI_CATCH }
int bodahip_copy_from_var(bodahip_ctx *ctx, void *host, const bodahip_dims *dims, const char *vn) {
  ABI_TRY if (!host) rt_err("null host pointer"); R(ctx).copy_var_to_nda(std::make_shared<nda_t>(to_dims(dims), host), S(vn, "vn")); ABI_CATCH }
int bodahip_get_raw_ptr(bodahip_ctx *ctx, const char *vn, void **p) { ABI_TRY if (!p) rt_err("null out"); *p = R(ctx).get_var_raw_native_pointer(S(vn, "vn"))->rp_elems(); ABI_CATCH }

int bodahip_graph_begin(bodahip_ctx *ctx) { ABI_TRY hip_any_graph_begin(&R(ctx)); ABI_CATCH }
int bodahip_graph_end(bodahip_ctx *ctx, uint32_t *graph_id, uint32_t *n_calls) {
  ABI_TRY if (!graph_id) rt_err("null graph_id_out"); *graph_id = hip_any_graph_end(&R(ctx)); if (n_calls) *n_calls = hip_any_graph_num_calls(&R(ctx), *graph_id); ABI_CATCH }
int bodahip_graph_launch(bodahip_ctx *ctx, uint32_t graph_id, uint32_t *call_id) {
  ABI_TRY uint32_t const id = hip_any_graph_launch(&R(ctx), graph_id); if (call_id) *call_id = id; ABI_CATCH }
int bodahip_graph_end_deps(bodahip_ctx *ctx, uint32_t n_calls, const uint32_t *dep_ptr, const uint32_t *dep_idx, uint32_t *graph_id) {
  ABI_TRY if (!graph_id || !dep_ptr) rt_err("null argument"); static uint32_t const none = 0;
  *graph_id = hip_any_graph_end_deps(&R(ctx), n_calls, dep_ptr, dep_idx ? dep_idx : &none); ABI_CATCH }
int bodahip_graph_destroy(bodahip_ctx *ctx, uint32_t graph_id) { ABI_TRY hip_any_graph_destroy(&R(ctx), graph_id); ABI_CATCH }
int bodahip_get_stream(bodahip_ctx *ctx, void **s) { ABI_TRY if (!s) rt_err("null out"); *s = hip_compute_stream(hip_multi_sub(&R(ctx), 0)); ABI_CATCH }
int bodahip_get_device_info(bodahip_ctx *ctx, char *arch_buf, size_t n, int *num_cus, int *clock_khz) {
  ABI_TRY
  native_kernels_t *nk = hip_compute_native(hip_multi_sub(&R(ctx), 0)); (void)nk;
  native_host_t *h = dynamic_cast<native_host_t *>(hip_multi_sub(&R(ctx), 0)); if (!h) rt_err("not a hip backend");
  put_str(arch_buf, n, h->nh_arch(), "arch"); if (num_cus) *num_cus = h->nh_num_cus();
  if (clock_khz) { int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, h->nh_device()); *clock_khz = khz; } // (the ctx's device, not the thread's current one)
  ABI_CATCH }
int bodahip_set_tune(bodahip_ctx *ctx, const char *key, const char *value) {
  ABI_TRY
  if (S(key, "key") == "timing") {   // how get_dur attributes stream time to calls (hip_compute.cc: timing_stream)
    string const v = value ? value : "";
    if (!v.empty() && v != "call" && v != "stream" && v != "kernel") rt_err("set_tune: timing must be call | stream | kernel");
    for (uint32_t i = 0; i < hip_multi_num_devices(&R(ctx)); ++i) hip_compute_set_timing(hip_multi_sub(&R(ctx), i), v == "stream" ? 1 : (v == "kernel" ? 2 : 0));
    return 0;
  }
  for (uint32_t i = 0; i < hip_multi_num_devices(&R(ctx)); ++i) hip_compute_native(hip_multi_sub(&R(ctx), i))->set_tune(S(key, "key"), value ? value : ""); ABI_CATCH }
int bodahip_last_launch(bodahip_ctx *ctx, char *kbuf, size_t kn, char *cbuf, size_t cn, uint32_t *grid, uint32_t *block, double *flops, double *algo_bytes) {
  ABI_TRY
  launch_info_t const &li = hip_compute_native(hip_multi_sub(&R(ctx), 0))->last_launch;   // (a multi-device backend: device 0's)
  put_str(kbuf, kn, li.kernel, "kernel"); put_str(cbuf, cn, li.cfg.str(), "cfg");
  if (grid) *grid = li.grid; if (block) *block = li.block; if (flops) *flops = li.flops; if (algo_bytes) *algo_bytes = li.algo_bytes;
  ABI_CATCH }
int bodahip_compile_offline(const char *src_or_opts, const char *native_template, const char *arch, int add_prelude, int use_cache, size_t *code_size_out, char *log_buf, size_t log_n) {
  ABI_TRY
  string log; std::vector<char> code;
  if (native_template && native_template[0]) {
    if (string(native_template) != "gemm_conv_f32") rt_err(string("unknown native kernel template '") + native_template + "'");
    vect_string opts; std::istringstream is(S(src_or_opts, "opts")); string tok; while (is >> tok) opts.push_back(tok);
    code = hiprtc_compile(k_src_gemm_conv_f32_ptr, "offline_native", S(arch, "arch"), opts, &log, use_cache != 0);
  } else {
    string src = (add_prelude ? cucl_prelude() : string()) + S(src_or_opts, "src");
    code = hiprtc_compile(src, "offline_cucl", S(arch, "arch"), {"-ffast-math"}, &log, use_cache != 0);
  }
  if (code_size_out) *code_size_out = code.size();
  if (log_buf && log_n) { size_t const c = std::min(log_n - 1, log.size()); memcpy(log_buf, log.data(), c); log_buf[c] = 0; }
  ABI_CATCH }
int bodahip_parse_op(const char *op_lexp, char *canon_buf, size_t canon_buf_sz) {
  ABI_TRY put_str(canon_buf, canon_buf_sz, op_to_str(parse_op_lexp(S(op_lexp, "op"))), "canonical op"); ABI_CATCH }
int bodahip_prebuild(const char *op_lexp, const char *arch, int num_cus, const char *tile, size_t *code_size_out) {
  ABI_TRY
  size_t const n = native_kernels_t::prebuild(parse_op_lexp(S(op_lexp, "op")), S(arch, "arch"), num_cus > 0 ? num_cus : 256, tile ? tile : "");
  if (code_size_out) *code_size_out = n;
  ABI_CATCH }
int bodahip_explain_plan(const char *op_lexp, int num_cus, const char *tile, char *plan_buf, size_t plan_buf_sz) {
  ABI_TRY
  string plan;
  native_kernels_t::prebuild(parse_op_lexp(S(op_lexp, "op")), "", num_cus > 0 ? num_cus : 256, tile ? tile : "", &plan);
  put_str(plan_buf, plan_buf_sz, plan, "plan");
  ABI_CATCH }
int bodahip_compile_stats(uint64_t *cache_hits_out, uint64_t *compiled_out, double *compile_ms_out) {
  ABI_TRY
  hiprtc_compile_stats(cache_hits_out, compiled_out, compile_ms_out);
  ABI_CATCH }
} // extern "C"

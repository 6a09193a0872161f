/* bodahip.h -- C ABI of libbodahip.so: the MI355X-native rtc_compute backend (be=hip) for Boda.
 *
 * Every entry point below is the flat-C form of one member of the reference's backend interface
 * `struct rtc_compute_t` (src/rtc_compute.H:35-97); the comment on each names the member it replaces.  A Boda checkout
 * binds them from a ~100-line `hip_compute_t : rtc_compute_t` adapter; tests bind them with ctypes.
 * Plain pointers and sizes only -- no C++ or torch types cross this boundary.
 *
 * Errors: the reference throws rt_err (fatal) or unsup_err ("this configuration is unsupported", which callers such
 * as ops-prof catch and record, src/rtc_prof.cc:287-296).  Here every call returns
 *     BODAHIP_OK (0) | BODAHIP_UNSUPPORTED (1) | BODAHIP_ERROR (2)
 * and bodahip_last_error() returns the message of the last failing call on this thread.
 *
 * Threading: one host thread per context, one in-order stream; bodahip_finish_and_sync() is the only barrier
 * (H2D copies are asynchronous on that stream, D2H copies are synchronous) -- as in the reference's backends.
 */
#ifndef BODAHIP_H_
#define BODAHIP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BODAHIP_OK 0
#define BODAHIP_UNSUPPORTED 1
#define BODAHIP_ERROR 2
#define BODAHIP_ABI_VERSION 1

typedef struct bodahip_ctx bodahip_ctx;

/* dims_t (src/boda_base.H:498-690): row-major named dims + element type name ("float","uint32_t","none",...) */
typedef struct bodahip_dims {
  const char *tn;
  uint32_t ndims;
  const uint32_t *sizes;
  const char *const *names;
} bodahip_dims;

/* rtc_arg_t (src/rtc_compute.H:103-115).  kind 0: `var` names a device var (its pointer is passed to the kernel).
 * kind 1: a value: `dims` + `data`; data != NULL -> dims.bytes_sz() raw bytes are passed BY VALUE; data == NULL -> a
 * null pointer is passed (REF / optional args: the information is the dims, e.g. stride = dims (y=4,x=4)). */
typedef struct bodahip_arg {
  const char *name;
  int32_t kind;
  const char *var;
  bodahip_dims dims;
  const void *data;
} bodahip_arg;

/* rtc_func_info_t (src/rtc_compute.H:23-28).  `op` is the (annotated) op_base_t as one lexp line
 * "(str_vals=(func_name=...,type=...),nda_vals=(...))"; op.func_name selects the native kernels
 * (hip_sgemm / hip_conv, aliases cublas_sgemm / cudnn_conv). */
typedef struct bodahip_func_info {
  const char *func_name;
  const char *func_src;
  uint32_t n_args;
  const char *const *arg_names;
  const char *op;
} bodahip_func_info;

/* rtc_compile_opts_t (src/rtc_compute.H:9-21) */
typedef struct bodahip_compile_opts {
  uint32_t show_compile_log, enable_lineinfo, show_func_attrs, show_rtc_calls;
} bodahip_compile_opts;

int bodahip_abi_version(void);
const char *bodahip_last_error(void);

/* construction: NESI would create the backend from "(be=hip)"; device_ordinal selects the GPU (one process per GPU) */
int bodahip_create(bodahip_ctx **out, int device_ordinal);
/* the same by NESI type id: be = "hip" (as above) | "cpu" -- the host-cores backend behind the same contract (blocked, vectorised, OpenMP sgemm /
 * conv + bias + ReLU on reference-layout tensors; native function names only; the CPU baseline timed beside the GPU, SURVEY.md section 8d: the
 * reference itself has no CPU path, src/rtc_fwd.cc:43-44, its one precedent is a cblas_sgemm loop, src/qblas-test.cc:33-41) */
int bodahip_create_be(bodahip_ctx **out, const char *be, int device_ordinal);
/* N GPUs behind ONE context ("one logical var <-> N shards", one host thread, one stream per device; the surface that hides it
 * is src/rtc_compute.H:48-80): vars with a leading dim `img` (sgemm: dim `M`) are split into n_devices contiguous chunks, everything else is
 * replicated (H2D once + peer-to-peer fan-out); copy_to_var scatters, copy_from_var gathers, run() of a native function enqueues on every device,
 * get_dur() is the slowest device's.  Generated CUCL functions run only on replicated vars (BODAHIP_UNSUPPORTED on sharded ones).  Ordinals
 * may repeat ({0,0}: two shards on one GPU).  No collective on the data path. */
int bodahip_create_multi(bodahip_ctx **out, uint32_t n_devices, const int *device_ordinals);
int bodahip_num_devices(bodahip_ctx *ctx, uint32_t *n_devices_out);
void bodahip_destroy(bodahip_ctx *ctx);
int bodahip_set_gen_src(bodahip_ctx *ctx, uint32_t gen_src, const char *gen_src_output_dir); /* fields gen_src, gen_src_output_dir (:39-40) */

int bodahip_init(bodahip_ctx *ctx);                                            /* rtc_compute_t::init() (:45) */
int bodahip_get_plat_tag(bodahip_ctx *ctx, char *buf, size_t buf_sz);           /* ::get_plat_tag() (:46) -> "hip:<device name>" */
int bodahip_create_var(bodahip_ctx *ctx, const char *vn, const bodahip_dims *dims); /* ::create_var_with_dims() (:48); zero-filled */
int bodahip_create_view(bodahip_ctx *ctx, const char *vn, const bodahip_dims *dims, const char *src_vn); /* ::create_var_with_dims_as_reshaped_view_of_var() (:49) */
int bodahip_release_var(bodahip_ctx *ctx, const char *vn);                     /* ::release_var() (:50) */
/* ::get_var_dims() (:51).  sizes[]/names_buf are caller storage: *ndims_inout is capacity in, count out; names are
 * written NUL-separated into names_buf; tn into tn_buf. */
int bodahip_get_var_dims(bodahip_ctx *ctx, const char *vn, char *tn_buf, size_t tn_buf_sz, uint32_t *ndims_inout, uint32_t *sizes,
                         char *names_buf, size_t names_buf_sz);
int bodahip_set_var_to_zero(bodahip_ctx *ctx, const char *vn);                 /* ::set_var_to_zero() (:52) */
int bodahip_compile(bodahip_ctx *ctx, uint32_t n_funcs, const bodahip_func_info *funcs, const bodahip_compile_opts *opts); /* ::compile() (:55) */
int bodahip_release_func(bodahip_ctx *ctx, const char *func_name);             /* ::release_func() (:56) */
int bodahip_release_all_funcs(bodahip_ctx *ctx);                               /* ::release_all_funcs() (:62) */
int bodahip_run(bodahip_ctx *ctx, const char *rtc_func_name, uint32_t n_args, const bodahip_arg *args, uint32_t tpb, uint32_t blks,
                uint32_t *call_id_out);                                        /* ::run(rtc_func_call_t) (:59) */
int bodahip_finish_and_sync(bodahip_ctx *ctx);                                 /* ::finish_and_sync() (:60) */
int bodahip_release_per_call_id_data(bodahip_ctx *ctx);                        /* ::release_per_call_id_data() (:61) */
int bodahip_get_dur(bodahip_ctx *ctx, uint32_t b, uint32_t e, float *ms_out);  /* ::get_dur() (:70), milliseconds */
int bodahip_profile_start(bodahip_ctx *ctx);                                   /* ::profile_start() (:72) */
int bodahip_profile_stop(bodahip_ctx *ctx);                                    /* ::profile_stop() (:73) */
int bodahip_copy_to_var(bodahip_ctx *ctx, const char *vn, const bodahip_dims *dims, const void *host_data);   /* ::copy_nda_to_var() (:80); dims must equal the var's */
int bodahip_copy_from_var(bodahip_ctx *ctx, void *host_data, const bodahip_dims *dims, const char *vn);       /* ::copy_var_to_nda() (:78) */
int bodahip_get_raw_ptr(bodahip_ctx *ctx, const char *vn, void **dev_ptr_out); /* ::get_var_raw_native_pointer() (:79) */

/* ---- additions with no counterpart in the reference interface (plumbing / tooling) ---- */
/* hipGraph capture of a call list (no counterpart in rtc_compute_t: the reference enqueues every call of a forward pass one by
 * one, src/rtc_fwd.cc:545-549).  Between graph_begin and graph_end, bodahip_run() records the launch -- arguments frozen as
 * passed, call_id_out = 0xfffffffe (no per-call events) -- instead of executing it; everything the calls need lazily (hiprtc
 * specialisations, gather tables) must already exist, i.e. the list has been run once.  graph_launch replays the list with one
 * host call; its call id times the whole replay through bodahip_get_dur. */
int bodahip_graph_begin(bodahip_ctx *ctx);
int bodahip_graph_end(bodahip_ctx *ctx, uint32_t *graph_id_out, uint32_t *n_calls_out);
int bodahip_graph_launch(bodahip_ctx *ctx, uint32_t graph_id, uint32_t *call_id_out);
/* graph_end with the true dependencies of the n_calls captured calls (CSR: call i runs after calls dep_idx[dep_ptr[i] ..
 * dep_ptr[i+1]), all earlier than i, and after nothing else): the chain a stream capture yields is re-wired so that independent
 * branches of a net become parallel branches of the graph.  Requires exactly one kernel per captured call. */
int bodahip_graph_end_deps(bodahip_ctx *ctx, uint32_t n_calls, const uint32_t *dep_ptr, const uint32_t *dep_idx, uint32_t *graph_id_out);
int bodahip_graph_destroy(bodahip_ctx *ctx, uint32_t graph_id);
int bodahip_get_stream(bodahip_ctx *ctx, void **hip_stream_out);   /* the backend's hipStream_t, for event timing / interop */
int bodahip_get_device_info(bodahip_ctx *ctx, char *arch_buf, size_t arch_buf_sz, int *num_cus_out, int *clock_khz_out);
/* tile override for the native kernels (the op_tune_t MNt/MNb/Kb analogue): key "sgemm_tile"|"conv_tile",
 * value "BIxBJxBKxWIxWJ[xMINW[xSPLITK[xMT[xPF[xSW[xKHO]]]]]]" (workgroup tile, K step, waves, min waves per SIMD, K slices, MFMA tile 32|16,
 * K-tiles prefetched 1|2|4|6|8, staging waves 0|1, sequential K segments per tile of an fp32 convolution) or "" to restore the heuristic; key "k1_stream" (the streaming kernel for short-K 1x1 / stride-1 convs,
 * kernels/k1_stream_f32.hip): "" automatic | "off" | "WIxWJxOCBxCB[xMINW]" (waves along out_chan / pel, 32-row / 32-pel blocks per wave);
 * key "timing": how get_dur attributes stream time to calls -- "" | "call" (markers around every call: the reference's semantics,
 * src/nvrtc_util.cc:355-385) | "kernel" (events bound to the call's own dispatches) | "stream" (end markers only: per-call durations add up to
 * the stream time, and a host stall between two launches is billed to the later call).  Switching the mode while call ids are outstanding is
 * an error (fatal, 2): call bodahip_release_per_call_id_data first; setting the mode already in force is a no-op. */
int bodahip_set_tune(bodahip_ctx *ctx, const char *key, const char *value);
int bodahip_last_launch(bodahip_ctx *ctx, char *kernel_buf, size_t kernel_buf_sz, char *cfg_buf, size_t cfg_buf_sz, uint32_t *grid, uint32_t *block,
                        double *flops, double *algo_bytes);
/* device-less hiprtc compile of CUCL-dialect source (prelude prepended iff add_prelude) or of a named native kernel
 * template ("gemm_conv_f32"; src = "-D..." option string); returns code-object size.  Works without a GPU. */
int bodahip_compile_offline(const char *src_or_opts, const char *native_template_or_null, const char *arch, int add_prelude, int use_cache,
                            size_t *code_size_out, char *log_buf, size_t log_buf_sz);

/* ::compile() for functions that arrive as a ready gfx950 code object instead of CUCL source (func_src of the infos is ignored): same
 * registration, arg marshalling and run() as source-compiled functions.  bodahip_compile_to_file writes such a code object from CUCL-dialect
 * source without a device (ahead-of-time generation on a machine that holds the templates; see oracle/ref_cucl.py). */
int bodahip_compile_code_object(bodahip_ctx *ctx, const void *code, size_t code_sz, uint32_t n_funcs, const bodahip_func_info *funcs);
int bodahip_compile_to_file(const char *src, const char *arch, int add_prelude, const char *out_path);

/* parse one op line (op_base_t lexp, current or legacy form -- what bodahip_compile does with bodahip_func_info.op) and
 * write it back in canonical form (sorted str_vals / nda_vals, as NESI prints std::map).  Host-only; for tests/tools. */
int bodahip_parse_op(const char *op_lexp, char *canon_buf, size_t canon_buf_sz);
/* AOT: compile, into the on-disk code-object cache the runtime reads, the native-kernel specialisation run() would pick
 * for the op described by `op_lexp` (sgemm / Convolution line) on a device of `arch` with `num_cus` CUs.  No GPU needed. */
int bodahip_prebuild(const char *op_lexp, const char *arch, int num_cus, const char *tile, size_t *code_size_out);
/* the planner's decision for an annotated op, without compiling or touching a device: "<kernel> <tile> <-D options ...>"
 * (variant / blocking selection is host logic: the counterpart of add_codegen_annotations' choice, src/cnn_op.cc:16-378) */
int bodahip_explain_plan(const char *op_lexp, int num_cus, const char *tile, char *plan_buf, size_t plan_buf_sz);
/* what run-time compilation cost this process so far (the reference publishes it per run: doc/ops-prof-and-wis-ana-usage-notes.txt:9-10, INSTALL.md:283; its compile is
 * nvrtc_compute_t::compile, src/nvrtc_util.cc:216-238): code objects served from the on-disk cache, code objects hiprtc had to compile, milliseconds spent compiling */
int bodahip_compile_stats(uint64_t *cache_hits_out, uint64_t *compiled_out, double *compile_ms_out);

#ifdef __cplusplus
}
#endif
#endif /* BODAHIP_H_ */

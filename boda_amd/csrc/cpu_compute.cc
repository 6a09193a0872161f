// cpu_compute.cc -- cpu_compute_t: `be=cpu`, the host-cores counterpart of be=hip behind the SAME rtc_compute_t contract and C ABI.
//
// the reference has no CPU conv / sgemm path to time beside the GPU (its only precedent is a bare cblas_sgemm
// loop, src/qblas-test.cc:33-41), so the CPU baseline of bench.py is this backend: cache-blocked, vectorised (AVX-512 or AVX2, chosen at
// init), OpenMP over all host cores, fp32, same epilogue -- reached exactly like the GPU kernels, through the native side door
// (op.func_name hip_sgemm / hip_conv, aliases cublas_sgemm / cudnn_conv; tensors in reference layout: a K:M, b K:N, c M:N; in / out
// img:chan:y:x, filts out_chan:in_chan:y:x).  It is a backend one SELECTS (`(be=cpu)`); be=hip never falls back to it.
//   * vars are host buffers (64-byte aligned, zero-filled); copy_nda_to_var / copy_var_to_nda are memcpys; get_dur() from steady_clock
//   * compile(): only the native function names; CUCL source cannot run on a CPU -> unsup_err (the reference's harness records such
//     failures and moves on, src/rtc_prof.cc:287-296)
// Numerics: every output is ONE fp32 fma chain in ascending k (K blocking continues the chain through the tile's partial sums), bias added
// after the chain, then ReLU -- bit-identical to the reference's per-thread fmaf loop, hence to the oracle and to be=hip's fp32 kernels
// (tests/test_cpu_backend.py).
//
// Built with g++ (-fopenmp), linked into libbodahip.so (boda_amd/build.py).
#include "rtc_types.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <omp.h>

namespace bodahip {

namespace {
typedef float vf __attribute__((vector_size(64), aligned(4)));   // 16 floats: one zmm, or two ymm under the AVX2 variant
constexpr int kMR = 6, kNR = 32;       // register tile: 6 rows x 32 columns = 12 accumulators of 16 floats
constexpr int kKC = 256, kMB = 96, kNB = 256;   // cache blocks: K step, rows (16 register tiles) and columns (8 register tiles) of a thread's tile

// ct[m][n] (+)= sum_{k < kc} ap[k*kMB + m] * bp[k*ldb + n]   for the kMR x kNR register tile at (m, n); one fma per (k, m, n), ascending k.
// ap: packed a block [kc][kMB] (zero beyond the matrix); bp: b panel, rows of >= kNR readable floats; ct: the thread's tile buffer [kMB][kNB].
#define MICRO_BODY                                                                                                     \
  vf acc[kMR][2];                                                                                                      \
  for (int m = 0; m < kMR; ++m) for (int h = 0; h < 2; ++h) {                                                            \
    if (first) { vf z = {0}; acc[m][h] = z; } else acc[m][h] = *(vf const *)(ct + m * kNB + 16 * h); }                   \
  for (int k = 0; k < kc; ++k) {                                                                                       \
    vf const b0 = *(vf const *)(bp + (long)k * ldb), b1 = *(vf const *)(bp + (long)k * ldb + 16);                        \
    float const *ak = ap + k * kMB;                                                                                    \
    for (int m = 0; m < kMR; ++m) {                                                                                    \
      float const av = ak[m];                                                                                          \
      vf const va = {av, av, av, av, av, av, av, av, av, av, av, av, av, av, av, av};                                  \
      acc[m][0] += va * b0; acc[m][1] += va * b1;   /* contracted to vfmadd (-ffp-contract=fast): one rounding */      \
    }                                                                                                                  \
  }                                                                                                                    \
  for (int m = 0; m < kMR; ++m) for (int h = 0; h < 2; ++h) *(vf *)(ct + m * kNB + 16 * h) = acc[m][h];
__attribute__((target("avx512f,fma"))) void micro_avx512(float const *ap, float const *bp, long ldb, float *ct, int kc, bool first) { MICRO_BODY }
__attribute__((target("avx2,fma"))) void micro_avx2(float const *ap, float const *bp, long ldb, float *ct, int kc, bool first) { MICRO_BODY }
#undef MICRO_BODY
typedef void (*micro_t)(float const *, float const *, long, float *, int, bool);

struct scratch_t { float *ap, *bp, *ct; };   // per thread: packed a block [kKC][kMB], b panel [kKC][kNB], tile [kMB][kNB]
void *aligned(size_t bytes) { void *p = nullptr; if (posix_memalign(&p, 64, std::max<size_t>(bytes, 64))) rt_err("be=cpu: out of memory"); return p; }

// D[i][j] = sum_k A(k, i) * B(k, j) over tiles of kMB x kNB; A(k, i) = at[k*lda + i] (k-major); the panel of B and the scatter of D are
// supplied by the caller (sgemm: rows of b / rows of c; conv: im2col gather / NCHW scatter with bias + ReLU).
template <typename PanelF, typename StoreF>
void tiled_contract(micro_t micro, float const *at, long lda, long Mi, long Nj, long K, PanelF panel, StoreF store) {
  long const tm = (Mi + kMB - 1) / kMB, tn = (Nj + kNB - 1) / kNB;
#pragma omp parallel
  {
    scratch_t s; s.ap = (float *)aligned(sizeof(float) * kKC * kMB); s.bp = (float *)aligned(sizeof(float) * kKC * kNB); s.ct = (float *)aligned(sizeof(float) * kMB * kNB);
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (long tj = 0; tj < tn; ++tj)
      for (long ti = 0; ti < tm; ++ti) {
        long const i0 = ti * kMB, j0 = tj * kNB, mb = std::min<long>(kMB, Mi - i0), nb = std::min<long>(kNB, Nj - j0);
        for (long k0 = 0; k0 < K; k0 += kKC) {
          int const kc = (int)std::min<long>(kKC, K - k0);
          for (int k = 0; k < kc; ++k) {     // pack the a block: rows are contiguous in i; zero beyond the matrix
            memcpy(s.ap + k * kMB, at + (k0 + k) * lda + i0, sizeof(float) * mb);
            if (mb < kMB) memset(s.ap + k * kMB + mb, 0, sizeof(float) * (kMB - mb));
          }
          panel(s.bp, k0, kc, j0, nb);       // [kc][kNB], zero beyond column nb
          for (long mo = 0; mo < mb; mo += kMR)
            for (long no = 0; no < nb; no += kNR) micro(s.ap + mo, s.bp + no, kNB, s.ct + mo * kNB + no, kc, k0 == 0);
        }
        store(s.ct, i0, mb, j0, nb);
      }
    free(s.ap); free(s.bp); free(s.ct);
  }
}

struct conv_geom_c { long B, C, H, W, OC, KH, KW, SY, SX, PY, PX, OH, OW; bool relu; };
} // namespace

struct cpu_var_t { std::shared_ptr<void> buf; dims_t dims; };
struct cpu_func_t { rtc_func_info_t info; };

struct cpu_compute_t : public rtc_compute_t {
  bool init_done = false;
  micro_t micro = nullptr;
  string isa;
  std::map<string, cpu_var_t> vis;
  std::map<string, cpu_func_t> funcs;
  std::vector<std::pair<double, double>> call_t;   // (begin, end) in ms since init
  std::chrono::steady_clock::time_point t0;
  cpu_compute_t() { be = "cpu"; }

  double now_ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  void init() override {
    assert_st(!init_done);
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma") && !getenv("BODACPU_NO_AVX512")) { micro = micro_avx512; isa = "avx512"; }
    else if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) { micro = micro_avx2; isa = "avx2"; }
    else rt_err("cpu backend: this host has neither AVX-512 nor AVX2 with FMA (the kernels need fused multiply-add for the reference's fmaf chain)");
    t0 = std::chrono::steady_clock::now(); init_done = true;
  }
  string get_plat_tag() override { assert_st(init_done); return "cpu:" + isa + ":" + std::to_string(omp_get_max_threads()) + "t"; }

  // ---- vars
  void create_var_with_dims(string const &vn, dims_t const &dims) override {
    assert_st(init_done);
    if (vis.count(vn)) rt_err("create_var_with_dims: var '" + vn + "' already exists");
    size_t const sz = dims.bytes_sz();
    cpu_var_t v; v.dims = dims; v.buf = std::shared_ptr<void>(aligned(sz), free); memset(v.buf.get(), 0, sz);
    vis.emplace(vn, std::move(v));
  }
  void create_var_with_dims_as_reshaped_view_of_var(string const &vn, dims_t const &dims, string const &src_vn) override {
    cpu_var_t const &src = must_find(vis, src_vn);
    rtc_reshape_check(dims, src.dims);
    if (vis.count(vn)) rt_err("create_var_with_dims_as_reshaped_view_of_var: var '" + vn + "' already exists");
    cpu_var_t v; v.dims = dims; v.buf = src.buf; vis.emplace(vn, std::move(v));
  }
  void release_var(string const &vn) override { must_erase(vis, vn); }
  dims_t get_var_dims(string const &vn) override { return must_find(vis, vn).dims; }
  void set_var_to_zero(string const &vn) override { cpu_var_t const &v = must_find(vis, vn); memset(v.buf.get(), 0, v.dims.bytes_sz()); }
  void copy_nda_to_var(string const &vn, p_nda_t const &nda) override {
    cpu_var_t const &v = must_find(vis, vn);
    if (!(v.dims == nda->dims)) rt_err("copy_nda_to_var: dims mismatch for var '" + vn + "': var " + v.dims.pretty_str() + " nda " + nda->dims.pretty_str());
    memcpy(v.buf.get(), nda->rp_elems(), v.dims.bytes_sz());
  }
  void copy_var_to_nda(p_nda_t const &nda, string const &vn) override {
    cpu_var_t const &v = must_find(vis, vn);
    if (!(v.dims == nda->dims)) rt_err("copy_var_to_nda: dims mismatch for var '" + vn + "': var " + v.dims.pretty_str() + " nda " + nda->dims.pretty_str());
    memcpy(nda->rp_elems(), v.buf.get(), v.dims.bytes_sz());
  }
  p_nda_t get_var_raw_native_pointer(string const &vn) override { cpu_var_t const &v = must_find(vis, vn); return std::make_shared<nda_t>(v.dims, v.buf.get()); }

  // ---- functions: the native side door only
  static bool is_sgemm(string const &fn) { return fn == "hip_sgemm" || fn == "cublas_sgemm" || fn == "cpu_sgemm"; }
  static bool is_conv(string const &fn) { return fn == "hip_conv" || fn == "cudnn_conv" || fn == "cpu_conv_fwd"; }
  void compile(vect_rtc_func_info_t const &func_infos, rtc_compile_opts_t const &) override {
    assert_st(init_done);
    for (auto const &fi : func_infos) {
      if (funcs.count(fi.func_name)) rt_err("compile: function '" + fi.func_name + "' already exists");
      string const fn = fi.op.has_func_name() ? fi.op.get_func_name() : string();
      if (!is_sgemm(fn) && !is_conv(fn)) unsup_err("be=cpu runs the native sgemm / Convolution functions only (hip_sgemm, hip_conv and their aliases); '" +
                                                    (fn.empty() ? fi.func_name : fn) + "' is generated CUCL source, which needs a GPU backend");
      if (is_conv(fn)) (void)fi.op.get_u32("conv_has_relu");
      funcs.emplace(fi.func_name, cpu_func_t{fi});
    }
  }
  void release_func(string const &func_name) override { must_erase(funcs, func_name); }
  void release_all_funcs() override { funcs.clear(); }

  static string var_of(map_str_rtc_arg_t const &am, string const &an) {
    auto i = am.find(an);
    if (i == am.end()) rt_err("cpu_compute_t: arg '" + an + "' not found in arg_map for call.");
    if (!i->second.is_valid() || !i->second.is_var()) rt_err("cpu_compute_t: arg '" + an + "' must be a var");
    return i->second.n;
  }
  static void need_float(dims_t const &d, char const *an) { if (d.tn != "float") unsup_err(string("be=cpu: arg '") + an + "' has type " + d.tn + "; only float is supported"); }

  void sgemm(float const *a, float const *b, float *c, long M, long N, long K) {
    if (!M || !N) return;
    if (!K) { memset(c, 0, sizeof(float) * M * N); return; }
    tiled_contract(micro, a, M, M, N, K,
      [&](float *bp, long k0, int kc, long j0, long nb) {
        for (int k = 0; k < kc; ++k) { memcpy(bp + k * kNB, b + (k0 + k) * N + j0, sizeof(float) * nb); if (nb < kNB) memset(bp + k * kNB + nb, 0, sizeof(float) * (kNB - nb)); } },
      [&](float const *ct, long i0, long mb, long j0, long nb) { for (long m = 0; m < mb; ++m) memcpy(c + (i0 + m) * N + j0, ct + m * kNB, sizeof(float) * nb); });
  }
  void conv(float const *filts, float const *biases, float const *in, float *out, conv_geom_c const &g) {
    long const K = g.C * g.KH * g.KW, Nj = g.B * g.OH * g.OW, OHW = g.OH * g.OW;
    if (!Nj || !g.OC) return;
    std::vector<float> ft((size_t)K * g.OC);     // filters k-major: ft[k][oc]
#pragma omp parallel for schedule(static)
    for (long oc = 0; oc < g.OC; ++oc) for (long k = 0; k < K; ++k) ft[k * g.OC + oc] = filts[oc * K + k];
    tiled_contract(micro, ft.data(), g.OC, g.OC, Nj, K,
      [&](float *bp, long k0, int kc, long j0, long nb) {     // im2col of columns j0 .. j0+nb, rows k0 .. k0+kc (cross-correlation, zero padding: test/rtc/conv.cucl:33-36)
        long base[kNB]; int iy0[kNB], ix0[kNB];               // per column: offset of (img, chan 0, iy0, ix0) and the window origin
        for (long n = 0; n < nb; ++n) {
          long const j = j0 + n, img = j / OHW, pel = j - img * OHW, oy = pel / g.OW, ox = pel - oy * g.OW;
          iy0[n] = (int)(oy * g.SY - g.PY); ix0[n] = (int)(ox * g.SX - g.PX);
          base[n] = (img * g.C * g.H + iy0[n]) * g.W + ix0[n];
        }
        for (int kk = 0; kk < kc; ++kk) {
          long const k = k0 + kk, c = k / (g.KH * g.KW), r = k - c * (g.KH * g.KW), ky = r / g.KW, kx = r - ky * g.KW;
          long const koff = (c * g.H + ky) * g.W + kx;
          float *row = bp + kk * kNB;
          for (long n = 0; n < nb; ++n) {
            int const iy = iy0[n] + (int)ky, ix = ix0[n] + (int)kx;
            row[n] = ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) ? in[base[n] + koff] : 0.f;
          }
          if (nb < kNB) memset(row + nb, 0, sizeof(float) * (kNB - nb));
        } },
      [&](float const *ct, long i0, long mb, long j0, long nb) {   // bias after the chain, then ReLU (src/cnn_codegen.cc:35-42)
        for (long m = 0; m < mb; ++m) {
          float const bias = biases[i0 + m];
          for (long n = 0; n < nb; ++n) {
            long const j = j0 + n, img = j / OHW, pel = j - img * OHW;
            float v = ct[m * kNB + n] + bias;
            if (g.relu) v = (v > 0.f) ? v : 0.f;
            out[(img * g.OC + i0 + m) * OHW + pel] = v;
          }
        } });
  }

  uint32_t run(rtc_func_call_t const &rfc) override {
    assert_st(init_done);
    auto fit = funcs.find(rfc.rtc_func_name);
    if (fit == funcs.end()) rt_err("run: unknown function '" + rfc.rtc_func_name + "' (not compiled, or released)");
    rtc_func_info_t const &fi = fit->second.info;
    string const &fn = fi.op.get_func_name();
    map_str_rtc_arg_t const &am = rfc.arg_map;
    double const tb = now_ms();
    if (is_sgemm(fn)) {
      string const an = var_of(am, "a"), bn = var_of(am, "b"), cn = var_of(am, "c");
      dims_t const a = get_var_dims(an), b = get_var_dims(bn), c = get_var_dims(cn);
      need_float(a, "a"); need_float(b, "b"); need_float(c, "c");
      assert_st(a.sz() == 2 && b.sz() == 2 && c.sz() == 2);
      assert_st(a.names(0) == "K" && a.names(1) == "M" && b.names(0) == "K" && b.names(1) == "N" && c.names(0) == "M" && c.names(1) == "N");
      uint32_t const M = a.dsz("M"), K = a.dsz("K"), N = b.dsz("N");
      assert_st(b.dsz("K") == K); assert_st(c.dsz("M") == M); assert_st(c.dsz("N") == N);
      sgemm((float const *)must_find(vis, an).buf.get(), (float const *)must_find(vis, bn).buf.get(), (float *)must_find(vis, cn).buf.get(), M, N, K);
    } else {
      string const fnm = var_of(am, "filts"), bnm = var_of(am, "biases"), inm = var_of(am, "in"), onm = var_of(am, "out");
      dims_t const f = get_var_dims(fnm), bi = get_var_dims(bnm), in = get_var_dims(inm), out = get_var_dims(onm);
      need_float(f, "filts"); need_float(bi, "biases"); need_float(in, "in"); need_float(out, "out");
      auto si = am.find("stride"), pi = am.find("in_pad");
      if (si == am.end() || pi == am.end()) rt_err("hip_conv: 'stride' and 'in_pad' REF args are required");
      dims_t const stride = si->second.get_dims(*this), in_pad = pi->second.get_dims(*this);
      assert_st(f.sz() == 4 && in.sz() == 4 && out.sz() == 4 && bi.sz() == 1 && stride.sz() == 2 && in_pad.sz() == 2);
      conv_geom_c g;
      g.B = in.dsz("img"); g.C = in.dsz("chan"); g.H = in.dsz("y"); g.W = in.dsz("x"); g.OC = f.dsz("out_chan"); g.KH = f.dsz("y"); g.KW = f.dsz("x");
      g.SY = stride.dsz("y"); g.SX = stride.dsz("x"); g.PY = in_pad.dsz("y"); g.PX = in_pad.dsz("x"); g.OH = out.dsz("y"); g.OW = out.dsz("x");
      g.relu = fi.op.get_u32("conv_has_relu") != 0;
      if (f.dsz("in_chan") != (uint32_t)g.C) rt_err("hip_conv: filts.in_chan != in.chan");
      if (bi.dsz("out_chan") != (uint32_t)g.OC || out.dsz("chan") != (uint32_t)g.OC || out.dsz("img") != (uint32_t)g.B) rt_err("hip_conv: inconsistent biases/out dims");
      if (!g.SY || !g.SX) rt_err("hip_conv: zero stride");
      if ((g.H + 2 * g.PY - g.KH) / g.SY + 1 != g.OH || (g.W + 2 * g.PX - g.KW) / g.SX + 1 != g.OW) rt_err("hip_conv: out dims do not match in/filts/stride/in_pad");
      conv((float const *)must_find(vis, fnm).buf.get(), (float const *)must_find(vis, bnm).buf.get(), (float const *)must_find(vis, inm).buf.get(),
           (float *)must_find(vis, onm).buf.get(), g);
    }
    call_t.emplace_back(tb, now_ms());
    return (uint32_t)call_t.size() - 1;
  }
  void finish_and_sync() override {}
  void release_per_call_id_data() override { call_t.clear(); }
  float get_dur(uint32_t const &b, uint32_t const &e) override {
    if (b >= call_t.size() || e >= call_t.size()) rt_err("invalid call_id");
    return (float)(call_t[e].second - call_t[b].first);
  }
  void profile_start() override {}
  void profile_stop() override {}
};

p_rtc_compute_t make_cpu_compute() { return std::make_shared<cpu_compute_t>(); }

} // namespace bodahip

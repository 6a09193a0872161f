// native_kernels.cc -- host side of the hand-written kernels: variant/tile selection, hiprtc specialisation, launch.
// See native_kernels.h for the contract and the reference precedent.
#include "native_kernels.h"
#include <algorithm>
#include <cstdlib>
#include <sstream>

namespace bodahip {

// kernel template sources, embedded at build time from kernels/*.hip (see build.py: kernels_embed.inc)
#include "kernels_embed.inc"

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

native_kernels_t::native_kernels_t(native_host_t *host_) : impl(new impl_t), host(host_) {
  if (char const *e = getenv("BODAHIP_SGEMM_TILE")) impl->tune["sgemm_tile"] = e;
  if (char const *e = getenv("BODAHIP_CONV_TILE")) impl->tune["conv_tile"] = e;
  if (char const *e = getenv("BODAHIP_K1_STREAM")) impl->tune["k1_stream"] = e;
  if (char const *e = getenv("BODAHIP_CONV_ALGO")) impl->tune["conv_algo"] = e;
  if (char const *e = getenv("BODAHIP_EXACT")) impl->tune["exact"] = e;
}
native_kernels_t::~native_kernels_t() {
  if (impl->ts_bytes && impl->ws) {
    char const *e = getenv("BODAHIP_CBIG_TSTAMP"); string fn = e ? string(e) : string(); if (fn.size() > 5 && fn.substr(fn.size() - 5) == ":late") fn.resize(fn.size() - 5);
    std::vector<unsigned long long> h(impl->ts_bytes / 8);
    if (!fn.empty() && hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), (char *)impl->ws + impl->ts_off, impl->ts_bytes, hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE *f = fopen(fn.c_str(), "a")) { fprintf(f, "%s\n", impl->ts_hdr.c_str());
        for (size_t w = 0; w < h.size() / 16; ++w) { for (int e2 = 0; e2 < 16; ++e2) fprintf(f, "%llu ", h[w * 16 + e2]); fprintf(f, "\n"); } fclose(f); }
    }
  }
  for (auto &kv : impl->kernels) { if (kv.second.mod) (void)hipModuleUnload(kv.second.mod); }
  if (impl->wino_mod) (void)hipModuleUnload(impl->wino_mod);
  if (impl->ws) (void)hipFree(impl->ws);
  for (void *r : impl->ws_retired) (void)hipFree(r);
  for (auto &kv : impl->ktabs) (void)hipFree(kv.second);
  delete impl;
}
uint32_t native_kernels_t::num_specialisations() const { return (uint32_t)impl->kernels.size(); }

bool native_kernels_t::is_native_func_name(string const &fn) {
  if (fn.find("_xpose_") != string::npos) return false; // (layout passes are generated CUCL functions, as the reference's <func>_xpose_<arg>)
  return fn == "hip_sgemm" || fn == "hip_conv" || fn == "cublas_sgemm" || fn == "cudnn_conv" || startswith(fn, "hip_");
}
void native_kernels_t::check_compile_time(rtc_func_info_t const &fi) {
  string const &fn = fi.op.get_func_name();
  if (fn == "hip_sgemm" || fn == "cublas_sgemm" || fn == "hip_sgemm_bf16") return;
  if (fn == "hip_conv" || fn == "cudnn_conv" || fn == "hip_conv_bf16" || fn == "hip_conv_winograd" || fn == "hip_conv_nhwc" || fn == "hip_conv_nhwc_grp" || fn == "hip_conv_nhwc_multi" || fn == "hip_conv_nhwc_set") { (void)fi.op.get_u32("conv_has_relu"); return; } // required, as src/culibs-wrap.cc:198
  if (fn == "hip_conv_k1_chain") { (void)fi.op.get_u32("conv_has_relu"); (void)fi.op.get_u32("conv_has_relu2"); return; }
  if (fn == "hip_conv_filts_kmajor") return;
  rt_err("unknown/unhandled native hip function: " + fn);
}
void native_kernels_t::set_tune(string const &key, string const &val) {
  if (key != "sgemm_tile" && key != "conv_tile" && key != "k1_stream" && key != "conv_algo" && key != "exact") rt_err("set_tune: unknown key '" + key + "'");
  if (key == "exact" && !val.empty() && val != "0" && val != "1") rt_err("set_tune: exact must be 0 | 1");
  if (key == "conv_algo" && !val.empty() && val != "direct" && val != "winograd" && val != "winograd_all") rt_err("set_tune: conv_algo must be direct | winograd | winograd_all");
  if (val.empty()) impl->tune.erase(key); else { impl->tune[key] = val; }
}

// "BIxBJxBKxWIxWJ[xMINW[xSPLITK[xMT[xPF[xSW[xKHO]]]]]]"
static bool parse_tile(string const &s, tile_cfg_t &c) {
  int v[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int n = 0; string cur;
  for (size_t i = 0; i <= s.size(); ++i) {
    if (i == s.size() || s[i] == 'x' || s[i] == ':') { if (cur.empty() || n >= 11) return false; v[n++] = atoi(cur.c_str()); cur.clear(); }
    else if (s[i] >= '0' && s[i] <= '9') cur.push_back(s[i]); else return false;
  }
  if (n < 5) return false;
  c.BI = v[0]; c.BJ = v[1]; c.BK = v[2]; c.WI = v[3]; c.WJ = v[4]; c.MINW = (n >= 6) ? v[5] : 1; c.SPLITK = (n >= 7) ? v[6] : 1; c.MT = (n >= 8) ? v[7] : 32; c.PF = (n >= 9) ? v[8] : 1; c.SW = (n >= 10) ? v[9] : 0; c.KHO = (n >= 11) ? v[10] : 0;
  return true;
}
// the static_asserts of the kernel, checked on the host so that a bad tune is an unsup_err, not a compile failure
static void check_cfg(tile_cfg_t const &c, bool gather, bool patch = false) {   // patch: the J image is an input patch sized (and checked) by the patch planner, not BK x BJ
  int const nt = c.WI * c.WJ * 64;   // multiplying threads (= staging threads)
  bool ok = c.BI > 0 && c.BJ > 0 && c.BK > 0 && c.WI > 0 && c.WJ > 0 && nt <= 1024 && c.threads() <= 1024 && (c.MT == 32 || c.MT == 16) && (c.BI % (c.WI * c.MT) == 0) && (c.BJ % (c.WJ * c.MT) == 0) &&
            (c.BK % 2 == 0) && (c.MT == 32 || c.BK % 4 == 0) && (c.BI % 4 == 0) && (c.BJ % 4 == 0);
  if (gather) ok = ok && ((c.BK * c.BJ) % nt == 0); // the gathers give every thread whole elements / rows
  if (gather) ok = ok && (nt % c.BJ == 0) && (c.BJ % 64 == 0);
  ok = ok && c.SPLITK >= 1 && c.SPLITK <= 64 && c.MINW >= 1 && (c.PF == 1 || c.PF == 2 || c.PF == 4 || c.PF == 6 || c.PF == 8);
  ok = ok && c.KHO >= 0 && c.KHO <= 64 && (c.KHO <= 1 || (c.SPLITK == 1 && c.SW == 0));   // K hand-off: an exact form (no K slices), not with staging waves
  if (c.PF > 2) ok = ok && ((long)c.PF * c.BK * (c.BI + c.BJ) / nt <= 192);   // (the ring of register sets: PF x staged elements per thread)
  int const accs = (c.BI / (c.WI * c.MT)) * (c.BJ / (c.WJ * c.MT));
  ok = ok && accs * (c.MT == 32 ? 16 : 4) <= 256;
  uint64_t const lds = 2ull * c.BK * (c.BI + 4 + (patch ? 0 : c.BJ + 4)) * 4;
  ok = ok && lds <= 160 * 1024;
  if (!ok) unsup_err("native kernel: unsupported tile configuration " + c.str());
}

// Tile heuristic: score = measured base rate of the tile shape x fraction of the padded tile grid that is real work x how evenly
// the workgroups deal out over the CUs (tiles / (num_cus * ceil(tiles / num_cus)); with fewer tiles than CUs this is the fraction
// of CUs that get one).  Base rates are steady-state MI355X measurements relative to 128x128 (sgemm 4096^3..12288^3, AlexNet /
// NiN / GoogLeNet layers, tools/tile_sweep.py):
//   128x128 w2x2 1.00 | 256x256 w2x4 with two K-tiles in flight 1.02 (k-major operands only: halves the HBM re-reads; eight 128x64 waves, 8192^3: 140.7 TF/s vs 135.6 as sixteen 64x64 waves with one; BK32 137.7) | 96x256 w1x4 0.95 (gathers; OC = 96-multiples)
//   64x64 w2x2 with two K-tiles in flight 0.93 (two-wave 64x128 / 32x128 workgroups measured 1.3-1.8x slower than this and are gone)
//   32x64 as eight 16x16x4-MFMA waves 0.60 (thin out_chan / tile-starved: 2-2.7x faster than 32x128 there) | 32x32 m16 w2x2 0.45
// Splitting K would fill the chip for tile-starved shapes too, but it re-associates the fp32 sum: the reference's golden digests
// (tolerance 2e-4 on max(1,|v|)) are only met robustly by the ascending-k chain, so SPLITK is an explicit tune, never the default.
static double const kShortTail = getenv("BODAHIP_SHORT_TAIL") ? atof(getenv("BODAHIP_SHORT_TAIL")) : 0.6; // tile-times per CU a short launch loses to ramp-up / tail
static tile_cfg_t choose_cfg(int Mi, int Nj, int K, int num_cus, bool gather, bool bf16 = false) {
  struct cand_t { int bi, bj, bk, wi, wj, minw, mt, pf; double base; bool gather_ok, plain_ok; };
  // bf16 kernel (32x32x16 MFMA only, staging-bound: large tiles matter more; 8192^3: 256x256 624 TF/s vs 128x128 457)
  static cand_t const cands_bf16[] = {
    {128, 128, 32, 2, 2, 2, 32, 1, 1.00, true, true}, {256, 256, 32, 4, 4, 1, 32, 1, 1.35, false, true}, {128, 256, 32, 2, 4, 1, 32, 1, 1.15, true, false},
    {96, 128, 32, 1, 2, 2, 32, 1, 0.85, true, true},  {64, 64, 32, 2, 2, 2, 32, 1, 0.70, true, true}};
  static cand_t const cands_f32[] = {
    {128, 128, 16, 2, 2, 2, 32, 1, 1.00, true, true},  {256, 256, 16, 2, 4, 1, 32, 2, 1.02, false, true}, {96, 256, 16, 1, 4, 2, 32, 1, 0.95, true, false},
    {96, 128, 16, 1, 2, 2, 32, 1, 0.90, false, true},  {64, 64, 16, 2, 2, 2, 32, 2, 0.93, true, true},    {32, 64, 32, 2, 4, 1, 16, 4, 0.60, true, true},
    {32, 32, 32, 2, 2, 1, 16, 8, 0.45, false, true}}; // (a ring of register-staged K tiles, round 4 -- tools/fc_pf_sweep.py, isolated layers, us: AlexNet fc8 32x32x64 two tiles in
  // flight 63.6 -> 32x32x32 eight in flight 51.1; GoogLeNet's classifier at 64 images 18.8 -> 16.3; fc6 at 128 images on 32x64x32: two in flight 181 -> four 148.5)
  tile_cfg_t best_c; double best = -1;
  // launches shorter than ~200 us at full rate also pay ramp-up / tail: about 0.6 tile-times per CU (measured NiN 1x1 layers at
  // B=128: 507 128x128 tiles 83 TF/s, 2028 64x64 tiles 88-91), which favours finer tiles there
  bool const short_kernel = !bf16 && 2.0 * Mi * (double)Nj * K < 2.4e10;
  cand_t const *cands = bf16 ? cands_bf16 : cands_f32;
  int const n_cands = bf16 ? (int)(sizeof(cands_bf16) / sizeof(cand_t)) : (int)(sizeof(cands_f32) / sizeof(cand_t));
  for (int ci = 0; ci < n_cands; ++ci) {
    cand_t const &cd = cands[ci];
    if (gather ? !cd.gather_ok : !cd.plain_ok) continue;
    if (cd.mt == 16 && 2.0 * Mi * (double)Nj * K < 1e8) continue; // tiny problems: launch-bound either way, keep the common kernel
    long const ti = (Mi + cd.bi - 1) / cd.bi, tj = (Nj + cd.bj - 1) / cd.bj, tiles = ti * tj;
    if (cd.bj == 256 && cd.bi >= 128 && tiles < num_cus) continue;
    double const pad = ((double)Mi / (double)(ti * cd.bi)) * ((double)Nj / (double)(tj * cd.bj));
    double const bal = ((double)tiles / num_cus) / (double)((tiles + num_cus - 1) / num_cus);
    double score = cd.base * pad * bal;
    if (short_kernel) { double const x = (double)tiles / num_cus; score *= x / (x + kShortTail); }
    if (score > best) { best = score; best_c.BI = cd.bi; best_c.BJ = cd.bj; best_c.BK = (cd.bi == 64 && cd.bj == 64 && !gather && !bf16) ? 32 : cd.bk; // (k-contiguous / plain operands, usually cold from HBM: twice the bytes in flight; AlexNet fc6/fc7 in sequence 63 -> 78 TF/s)
      best_c.WI = cd.wi; best_c.WJ = cd.wj; best_c.MINW = cd.minw; best_c.MT = cd.mt; best_c.PF = cd.pf; best_c.SPLITK = 1;
      // one workgroup per CU and a long K loop (fully-connected layers: 256 tiles, K = 4096 / 9216): four K tiles in flight instead of two (fc6 252.7 -> 237.8 us,
      // fc7 116.6 -> 111.6; with several workgroups per CU -- sgemm 2048^3 / 3072^3 -- the deeper ring only costs registers: 164 -> 170 us)
      if (!bf16 && cd.bi == 64 && cd.bj == 64 && cd.mt == 32 && !gather && tiles <= num_cus && K >= 2048) best_c.PF = 4;
      if (cd.mt == 16 && gather) best_c.PF = 2; }   // (the deeper rings were measured on k-contiguous operands only)
  }
  return best_c;
}

// 64x64 workgroups of four 32x32 wave tiles are what tile-starved shapes get (often a single workgroup per CU): their ~0.5 us
// MFMA phase per K step cannot cover HBM latency with one K-tile in flight, two can (measured fc6 70 -> 81, fc7 67 -> 76,
// sgemm 2048^3 95 -> 107 TF/s); larger tiles gain nothing and pay registers.
static int pf_for(tile_cfg_t const &c) { return (c.BI == 64 && c.BJ == 64 && c.WI == 2 && c.WJ == 2 && c.MT == 32 && c.SPLITK == 1) ? 2 : 1; }
static vect_string cfg_defs(tile_cfg_t const &c) {
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { // experiment hook: extra -D options for the native kernels
    vect_string r = {"-DBI=" + std::to_string(c.BI), "-DBJ=" + std::to_string(c.BJ), "-DBK=" + std::to_string(c.BK), "-DWI=" + std::to_string(c.WI),
                     "-DWJ=" + std::to_string(c.WJ), "-DMINW=" + std::to_string(c.MINW), "-DMT=" + std::to_string(c.MT), "-DPF=" + std::to_string(c.PF), "-DSPECW=" + std::to_string(c.SW)};
    if (c.KHO > 1) r.push_back("-DKHO=1");
    std::istringstream is(e); string tok; while (is >> tok) r.push_back(tok);
    return r;
  }
  vect_string r = {"-DBI=" + std::to_string(c.BI), "-DBJ=" + std::to_string(c.BJ), "-DBK=" + std::to_string(c.BK), "-DWI=" + std::to_string(c.WI),
          "-DWJ=" + std::to_string(c.WJ), "-DMINW=" + std::to_string(c.MINW), "-DMT=" + std::to_string(c.MT), "-DPF=" + std::to_string(c.PF), "-DSPECW=" + std::to_string(c.SW)};
  if (c.KHO > 1) r.push_back("-DKHO=1");
  return r;
}
struct plan_t;
static kernel_t &get_kernel(native_kernels_t::impl_t *impl, native_host_t *host, plan_t const &p);
static void launch(native_host_t *host, kernel_t &k, gemm_args_t &a, tile_cfg_t const &c) {
  void *params[] = {&a};
  uint32_t const grid = (uint32_t)a.tiles_i * (uint32_t)a.tiles_j * (uint32_t)std::max(1, a.splitk);
  hip_err_chk(host->nh_launch(k.func, grid, 1, (uint32_t)c.threads(), params), "hipModuleLaunchKernel(native)");
}

struct plan_t { tile_cfg_t cfg; vect_string defs; string kname; long split_pels = 0; tile_cfg_t tail_cfg; vect_string tail_defs;   /* split_pels > 0 (staging-wave convolution, round 6): two-level tiling along the pels -- this plan's tiles over the first split_pels pels (whole rounds of the CUs), tail_cfg's over the rest */ bool ipconv = false, k1 = false, bf16 = false, patch = false, stream = false, quad = false, fc = false, big = false, cbig = false, rdec = false, patch16 = false, nhwc = false, nhwc_patch = false, nhwc_multi = false, nhwc_rows = false, ksl = false; int rows = 0, cg = 0; };

// Streaming kernel for short-K 1x1 convolutions (kernels/k1_stream_f32.hip): resident filters, persistent waves, no K tiling.
//   spec: "" = automatic | "off" | "WIxWJxOCBxCB[xMINW]" (waves along out_chan / pel, 32-row and 32-pel blocks per wave)
// In the returned plan cfg.BI = out_chans per workgroup, cfg.BJ = pels per super-block, cfg.BK = in_chans.
static bool plan_k1_stream(conv_geom_t const &g, int num_cus, string const &spec, plan_t &p) {
  (void)num_cus;
  if (spec == "off" || getenv("BODAHIP_NO_K1_STREAM")) return false;
  if (!(g.KH == 1 && g.KW == 1 && g.SY == 1 && g.SX == 1 && g.PY == 0 && g.PX == 0)) return false; // (a spec only applies to the shapes the kernel covers)
  long const Nj = (long)g.B * g.OH * g.OW;
  int WI = 0, WJ = 0, OCB = 0, CB = 0, MINW = 0;
  auto regs = [&](int ocb, int cb) { return (g.C + 1) / 2 * cb + 2 * 16 * ocb * cb + 30; }; // operand ring + two accumulator sets
  auto lds = [&](int wi, int ocb) { long const oct = wi * ocb * 32; return 4 * ((long)((g.C + 1) / 2 * 2) * (oct | 1) + oct); };
  // kernels/k1_quad_f32.hip (16 bytes per lane both ways: 128-pel blocks of one image per wave, every wave all of the workgroup's out_chans): spec "qWJxOCBxRING[xMINW]";
  // automatic for the NiN cccp1/2 class -- at most 96 out_chans (one accumulator set of OCB*64 registers, two waves per SIMD), a short K loop, a long pel axis
  {
    int QWJ = 0, QOCB = 0, QRING = 0, QMINW = 0;
    int const ksteps = (g.C + 1) / 2;
    if (!spec.empty() && spec[0] == 'q') {
      int v[4] = {0, 0, 0, 0}, n = 0; size_t i = 1;
      while (i < spec.size() && n < 4) { size_t j = spec.find('x', i); if (j == string::npos) j = spec.size(); v[n++] = atoi(spec.substr(i, j - i).c_str()); i = j + 1; }
      if (n < 3) rt_err("bad k1_stream spec '" + spec + "' (qWJxOCBxRING[xMINW])");
      QWJ = v[0]; QOCB = v[1]; QRING = v[2]; QMINW = v[3];
      if (QWJ < 1 || QWJ > 16 || QOCB < 1 || QOCB > 4 || QRING < 1 || QRING > 16 || ksteps % QRING || g.OH * g.OW < 4 || lds(1, QOCB) > 160 * 1024)
        unsup_err("k1_stream: unsupported configuration '" + spec + "' for this shape");
    } else if (spec.empty() && !getenv("BODAHIP_NO_K1_QUAD") && g.OC > 64 && g.OC <= 96 && g.C <= 128 && g.OH * g.OW >= 512 && Nj >= 150000) {
      // measured (MI355X, tools/k1s_probe.py, NiN cccp1 at 256 / 128 images, us): tiled kernel 159.5 / 88.6; q4x3x8 two workgroups per CU 156.5 / 95.7; q8x3x8 153.5 / 93.7;
      // q4x3x8x1 (one workgroup of four waves per CU, six / three blocks per wave) 153.0 / 84.6; rings of 12 / 16 steps 204 / 208 (with the epilogue's 48 stores
      // they overflow the 6-bit vmcnt and every wait becomes a drain)
      QWJ = 4; QOCB = (g.OC + 31) / 32; QRING = 8; QMINW = 1; while (ksteps % QRING) --QRING;
      if (lds(1, QOCB) > 80 * 1024) QWJ = 0;
    } else if (spec.empty() && !getenv("BODAHIP_NO_K1_QUAD") && g.OC > 96 && g.OC <= 256 && g.OC % 64 == 0 && g.C > 128 && g.C <= 256 && g.OH * g.OW >= 512 && Nj >= 180000) {
      // NiN cccp3 / cccp4 class (256 -> 256 on 27 x 27) at 256 images, once the filter image was staged with all its loads in flight (round 4c; before, 64 serial round
      // trips cost this kernel 40 us per launch): eight waves, 64 out_chans per workgroup (four passes over the input, from L2), two workgroups per CU.  Measured
      // (tools/k1s_probe.py, us): tiled 128x128 219.1 | q8x2x8x2 203.6 | q4x2x8x1 210.5 | q4x4x8x1 211.9 | q4x3x8x1 242.1; at 128 images the tiled kernel leads (116 vs 126)
      QWJ = 8; QOCB = 2; QRING = 8; QMINW = 2; while (ksteps % QRING) --QRING;
      if (lds(1, QOCB) > 80 * 1024) QWJ = 0;
    }
    if (QWJ) {
      if (!QMINW) QMINW = (QOCB == 3) ? 2 : ((QOCB == 2) ? 3 : 4);   // registers: OCB*64 accumulators + 4*RING operands + ~30
      QMINW = (int)std::max(1l, std::min((long)QMINW, std::max(1l, (160l * 1024) / lds(1, QOCB)) * ((QWJ + 3) / 4)));
      p.stream = true; p.quad = true; p.kname = "bodahip_k1_quad_f32";
      p.cfg.BI = QOCB * 32; p.cfg.BJ = QWJ * 128; p.cfg.BK = g.C; p.cfg.WI = 1; p.cfg.WJ = QWJ; p.cfg.MINW = QMINW; p.cfg.SPLITK = 1; p.cfg.MT = 32; p.cfg.PF = QRING;
      p.defs = {"-DKC=" + std::to_string(g.C), "-DHW=" + std::to_string(g.OH * g.OW), "-DWJ=" + std::to_string(QWJ), "-DOCB=" + std::to_string(QOCB),
                "-DRING=" + std::to_string(QRING), "-DMINW=" + std::to_string(QMINW), string("-DRELU=") + (g.relu ? "1" : "0"), string("-DEDGE_OC=") + ((g.OC % (QOCB * 32)) ? "1" : "0")};
      if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
      return true;
    }
  }
  if (!spec.empty()) {
    int v[5] = {0, 0, 0, 0, 0}, n = 0; size_t i = 0;
    while (i < spec.size() && n < 5) { size_t j = spec.find('x', i); if (j == string::npos) j = spec.size(); v[n++] = atoi(spec.substr(i, j - i).c_str()); i = j + 1; }
    if (n < 4) rt_err("bad k1_stream spec '" + spec + "' (WIxWJxOCBxCB[xMINW])");
    WI = v[0]; WJ = v[1]; OCB = v[2]; CB = v[3]; MINW = v[4];
    if (WI < 1 || WJ < 1 || WI * WJ > 16 || OCB < 1 || OCB > 4 || CB < 1 || CB > 2 || regs(OCB, CB) > 256 || lds(WI, OCB) > 160 * 1024)
      unsup_err("k1_stream: unsupported configuration '" + spec + "' for this shape");
  } else {
    // automatic: only where it measures ahead of the tiled kernel (MI355X, in the layer sequence of the bench): few K steps, a long
    // pel axis, one out_chan tile (the input is streamed once): ResNet-50 res2 64->256 @56x56 B=64: 89 vs 94 us.  For NiN cccp1/2
    // (96->96 @55x55) it led by 5-12 % until the tiled kernel got its paired 256-byte stores; now the two tie (B=256: 153 + 178 vs
    // 157 + 159 us), so those layers stay on the tiled kernel.  The layout's R+W ceiling (tools/mem_pattern_probe.py: 2.8-4.2 TB/s on
    // planes that are not a multiple of 128 bytes) bounds both.
    if (g.C <= 64 && g.OC > 128 && g.OC <= 512 && g.OC % 256 == 0 && Nj >= 150000) { WI = 8; WJ = 1; OCB = g.OC / 256; CB = 2; }
    else return false;
    if (regs(OCB, CB) > 250 || lds(WI, OCB) > 80 * 1024) return false;
  }
  if (!MINW) { int const r = regs(OCB, CB); MINW = std::max(1, std::min(8, 512 / r)); int const wpw = (WI * WJ + 3) / 4; // waves per SIMD of one workgroup
               long const by_lds = std::max(1l, (160l * 1024) / lds(WI, OCB)); MINW = (int)std::max(1l, std::min((long)MINW, by_lds * wpw)); }
  p.stream = true; p.kname = "bodahip_k1_stream_f32";
  p.cfg.BI = WI * OCB * 32; p.cfg.BJ = WJ * CB * 32; p.cfg.BK = g.C; p.cfg.WI = WI; p.cfg.WJ = WJ; p.cfg.MINW = MINW; p.cfg.SPLITK = 1; p.cfg.MT = 32; p.cfg.PF = 1;
  p.defs = {"-DKC=" + std::to_string(g.C), "-DHW=" + std::to_string(g.OH * g.OW), "-DWI=" + std::to_string(WI), "-DWJ=" + std::to_string(WJ),
            "-DOCB=" + std::to_string(OCB), "-DCB=" + std::to_string(CB), "-DMINW=" + std::to_string(MINW), string("-DRELU=") + (g.relu ? "1" : "0"),
            string("-DEDGE_OC=") + ((g.OC % (WI * OCB * 32)) ? "1" : "0")};
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return true;
}

// bf16 variant (kernels/gemm_conv_bf16.hip): BK = 32, 32x32x16 MFMA only, chunked staging
static void bf16_cfg(tile_cfg_t &c, bool gather, long Mi = 0, long Nj = 0, long K = 0, int num_cus = 0, bool explicit_tile = false) {
  if (c.MT != 32) { c.MT = 32; c.BI = 64; c.BJ = 64; c.WI = 2; c.WJ = 2; }
  if (c.BK != 32 && c.BK != 64) c.BK = 32;
  c.PF = 1;
  // split-K by default for tile-starved shapes with a long K loop (fully-connected layers: AlexNet fc6 = 4096 x 256 outputs, K = 9216):
  // the bf16 path has no order-exactness to lose.  Such shapes take 128x128 tiles (the LDS reuse the bf16 MFMA rate needs) and enough
  // K slices for ~2 workgroups per CU with >= 8 K steps each.  Measured (AlexNet B=256, us): fc6 / fc7 64x64 unsplit 326 / 145,
  // 64x64 x4 169 / 90, 128x128 x8 114 / 70; mid-size layers lose (ResNet res4 1x1 1024->256, 196 tiles x 32 steps: 47 split vs 42).
  if (!explicit_tile) {
    c.SPLITK = 1;
    if (Mi > 0 && num_cus > 0 && getenv("BODAHIP_NO_BF16_SPLITK") == nullptr) {
      long const nkt = (K + c.BK - 1) / c.BK;
      long const tiles = ((Mi + c.BI - 1) / c.BI) * ((Nj + c.BJ - 1) / c.BJ), tiles128 = ((Mi + 127) / 128) * ((Nj + 127) / 128);
      if (tiles <= num_cus && nkt >= 64) {
        long s = std::min<long>(16, (2l * num_cus + tiles128 - 1) / tiles128);
        s = std::min<long>(s, nkt / 8);
        if (s >= 2) { c.BI = 128; c.BJ = 128; c.WI = 2; c.WJ = 2; c.MINW = 2; c.SPLITK = (int)s; }
      }
    }
  }
  int const nt = c.threads();
  bool ok = (c.BI % (c.WI * 32) == 0) && (c.BJ % (c.WJ * 32) == 0) && ((c.BI * c.BK / 8) % nt == 0) && ((c.BJ * c.BK / 8) % nt == 0) && nt <= 1024 &&
            (c.BI / (c.WI * 32)) * (c.BJ / (c.WJ * 32)) * 16 <= 256 && 4ull * (c.BK + 8) * (c.BI + c.BJ) <= 160 * 1024;
  if (gather) ok = ok && (c.BJ % 64 == 0) && (nt % c.BJ == 0);
  if (!ok) unsup_err("native bf16 kernel: unsupported tile configuration " + c.str());
}

static char const *const kStg64 = "64x64x16x2x2x4x1x32x2x3";   // the staging-wave kernel's 64 x 64 form (tile field 10 == 3: multiplying waves spelled out)
static plan_t plan_sgemm(uint32_t M, uint32_t N, uint32_t K, int num_cus, string const &tile, bool bf16 = false, int batch = 1, bool allow_big = true) {
  plan_t p; p.kname = bf16 ? "bodahip_sgemm_bf16" : "bodahip_sgemm_f32"; p.bf16 = bf16;
  p.cfg = choose_cfg((int)M, (int)std::min<uint64_t>((uint64_t)N * batch, 0x7fffffffull), (int)K, num_cus, false, bf16); // (a batch deals batch x the tiles)
  if (!tile.empty()) { if (!parse_tile(tile, p.cfg)) rt_err("bad sgemm_tile '" + tile + "'"); }
  if (bf16) {
    bf16_cfg(p.cfg, false, M, N, K, num_cus, !tile.empty());
    p.defs = cfg_defs(p.cfg); p.defs.push_back("-DI_MODE=0"); p.defs.push_back("-DJ_MODE=0"); p.defs.push_back("-DEPI=0");
    if (p.cfg.SPLITK > 1) p.defs.push_back("-DSPLITK=1");
    return p;
  }
  // Plain fp32 operands on the staging-wave kernel (kernels/sgemm_big_f32.hip: eight multiplying waves + four staging waves, four LDS stages; BODAHIP_SGEMM_BIG = off |
  // BKSxPF): 256 x 256 tiles where they fill the chip, and -- round 5 -- its 128 x 128 (two workgroups per CU), 256 x 128 and 128 x 256 forms.  cfg.WI x WJ = 3 x 4 stands for
  // the twelve waves; an explicit tile asks for the kernel that way ("128x128x16x3x4x2").
  if (!tile.empty() && p.cfg.SW == 3) {   // round 6: the staging-wave kernel with WI x WJ = 8 | 4 multiplying waves spelled out ("64x128x16x2x2x4x1x32x2x3"): its small forms
    tile_cfg_t const &c = p.cfg;
    int const nmw = c.WI * c.WJ, ti = (c.WI > 0 && c.BI % (c.WI * 32) == 0) ? c.BI / (c.WI * 32) : 0, tj = (c.WJ > 0 && c.BJ % (c.WJ * 32) == 0) ? c.BJ / (c.WJ * 32) : 0;
    bool const ok = (nmw == 8 || nmw == 4) && (ti == 4 || ti == 2 || ti == 1) && (tj == 2 || tj == 1) && c.BK >= 4 && c.BK <= 32 && c.BK % 4 == 0 && (c.BK * (c.BI / 4)) % 256 == 0 &&
                    (c.BK * (c.BJ / 4)) % 256 == 0 && (c.PF == 2 || c.PF == 4) && c.MT == 32 && c.SPLITK == 1 && c.MINW >= 1 && c.MINW <= 8 && 4l * 4 * c.BK * (c.BI + c.BJ + 8) <= 160 * 1024;
    if (!ok || batch != 1 || M % 4 || N % 4 || !allow_big) unsup_err("hip_sgemm: unsupported staging-wave tile " + c.str() + " (8 | 4 multiplying waves of 4 | 2 | 1 x 2 | 1 blocks, whole float4 units per staging thread, M and N multiples of 4)");
    p.big = true; p.kname = "bodahip_sgemm_big_f32"; p.cfg.KHO = 0;
    p.defs = {"-DBKS=" + std::to_string(c.BK), "-DPF=" + std::to_string(c.PF), "-DTBI=" + std::to_string(c.BI), "-DTBJ=" + std::to_string(c.BJ), "-DWI=" + std::to_string(c.WI),
              "-DWJ=" + std::to_string(c.WJ), "-DMINW=" + std::to_string(c.MINW)};
    if (char const *x = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(x); string tok; while (is >> tok) p.defs.push_back(tok); }
    return p;
  }
  bool const big_tile = (p.cfg.BI == 256 || p.cfg.BI == 128) && (p.cfg.BJ == 256 || p.cfg.BJ == 128);
  bool const want_big = big_tile && ((p.cfg.WI == 3 && p.cfg.WJ == 4) || (p.cfg.BI == 256 && p.cfg.BJ == 256));
  if (allow_big && batch == 1 && want_big && p.cfg.MT == 32 && p.cfg.SPLITK == 1 && M % 4 == 0 && N % 4 == 0) {
    char const *e = getenv("BODAHIP_SGEMM_BIG");
    if (!(e && string(e) == "off")) {
      int bks = 8, pf = 2;   // measured (MI355X, 12288^3 / 8192^3 / 6144^3, TF/s): gemm_conv_f32.hip on the same tile 140.2 / 140.4 / 133.6; 16x2 144.7 / 144.5 / 129.0; 8x2 145.1 / 144.8 / 137.7; 8x4 145.1 / 144.8 / 135.4; 16x4 142.0 / 142.2 / 136.5 (four LDS stages)
      if (!tile.empty() && p.cfg.WI == 3 && p.cfg.BK >= 4 && p.cfg.BK <= 32 && p.cfg.BK % 4 == 0) bks = p.cfg.BK;   // (asked for by its own tile string: the K step too)
      if (e && *e) { if (sscanf(e, "%dx%d", &bks, &pf) != 2 || bks < 4 || bks > 32 || bks % 4 || (pf != 2 && pf != 4)) rt_err(string("bad BODAHIP_SGEMM_BIG '") + e + "' (off | BKSxPF)"); }
      if (bks * (p.cfg.BI / 4) % 256 || bks * (p.cfg.BJ / 4) % 256) bks = 8;   // (whole float4 units per staging thread: BKS x TB / 4 a multiple of 256)
      int const minw = (p.cfg.BI == 128 && p.cfg.BJ == 128) ? ((!tile.empty() && p.cfg.MINW >= 1) ? std::min(p.cfg.MINW, 2) : 2) : 1;
      p.big = true; p.kname = "bodahip_sgemm_big_f32"; p.cfg.BK = bks; p.cfg.PF = pf; p.cfg.WI = 3; p.cfg.WJ = 4; p.cfg.MINW = minw;
      p.defs = {"-DBKS=" + std::to_string(bks), "-DPF=" + std::to_string(pf)};
      if (!(p.cfg.BI == 256 && p.cfg.BJ == 256)) {   // (the 256 x 256 form keeps its option string: its code objects stay the cached ones)
        bool const tall = (p.cfg.BI == 256 && p.cfg.BJ == 128);    // 256 x 128: 4 x 2 multiplying waves of 64 x 64; the others 2 x 4 (128 x 128: 64 x 32; 128 x 256: 64 x 64)
        p.defs.push_back("-DTBI=" + std::to_string(p.cfg.BI)); p.defs.push_back("-DTBJ=" + std::to_string(p.cfg.BJ));
        p.defs.push_back(string("-DWI=") + (tall ? "4" : "2")); p.defs.push_back(string("-DWJ=") + (tall ? "2" : "4")); p.defs.push_back("-DMINW=" + std::to_string(minw));
      }
      if (char const *x = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(x); string tok; while (is >> tok) p.defs.push_back(tok); }
      return p;
    }
  }
  // Round 6: where the general kernel would run 64 x 64 tiles (the sizes that give 256 CUs less than two 128 x 128 tiles each: 768^3 .. 3072^3 of sgemm-ops-full), the
  // staging-wave kernel's 64 x 64 form -- four multiplying waves of one 32 x 32 block, four staging waves, four workgroups per CU -- runs ahead of it: in the layer sequence
  // of the list (tools/sgemm_stg_ab.sh, TF/s) 1024^3 75 -> 87, 1536^3 88 -> 93, 2048^3 123.5 -> 127.4, 3072^3 128.1 -> 131.5; its 64 x 128 / 128 x 128 forms measured
  // level at 2048^3 and behind below.  Bit-identical (the same ascending-k chain per output).
  if (tile.empty() && allow_big && batch == 1 && p.cfg.BI == 64 && p.cfg.BJ == 64 && p.cfg.MT == 32 && p.cfg.SPLITK == 1 && M % 4 == 0 && N % 4 == 0 && K >= 512 && getenv("BODAHIP_NO_SGEMM_STG64") == nullptr) {
    char const *e = getenv("BODAHIP_SGEMM_BIG");
    if (!(e && string(e) == "off")) return plan_sgemm(M, N, K, num_cus, kStg64, false, 1, true);
  }
  p.cfg.KHO = 0;   // (K hand-off is a convolution form: the sgemm launches are plain grids)
  check_cfg(p.cfg, false);
  p.defs = cfg_defs(p.cfg);
  p.defs.push_back(string("-DI_MODE=") + ((M % 4 == 0) ? "0" : "1"));
  p.defs.push_back(string("-DJ_MODE=") + ((N % 4 == 0) ? "0" : "1"));
  p.defs.push_back("-DEPI=0");
  if (p.cfg.SPLITK > 1) p.defs.push_back("-DSPLITK=1");
  return p;
}
// bf16 convolution from a channel-innermost LDS patch (kernels/conv_patch_bf16.hip): KH x KW kernels, stride 1 in x, in_chan % 8 == 0
static bool plan_patch_bf16(conv_geom_t const &g, int num_cus, plan_t &p) {
  if (getenv("BODAHIP_NO_PATCH16")) return false;
  int const taps = g.KH * g.KW;
  if (!(g.SX == 1 && taps >= 2 && g.KH >= g.SY && g.C % 8 == 0 && g.C >= 16)) return false;
  if (g.OH == 1 && g.OW == 1 && g.PY == 0 && g.PX == 0 && g.KH == g.H && g.KW == g.W) return false; // ("ipconv" shapes stay on the j-major path)
  int cg = 1; while (cg * taps < 16) ++cg;                                    // k-slots per K step: >= 16 (an odd count is padded with one zero slot)
  if (char const *e = getenv("BODAHIP_PATCH16_CG")) { if (atoi(e) > 0) cg = atoi(e); }   // (experiments)
  if (cg > 8) return false;
  int const np = cg * taps + ((cg * taps) & 1);
  long const Nj = (long)g.B * g.OH * g.OW;
  int const wp = g.W + 2 * g.PX;
  auto lds = [&](int bi, int bj) {
    int const rows_max = (bj - 2) / g.OW + 2, seg_max0 = (g.OH - 1 + rows_max - 1) / g.OH + 1, seg_max = std::min(seg_max0, rows_max);
    long const cs = (long)((rows_max - seg_max) * g.SY + seg_max * g.KH) * wp;
    return 16l * ((long)np * bi + cg * cs);
  };
  struct cand_t { int bi, bj, wi, wj; };
  static cand_t const cands[] = {{128, 128, 2, 2}, {96, 128, 1, 4}, {64, 128, 1, 4}, {64, 64, 2, 2}};
  int pick = -1;
  for (int ci = 0; ci < 4; ++ci) {
    if (cands[ci].bi == 96 && !(g.OC % 128 > 64 && g.OC % 128 <= 96)) continue;   // 96-row tiles only where they remove padding (out_chan = 96, 224, ...)
    if (cands[ci].bi == 128 && g.OC % 128 > 64 && g.OC % 128 <= 96 && g.OC < 256) continue;
    if (cands[ci].bi > 64 && g.OC <= 64) continue;                                       // (no rows of padding for thin layers)
    cand_t const &c = cands[ci];
    if (lds(c.bi, c.bj) > 64 * 1024) continue;
    long const tiles = (long)((g.OC + c.bi - 1) / c.bi) * ((Nj + c.bj - 1) / c.bj);
    if (pick < 0) pick = ci;
    if (tiles >= (long)num_cus * 3 / 2) { pick = ci; break; }                 // the largest tile that still gives every CU work
    pick = ci;
  }
  if (pick < 0) return false;
  cand_t const &c = cands[pick];
  p.patch16 = true; p.bf16 = true; p.cg = cg; p.kname = "bodahip_conv_patch_bf16";
  p.cfg.BI = c.bi; p.cfg.BJ = c.bj; p.cfg.BK = cg * 8 * taps; p.cfg.WI = c.wi; p.cfg.WJ = c.wj; p.cfg.MINW = 2; p.cfg.SPLITK = 1; p.cfg.MT = 32; p.cfg.PF = 1;
  p.defs = {"-DBI=" + std::to_string(c.bi), "-DBJ=" + std::to_string(c.bj), "-DWI=" + std::to_string(c.wi), "-DWJ=" + std::to_string(c.wj), "-DMINW=2",
            "-DCG=" + std::to_string(cg), "-DKH=" + std::to_string(g.KH), "-DKW=" + std::to_string(g.KW), "-DSY=" + std::to_string(g.SY),
            "-DPY=" + std::to_string(g.PY), "-DPX=" + std::to_string(g.PX), "-DCH=" + std::to_string(g.H), "-DCW=" + std::to_string(g.W),
            "-DCOH=" + std::to_string(g.OH), "-DCOW=" + std::to_string(g.OW), string("-DRELU=") + (g.relu ? "1" : "0")};
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return true;
}

// Channels-last bf16 convolution (kernels/conv_nhwc_bf16.hip): implicit GEMM D[oc][pel], operands straight from HBM into LDS
// (buffer_load ... lds), 32x32x16 bf16 MFMA.  g.C is the STORED channel count (a multiple of 8).  tile: "BIxBJxBKxWIxWJ[xMINW]" or "".
// grp_pad > 0: horizontally fused convolutions (-DGROUPS=1): g.OC is the stacked, padded out_chan count, every member starts at a multiple of grp_pad -> tiles
// may not be taller than grp_pad and must divide it; no K slices.
// In-launch K slices (KSL of kernels/conv_nhwc_bf16.hip, conv_nhwc_patch_bf16.hip): the count is a compile-time constant of the kernel, every slice runs
// ceil(nk / slices) K steps -- so the count is lowered until no slice is empty (32 at most).
static int ksl_normalise(int want, long nk) {
  int s = (int)std::max<long>(1, std::min<long>(std::min(want, 32), nk));
  while (s > 1) { long const per = (nk + s - 1) / s; if ((nk + per - 1) / per == s) break; --s; }
  return s;
}
static plan_t plan_conv_nhwc(conv_geom_t const &g, int num_cus, string const &tile, bool out_f32, int grp_pad = 0, bool allow_split = true) {
  if (g.C % 8) unsup_err("hip_conv_nhwc: in_chan of a channels-last bf16 tensor must be a multiple of 8 (the layout pass pads)");
  if (g.H >= 32768 || g.W >= 32768) unsup_err("hip_conv_nhwc: planes of 32768 rows / columns or more are not supported");
  long const Nj = (long)g.B * g.OH * g.OW;
  int const cg = g.C / 8, kc = cg * g.KH * g.KW;
  plan_t p; p.nhwc = true; p.bf16 = true; p.kname = "bodahip_conv_nhwc_bf16";
  tile_cfg_t c; c.MT = 32; c.SPLITK = 1; c.PF = 1;
  // K step: 64 (8 chunks) when a tap's chunks divide into it -- or when K is long anyway; 32 otherwise
  // Round 5 (tools/ksl_sweep.py, every 1x1 layer of GoogLeNet at 64 images under ten tiles): on SMALL maps (14 x 14 and 7 x 7 at 64 images: 3-12 k pels, 100-600
  // tiles) the launches are latency-bound, not HBM-bound, and the 64-deep step with half as many barriers wins by 15-25 % (512 -> 128 channels at 14 x 14: 64x128x32
  // ring 4 10.5 us, 64x64x64 ring 3 7.9; 480 -> 96: 10.3 -> 8.6), ragged taps included (480 channels = 7.5 steps); the 32-deep rules stay for the large maps.
  bool const big_map = Nj >= 32768;
  c.BK = (cg % 8 == 0) ? 64 : ((cg % 4 == 0 && big_map) ? 32 : (kc >= 32 ? 64 : 32));
  // short K (<= 512) on large maps: a 32-deep step -- half the LDS per ring slot, so a deeper ring and more workgroups per CU for what are HBM-bound
  // launches (measured at 64 images: 1x1 layers with 64-512 input channels at 28 x 28 / 56 x 56 5-10 % faster, the 7x7 / 2 conv1 on 8 stored channels 93 -> 75 us;
  // from 1024 channels up and on 3x3 layers the 64-deep step wins)
  if (kc <= 64 && (big_map || kc <= 32)) c.BK = 32;
  int nbuf = 0;   // LDS ring depth (the tile string's 9th field; 0 = choose below)
  if (!tile.empty()) {
    if (!parse_tile(tile, c)) rt_err("bad conv_tile '" + tile + "'");
    { int nf = 1; for (char ch : tile) if (ch == 'x' || ch == ':') ++nf; if (nf >= 9) nbuf = c.PF; }
    c.MT = 32; c.PF = 1;
    if (c.SPLITK < 1 || c.SPLITK > 64) unsup_err("hip_conv_nhwc: unsupported K split " + std::to_string(c.SPLITK));
    if (grp_pad && (c.BI > grp_pad || grp_pad % c.BI)) unsup_err("hip_conv_nhwc_grp: tile " + c.str() + " does not fit the members' padding of " + std::to_string(grp_pad) + " out_chans");
  } else {
    // score = base rate of the tile x fraction of the padded tile grid that is real work x how evenly the tiles deal out over the CUs
    // (the rule of choose_cfg); base rates are first MI355X measurements of this kernel relative to 128x128
    // Tile and K split by a small time model (us), calibrated on MI355X (tools/nhwc_sweep.py, ResNet-50 / GoogLeNet at 64 images):
    //   * the K loop is bound by the L2 -> LDS operand stream, not by the MFMAs: one K step of a workgroup costs ~0.7 us per 32 KB of
    //     operand tiles (128x128x64), workgroups that share a CU share that rate -> t_main = ceil(wgs / CUs) * steps * 0.7 * (BI + BJ) * BK / 16384;
    //   * ~4 us per launch of ramp-up, prologue and epilogue;
    //   * K slices (tile-starved layers with a long K loop: 7x7-map layers, fully-connected layers; this path has no summation order to
    //     keep) cost a second launch (~3 us) and the fp32 partial tiles written and read once each at ~5 TB/s.
    struct cand_t { int bi, bj, wi, wj, minw; };
    //   * large tiles (8 / 16 waves) move fewer operand bytes per flop but then meet the matrix pipes: a K step is never faster than its flops at
    //     ~45 % of the CU's bf16 MFMA rate (AlexNet / NiN conv2, 5x5 96->256 at 256 images: 128x128 305, 128x256 254, 256x256 243 us).
    //   * (round 5) the two-wave 32 x 64 tile is gone from the list: never ahead of 64 x 64 x 64 / 32 x 128 x 64 in the sweep of GoogLeNet's 1x1 layers (16-32 out_chans at
    //     14 x 14: 8.2-8.5 us against 6.6-7.4), and a 128-thread member cannot share a level set's wrapper kernel
    static cand_t const cands[] = {{128, 128, 2, 2, 2}, {64, 128, 1, 4, 2}, {64, 64, 2, 2, 2}, {32, 128, 1, 4, 2}, {128, 256, 2, 4, 1}, {256, 256, 4, 4, 1}};
    long const nk = (kc + c.BK / 8 - 1) / (c.BK / 8);
    bool const two_kernel = getenv("BODAHIP_NHWC_SPLITK2") != nullptr && !grp_pad;
    // (no slices for a fused sibling group: its outputs stay bit-identical to its members' own launches, which may slice differently or not at all)
    bool const may_split = getenv("BODAHIP_NO_NHWC_SPLITK") == nullptr && allow_split && !grp_pad;
    double best = 1e30;
    for (cand_t const &cd : cands) {
      if (grp_pad && (cd.bi > grp_pad || grp_pad % cd.bi)) continue;
      long const ti = (g.OC + cd.bi - 1) / cd.bi, tj = (Nj + cd.bj - 1) / cd.bj, tiles = ti * tj;
      if (cd.bi * cd.bj > 128 * 128 && ((long)(cd.bi + cd.bj) * c.BK * 2 * 2 > 140 * 1024 || !getenv("BODAHIP_NHWC_BIG_TILES"))) continue;   // 128x256 / 256x256 at one workgroup per CU:
      // opt-in.  Measured (MI355X, same box, A/B): NiN whole net +2 %, ResNet-50 / GoogLeNet lists and AlexNet net within noise, single layers both ways --
      // the wider tile halves the operand re-reads per MFMA but leaves one workgroup per CU with nothing to hide its barriers behind.
      double const tau = std::max(0.7 * (double)(cd.bi + cd.bj) * c.BK / 16384.0, 2.0 * cd.bi * cd.bj * c.BK / (0.45 * 2.5e9 / num_cus * 1e3));
      for (int sk = 1; sk <= 16; sk *= 2) {
        if (sk > 1 && (!may_split || nk / sk < 4)) break;
        long const wgs = tiles * sk, steps = (nk + sk - 1) / sk;
        // (two workgroups share a CU without slowing each other at these sizes -- the launches are latency-bound: 392 tiles of 64 x 64 run like 196 --, so a round is 2 x CUs)
        double t = (double)((wgs + 2 * num_cus - 1) / (2 * num_cus)) * (double)steps * tau + 4.0;
        // K slices reduced inside the launch (KSL): no second launch, but the slabs leave write-through and are read back by the last arriver -- measured (tools/ksl_sweep.py,
        // GoogLeNet's 14 x 14 / 7 x 7 layers at 64 images): ~1.5 us + ~1 us per MB of slabs (slices x out_chans x pels x 4 bytes), whatever the tile
        if (sk > 1) t += two_kernel ? (3.0 + 2.0 * sk * (double)Nj * g.OC * 4.0 / 5e6) : (1.5 + sk * (double)Nj * g.OC * 4.0 / 1e6);
        if (t < best) { best = t; c.BI = cd.bi; c.BJ = cd.bj; c.WI = cd.wi; c.WJ = cd.wj; c.MINW = cd.minw; c.SPLITK = sk; }
      }
    }
  }
  int const cpr = c.BK / 8, nt = c.threads();
  {
    // ring depth: 3 (one K step of loads in flight across each barrier) where every wave issues the same number of loads per step and two
    // workgroups still fit a CU's LDS; 2 otherwise
    bool const even = c.WI > 0 && c.WJ > 0 && ((c.BI * cpr / 64) % (c.WI * c.WJ) == 0) && ((c.BJ * cpr / 64) % (c.WI * c.WJ) == 0);
    long const per_buf = (long)(c.BI + c.BJ) * c.BK * 2;
    // (64-deep steps: a third slot only for the small tiles -- 64 x 128 x 64 and 128 x 128 x 64 measured level or FASTER on a ring of two at 28 x 28 / 14 x 14 / 7 x 7:
    //  832 -> 384 at 7 x 7 11.3 -> 9.4 us, 192 -> 96 at 28 x 28 on 128 x 128 12.9 -> 10.4)
    if (!nbuf) nbuf = (even && c.BK == 32 && 4 * per_buf <= 48 * 1024) ? 4 : ((even && 3 * per_buf <= 80 * 1024 && (c.BK == 32 || per_buf <= 20 * 1024)) ? 3 : 2);
    if (nbuf < 2 || nbuf > 4 || (nbuf > 2 && !even)) unsup_err("hip_conv_nhwc: unsupported LDS ring depth " + std::to_string(nbuf) + " for tile " + c.str());
    c.PF = nbuf;   // (reported as _pN in the launch info)
  }
  bool ok = (c.BK == 32 || c.BK == 64) && c.BI > 0 && c.BJ > 0 && c.WI > 0 && c.WJ > 0 && nt <= 1024 && (c.BI % (c.WI * 32) == 0) && (c.BJ % (c.WJ * 32) == 0) &&
            ((c.BI * cpr) % 64 == 0) && ((c.BJ * cpr) % 64 == 0) && (c.BI / (c.WI * 32)) * (c.BJ / (c.WJ * 32)) * 16 <= 256 && c.MINW >= 1;
  long const lds = std::max<long>((long)nbuf * (c.BI + c.BJ) * c.BK * 2, out_f32 ? 0 : (long)c.BJ * (c.BI * 2 + 16));
  ok = ok && lds <= 160 * 1024;
  if (!ok) unsup_err("hip_conv_nhwc: unsupported tile configuration " + c.str());
  p.cfg = c;
  p.defs = {"-DBI=" + std::to_string(c.BI), "-DBJ=" + std::to_string(c.BJ), "-DBK=" + std::to_string(c.BK), "-DWI=" + std::to_string(c.WI), "-DWJ=" + std::to_string(c.WJ),
            "-DMINW=" + std::to_string(c.MINW), "-DCIN=" + std::to_string(g.C), "-DKH=" + std::to_string(g.KH), "-DKW=" + std::to_string(g.KW),
            "-DSY=" + std::to_string(g.SY), "-DSX=" + std::to_string(g.SX), "-DPY=" + std::to_string(g.PY), "-DPX=" + std::to_string(g.PX),
            "-DCH=" + std::to_string(g.H), "-DCW=" + std::to_string(g.W), "-DCOH=" + std::to_string(g.OH), "-DCOW=" + std::to_string(g.OW),
            string("-DRELU=") + (g.relu ? "1" : "0"), string("-DOUT_F32=") + (out_f32 ? "1" : "0"), "-DNBUF=" + std::to_string(nbuf)};
  if (c.SPLITK > 1) {   // K slices: reduced inside the launch (KSL, round 5) unless the two-kernel form is asked for (BODAHIP_NHWC_SPLITK2=1: slabs in the shared scratch + bodahip_nhwc_splitk_reduce)
    long const nk2 = ((long)kc + c.BK / 8 - 1) / (c.BK / 8);
    if (getenv("BODAHIP_NHWC_SPLITK2") && !grp_pad) p.defs.push_back("-DSPLITK=1");
    else { c.SPLITK = ksl_normalise(c.SPLITK, nk2); p.cfg = c; if (c.SPLITK > 1) { p.defs.push_back("-DKSL=" + std::to_string(c.SPLITK)); p.ksl = true; } }
  }
  if (grp_pad) p.defs.push_back("-DGROUPS=1");
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return p;
}
// Channels-last bf16 convolution from an LDS input patch (kernels/conv_nhwc_patch_bf16.hip): KH x KW kernels with more than one tap, stride 1 in x; filters in the
// F'[in_grp][ky][kx][out_chan][8] form.  A K step is CG groups of 8 channels x all taps.  tile: "BIxBJx0xWIxWJ[xMINW]" or "".
// pool: g.KH x g.KW / g.PY, g.PX describe a MAX-POOLING window fused in front of a 1x1 convolution (-DPOOL=1: the filters hold one k-slot per channel group).
// channel groups per K step of the input-patch forms (shared by the patch kernel and the rolling-rows kernel: same k-slot order, same MFMA chain, same bits)
static int patch_cg(int ncg, int taps) {
  int cg = 1; long best = -1;
  for (int c = std::min(ncg, 4); c >= 1; --c) {
    if (c > 1 && c * taps > 50) continue;
    long const slots = (long)((ncg + c - 1) / c) * (c * taps + ((c * taps) & 1));
    if (best < 0 || slots < best) { best = slots; cg = c; }
  }
  return cg;
}
static plan_t plan_conv_nhwc_patch(conv_geom_t const &g, int num_cus, string const &tile_arg, bool out_f32, bool pool = false) {
  if (g.C % 8) unsup_err("hip_conv_nhwc: in_chan of a channels-last bf16 tensor must be a multiple of 8");
  string tile = tile_arg;
  if (pool && tile.empty()) { if (char const *e = getenv("BODAHIP_NHWC_POOL_TILE")) tile = e; }   // (experiments: the tile of the fused-pooling form)
  int const taps = g.KH * g.KW, ncg = g.C / 8;
  if (!(g.SX == 1 && taps >= 2 && g.KH >= g.SY)) unsup_err("hip_conv_nhwc (patch form of filts): needs stride 1 in x and more than one tap");
  long const Nj = (long)g.B * g.OH * g.OW;
  // ADIRECT (default): filter fragments straight from global memory, the LDS holds the (double-buffered) patch only.  BODAHIP_NHWC_ADIRECT=0: both operands staged.
  bool adirect = true; if (char const *e = getenv("BODAHIP_NHWC_ADIRECT")) adirect = atoi(e) != 0;
  // Channel groups per K step: at most 4, at most 50 k-slots, the count that wastes the fewest zero k-slots over the layer (a ragged last step and the zero slot
  // of an odd step are MFMAs on zeros: 6 groups of a 3x3 as 4 + 2 cost 72 slots, as 2 + 2 + 2 54 -- AlexNet's space-to-depth conv1 at 256 images 128.7 -> 111.9 us;
  // 5x5 on 12 groups as 6 x 50 instead of 12 x 26: 209 -> 200 us); ties go to the larger step (fewer barriers).
  int cg = patch_cg(ncg, taps);
  if (pool) { if (!adirect) unsup_err("hip_conv_nhwc (fused pooling): needs the direct filter path"); cg = std::min(ncg, 4);    // (one k-slot per group: four groups = two MFMA k-iterations per step)
    if (char const *e = getenv("BODAHIP_NHWC_POOL_CG")) { if (atoi(e) > 0) cg = std::min(ncg, atoi(e)); } }
  if (char const *e = getenv("BODAHIP_NHWC_PATCH_CG")) { if (atoi(e) > 0) cg = std::min(ncg, atoi(e)); }   // (experiments)
  int wp = g.W + 2 * g.PX;                                              // slot pitch: as the kernel's wpitch()
  for (int p2 = wp; p2 < wp + 16; ++p2) if ((g.SY * p2 - g.OW) % 16 == 0) { wp = p2; break; }
  auto lds_cg = [&](int bi, int bj, int cgx) {
    int const npx = cgx * taps + ((cgx * taps) & 1);
    int const rows_max = (bj - 2) / g.OW + 2, seg_max0 = (g.OH - 1 + rows_max - 1) / g.OH + 1, seg_max = std::min(seg_max0, rows_max);
    long const cs = (long)((rows_max - seg_max) * g.SY + seg_max * g.KH) * wp, csp = cs + ((2 - cs % 16) + 16) % 16;
    long const ops = adirect ? 2l * 16l * cgx * csp : 16l * ((long)npx * bi + cgx * csp);
    return std::max<long>(ops, out_f32 ? 0 : (long)bj * (bi * 2 + 16));
  };
  auto lds = [&](int bi, int bj) { return lds_cg(bi, bj, cg); };
  struct cand_t { int bi, bj, wi, wj, minw, pf; };
  static cand_t const cands_staged[] = {{64, 256, 1, 4, 2, 0}, {64, 128, 1, 4, 2, 0}, {32, 256, 1, 4, 2, 0}, {128, 128, 2, 2, 2, 0}, {32, 128, 1, 4, 2, 0}, {64, 64, 2, 2, 2, 0}};
  // ADIRECT: wave tiles wide in pels first (32 x 128: one 1-KB filter fragment load per four MFMAs), in order of preference on equal cost; last the 64 x 128 wave
  // tile (half the operand bytes per MFMA; 247 registers with four fragments in flight: still two waves per SIMD) for layers with tiles to spare
  static cand_t const cands_direct[] = {{128, 128, 4, 1, 2, 0}, {64, 256, 2, 2, 2, 0}, {64, 128, 2, 2, 2, 0}, {128, 64, 4, 1, 2, 0}, {32, 128, 1, 4, 2, 0}, {128, 256, 2, 2, 2, 4}, {256, 128, 4, 1, 2, 4}};
  cand_t const *const cands = adirect ? cands_direct : cands_staged;
  int const n_cands = adirect ? (int)(sizeof(cands_direct) / sizeof(cand_t)) : (int)(sizeof(cands_staged) / sizeof(cand_t));
  plan_t p; p.nhwc = true; p.nhwc_patch = true; p.bf16 = true; p.kname = "bodahip_conv_nhwc_patch_bf16";
  tile_cfg_t c; c.MT = 32; c.SPLITK = 1; c.PF = 1;
  int pick_pf = 0;
  if (!tile.empty()) {
    if (!parse_tile(tile, c)) rt_err("bad conv_tile '" + tile + "'");
    c.MT = 32; c.PF = 1;
    if (c.SPLITK < 1 || c.SPLITK > 32 || (c.SPLITK > 1 && !adirect)) unsup_err("hip_conv_nhwc (patch form of filts): unsupported K slices " + std::to_string(c.SPLITK));
  } else if (pool && getenv("BODAHIP_NHWC_POOL_R4PLAN") == nullptr) {
    // The fused-pooling form is bound by its LDS reads: a B fragment is KH x KW patch reads + maxima, and a wave forms it for every pel block of its tile -- so waves
    // must not SHARE pels (WI = 1: the 4 x 1 and 2 x 2 layouts redo the window maxima four / two times per tile) and the wave tile is one pel block wide.  Round 5,
    // GoogLeNet's nine pool projections at 64 images, us per launch (tools/pool_tile_ab.sh; round-4 plan | 64x128 as 1 x 4 waves): 24.5 | 15.1 (256 -> 64 at 28 x 28),
    // 26.0 | 17.1 (528 -> 128 at 14 x 14); level elsewhere.  Tile-starved layers with a long K (832 -> 128 at 7 x 7: 50 tiles, 26 steps -- every workgroup pulls 400 KB
    // through ONE CU's load path) take 64 x 64 tiles and four K slices reduced inside the launch: 27.5 -> 15.6 us.
    c.BI = (g.OC <= 32) ? 32 : 64; c.BJ = 128; c.WI = 1; c.WJ = 4; c.MINW = 2;
    long const tiles = (long)((g.OC + c.BI - 1) / c.BI) * ((Nj + c.BJ - 1) / c.BJ), steps = (ncg + 3) / 4;
    if (tiles * 4 < num_cus && steps >= 16 && getenv("BODAHIP_NO_NHWC_SPLITK") == nullptr) { c.BI = 64; c.BJ = 64; c.WI = 2; c.WJ = 2; c.SPLITK = 4; }
    cg = std::min(ncg, 4);
    while (cg > 1 && lds_cg(c.BI, c.BJ, cg) > 80 * 1024) cg = (cg + 1) / 2;
    if (cg < 4) { cg = std::min(ncg, 4); while (cg > 1 && lds_cg(c.BI, c.BJ, cg) > 160 * 1024) cg = (cg + 1) / 2; }   // (narrow maps with a padded pitch: one workgroup per CU rather than twice the steps)
  } else if (!adirect) {
    // Narrow in out_chan, wide in pels: the filter tile -- the larger operand stream here -- is staged once per BJ pels.  score = padding efficiency x share of
    // the CUs that get a workgroup / operand bytes per flop (filter stream ~ 1/BJ, patch stream ~ 1/(4 BI)).  Measured on MI355X (tools/patch_sweep.sh, 64
    // images, us incl. the ~6 us launch floor): ResNet-50 3x3 at 56^2 / 28^2 / 14^2 / 7^2: 64x256 30 / 26 / 27 / 45, 64x128 34 / 29 / 27 / 32.5, 32x128 35 / 29 /
    // 30 / 34; GoogLeNet 3x3 64->192 at 56^2: 64x256 63-66, 32x128 91.  Two workgroups per CU must fit the LDS (80 KB each): wide planes take 2 channel groups
    // per K step instead of 4 (level with each other where both fit; 8 groups measured 10-50 % slower).
    int pick = -1, pick_cg = cg; double best = -1;
    for (long lim = 80 * 1024; pick < 0 && lim <= 160 * 1024; lim *= 2)
    for (int ci = 0; ci < n_cands; ++ci) {
      cand_t const &cd = cands[ci];
      int cgx = cg; while (cgx > 1 && lds_cg(cd.bi, cd.bj, cgx) > lim) cgx = (cgx + 1) / 2;
      if (lds_cg(cd.bi, cd.bj, cgx) > lim) continue;
      if (cd.bi > 64 && g.OC <= 64) continue;
      long const ti = (g.OC + cd.bi - 1) / cd.bi, tj = (Nj + cd.bj - 1) / cd.bj, tiles = ti * tj;
      double const pad = ((double)g.OC / (double)(ti * cd.bi)) * ((double)Nj / (double)(tj * cd.bj));
      double const fill = std::min(1.0, (double)tiles / (double)num_cus);
      double const bytes_per_flop = 1.0 / cd.bj + 0.25 / cd.bi;
      double const score = pad * fill / bytes_per_flop;
      if (score > best) { best = score; pick = ci; pick_cg = cgx; }
    }
    if (pick < 0) unsup_err("hip_conv_nhwc (patch form of filts): no tile fits the LDS for this plane width");
    c.BI = cands[pick].bi; c.BJ = cands[pick].bj; c.WI = cands[pick].wi; c.WJ = cands[pick].wj; c.MINW = cands[pick].minw; cg = pick_cg;
  } else {
    // What a launch costs here (tools/adirect_ablate.sh, ResNet-50 256 -> 256 at 14^2, 64 images, 128 x 128 tiles: 20.6 us = 7.3 without the K loop + 7.7 of MFMA
    // issue + 5.3 of operand loads, 1.5 us of which overlap): the MFMA work of the busiest SIMD -- rounds of workgroups over the CUs x the wave tile -- inflated by
    // the filter fragments its waves pull through the CU's 64 B/clk L1 path per MFMA (1 / pel blocks of the wave tile).  tiles <= CUs: one round; <= 2 CUs: the CUs
    // that hold two workgroups set the pace (1.7: two waves per SIMD overlap better than one); beyond that workgroups are handed out as CUs free up.  Measured on
    // MI355X (tools/adirect_sweep2.sh, 64 images, us incl. the ~6 us launch floor, staged -> direct): ResNet-50 3x3 at 28^2 / 14^2 / 7^2 25.4 -> 21.3 / 26.8 -> 21.2 /
    // 32.3 -> 24.4-25.1; GoogLeNet 96->208 / 128->256 / 160->320 at 14^2 14.7 -> 11.8 / 17.1 -> 13.8 / 20.7 -> 16.7, 64->192 at 56^2 64.6 -> 60.5; 4 channel groups
    // per K step (8: 5-100 % slower, 2: level or 10 % slower); 4 / 8 / 12 fragments in flight: level.
    int pick = -1, pick_cg = cg; double best = 1e30;
    for (long lim = 80 * 1024; pick < 0 && lim <= 160 * 1024; lim *= 2)      // two workgroups per CU; one where nothing fits that (maps one or two positions wide)
    for (int ci = 0; ci < n_cands; ++ci) {
      cand_t const &cd = cands[ci];
      int cgx = cg; while (cgx > 1 && lds_cg(cd.bi, cd.bj, cgx) > lim) cgx = (cgx + 1) / 2;
      if (lds_cg(cd.bi, cd.bj, cgx) > lim) continue;
      long const ti = (g.OC + cd.bi - 1) / cd.bi, tj = (Nj + cd.bj - 1) / cd.bj, tiles = ti * tj;
      double const rounds = (tiles <= num_cus) ? 1.0 : std::max(1.7, (double)tiles / (double)num_cus + 0.25);
      int const ktj = cd.bj / (cd.wj * 32), kti = cd.bi / (cd.wi * 32);
      // (fitted to the sweeps; the 64 x 128 wave tile, as 128 x 256 or 256 x 128 workgroup tiles: AlexNet conv2 at 256 images, 1458 tiles, 201 -> 181 us; the
      //  space-to-depth conv1, 3025 tiles, 110.6 -> 99.4; conv5, 338 tiles, 72 -> 79-88)
      double const cost = rounds * (double)(cd.bi * cd.bj) / (double)(cd.wi * cd.wj) * ((kti >= 2 && ktj >= 4) ? 1.02 : (ktj >= 4) ? 1.1 : (ktj >= 2) ? 1.3 : 2.0);
      if (cost < best * 0.97) { best = cost; pick = ci; pick_cg = cgx; pick_pf = cd.pf; }   // (a later candidate must be clearly cheaper)
    }
    if (pick < 0) unsup_err("hip_conv_nhwc (patch form of filts): no tile fits the LDS for this plane width");
    c.BI = cands[pick].bi; c.BJ = cands[pick].bj; c.WI = cands[pick].wi; c.WJ = cands[pick].wj; c.MINW = cands[pick].minw; cg = pick_cg;
  }
  while (cg > 1 && lds(c.BI, c.BJ) > 160 * 1024) cg = (cg + 1) / 2;
  p.cg = cg; c.BK = pool ? cg * 8 : cg * 8 * taps;
  bool ok = c.BI > 0 && c.BJ > 0 && c.WI > 0 && c.WJ > 0 && c.threads() <= 1024 && (c.BI % (c.WI * 32) == 0) && (c.BJ % (c.WJ * 32) == 0) &&
            (c.BI / (c.WI * 32)) * (c.BJ / (c.WJ * 32)) * 16 <= 256 && c.MINW >= 1 && lds(c.BI, c.BJ) <= 160 * 1024;
  if (!ok) unsup_err("hip_conv_nhwc (patch form of filts): unsupported tile configuration " + c.str());
  p.cfg = c;
  p.defs = {"-DBI=" + std::to_string(c.BI), "-DBJ=" + std::to_string(c.BJ), "-DWI=" + std::to_string(c.WI), "-DWJ=" + std::to_string(c.WJ), "-DMINW=" + std::to_string(c.MINW),
            "-DCG=" + std::to_string(cg), "-DCIN=" + std::to_string(g.C), "-DKH=" + std::to_string(g.KH), "-DKW=" + std::to_string(g.KW), "-DSY=" + std::to_string(g.SY),
            "-DPY=" + std::to_string(g.PY), "-DPX=" + std::to_string(g.PX), "-DCH=" + std::to_string(g.H), "-DCW=" + std::to_string(g.W),
            "-DCOH=" + std::to_string(g.OH), "-DCOW=" + std::to_string(g.OW), string("-DRELU=") + (g.relu ? "1" : "0"), string("-DOUT_F32=") + (out_f32 ? "1" : "0")};
  if (adirect) p.defs.push_back("-DADIRECT=1");
  if (pool) p.defs.push_back("-DPOOL=1");
  if (c.SPLITK > 1) { c.SPLITK = ksl_normalise(c.SPLITK, (ncg + cg - 1) / cg); p.cfg = c; if (c.SPLITK > 1) { p.defs.push_back("-DKSL=" + std::to_string(c.SPLITK)); p.ksl = true; } }
  if (adirect && (pick_pf || (tile.empty() ? 0 : ((c.BI / (c.WI * 32)) * (c.BJ / (c.WJ * 32)) >= 8)))) p.defs.push_back("-DPF=4");   // (128 accumulators: four fragments in flight)
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return p;
}
// Rolling-rows form of the channels-last bf16 convolution (kernels/conv_nhwc_rows_bf16.hip): short-K layers of at most 64 out_chans whose time is their output --
// the 7x7/2 stems after space-to-depth.  A workgroup of eight waves walks down a run of output rows of one image with the filters in registers; the rows land in an LDS
// ring from which they are stored -- or pooled (+ LRN'd) without ever reaching memory (post).  Same k-slot order as the patch kernel (patch_cg): with an even tap count
// the MFMA chains are the same for every CG, i.e. the same bits as bodahip_conv_nhwc_patch_bf16 whatever that one's plan.
// Returns false (why = the reason) where the form does not apply.  cfg: BI x BJ = 64 x 256 positions per tile, BK = K of the layer, WJ waves.
struct rows_args_t { // must match kernels/conv_nhwc_rows_bf16.hip
  void const *filts; void const *in; void *out; float const *bias;
  int n_img, oc; unsigned filts_bytes, in_bytes, out_bytes; int out_ctot, out_coff; int n_chunks, rows_per_chunk; float lrn_alpha, lrn_beta, lrn_k;
};
static bool plan_conv_nhwc_rows(conv_geom_t const &g, post_ops_t const &post, int num_cus, plan_t &p, string *why = nullptr) {
  auto no = [&](char const *w) { if (why) *why = w; return false; };
  (void)num_cus;
  int const taps = g.KH * g.KW, ncg = g.C / 8;
  if (g.C % 8) return no("in_chan must be a multiple of 8");
  if (!(g.SX == 1 && g.SY == 1 && taps >= 2)) return no("needs stride 1 and more than one tap");
  if (g.OC > 64) return no("at most 64 out_chans (every wave multiplies all of them)");
  if (g.OW > 256 || g.OW < 1) return no("output rows of at most 256 positions");
  int cg = patch_cg(ncg, taps);
  if (char const *e = getenv("BODAHIP_NHWC_PATCH_CG")) { if (atoi(e) > 0) cg = std::min(ncg, atoi(e)); }
  int const nkt = (ncg + cg - 1) / cg, npr = cg * taps, kn = (npr + (npr & 1)) / 2, nit = nkt * kn;
  if (nit * 2 * 4 > 160) return no("the filters do not fit the registers (K too long)");
  if (post.pooled() && !g.relu) return no("the fused pooling needs the convolution's ReLU (non-negative values)");
  if (post.LRN_N && !(post.pooled() && (post.LRN_N & 1) && post.LRN_N <= 9)) return no("the fused LRN follows a fused pooling; odd local sizes up to 9");
  int wj = 8; if (char const *e = getenv("BODAHIP_NHWC_ROWS_WJ")) { int const v = atoi(e); if (v == 2 || v == 4 || v == 8) wj = v; }
  int tr = std::max(1, std::min(256 / g.OW, 8)); if (char const *e = getenv("BODAHIP_NHWC_ROWS_TR")) { int const v = atoi(e); if (v >= 1 && v * g.OW <= 256) tr = v; }
  auto lds = [&](int trx) {
    int const wr = g.W + 2 * g.PX; int wp = wr; for (int q = wr; q < wr + 16; ++q) if ((q - g.OW) % 16 == 0) { wp = q; break; }
    long const cs = (long)(trx + g.KH - 1) * wp, csp = cs + ((2 - cs % 16) + 16) % 16;
    return 2l * 16l * ncg * csp + (long)(post.pooled() ? trx + post.PKH - 1 : trx) * g.OW * (64 * 2 + 16) + (post.pooled() ? (long)post.POW * 160 : 0l) + 256l;
  };
  while (tr > 1 && lds(tr) > 160 * 1024) --tr;
  if (lds(tr) > 160 * 1024) return no("the input rows of one output row do not fit the LDS");
  p = plan_t(); p.nhwc = true; p.nhwc_rows = true; p.bf16 = true; p.kname = "bodahip_conv_nhwc_rows_bf16"; p.cg = cg; p.rows = tr;
  tile_cfg_t c; c.MT = 32; c.SPLITK = 1; c.PF = 1; c.BI = 64; c.BJ = 256; c.BK = g.C * taps; c.WI = 1; c.WJ = wj; c.MINW = 1;
  p.cfg = c;
  p.defs = {"-DCIN=" + std::to_string(g.C), "-DCG=" + std::to_string(cg), "-DKH=" + std::to_string(g.KH), "-DKW=" + std::to_string(g.KW), "-DPY=" + std::to_string(g.PY), "-DPX=" + std::to_string(g.PX),
            "-DCH=" + std::to_string(g.H), "-DCW=" + std::to_string(g.W), "-DCOH=" + std::to_string(g.OH), "-DCOW=" + std::to_string(g.OW), string("-DRELU=") + (g.relu ? "1" : "0"),
            "-DWJ=" + std::to_string(wj), "-DTR=" + std::to_string(tr)};
  if (post.pooled()) for (auto const &kv : std::vector<std::pair<char const *, int>>{{"PKH", post.PKH}, {"PKW", post.PKW}, {"PSY", post.PSY}, {"PSX", post.PSX}, {"PPY", post.PPY}, {"PPX", post.PPX}, {"POH", post.POH}, {"POW", post.POW}})
    p.defs.push_back(string("-D") + kv.first + "=" + std::to_string(kv.second));
  if (post.LRN_N) { p.defs.push_back("-DLRN_N=" + std::to_string(post.LRN_N)); p.defs.push_back("-ffast-math"); }   // (the LRN expression under the flags of the generated LRN functions: see the kernel)
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return true;
}
// Plain convolutions (no post ops) that take the rolling-rows kernel by themselves: output-bound stems -- a short K (the filters in registers), few out_chans, a large
// output.  BODAHIP_NHWC_ROWS=0: never (the patch kernel, as before round 5); =1: wherever the form applies.
static bool rows_auto(conv_geom_t const &g, int num_cus, string const &tile) {
  char const *e = getenv("BODAHIP_NHWC_ROWS");
  if ((e && atoi(e) == 0) || !tile.empty()) return false;
  plan_t p;
  if (!plan_conv_nhwc_rows(g, post_ops_t(), num_cus, p)) return false;
  if (e && atoi(e) == 1) return true;
  return (long)g.C * g.KH * g.KW <= 512 && (double)g.B * g.OH * g.OW * g.OC * 2.0 >= 32e6 && g.B * 2 >= num_cus / 4;
}

void native_kernels_t::conv_nhwc_rows(void const *filts, float const *biases, void const *in, void *out, conv_geom_t const &g, post_ops_t const &post, int out_ctot, int out_coff) {
  if (out_ctot <= 0) { out_ctot = g.OC; out_coff = 0; }
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
  if (!Nj || !g.OC) return;
  plan_t p; string why;
  if (!plan_conv_nhwc_rows(g, post, host->nh_num_cus(), p, &why)) unsup_err("hip_conv_nhwc (rolling-rows form): " + why);
  kernel_t &k = get_kernel(impl, host, p);
  long const out_pels = post.pooled() ? (long)g.B * post.POH * post.POW : Nj;
  uint64_t const in_bytes = (uint64_t)g.B * g.C * g.H * g.W * 2, f_bytes = (uint64_t)g.OC * Kt * 2, out_bytes = (uint64_t)out_pels * out_ctot * 2;
  if (in_bytes >= 0x7ffffff0ull || f_bytes >= 0x7ffffff0ull || out_bytes >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
  rows_args_t ra; memset(&ra, 0, sizeof(ra));
  ra.filts = filts; ra.in = in; ra.out = out; ra.bias = biases; ra.n_img = g.B; ra.oc = g.OC;
  ra.filts_bytes = (unsigned)f_bytes; ra.in_bytes = (unsigned)in_bytes; ra.out_bytes = (unsigned)out_bytes; ra.out_ctot = out_ctot; ra.out_coff = out_coff;
  // one workgroup per CU: the chunks of an image each redo the rows they share with the next one (overlapping pooling windows), so no more of them than fill the chip
  int const rows = post.pooled() ? post.POH : g.OH, cus = host->nh_num_cus();
  int nch = std::max(1, std::min(rows, (cus + g.B - 1) / g.B));
  if (char const *e = getenv("BODAHIP_NHWC_ROWS_CHUNKS")) { int const v = atoi(e); if (v >= 1) nch = std::min(rows, v); }
  ra.rows_per_chunk = (rows + nch - 1) / nch; ra.n_chunks = (rows + ra.rows_per_chunk - 1) / ra.rows_per_chunk;
  ra.lrn_alpha = post.alpha; ra.lrn_beta = post.beta; ra.lrn_k = post.k;
  void *params[] = {&ra};
  uint32_t const grid = (uint32_t)(g.B * ra.n_chunks);
  hip_err_chk(host->nh_launch(k.func, grid, 1, (uint32_t)p.cfg.threads(), params), "hipModuleLaunchKernel(conv_nhwc_rows_bf16)");
  last_launch.kernel = p.kname; last_launch.cfg = p.cfg; last_launch.grid = grid; last_launch.block = p.cfg.threads();
  last_launch.flops = 2.0 * Nj * g.OC * Kt;
  last_launch.algo_bytes = 2.0 * ((double)g.B * g.C * g.H * g.W + (double)g.OC * Kt) + 2.0 * (double)out_pels * g.OC + 4.0 * g.OC;
}

// What follows the convolution inside the rolling-rows launch (annotations of the function's op, boda_amd/nhwc.py fuse_post): uint32 nhwc_post_pool = 1 with dims
// post_pool_sz / post_pool_stride / post_pool_pad (y, x); uint32 nhwc_post_lrn = local size with floats post_lrn_alpha / post_lrn_beta / post_lrn_k.  The function's
// `out` is then the POOLED tensor: g arrives with OH x OW read from it; they become post's planes and g gets the convolution's own.
static float op_f32(op_base_t const &op, string const &an) {
  p_nda_t const &n = op.get(an); if (n->dims.tn != "float" || n->dims.sz() != 0 || !n->rp) rt_err("op: '" + an + "' is not a float scalar");
  return *static_cast<float const *>(n->rp);
}
static bool apply_post_ops(op_base_t const &op, conv_geom_t &g, post_ops_t &post, char const *what) {
  post = post_ops_t();
  if (!op.has("nhwc_post_pool") || !op.get_u32("nhwc_post_pool")) return false;
  dims_t const &ks = op.get_dims("post_pool_sz"), &st = op.get_dims("post_pool_stride"), &pp = op.get_dims("post_pool_pad");
  post.PKH = (int)ks.dsz("y"); post.PKW = (int)ks.dsz("x"); post.PSY = (int)st.dsz("y"); post.PSX = (int)st.dsz("x"); post.PPY = (int)pp.dsz("y"); post.PPX = (int)pp.dsz("x");
  post.POH = g.OH; post.POW = g.OW;
  if (!g.SY || !g.SX) rt_err(string(what) + ": zero stride");
  g.OH = (g.H + 2 * g.PY - g.KH) / g.SY + 1; g.OW = (g.W + 2 * g.PX - g.KW) / g.SX + 1;
  if (post.PKH < 1 || post.PKW < 1 || post.PKH > 7 || post.PKW > 7 || post.PSY < 1 || post.PSX < 1 || post.PPY < 0 || post.PPX < 0 || post.PPY >= post.PKH || post.PPX >= post.PKW)
    unsup_err(string(what) + ": fused pooling behind the convolution takes windows of at most 7 x 7 with a padding smaller than the window");
  if (post.POH < 1 || post.POW < 1 || (post.POH - 1) * post.PSY - post.PPY >= g.OH || (post.POW - 1) * post.PSX - post.PPX >= g.OW)
    rt_err(string(what) + ": the pooled planes of `out` have windows outside the convolution's output");
  if (op.has("nhwc_post_lrn") && op.get_u32("nhwc_post_lrn")) {
    post.LRN_N = (int)op.get_u32("nhwc_post_lrn"); post.alpha = op_f32(op, "post_lrn_alpha"); post.beta = op_f32(op, "post_lrn_beta"); post.k = op_f32(op, "post_lrn_k");
  }
  return true;
}

// Exact fp32 convolutions whose operands are k-contiguous in the REFERENCE layout -- output 1x1, no padding, kernel == whole input (AlexNet
// fc6-fc8: in[img][K], filts[out_chan][K]) -- through the LDS-DMA kernel's IN_F32 variant (kernels/conv_nhwc_bf16.hip): 64x64 tiles of four
// waves, 32-deep K steps, an 8-slot LDS ring (six K steps of loads in flight).  Same ascending-k fma chain: bit-exact (tested).  MEASURED SLOWER
// than the register-staged gather kernel on MI355X and therefore opt-in (BODAHIP_IPCONV_DMA=1): AlexNet fc6 / fc7 at 256 images 65.5 / 63.8 TF/s
// with 8 ring slots (37 / 37 with 2 or 4) against 80.5 / 78.6 -- a workgroup alone on its CU streaming 16 KB per K step through LDS-DMA gets
// ~16 GB/s however many steps are in flight (the per-CU LDS-DMA fill rate from HBM), less than two register-staged K-tiles deliver.
static bool plan_ipconv_dma(conv_geom_t const &g, int num_cus, plan_t &p) {
  if (!getenv("BODAHIP_IPCONV_DMA")) return false;
  long const Kt = (long)g.C * g.KH * g.KW;
  if (!(g.OH == 1 && g.OW == 1 && g.PY == 0 && g.PX == 0 && g.KH == g.H && g.KW == g.W)) return false;
  if (Kt % 4 || Kt < 512) return false;
  long const tiles = (long)((g.OC + 63) / 64) * ((g.B + 63) / 64);
  if (tiles * 4 < num_cus * 3 && string(getenv("BODAHIP_IPCONV_DMA")) != "force") return false;   // fewer workgroups than 3/4 of the CUs: finer (16x16-MFMA) tiles of the gather kernel do better
  tile_cfg_t c; c.BI = 64; c.BJ = 64; c.BK = 32; c.WI = 2; c.WJ = 2; c.MINW = 1; c.SPLITK = 1; c.MT = 32; c.PF = 8;
  if (char const *e = getenv("BODAHIP_IPCONV_DMA_NBUF")) c.PF = std::max(2, std::min(8, atoi(e)));
  p = plan_t(); p.nhwc = true; p.kname = "bodahip_conv_nhwc_f32"; p.cfg = c;
  p.defs = {"-DBI=64", "-DBJ=64", "-DBK=32", "-DWI=2", "-DWJ=2", "-DMINW=1", "-DCIN=" + std::to_string(Kt), "-DKH=1", "-DKW=1", "-DSY=1", "-DSX=1", "-DPY=0", "-DPX=0",
            "-DCH=1", "-DCW=1", "-DCOH=1", "-DCOW=1", string("-DRELU=") + (g.relu ? "1" : "0"), "-DOUT_F32=1", "-DIN_F32=1", "-DNBUF=" + std::to_string(c.PF)};
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return true;
}
// Tolerance mode (tune `exact` = 0, op_tune hip_exact=0): the default fp32 plan keeps every output ONE ascending-k fma chain -- bit-identical to the
// reference's per-thread loop -- which leaves tile-starved layers with a long K (AlexNet fc6 / fc7 / fc8: 256 / 256 / 64 tiles of 64x64 for 256
// CUs, 128-288 barrier-separated K steps each) on one workgroup per CU.  The reference's own bar is a tolerance, not bit equality; with exact = 0
// the planner may cut K into slices (deterministic: slice s owns K-tiles [s*kt_per, (s+1)*kt_per), the slabs are summed in ascending slice
// order by bodahip_splitk_reduce) when the tiles do not fill the chip and K >= 512 (>= 8 K steps per slice).  That re-associates the sum: on the reference's U(-5,5)
// data fc6 (K = 9216) then differs from the single chain by mrd 8.6e-4 -- inside the reference's bound for re-associating kernels (2e-3,
// src/rtc_prof.cc:317-319,436), outside its 2e-4 default (:161), and no farther from the exact fp64 product than the chain itself (tested).
// Measured (MI355X, AlexNet at 256 images, TF/s): fc6 75 -> 94 alone, 80 -> 106 in the layer sequence (64x64, 4 slices of 72 K steps),
// fc7 78 -> 94, fc8 33 -> 57.
static void tolerance_splitk(plan_t &p, conv_geom_t const &g, int num_cus, long Nj, long Kt) {
  static long const min_k = getenv("BODAHIP_TOL_MIN_K") ? atol(getenv("BODAHIP_TOL_MIN_K")) : 512, min_steps = getenv("BODAHIP_TOL_MIN_STEPS") ? atol(getenv("BODAHIP_TOL_MIN_STEPS")) : 8;
  // (swept on MI355X, fp32 lists at 64 images, effective TF/s with (min K, min K steps per slice) = (2048, 32) / (1024, 16) / (512, 8): GoogLeNet 76.7 / 78.3 / 79.9, ResNet-50 100.5 / 101.5 /
  //  101.7, AlexNet at 256 images 139.2 / 141.0 / 139.6)
  if (p.bf16 || p.stream || p.patch16 || p.patch || p.nhwc || p.rows || p.cfg.SPLITK != 1 || Kt < min_k) return;
  tile_cfg_t c = p.cfg;
  long tiles = (long)((g.OC + c.BI - 1) / c.BI) * ((Nj + c.BJ - 1) / c.BJ);
  if (tiles > num_cus) return;
  if (c.MT != 32 || c.BI < 64 || c.BJ < 64) {   // the thin tiles tile starvation chose: with K slices the common 64x64 tile fills the chip
    c.BI = 64; c.BJ = 64; c.BK = 32; c.WI = 2; c.WJ = 2; c.MINW = 2; c.MT = 32; c.PF = 2;
    tiles = (long)((g.OC + 63) / 64) * ((Nj + 63) / 64);
  }
  long const nkt = (Kt + c.BK - 1) / c.BK;
  int sk = 1;
  while (sk < 16 && nkt / (sk * 2) >= min_steps && tiles * sk * 2 <= 4l * num_cus) sk *= 2;
  if (sk < 2) return;
  c.SPLITK = sk; p.cfg = c;
}
// Strided convolutions without padding and with wide kernels (conv1 layers: 11x11 / 4): the row-decimated LDS patch of gemm_conv_f32.hip (-DRDEC=1, J_MODE 7 presented with
// C*KH row sets of 1 x KW kernels).  BODAHIP_RDEC = off | BIxBJxWIxWJxMINW.  In the returned plan cfg.BK = row sets per K step x KW.
static bool plan_rdec(conv_geom_t const &g, int num_cus, plan_t &p) {
  char const *e = getenv("BODAHIP_RDEC");
  if (e && string(e) == "off") return false;
  if (!(g.PY == 0 && g.PX == 0 && g.SY > 1 && g.KH >= 2 && g.KW >= 6 && g.KW <= 16 && g.OH > 1)) return false;
  int cb = 1; while (cb * g.KW < 20 || ((cb * g.KW) & 1)) ++cb;
  int const bk = cb * g.KW; if (bk > 128) return false;
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
  auto fits = [&](tile_cfg_t const &c) {
    int const rows_max = (c.BJ - 2) / g.OW + 2; long const cs = (long)rows_max * g.W;
    return 2l * 4 * ((long)bk * (c.BI + 4) + cb * cs) <= 64 * 1024 && cs <= 16l * (c.WI * c.WJ * 64);
  };
  struct cand_t { int bi, bj, wi, wj, minw; double base; };
  // measured (MI355X, AlexNet conv1 at 256 images, isolated, us): row gather 96x256 531 | row-decimated patch 32x256 508, 96x256 546, 96x128 549-562, 64x256 (padded out_chans) 628
  static cand_t const cands[] = {{32, 256, 1, 4, 2, 1.00}, {96, 256, 1, 4, 2, 0.93}, {64, 256, 1, 4, 2, 0.93}, {128, 256, 2, 4, 1, 0.90}};
  tile_cfg_t best_c; double best = -1;
  if (e && *e) { int v[5]; if (sscanf(e, "%dx%dx%dx%dx%d", &v[0], &v[1], &v[2], &v[3], &v[4]) != 5) rt_err(string("bad BODAHIP_RDEC '") + e + "' (off | BIxBJxWIxWJxMINW)");
    best_c.BI = v[0]; best_c.BJ = v[1]; best_c.WI = v[2]; best_c.WJ = v[3]; best_c.MINW = v[4]; best_c.BK = bk; best_c.MT = 32; best_c.SPLITK = 1; best_c.PF = 1;
    if (!fits(best_c)) unsup_err(string("BODAHIP_RDEC '") + e + "': the tile does not fit"); best = 1; }
  else for (cand_t const &cd : cands) {
    tile_cfg_t c; c.BI = cd.bi; c.BJ = cd.bj; c.WI = cd.wi; c.WJ = cd.wj; c.MINW = cd.minw; c.BK = bk; c.MT = 32; c.SPLITK = 1; c.PF = 1;
    if (!fits(c)) continue;
    long const ti = (g.OC + c.BI - 1) / c.BI, tj = (Nj + c.BJ - 1) / c.BJ, tiles = ti * tj;
    double const pad = ((double)g.OC / (double)(ti * c.BI)) * ((double)Nj / (double)(tj * c.BJ));
    double const bal = ((double)tiles / num_cus) / (double)((tiles + num_cus - 1) / num_cus);
    double const score = cd.base * pad * bal;
    if (score > best) { best = score; best_c = c; }
  }
  if (best < 0) return false;
  check_cfg(best_c, false, true);
  p = plan_t(); p.kname = "bodahip_conv_f32"; p.patch = true; p.rdec = true; p.cfg = best_c;
  p.defs = cfg_defs(p.cfg);
  p.defs.push_back(string("-DI_MODE=") + ((Kt % 4 == 0 && bk % 4 == 0) ? "2" : ((Kt % 2 == 0) ? "4" : "3")));
  for (string const &d : {string("-DJ_MODE=7"), string("-DRDEC=1"), "-DC0=" + std::to_string(g.C), "-DH0=" + std::to_string(g.H), "-DKH0=" + std::to_string(g.KH), "-DSY0=" + std::to_string(g.SY),
                          "-DCH=" + std::to_string(g.OH), "-DCW=" + std::to_string(g.W), "-DCOH=" + std::to_string(g.OH), "-DCOW=" + std::to_string(g.OW), string("-DEPI=1"),
                          string("-DKH=1"), "-DKW=" + std::to_string(g.KW), string("-DSY=1"), "-DSX=" + std::to_string(g.SX), string("-DPY=0"), string("-DPX=0"),
                          string("-DRELU=") + (g.relu ? "1" : "0")}) p.defs.push_back(d);
  return true;
}
// Round 6: kernels/conv_big_f32.hip -- WI x WJ multiplying waves + four staging waves (tile field SW == 2): "TBIxTBJxBKSxWIxWJxMINWx1x32xPFx2".  The pel side takes the
// cheapest form the geometry allows: the LDS input patch (stride 1 in x, more than one tap: BKS becomes whole channels, the smallest even multiple of KH KW that is >= the
// tile's BKS), the plain 1x1 form, else the table gather.  The host-side checks are the kernel's static_asserts (a bad tune is an unsup_err, not a compile failure).
struct conv_big_form_t { int jmode = 2, bks = 16, nstg = 4, ivw = 1; long lds = 0; bool rdec = false; };
static bool conv_big_form(conv_geom_t const &g, tile_cfg_t const &c, conv_big_form_t &f, string *why = nullptr) {
  auto bad = [&](char const *m) { if (why) *why = m; return false; };
  int const nmw = c.WI * c.WJ;
  if (!(nmw == 8 || nmw == 4)) return bad("eight (or four) multiplying waves");
  if (c.BI <= 0 || c.BJ <= 0 || c.BI % (c.WI * 32) || c.BJ % (c.WJ * 32)) return bad("tile not a multiple of the waves' 32 x 32 blocks");
  int const ti = c.BI / (c.WI * 32), tj = c.BJ / (c.WJ * 32);
  if (ti > 4 || tj > 4 || ti * tj > 8) return bad("more than 4 x 2 | 2 x 4 blocks per wave");
  if (c.BK % 2 || c.BK < 4 || c.BK > 64) return bad("BKS: even, 4 .. 64");
  if (!(c.PF == 1 || c.PF == 2 || c.PF == 4) || c.MT != 32 || c.SPLITK != 1 || c.KHO > 1 || c.MINW < 1 || c.MINW > 2) return bad("PF 1 | 2 | 4, MT 32, no K slices / hand-off, MINW 1 | 2");
  long const Kt = (long)g.C * g.KH * g.KW;
  int const ldi = (c.BI / ti) * (ti == 3 ? 4 : ti) + 4, ldj = (c.BJ / tj) * (tj == 3 ? 4 : tj) + 4;
  long img_j = 0;
  bool const k1 = g.KH == 1 && g.KW == 1 && g.PY == 0 && g.PX == 0;
  // strided, unpadded, wide kernels (conv1 layers: 11x11 / 4): the row-decimated patch -- C * KH row sets of 1 x KW kernels over the OH decimated rows (-DRDEC=1)
  bool const rdec = !k1 && g.PY == 0 && g.PX == 0 && g.SY > 1 && g.KH >= 2 && g.KW >= 6 && g.KW <= 16 && g.OH > 1 && getenv("BODAHIP_CBIG_NO_PATCH") == nullptr;
  bool const patch = rdec || (!k1 && g.SX == 1 && g.KH * g.KW >= 2 && g.KH >= g.SY && getenv("BODAHIP_CBIG_NO_PATCH") == nullptr);
  f.bks = c.BK;
  bool patch_ok = false;
  if (patch) {   // (a patch that does not fit -- whole-input windows, very wide planes -- gives way to the table gather)
    int const taps = rdec ? g.KW : g.KH * g.KW; int cb = 1; while (cb * taps < c.BK || (cb * taps) % 2) ++cb;
    int const wp = rdec ? g.SX * ((g.W + g.SX - 1) / g.SX) : g.W + 2 * g.PX, rows_max = (c.BJ - 2) / g.OW + 2, seg_max0 = (g.OH - 1 + rows_max - 1) / g.OH + 1, seg_max = std::min(seg_max0, rows_max);
    long const cs = rdec ? (long)rows_max * wp : (long)((rows_max - seg_max) * g.SY + seg_max * g.KH) * wp;
    long const stage4 = ((long)cb * taps * ldi + (cb * cs + 3) / 4 * 4) * 4;
    if (cb * taps <= 128 && cs <= 16 * 256 && 3 * stage4 <= 160l * 1024 / c.MINW) { patch_ok = true; f.bks = cb * taps; f.jmode = 7; f.rdec = rdec; img_j = (cb * cs + 3) / 4 * 4; }
  }
  if (!patch_ok) {
    int const cpt = (c.BJ + 255) / 256; if (c.BJ % cpt) return bad("pel columns per staging thread");
    int const tw = c.BJ / cpt; if (tw % 64 || 256 / tw < 1 || c.BK % (256 / tw)) return bad("pel staging: whole waves per k row");
    if (!k1 && (c.BK / (256 / tw)) % 4) return bad("table gather: whole quads of k rows per staging thread");
    f.jmode = k1 ? 5 : 2; img_j = (long)f.bks * ldj;
  }
  // filters: k-major from the call's scratch (0; bodahip_conv_big_xpose runs first) unless BODAHIP_CBIG_IVW=direct asks for loads straight from OIHW rows (4 | 2 | 1 floats along k)
  f.ivw = 0; if (char const *e = getenv("BODAHIP_CBIG_IVW")) { if (string(e) == "direct") f.ivw = (Kt % 4 == 0 && f.bks % 4 == 0) ? 4 : ((Kt % 2 == 0 && f.bks % 2 == 0) ? 2 : 1); }
  long const stage = ((long)f.bks * ldi + img_j) * 4, cap = 160l * 1024 / c.MINW;
  f.nstg = 4; if (char const *e = getenv("BODAHIP_CBIG_NSTG")) f.nstg = atoi(e); else if (4 * stage > cap) f.nstg = 3;
  if (!(f.nstg == 3 || f.nstg == 4) || f.nstg * stage > cap) return bad("LDS stages exceed the CU's 160 KB");
  f.lds = f.nstg * stage;
  return true;
}
static plan_t plan_conv_big(conv_geom_t const &g0, tile_cfg_t const &c) {
  string why; conv_big_form_t f;
  if (!conv_big_form(g0, c, f, &why)) unsup_err("native kernel: unsupported staging-wave tile " + c.str() + " (" + why + ")");
  plan_t p; p.cbig = true; p.k1 = (f.jmode == 5); p.patch = (f.jmode == 7); p.rdec = f.rdec; p.kname = "bodahip_conv_big_f32"; p.cfg = c; p.cfg.BK = f.bks;
  conv_geom_t g = g0; if (f.rdec) { g.KH = 1; g.SY = 1; g.H = g0.OH; }   // (the decimated presentation: 1 x KW kernels, stride 1 in y, OH rows)
  p.defs = {"-DTBI=" + std::to_string(c.BI), "-DTBJ=" + std::to_string(c.BJ), "-DWI=" + std::to_string(c.WI), "-DWJ=" + std::to_string(c.WJ), "-DBKS=" + std::to_string(f.bks),
            "-DPF=" + std::to_string(c.PF), "-DNSTG=" + std::to_string(f.nstg), "-DMINW=" + std::to_string(c.MINW), "-DI_VW=" + std::to_string(f.ivw), "-DJ_MODE=" + std::to_string(f.jmode),
            "-DKH=" + std::to_string(g.KH), "-DKW=" + std::to_string(g.KW), "-DSY=" + std::to_string(g.SY), "-DSX=" + std::to_string(g.SX),
            "-DPY=" + std::to_string(g.PY), "-DPX=" + std::to_string(g.PX), string("-DRELU=") + (g.relu ? "1" : "0")};
  if (f.jmode == 7) for (auto const &kv : {std::make_pair("CH", g.H), std::make_pair("CW", g.W), std::make_pair("COH", g.OH), std::make_pair("COW", g.OW)}) p.defs.push_back(string("-D") + kv.first + "=" + std::to_string(kv.second));
  if (f.rdec) for (auto const &kv : {std::make_pair("RDEC", 1), std::make_pair("C0", g0.C), std::make_pair("H0", g0.H), std::make_pair("KH0", g0.KH), std::make_pair("SY0", g0.SY)}) p.defs.push_back(string("-D") + kv.first + "=" + std::to_string(kv.second));
  if (char const *x = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(x); string tok; while (is >> tok) p.defs.push_back(tok); }
  return p;
}
static plan_t plan_conv_tiled(conv_geom_t const &g, int num_cus, string const &tile, bool bf16, string const &k1s, bool allow_splitk, bool exact) {
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
  plan_t p;
  if (!bf16 && tile.empty() && plan_ipconv_dma(g, num_cus, p)) return p;
  if (!bf16 && tile.empty() && plan_k1_stream(g, num_cus, k1s, p)) return p;
  if (!bf16 && tile.empty() && !g.pooled() && plan_rdec(g, num_cus, p)) return p;
  if (bf16 && tile.empty() && plan_patch_bf16(g, num_cus, p)) return p; p.kname = bf16 ? "bodahip_conv_bf16" : "bodahip_conv_f32"; p.bf16 = bf16;
  // output 1x1, no padding, kernel == whole input ("ipconv" case): the im2col row of image j is the contiguous image
  p.ipconv = (g.OH == 1 && g.OW == 1 && g.PY == 0 && g.PX == 0 && g.KH == g.H && g.KW == g.W);
  p.cfg = choose_cfg(g.OC, (int)Nj, (int)Kt, num_cus, !p.ipconv, bf16);
  if (!tile.empty()) { if (!parse_tile(tile, p.cfg)) rt_err("bad conv_tile '" + tile + "'"); }
  if (!bf16 && p.cfg.SW == 2) {   // the staging-wave kernel, asked for by its own tile string
    if (g.pooled()) unsup_err("hip_conv: fused pooling (hip_pool) is a form of the LDS-patch kernel, not of the staging-wave kernel");
    return plan_conv_big(g, p.cfg);
  }
  // wide kernels: row gather (one address + wide loads per (in_chan,ky) row of KW taps): a K step is `rows` whole rows, BK = rows*KW.
  // Measured (MI355X, B=256): 11x11/s4 +8%, 5x5 -3%, 3x3 -9% vs the per-element table gather (unaligned x3 loads cost more than the
  // address arithmetic they save), so the default takes it for KW >= 6 only; BODAHIP_ROW_GATHER_MIN_KW overrides (>= 2).
  p.rows = 0;
  int rg_min_kw = 6; if (char const *e = getenv("BODAHIP_ROW_GATHER_MIN_KW")) rg_min_kw = std::max(2, atoi(e));
  if (!bf16 && !p.ipconv && g.KW >= rg_min_kw && g.KW <= 16 && getenv("BODAHIP_NO_ROW_GATHER") == nullptr) {
    int const rpp = std::max(1, p.cfg.WI * p.cfg.WJ * 64 / p.cfg.BJ);               // row groups per K step
    int rows = (g.KW <= 3) ? 8 : (g.KW <= 8 ? 4 : 2);                        // BK = 16..28 (22 for 11x11)
    while (rows % rpp) rows += 2;
    if ((rows * g.KW) % 2 == 0 && rows % rpp == 0 && (p.cfg.WI * p.cfg.WJ * 64) % p.cfg.BJ == 0 && p.cfg.BJ % 64 == 0 && p.cfg.MT == 32) { p.rows = rows; p.cfg.BK = rows * g.KW; }
  }
  // 1x1 kernel, no padding (any stride): the reference's k1conv case -- one add per gathered element, no table
  p.k1 = !p.ipconv && g.KH == 1 && g.KW == 1 && g.PY == 0 && g.PX == 0;
  // long-K 1x1 layers on 64x64 tiles: a 32-deep K step (tools/tune_tiles.py over GoogLeNet / ResNet-50 at B=64: every 1x1 layer with
  // >= 480 input channels gains 4-6 % over the 16-deep step; with 256 channels and fewer it does not)
  if (p.k1 && !bf16 && tile.empty() && p.cfg.BI == 64 && p.cfg.BJ == 64 && p.cfg.MT == 32 && p.cfg.BK == 16 && g.C >= 448) p.cfg.BK = 32;
  // Round 5, in-sequence A/Bs through the tile-wisdom path (tools/wisdom_ab.sh; isolated sweeps mislead here): a 1x1 layer whose 128 x 128 tiles still make >= 3.5 rounds
  // of the CUs runs 8-9 % faster on them than on the 64 x 64 tiles the short-launch rule prefers (half the operand bytes per flop through the L2) -- NiN cccp5 / cccp6 at
  // 256 images 128 -> 116 us, cccp3 / cccp4 at 128 images 129 -> 119 us; with fewer tiles (cccp5 at 128 images: 507, cccp7 / cccp8: 288-576) the finer tiles stay ahead.
  if (p.k1 && !bf16 && tile.empty() && p.cfg.BI == 64 && p.cfg.BJ == 64 && p.cfg.MT == 32 && g.OC % 128 == 0 && getenv("BODAHIP_NO_K1_128") == nullptr) {
    long const t128 = (long)(g.OC / 128) * ((Nj + 127) / 128);
    if (t128 * 2 >= 7l * num_cus) { p.cfg.BI = 128; p.cfg.BJ = 128; p.cfg.BK = 16; p.cfg.WI = 2; p.cfg.WJ = 2; p.cfg.MINW = 2; p.cfg.PF = 1; }
  }
  // SX == 1, more than one tap: LDS input patch (J_MODE 7) -- a K step is CB whole input channels, staged as padded input rows
  // (coalesced, ~KH*KW x fewer loads than an im2col image) and read by the MFMAs in place.  Needs compile-time plane sizes.
  p.patch = false;
  if (!bf16 && !p.ipconv && !p.k1 && !p.rows && g.SX == 1 && g.KH * g.KW >= 2 && g.KH >= g.SY && (tile.empty() || (p.cfg.MT == 32 && p.cfg.SPLITK == 1)) &&
      getenv("BODAHIP_NO_PATCH") == nullptr) { // (an explicit tile keeps its BI/BJ/waves; its BK is replaced by whole channels)
    int const taps = g.KH * g.KW;
    int const bk_min = tile.empty() ? 32 : p.cfg.BK;                          // an explicit tile's BK is the lower bound for the K step
    int cb = 1; while (cb * taps < bk_min || (cb * taps) % 2) ++cb;          // BK = cb*taps: even, >= 32 (3x3 -> 36, 5x5 -> 50, 2x2 -> 32)
    int const bk = cb * taps, wp = g.W + 2 * g.PX;
    auto fits = [&](tile_cfg_t const &c, long lds_max) {
      int const rows_max = (c.BJ - 2) / g.OW + 2, seg_max0 = (g.OH - 1 + rows_max - 1) / g.OH + 1, seg_max = std::min(seg_max0, rows_max);
      long const cs = (long)((rows_max - seg_max) * g.SY + seg_max * g.KH) * wp;
      long const lds = 2l * 4 * ((long)bk * (c.BI + 4) + cb * cs);
      return bk <= 128 && lds <= lds_max && cs <= 16l * (c.WI * c.WJ * 64);
    };
    if (!tile.empty()) { if (fits(p.cfg, 160 * 1024)) { p.patch = true; p.cfg.BK = bk; } }
    else {
      // Staging the pel side is nearly free here, so the best tiles are narrow in out_chan and wide in pels (the filter tile is
      // staged once per 256 pels) with few accumulators per wave (<= 128 VGPRs: four waves per SIMD).  Measured steady state on
      // MI355X (B=256 AlexNet conv2-5, TF/s): 64x256 126-131 | 128x256 125-131 | 32x256 110-129 | 64x64 116-122 | 128x128 109-120;
      // the score is that base rate x tile padding x how evenly the tiles deal out over the CUs.
      struct cand_t { int bi, bj, wi, wj, minw; double base; };
      // tools/tune_tiles.py over every distinct GoogLeNet / ResNet-50 layer at B=64 (ten candidate tiles each) corrected two entries: with a
      // short K loop (< 2048: 64-192 input channels) the 64x64 tile is worth 0.82, not 0.93, of a 64x256 one (GoogLeNet conv2 3x3 64->192
      // @56x56: 471 vs 380 us), and the 32x128 tile wants four waves (one 32x32 block each), not two (3x3 / 5x5 layers with 32-224 out_chans:
      // 13-57 % faster).  What the sweep still finds after that is within a few percent of the planner's choice.
      static double const k32x256 = getenv("BODAHIP_BASE_32X256") ? atof(getenv("BODAHIP_BASE_32X256")) : 0.97; // (0.90 until the planner A/B of tools/tune_all.sh: AlexNet / NiN conv2 5x5 at B=256 1760 -> 1664 us on 32x256, -2 % on both lists)
      static cand_t const cands[] = {{64, 256, 1, 4, 2, 1.00}, {128, 256, 2, 4, 1, 0.99}, {32, 256, 1, 4, 2, k32x256}, {64, 64, 2, 2, 2, 0.93}, {128, 128, 2, 2, 2, 0.92}, {32, 128, 1, 4, 2, 0.80}};
      double best = -1;
      for (cand_t const &cd : cands) {
        tile_cfg_t c = p.cfg; c.BI = cd.bi; c.BJ = cd.bj; c.WI = cd.wi; c.WJ = cd.wj; c.MINW = cd.minw; c.BK = bk; c.MT = 32; c.SPLITK = 1;
        if (!fits(c, 64 * 1024)) continue;
        long const ti = (g.OC + c.BI - 1) / c.BI, tj = (Nj + c.BJ - 1) / c.BJ, tiles = ti * tj;
        double const pad = ((double)g.OC / (double)(ti * c.BI)) * ((double)Nj / (double)(tj * c.BJ));
        double const bal = ((double)tiles / num_cus) / (double)((tiles + num_cus - 1) / num_cus);
        double score = cd.base * pad * bal * ((cd.bi == 64 && cd.bj == 64 && Kt < 2048) ? (0.82 / 0.93) : 1.0);
        if (2.0 * g.OC * (double)Nj * Kt < 2.4e10) { double const x = (double)tiles / num_cus; score *= x / (x + kShortTail); } // short launches: see choose_cfg
        if (score > best) { best = score; p.cfg = c; p.patch = true; }
      }
    }
  }
  if (p.patch && tile.empty()) p.cfg.PF = pf_for(p.cfg);
  if (g.pooled()) {   // a max pooling fused in front (round 5): the LDS-patch form only, one K tile in flight (the window rows are gathered under thirds of a step's MFMAs)
    if (bf16 || !p.patch || p.rdec) unsup_err("hip_conv: fused pooling (hip_pool) needs an fp32 convolution that takes the LDS-patch form (stride 1 in x, more than one tap)");
    p.cfg.PF = 1; p.cfg.SW = 0; p.cfg.SPLITK = 1;
  }
  if (!bf16 && !exact && tile.empty() && allow_splitk && !g.pooled()) tolerance_splitk(p, g, num_cus, Nj, Kt);
  // fully-connected layers (whole-input windows, both operands k-contiguous): kernels/fc_f32.hip -- four multiplying + four staging waves, three LDS stages, 16x16x4
  // MFMA chains.  Tile TM images x TN out_chans: the largest of 64x64 / 64x32 / 32x32 that still gives (nearly) every CU a workgroup.  BODAHIP_FC = off | TMxTNxBKFxPF
  // (an explicit spec forces the kernel onto every layer it covers: tests).  Measured (MI355X, AlexNet at 256 images, layer sequence, us): fc6 223 -> 170, fc7 104 -> 82.
  char const *fc_env = getenv("BODAHIP_FC");
  bool const fc_forced = fc_env && *fc_env && string(fc_env) != "off";
  if (!bf16 && p.ipconv && tile.empty() && Kt % 4 == 0 && p.cfg.SPLITK == 1 && !(fc_env && string(fc_env) == "off") &&
      (fc_forced || (Kt >= 512 && (long)g.OC * Nj >= 65536))) {
    int tm = 64, tn = 64, bkf = 64, pf = 2;
    if (fc_forced) {
      if (sscanf(fc_env, "%dx%dx%dx%d", &tm, &tn, &bkf, &pf) != 4 || (tm != 32 && tm != 64) || (tn != 32 && tn != 64) || (bkf != 32 && bkf != 64) || (pf != 2 && pf != 4))
        rt_err(string("bad BODAHIP_FC '") + fc_env + "' (off | TMxTNxBKFxPF: 32|64 x 32|64 x 32|64 x 2|4)");
    } else {
      auto tiles = [&](int m, int n) { return (long)((Nj + m - 1) / m) * ((g.OC + n - 1) / n); };
      if (tiles(64, 64) * 4 < (long)num_cus * 3) { tn = 32; if (tiles(64, 32) * 4 < (long)num_cus * 3) tm = 32; }
    }
    p.fc = true; p.kname = "bodahip_fc_f32"; p.cfg.BI = tn; p.cfg.BJ = tm; p.cfg.MT = 16; p.cfg.BK = bkf; p.cfg.PF = pf; p.cfg.MINW = 1; p.cfg.WI = 2; p.cfg.WJ = 4;   // (eight waves)
    p.defs = {"-DTM=" + std::to_string(tm), "-DTN=" + std::to_string(tn), "-DBKF=" + std::to_string(bkf), "-DPF=" + std::to_string(pf), string("-DRELU=") + (g.relu ? "1" : "0")};
    if (char const *x = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(x); string tok; while (is >> tok) p.defs.push_back(tok); }
    return p;
  }
  if (bf16 || p.cfg.SPLITK > 1 || g.pooled()) p.cfg.KHO = 0;
  if (p.cfg.KHO > 1) {   // K hand-off: no empty segment
    long const nkt = (Kt + p.cfg.BK - 1) / p.cfg.BK, per = (nkt + p.cfg.KHO - 1) / p.cfg.KHO;
    p.cfg.KHO = (int)((nkt + per - 1) / per); if (p.cfg.KHO <= 1) p.cfg.KHO = 0;
  }
  if (bf16) bf16_cfg(p.cfg, !p.ipconv, g.OC, Nj, Kt, allow_splitk ? num_cus : 0, !tile.empty());
  else check_cfg(p.cfg, !p.ipconv && !p.patch, p.patch);
  p.defs = cfg_defs(p.cfg);
  p.defs.push_back(string("-DI_MODE=") + ((Kt % 4 == 0 && p.cfg.BK % 4 == 0) ? "2" : ((p.patch && Kt % 2 == 0) ? "4" : "3")));
  p.defs.push_back(p.ipconv ? (string("-DJ_MODE=") + ((Kt % 4 == 0) ? "3" : "4")) : string(p.k1 ? "-DJ_MODE=5" : (p.patch ? "-DJ_MODE=7" : (p.rows ? "-DJ_MODE=6" : "-DJ_MODE=2"))));
  if (p.rows) p.defs.push_back("-DJROWS=" + std::to_string(p.rows));
  if (p.patch) { p.defs.push_back("-DCH=" + std::to_string(g.H)); p.defs.push_back("-DCW=" + std::to_string(g.W));
                 p.defs.push_back("-DCOH=" + std::to_string(g.OH)); p.defs.push_back("-DCOW=" + std::to_string(g.OW)); }
  if (g.pooled()) for (auto const &kv : {std::make_pair("PKH", g.PKH), std::make_pair("PKW", g.PKW), std::make_pair("PSY", g.PSY), std::make_pair("PSX", g.PSX), std::make_pair("UH", g.UH), std::make_pair("UW", g.UW)})
    p.defs.push_back(string("-D") + kv.first + "=" + std::to_string(kv.second));
  p.defs.push_back("-DEPI=1");
  if (p.cfg.SPLITK > 1) p.defs.push_back("-DSPLITK=1");
  p.defs.push_back("-DKH=" + std::to_string(g.KH)); p.defs.push_back("-DKW=" + std::to_string(g.KW));
  p.defs.push_back("-DSY=" + std::to_string(g.SY)); p.defs.push_back("-DSX=" + std::to_string(g.SX));
  p.defs.push_back("-DPY=" + std::to_string(g.PY)); p.defs.push_back("-DPX=" + std::to_string(g.PX));
  p.defs.push_back(string("-DRELU=") + (g.relu ? "1" : "0"));
  return p;
}
// plan_conv: the planner of hip_conv.  Round 6: stride-1 KH x KW >= 2 layers (the reference's tconv / conv cases) and 1 x 1 / stride-1 layers with K >= 384 (NiN cccp5-8;
// shorter K belongs to the streaming kernels) go to the staging-wave kernel (kernels/conv_big_f32.hip) in its two-workgroups-per-CU tiles when those deal out evenly.
// In-sequence A/Bs on MI355X (tools/cbig_ab_sets*.sh, tools/env_ab_ops.sh; us, round-5 plan -> this): AlexNet at 256 images conv2 1677 -> 1657, conv3 571 -> 532, conv4
// 849 -> 784, conv5 649 -> 585; NiN at 256 images conv4 546 -> 463, cccp7 / cccp8 175 -> 152 (64 x 192 tiles of FOUR multiplying waves: 768 tiles = exactly three per CU),
// at 128 images conv2 908 -> 850, conv3 315 -> 277, conv4 314 -> 267; its one-workgroup tiles (128 x 512, 256 x 256) measured level or slower.  A lone workgroup of
// this kernel keeps the matrix pipe as busy as two co-resident ones (the staging waves hide the loads), so what counts is the deal over CUs, not over workgroup slots:
// score = base x padding x (tiles / CUs) / ceil(tiles / CUs), taken when >= 0.76.  The tiles of one or two 32 x 32 blocks per wave only stand in for the tiled kernel's
// tile-starvation choice (64 x 64): against its 32 x 256 / 64 x 256 tiles they measured slower (GoogLeNet 3x3 128 -> 192 at 28 x 28: 191 -> 212 us), and strided 1 x 1
// layers stay where they were (ResNet-50 res4a_branch1 on 32 x 128 tiles: 141 -> 230 us).  BODAHIP_CBIG = off | force (any score).
static plan_t plan_conv(conv_geom_t const &g, int num_cus, string const &tile, bool bf16 = false, string const &k1s = string(), bool allow_splitk = true, bool exact = true) {
  plan_t const old = plan_conv_tiled(g, num_cus, tile, bf16, k1s, allow_splitk, exact);
  char const *e = getenv("BODAHIP_CBIG");
  if (bf16 || !tile.empty() || g.pooled() || (e && string(e) == "off") || old.kname != "bodahip_conv_f32" || old.ipconv || old.cfg.SPLITK > 1 || old.cfg.KHO > 1) return old;
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
  if (old.rdec) {   // conv1 layers (11x11 / 4): the row-decimated patch of the staging-wave kernel on 96 x 256 tiles, one K tile in flight so that two workgroups share a CU (70
    // registers) and one's prologue / stores hide under the other's 17 K steps: AlexNet / NiN conv1 at 256 images 472 -> 450 us in the layer sequence (round 6)
    tile_cfg_t c; c.BI = 96; c.BJ = 256; c.BK = 16; c.WI = 1; c.WJ = 8; c.MINW = 2; c.SPLITK = 1; c.MT = 32; c.PF = 1; c.SW = 2; c.KHO = 0;
    conv_big_form_t f; long const ti = (g.OC + 95) / 96;
    bool const force_r = e && string(e) == "force";
    if (conv_big_form(g, c, f) && f.rdec && ((double)g.OC / (double)(ti * 96) >= 0.95 || force_r) && (Nj >= 64l * num_cus || force_r)) return plan_conv_big(g, c);
    return old;
  }
  bool const k1big = g.KH == 1 && g.KW == 1 && g.PY == 0 && g.PX == 0 && g.SY == 1 && g.SX == 1 && Kt >= 384;
  bool const patchy = g.SX == 1 && g.KH * g.KW >= 2 && g.KH >= g.SY && !(g.KH == g.H && g.KW == g.W && g.OH == 1);
  if (!k1big && !patchy) return old;
  bool const force = e && string(e) == "force", old_small = (old.cfg.BI <= 64 && old.cfg.BJ <= 64) || force;
  struct cand_t { int bi, bj, wi, wj; double base; bool small; };
  static cand_t const cands[] = {{128, 256, 2, 4, 1.00, false}, {64, 512, 1, 8, 1.00, false}, {64, 256, 1, 8, 0.98, false}, {128, 128, 2, 4, 0.97, false},   // eight multiplying waves: 2 x 2 | 2 x 1 blocks each
                                 {64, 192, 2, 2, 0.97, false}, {128, 128, 2, 2, 0.96, false},                                                             // four: 1 x 3 | 2 x 2
                                 {64, 128, 2, 2, 0.92, true}, {32, 128, 1, 4, 0.86, true}, {64, 64, 2, 2, 0.85, true}};                                   // four: 1 x 2 | 1 x 1 | 1 x 1
  double best = -1; tile_cfg_t best_c;
  for (cand_t const &cd : cands) {
    if (cd.small && !old_small) continue;
    tile_cfg_t c; c.BI = cd.bi; c.BJ = cd.bj; c.BK = 16; c.WI = cd.wi; c.WJ = cd.wj; c.MINW = 2; c.SPLITK = 1; c.MT = 32; c.PF = 2; c.SW = 2; c.KHO = 0;
    conv_big_form_t f; if (!conv_big_form(g, c, f) || f.jmode != (patchy ? 7 : 5)) continue;
    long const ti = (g.OC + c.BI - 1) / c.BI, tj = (Nj + c.BJ - 1) / c.BJ, tiles = ti * tj;
    double const pad = ((double)g.OC / (double)(ti * c.BI)) * ((double)Nj / (double)(tj * c.BJ));
    double const deal = ((double)tiles / num_cus) / (double)((tiles + num_cus - 1) / num_cus);
    double const score = cd.base * pad * deal;
    if (score > best) { best = score; best_c = c; }
  }
  if (best < (force ? 0.0 : 0.76)) return old;
  plan_t bp = plan_conv_big(g, best_c);
  // Two-level tiling along the pels (the sgemm path's idea, plan_sgemm_split): when the tiles leave a mostly idle last round, the main tile takes whole rounds of the CUs and
  // a launch of smaller tiles the remaining pels -- every output is still one launch's one fma chain.  AlexNet / NiN conv2 at 256 images: 1460 tiles of 64 x 512 = 5.7 rounds
  // -> 5 rounds + 720 tiles of 64 x 128 (2.8 quarter-size rounds): 6 -> 5.75 tile-times.  Taken when the model says >= 3 % on a launch of >= 50 GFLOP (the second launch
  // costs ~5 us).  BODAHIP_CBIG_SPLIT=off.
  char const *se = getenv("BODAHIP_CBIG_SPLIT");
  static double const split_min = (getenv("BODAHIP_CBIG_SPLIT_MIN_GFLOP") ? atof(getenv("BODAHIP_CBIG_SPLIT_MIN_GFLOP")) : 50.0) * 1e9;   // (tests lower it)
  if (!(se && string(se) == "off") && 2.0 * g.OC * (double)Nj * Kt >= split_min) {
    auto t_of = [&](cand_t const &cd, long n_pels, long &tiles_out) {
      long const ti = (g.OC + cd.bi - 1) / cd.bi, tj = (n_pels + cd.bj - 1) / cd.bj; tiles_out = ti * tj;
      return (double)((tiles_out + num_cus - 1) / num_cus) * cd.bi * cd.bj / cd.base;
    };
    cand_t const *bc = nullptr; for (cand_t const &cd : cands) if (cd.bi == best_c.BI && cd.bj == best_c.BJ && cd.wi == best_c.WI && cd.wj == best_c.WJ) bc = &cd;
    long tl = 0; double const t_single = bc ? t_of(*bc, Nj, tl) : 0;
    double t_best = t_single * 0.97; long best_n1 = 0; cand_t const *m_best = nullptr, *t_best_c = nullptr;
    for (cand_t const &m : cands) {
      if (m.small || m.wi * m.wj != 8) continue;
      tile_cfg_t c1; c1.BI = m.bi; c1.BJ = m.bj; c1.BK = 16; c1.WI = m.wi; c1.WJ = m.wj; c1.MINW = 2; c1.SPLITK = 1; c1.MT = 32; c1.PF = 2; c1.SW = 2; c1.KHO = 0;
      conv_big_form_t f1; if (!conv_big_form(g, c1, f1) || f1.jmode != (patchy ? 7 : 5)) continue;
      long const ti1 = (g.OC + m.bi - 1) / m.bi, tj_all = (Nj + m.bj - 1) / m.bj;
      long const R = (ti1 * tj_all) / num_cus; if (R < 1) continue;
      long const tj1 = std::min(tj_all - 1, (R * num_cus) / ti1); if (tj1 < 1) continue;
      long const n1 = tj1 * m.bj; if (n1 >= Nj) continue;
      double const t_main = (double)((ti1 * tj1 + num_cus - 1) / num_cus) * m.bi * m.bj / m.base;
      for (cand_t const &t : cands) {
        tile_cfg_t c2; c2.BI = t.bi; c2.BJ = t.bj; c2.BK = 16; c2.WI = t.wi; c2.WJ = t.wj; c2.MINW = 2; c2.SPLITK = 1; c2.MT = 32; c2.PF = 2; c2.SW = 2; c2.KHO = 0;
        conv_big_form_t f2; if (!conv_big_form(g, c2, f2) || f2.jmode != (patchy ? 7 : 5)) continue;
        long tt = 0; double const tsum = t_main + t_of(t, Nj - n1, tt);
        if (tsum < t_best) { t_best = tsum; best_n1 = n1; m_best = &m; t_best_c = &t; }
      }
    }
    if (m_best) {
      tile_cfg_t c1; c1.BI = m_best->bi; c1.BJ = m_best->bj; c1.BK = 16; c1.WI = m_best->wi; c1.WJ = m_best->wj; c1.MINW = 2; c1.SPLITK = 1; c1.MT = 32; c1.PF = 2; c1.SW = 2; c1.KHO = 0;
      tile_cfg_t c2 = c1; c2.BI = t_best_c->bi; c2.BJ = t_best_c->bj; c2.WI = t_best_c->wi; c2.WJ = t_best_c->wj;
      bp = plan_conv_big(g, c1); plan_t const tp = plan_conv_big(g, c2);
      bp.split_pels = best_n1; bp.tail_cfg = tp.cfg; bp.tail_defs = tp.defs;
    }
  }
  return bp;
}
static std::vector<char> compile_plan(plan_t const &p, string const &arch, string *log) {
  vect_string opts = p.defs; opts.push_back("-DKNAME=" + p.kname);
  return hiprtc_compile(p.nhwc_rows ? k_src_conv_nhwc_rows_bf16 : p.nhwc_multi ? k_src_conv_nhwc_multi_bf16 : p.nhwc_patch ? k_src_conv_nhwc_patch_bf16 : p.nhwc ? k_src_conv_nhwc_bf16 : p.patch16 ? k_src_conv_patch_bf16 : (p.cbig ? k_src_conv_big_f32 : p.big ? k_src_sgemm_big_f32 : p.fc ? k_src_fc_f32 : p.stream ? (p.quad ? k_src_k1_quad_f32 : k_src_k1_stream_f32) : (p.bf16 ? k_src_gemm_conv_bf16 : k_src_gemm_conv_f32)), p.kname, arch, opts, log, true);
}

// grow-only scratch shared by the split-K slabs and the Winograd-domain tensors (like the reference's cudnn scratch var)
static void ensure_ws(native_kernels_t::impl_t *impl, native_host_t *host, size_t need) {
  if (impl->ws_bytes >= need) return;
  if (host->nh_capturing()) rt_err("graph capture: kernel workspace not allocated yet -- run the call list once before capturing it");
  if (impl->ws) {
    hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize");
    // a captured hipGraph has the scratch pointer frozen into its kernel arguments (split-K slabs, Winograd U/V/M, bf16 filter
    // re-layout, space-to-depth buffers): while any graph is alive the outgrown buffer is retired, not freed, so that replaying
    // an older graph after a later call grew the scratch still works on valid memory
    if (host->nh_live_graphs() > 0) impl->ws_retired.push_back(impl->ws); else hip_err_chk(hipFree(impl->ws), "hipFree");
    impl->ws = nullptr; impl->ws_bytes = 0;
  }
  hip_err_chk(hipMalloc(&impl->ws, need), "hipMalloc(kernel scratch)"); impl->ws_bytes = need;
}
static void setup_splitk(native_kernels_t::impl_t *impl, native_host_t *host, gemm_args_t &ga, tile_cfg_t const &cfg, size_t out_elems) {
  ga.splitk = cfg.SPLITK; ga.kt_per = 0; ga.ws = nullptr; ga.ws_slab = 0;
  if (cfg.SPLITK <= 1) { ga.splitk = 1; return; }
  int const nkt = (ga.K + cfg.BK - 1) / cfg.BK;
  ga.kt_per = (nkt + cfg.SPLITK - 1) / cfg.SPLITK;
  size_t const slab = (out_elems + 3) & ~size_t(3);
  size_t const need = slab * (size_t)cfg.SPLITK * sizeof(float);
  ensure_ws(impl, host, need);
  ga.ws = (float *)impl->ws; ga.ws_slab = (long)slab;
}

// The workspace of a call whose K slices are reduced inside the launch (KSL): one ticket word per tile (zero between launches: the last arriver of a tile resets
// its ticket), then one slab of raw fp32 accumulators per (tile, slice).  It belongs to the CALL (key: its operands and plan), not to the backend's shared scratch:
// calls of an edge-free graph and members of a level set run at the same time.  Allocated and zeroed on the call's first run (not inside a capture), kept until the
// backend goes; `key_ptr` tells calls on the same tensors' shapes apart.
// Per-call workspaces (K slices, K hand-off) are keyed by the call's operands and plan and used to live until the backend went: every init() / release() cycle at a new
// batch size and every tile tried in a sweep left another one behind, up to 4 GiB each (round-5 advisor finding).  Now their sum is bounded: before a new one is made that
// would take the sum past 2 GiB (BODAHIP_CALL_WS_MB), all of them are dropped -- after a stream synchronisation, and only while no captured graph can still point into one;
// a call finds its workspace missing, makes it again and zeroes its tickets, exactly like a first run.
static void call_ws_make_room(native_kernels_t::impl_t *impl, native_host_t *host, size_t need) {
  static size_t const cap = (getenv("BODAHIP_CALL_WS_MB") ? (size_t)atol(getenv("BODAHIP_CALL_WS_MB")) : 2048) << 20;
  if (impl->call_ws_bytes + need <= cap || impl->call_ws_bytes == 0 || impl->call_ws_hold > 0 || host->nh_capturing() || host->nh_live_graphs() > 0) return;
  hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize");
  for (auto it = impl->ktabs.begin(); it != impl->ktabs.end();) {
    if (it->first.compare(0, 4, "ksl:") == 0 || it->first.compare(0, 4, "kho:") == 0) { (void)hipFree(it->second); it = impl->ktabs.erase(it); } else ++it;
  }
  impl->call_ws_bytes = 0;
}
static void setup_ksl(native_kernels_t::impl_t *impl, native_host_t *host, gemm_args_t &ga, tile_cfg_t const &cfg, long nk, void const *key_ptr, char const *what) {
  long const tiles = (long)ga.tiles_i * ga.tiles_j;
  size_t const tick_b = ((size_t)tiles * 4 + 255) & ~size_t(255);
  size_t const slab_b = (size_t)cfg.BI * cfg.BJ * 4;   // (kTI * kTJ * 16 floats x threads = BI x BJ floats)
  size_t const total = tick_b + (size_t)tiles * cfg.SPLITK * slab_b;
  if ((size_t)cfg.SPLITK * slab_b >= 0x7ffffff0ull || total >= (size_t(1) << 32)) unsup_err(string(what) + ": K-slice workspace too large");
  string const key = "ksl:" + std::to_string((uintptr_t)key_ptr) + ":" + std::to_string((uintptr_t)ga.J) + ":" + std::to_string((uintptr_t)ga.I) + ":" + std::to_string(ga.out_coff) + ":" +
                     std::to_string(total) + ":" + cfg.str();
  auto it = impl->ktabs.find(key);
  if (it == impl->ktabs.end()) {
    if (host->nh_capturing()) rt_err("graph capture: the K-slice workspace of this call is not allocated yet -- run the call list once before capturing it");
    void *dev = nullptr;
    call_ws_make_room(impl, host, total);
    hip_err_chk(hipMalloc(&dev, total), "hipMalloc(K-slice workspace)"); impl->call_ws_bytes += total;
    hip_err_chk(hipMemsetAsync(dev, 0, tick_b, host->nh_stream()), "hipMemsetAsync(K-slice tickets)");
    it = impl->ktabs.emplace(key, dev).first;
  }
  ga.splitk = cfg.SPLITK; ga.kt_per = (int)((nk + cfg.SPLITK - 1) / cfg.SPLITK); ga.ws = (float *)it->second; ga.ws_slab = (long)(tick_b / 4);
}

// The workspace of a call that runs as (tile, segment) jobs with sequential K hand-off (gemm_conv_f32.hip -DKHO=1): 16 counter words (job counter, exit counter), one flag
// word per tile, then ONE slab of raw fp32 accumulators per tile (a tile's segments run one after the other).  Like the K-slice workspace it belongs to the call, is
// allocated and zeroed on the call's first run (not inside a capture) and is left zeroed by every launch.
static void setup_kho(native_kernels_t::impl_t *impl, native_host_t *host, gemm_args_t &ga, tile_cfg_t const &cfg, void const *key_ptr) {
  long const tiles = (long)ga.tiles_i * ga.tiles_j, nkt = (ga.K + cfg.BK - 1) / cfg.BK;
  size_t const tick_b = ((size_t)(16 + tiles) * 4 + 255) & ~size_t(255);
  size_t const slab_b = (size_t)cfg.BI * cfg.BJ * 4;
  size_t const total = tick_b + (size_t)tiles * slab_b;
  string const key = "kho:" + std::to_string((uintptr_t)key_ptr) + ":" + std::to_string((uintptr_t)ga.J) + ":" + std::to_string((uintptr_t)ga.I) + ":" + std::to_string(ga.out_coff) + ":" +
                     std::to_string(total) + ":" + cfg.str();
  auto it = impl->ktabs.find(key);
  if (it == impl->ktabs.end()) {
    if (host->nh_capturing()) rt_err("graph capture: the K hand-off workspace of this call is not allocated yet -- run the call list once before capturing it");
    void *dev = nullptr;
    call_ws_make_room(impl, host, total);
    hip_err_chk(hipMalloc(&dev, total), "hipMalloc(K hand-off workspace)"); impl->call_ws_bytes += total;
    hip_err_chk(hipMemsetAsync(dev, 0, tick_b, host->nh_stream()), "hipMemsetAsync(K hand-off counters)");
    it = impl->ktabs.emplace(key, dev).first;
  }
  ga.splitk = cfg.KHO; ga.kt_per = (int)((nkt + cfg.KHO - 1) / cfg.KHO); ga.ws = (float *)it->second; ga.ws_slab = (long)(tick_b / 4);
}
// persistent launch of a K hand-off call: as many workgroups as are resident at once (more would only find the job queue empty), never more than there are jobs
static uint32_t launch_kho(native_host_t *host, kernel_t &k, gemm_args_t &a, tile_cfg_t const &c) {
  if (!k.occ) {
    int nb = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k.func, c.threads(), 0) != hipSuccess || nb < 1) nb = 1;
    k.occ = nb;
  }
  if (char const *e = getenv("BODAHIP_KHO_OCC")) k.occ = std::max(1, atoi(e));
  long const jobs = (long)a.tiles_i * a.tiles_j * a.splitk;
  uint32_t const grid = (uint32_t)std::min<long>(jobs, (long)host->nh_num_cus() * k.occ);
  void *params[] = {&a};
  hip_err_chk(host->nh_launch(k.func, grid, 1, (uint32_t)c.threads(), params), "hipModuleLaunchKernel(native, K hand-off)");
  return grid;
}

static kernel_t &get_reduce_kernel(native_kernels_t::impl_t *impl, native_host_t *host, bool epi, bool relu) {
  plan_t p; p.kname = "bodahip_splitk_reduce";
  p.defs = {"-DREDUCE_ONLY=1", string("-DRED_EPI=") + (epi ? "1" : "0"), string("-DRED_RELU=") + (relu ? "1" : "0")};
  return get_kernel(impl, host, p);
}
static void reduce_splitk(native_kernels_t::impl_t *impl, native_host_t *host, gemm_args_t const &ga, long n, bool epi, bool relu, int chan_stride, int n_chan) {
  kernel_t &k = get_reduce_kernel(impl, host, epi, relu);
  float const *ws = ga.ws; long ws_slab = ga.ws_slab; int splitk = ga.splitk; float *D = ga.D; float const *bias = ga.bias;
  void *params[] = {&ws, &ws_slab, &splitk, &D, &n, &bias, &chan_stride, &n_chan};
  long const groups = (n / 4 + 255) / 256;
  uint32_t const grid = (uint32_t)std::max<long>(1, std::min<long>(groups, 2048));
  hip_err_chk(host->nh_launch(k.func, grid, 1, 256, params), "hipModuleLaunchKernel(splitk_reduce)");
}

// per-k tables of the im2col gather (three arrays of n ints): offset of (in_chan,ky,kx) inside one image | ky | kx.
// Rows k >= K carry ky = 2^30 so that they fail the kernel's row-range test (zero contribution); n is padded so that any
// K-tile read stays inside the table.
struct ktab_t { void *d; int n; };
static ktab_t get_ktab(native_kernels_t::impl_t *impl, native_host_t *host, conv_geom_t const &g) {
  string const key = std::to_string(g.C) + "," + std::to_string(g.H) + "," + std::to_string(g.W) + "," + std::to_string(g.KH) + "," + std::to_string(g.KW);
  long const K = (long)g.C * g.KH * g.KW, n = ((K + 255) / 256 + 1) * 256;
  auto it = impl->ktabs.find(key);
  if (it != impl->ktabs.end()) return ktab_t{it->second, (int)n};
  if (host->nh_capturing()) rt_err("graph capture: gather table not built yet -- run the call list once before capturing it");
  std::vector<int> h((size_t)n * 3);
  for (long k = 0; k < n; ++k) {
    if (k < K) { long const ic = k / (g.KH * g.KW), rem = k % (g.KH * g.KW), ky = rem / g.KW, kx = rem % g.KW;
      h[k] = (int)((ic * g.H + ky) * g.W + kx); h[n + k] = (int)ky; h[2 * n + k] = (int)kx; }
    else { h[k] = 0; h[n + k] = 1 << 30; h[2 * n + k] = 0; }
  }
  void *d = nullptr;
  hip_err_chk(hipMalloc(&d, h.size() * sizeof(int)), "hipMalloc(ktab)");
  hip_err_chk(hipMemcpyAsync(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(ktab)");
  hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize");
  impl->ktabs.emplace(key, d);
  return ktab_t{d, (int)n};
}

// per-row tables of the row gather (two arrays of n ints): element offset of (in_chan,ky,kx=0) inside one image | ky.
// Rows >= C*KH carry ky = 2^30 (fail the kernel's row-range test -> zero contribution).
static ktab_t get_rtab(native_kernels_t::impl_t *impl, native_host_t *host, conv_geom_t const &g) {
  string const key = "rows:" + std::to_string(g.C) + "," + std::to_string(g.H) + "," + std::to_string(g.W) + "," + std::to_string(g.KH);
  long const R = (long)g.C * g.KH, n = ((R + 63) / 64 + 1) * 64;
  auto it = impl->ktabs.find(key);
  if (it != impl->ktabs.end()) return ktab_t{it->second, (int)n};
  if (host->nh_capturing()) rt_err("graph capture: gather table not built yet -- run the call list once before capturing it");
  std::vector<int> h((size_t)n * 2);
  for (long r = 0; r < n; ++r) {
    if (r < R) { long const ic = r / g.KH, ky = r % g.KH; h[r] = (int)((ic * g.H + ky) * g.W); h[n + r] = (int)ky; }
    else { h[r] = 0; h[n + r] = 1 << 30; }
  }
  void *d = nullptr;
  hip_err_chk(hipMalloc(&d, h.size() * sizeof(int)), "hipMalloc(rtab)");
  hip_err_chk(hipMemcpyAsync(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(rtab)");
  hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize");
  impl->ktabs.emplace(key, d);
  return ktab_t{d, (int)n};
}

static string tune_of(native_kernels_t::impl_t *impl, char const *key) { auto t = impl->tune.find(key); return (t == impl->tune.end()) ? string() : t->second; }

// Two-level tiling for the large fp32 sgemms.  A grid of 256x256 tiles runs one workgroup per CU, so a tile count that is not a
// multiple of the CU count ends in a mostly idle round (7168^3: 784 tiles = 3 rounds + 16 tiles, measured 102 TF/s against 136 at
// 8192^3 = 4 rounds exactly); 128x128 tiles quantise finer but run ~8 % slower per flop.  The split gives the first `m_main` rows of c
// (whole rounds of 256x256 tiles) to the large tile and the remaining rows to a second launch of small tiles that fills the chip for
// a fraction of a tile-time.  Each output is still ONE ascending-k chain in one thread: results are bit-identical to the unsplit launch.
// Time model (units: one 256x256 tile on one CU at rate 1): a launch of n tiles of relative area a, s workgroups per CU, relative
// rate r costs floor(n / (cus*s)) * s*a/r for its full rounds plus k*a/(r*eff) for the last, k = ceil(rest / cus) workgroups on the
// busiest CU, eff = 0.6 for one of two co-resident workgroups running alone (its MFMAs no longer hide the other's barriers).
struct sgemm_split_t { uint32_t m_main = 0; string tail_tile; double t_single = 0, t_split = 0; };
static double launch_model(long n, double a, int s, double r, int cus) {
  long const per = (long)cus * s, full = n / per, rest = n - full * per;
  double t = (double)full * s * a / r;
  if (rest) { long const k = (rest + cus - 1) / cus; double const eff = (k >= s) ? 1.0 : 0.6 + 0.4 * (double)(k - 1) / (double)(s - 1); t += (double)k * a / (r * eff); }
  return t;
}
static char const *const kBigTile = "256x256x16x2x4x1x1x32x2";
// (the rest launch's 64 x 64 tiles: the staging-wave kernel's form since round 6 -- 5120^3 133.1 -> 133.6, 6144^3 136.7 -> 137.3, 7168^3 139.8 -> 140.4 TF/s in the list)
static char const *const kTail64 = getenv("BODAHIP_SGEMM_TAIL64") ? getenv("BODAHIP_SGEMM_TAIL64") : ((getenv("BODAHIP_SGEMM_BIG") && string(getenv("BODAHIP_SGEMM_BIG")) == "off") || getenv("BODAHIP_NO_SGEMM_STG64")) ? "64x64x32x2x2x2x1x32x2" : kStg64;
static sgemm_split_t plan_sgemm_split(uint32_t M, uint32_t N, uint32_t K, int num_cus) {
  sgemm_split_t sp;
  if (getenv("BODAHIP_NO_SGEMM_SPLIT") || M % 4 || N % 4 || K < 512 || M < 1024 || N < 1024) return sp;
  long const ti = (M + 255) / 256, tj = (N + 255) / 256;
  if (ti * tj < num_cus) return sp;
  double const r_big = getenv("BODAHIP_SGEMM_RBIG") ? atof(getenv("BODAHIP_SGEMM_RBIG")) : 1.04, r_mid = 1.0, r_small = 0.93;
  auto small_n = [&](uint32_t rows, int b) { return (long)((rows + b - 1) / b) * (long)((N + b - 1) / b); };
  sp.t_single = std::min(launch_model(ti * tj, 1.0, 1, r_big, num_cus), launch_model(small_n(M, 128), 0.25, 2, r_mid, num_cus));
  double best = sp.t_single * 0.975;   // (a split must buy at least 2.5 %)
  for (long R = 1; R < ti; ++R) {
    uint32_t const m_main = (uint32_t)(R * 256), rows = M - m_main;
    double const tm = launch_model(R * tj, 1.0, 1, r_big, num_cus);
    double const t128 = tm + launch_model(small_n(rows, 128), 0.25, 2, r_mid, num_cus) + 0.004;
    double const t64 = tm + launch_model(small_n(rows, 64), 0.0625, 2, r_small, num_cus) + 0.004;
    char const *const force = getenv("BODAHIP_SGEMM_SPLIT_TAIL");   // (experiments: "128" | "64")
    if (force && atoi(force) == 128) { if (t128 < best) { best = t128; sp.m_main = m_main; sp.tail_tile = "128x128x16x2x2x2"; } continue; }
    if (force && atoi(force) == 64) { if (t64 < best) { best = t64; sp.m_main = m_main; sp.tail_tile = kTail64; } continue; }
    if (t128 < best) { best = t128; sp.m_main = m_main; sp.tail_tile = "128x128x16x2x2x2"; }
    if (t64 < best) { best = t64; sp.m_main = m_main; sp.tail_tile = kTail64; }
  }
  sp.t_split = best;
  return sp;
}

// the 256 x 128 form of the staging-wave kernel where its tiles deal out in (nearly) whole rounds -- see sgemm(); "" = not here
static string sgemm_wide_tile(uint32_t M, uint32_t N, uint32_t K, long cus) {
  if (M % 4 || N % 4 || K < 512 || getenv("BODAHIP_NO_SGEMM_256X128")) return string();
  if (char const *e = getenv("BODAHIP_SGEMM_BIG")) { if (string(e) == "off") return string(); }   // (the x3x4 tile is a form of the staging-wave kernel only: with the kernel switched off the general kernel's own choice stands)
  long const t256 = (long)((M + 255) / 256) * ((N + 255) / 256), t128 = (long)((M + 255) / 256) * ((N + 127) / 128);
  double const eff = (double)t128 / (double)(((t128 + cus - 1) / cus) * cus);
  return (t256 >= cus && eff >= 0.95) ? string("256x128x8x3x4x1") : string();
}

// Round 6: the two-level tiling generalised to guillotine cuts -- a list of rectangles [m0, m0 + rows) x [n0, n0 + cols) of c, each ONE launch of one tile form (operands
// and output addressed through pointer offsets: a and b are k-major, a sub-rectangle is a column range of both).  Every output still belongs to exactly one launch and is
// one ascending-k chain: bit-identical to the single launch.  10240^3: 3200 tiles of 256 x 128 = 12.5 rounds of 256 CUs -> rows < 8192 (10 rounds) + the last 2048 rows'
// first 8192 columns (2 rounds) + a 2048 x 2048 corner on 64 x 64 tiles (1024 = one round of four per CU).
struct sgemm_part_t { uint32_t m0 = 0, rows = 0, n0 = 0, cols = 0; string tile; };
static bool parse_parts_env(uint32_t M, uint32_t N, uint32_t K, std::vector<sgemm_part_t> &out) {   // experiments: BODAHIP_SGEMM_PARTS="<size>:m0,rows,n0,cols,tile/m0,rows,...;<size>:..."
  char const *e = getenv("BODAHIP_SGEMM_PARTS"); if (!e || M != N || N != K) return false;
  string const v = e, key = std::to_string(M) + ":"; size_t const at = (";" + v).find(";" + key); if (at == string::npos) return false;
  size_t const b = at + key.size(), en = v.find(';', b); string const spec = v.substr(b, en == string::npos ? string::npos : en - b);
  std::istringstream is(spec); string one;
  while (std::getline(is, one, '/')) {
    sgemm_part_t q; char tl[128];
    if (sscanf(one.c_str(), "%u,%u,%u,%u,%127s", &q.m0, &q.rows, &q.n0, &q.cols, tl) != 5) rt_err("bad BODAHIP_SGEMM_PARTS entry '" + one + "'");
    q.tile = tl; out.push_back(q);
  }
  uint64_t area = 0; for (auto const &q : out) { if (q.m0 % 4 || q.n0 % 4 || q.m0 + q.rows > M || q.n0 + q.cols > N || !q.rows || !q.cols) rt_err("bad BODAHIP_SGEMM_PARTS rectangle"); area += (uint64_t)q.rows * q.cols; }
  if (area != (uint64_t)M * N) rt_err("BODAHIP_SGEMM_PARTS: the rectangles do not add up to c");
  return true;
}

// The planner of those rectangles.  What the launches cost was measured (tools/sgemm_rounds_probe.py, MI355X, K = 6144): the 256 x 256 form runs ONE workgroup per CU --
// n tiles take ceil(n / 256) rounds of 1.434 ms; of the 256 x 128 form TWO share a CU (50 KB of LDS, 6 waves per SIMD) -- ceil(n / 512) double rounds of 1.416 ms, and the
// dispatcher packs a last partial round two-per-CU onto the CUs that free up first instead of spreading it (768 tiles = three whole rounds of 256 take FOUR rounds' time:
// 2.90 ms; 1280: 4.29 ms); only a launch that fits the chip at once is spread (256 tiles: 0.883 ms).  The 64 x 64 form runs four per CU (a round of 1024: 0.274 of a
// 256 x 256 round).  So: a cost per form = rounds of (CUs x workgroups per CU) tiles, a partial round costs a whole one unless the launch has no full round at all; the
// search tries guillotine cuts (rows | columns, the first part one launch of a large form, the remainder cut again: four launches at most) and takes what the model says
// is >= 2 % ahead of the single launch / the row split above.  10240^3 140.4 -> 146.0 TF/s, 5120^3 135.2 -> 139.5 in the list (tools/sgemm_parts_ab.sh).
struct sgemm_form_t { char const *tile; int bi, bj, s; double r, lone; };   // s workgroups per CU, relative rate r, efficiency of ONE workgroup alone on a CU against its share of a full CU
static sgemm_form_t const kFormQ = {"256x256x16x2x4x1x1x32x2", 256, 256, 1, 1.0, 1.0}, kFormW = {"256x128x8x3x4x1", 256, 128, 2, 1.0127, 0.82}, kFormS = {kStg64, 64, 64, 4, 0.913, 0.5};
static double form_cost(sgemm_form_t const &f, uint32_t rows, uint32_t cols, int cus) {
  long const n = (long)((rows + f.bi - 1) / f.bi) * ((cols + f.bj - 1) / f.bj), per = (long)cus * f.s;
  double const a = (double)f.bi * f.bj / 65536.0, round = f.s * a / f.r;
  if (n <= per) { long const k = (n + cus - 1) / cus; double const eff = (f.s == 1 || k >= f.s) ? 1.0 : f.lone + (1.0 - f.lone) * (double)(k - 1) / (double)(f.s - 1);
                  return (double)k * a / (f.r * eff); }   // (a launch the chip takes at once is spread over the CUs)
  if (f.s >= 4) return (double)n / (double)per * round;    // four staggered workgroups per CU: no rounds to speak of -- 5120^3's rest launch of 2560 tiles (2.5 x 1024) took 2.5 x a 1024-tile launch
  return (double)((n + per - 1) / per) * round;
}
struct sgemm_parts_plan_t { std::vector<sgemm_part_t> parts; double t = 1e30; };
static sgemm_parts_plan_t plan_parts_rec(uint32_t m0, uint32_t rows, uint32_t n0, uint32_t cols, int cus, double ovh, int depth) {
  sgemm_parts_plan_t best;
  for (sgemm_form_t const *f : {&kFormQ, &kFormW, &kFormS}) {
    double const t = form_cost(*f, rows, cols, cus) + ovh;
    if (t < best.t) { best.t = t; sgemm_part_t q; q.m0 = m0; q.rows = rows; q.n0 = n0; q.cols = cols; q.tile = f->tile; best.parts = {q}; }
  }
  if (depth <= 0) return best;
  for (sgemm_form_t const *f : {&kFormQ, &kFormW}) {
    long const per = (long)cus * f->s, tj = (cols + f->bj - 1) / f->bj, ti = (rows + f->bi - 1) / f->bi;
    for (long R = 1; R < ti; ++R) {     // rows [0, R tile rows) x all columns: one launch of form f -- whole rounds only (anything else is what the remainder's own search covers)
      if ((R * tj) % per) continue;
      uint32_t const r1 = (uint32_t)(R * f->bi);
      sgemm_parts_plan_t rest = plan_parts_rec(m0 + r1, rows - r1, n0, cols, cus, ovh, depth - 1);
      double const t = form_cost(*f, r1, cols, cus) + ovh + rest.t;
      if (t < best.t) { best.t = t; sgemm_part_t q; q.m0 = m0; q.rows = r1; q.n0 = n0; q.cols = cols; q.tile = f->tile; best.parts = {q}; best.parts.insert(best.parts.end(), rest.parts.begin(), rest.parts.end()); }
    }
    for (long C = 1; C < tj; ++C) {     // all rows x columns [0, C tile columns)
      if ((C * ti) % per) continue;
      uint32_t const c1 = (uint32_t)(C * f->bj);
      sgemm_parts_plan_t rest = plan_parts_rec(m0, rows, n0 + c1, cols - c1, cus, ovh, depth - 1);
      double const t = form_cost(*f, rows, c1, cus) + ovh + rest.t;
      if (t < best.t) { best.t = t; sgemm_part_t q; q.m0 = m0; q.rows = rows; q.n0 = n0; q.cols = c1; q.tile = f->tile; best.parts = {q}; best.parts.insert(best.parts.end(), rest.parts.begin(), rest.parts.end()); }
    }
  }
  return best;
}
// "" = no decomposition beats what sgemm() would do anyway
static std::vector<sgemm_part_t> plan_sgemm_parts(uint32_t M, uint32_t N, uint32_t K, int cus) {
  std::vector<sgemm_part_t> none;
  if (getenv("BODAHIP_NO_SGEMM_PARTS") || getenv("BODAHIP_NO_SGEMM_SPLIT") || M % 4 || N % 4 || K < 512 || M < 1024 || N < 1024 || (long)((M + 255) / 256) * ((N + 255) / 256) < cus) return none;
  if (char const *e = getenv("BODAHIP_SGEMM_BIG")) { if (string(e) == "off") return none; }
  if (M % 64 || N % 64) return none;                      // (cuts at multiples of the forms' tiles; the last parts reach the edges)
  double const ovh = 25.7 / (double)K;                    // ~6 us per launch in 256 x 256 rounds of this K
  sgemm_parts_plan_t pp = plan_parts_rec(0, M, 0, N, cus, ovh, 3);
  // ... and one free cut on top (neither side a launch of its own; coarse positions): 5120^3 = rows < 4096 { 4096 x 4096 on 256 x 256 tiles (256 = one round) | the other
  // 1024 columns on 64 x 64 } over the last 1024 rows on 64 x 64
  for (uint32_t R = 1024; R < M; R += 1024) {
    sgemm_parts_plan_t const t1 = plan_parts_rec(0, R, 0, N, cus, ovh, 2), t2 = plan_parts_rec(R, M - R, 0, N, cus, ovh, 2);
    if (t1.t + t2.t < pp.t) { pp.t = t1.t + t2.t; pp.parts = t1.parts; pp.parts.insert(pp.parts.end(), t2.parts.begin(), t2.parts.end()); }
  }
  for (uint32_t C = 1024; C < N; C += 1024) {
    sgemm_parts_plan_t const t1 = plan_parts_rec(0, M, 0, C, cus, ovh, 2), t2 = plan_parts_rec(0, M, C, N - C, cus, ovh, 2);
    if (t1.t + t2.t < pp.t) { pp.t = t1.t + t2.t; pp.parts = t1.parts; pp.parts.insert(pp.parts.end(), t2.parts.begin(), t2.parts.end()); }
  }
  if (pp.parts.size() < 2) return none;
  // what sgemm() does without it: the 256 x 128 single launch where it deals out, else the row split / the single launch (same cost model)
  double t_now = std::min(form_cost(kFormQ, M, N, cus), form_cost(kFormW, M, N, cus)) + ovh;
  sgemm_split_t const sp = plan_sgemm_split(M, N, K, cus);
  if (sp.m_main && sgemm_wide_tile(M, N, K, cus).empty()) t_now = std::min(t_now, form_cost(kFormQ, sp.m_main, N, cus) + form_cost(kFormS, M - sp.m_main, N, cus) + 2 * ovh);
  return (pp.t < 0.98 * t_now) ? pp.parts : none;
}

static void sgemm_parts(native_kernels_t *nk, native_kernels_t::impl_t *impl, native_host_t *host, float const *a, float const *b, float *c, uint32_t M, uint32_t N, uint32_t K, std::vector<sgemm_part_t> const &parts) {
  if ((uint64_t)K * M * 4 > 0x80000000ull || (uint64_t)K * N * 4 > 0x80000000ull || (uint64_t)M * N * 4 >= 0x7ffffff0ull) unsup_err("hip_sgemm: operands / c of 2 GiB or more are not supported (32-bit buffer offsets)");
  uint32_t grid = 0; bool first = true;
  for (sgemm_part_t const &q : parts) {
    plan_t const p = plan_sgemm(q.rows, q.cols, K, host->nh_num_cus(), q.tile, false);
    kernel_t &k = get_kernel(impl, host, p);
    gemm_args_t ga; memset(&ga, 0, sizeof(ga));
    ga.I = a + q.m0; ga.J = b + q.n0; ga.D = c + (size_t)q.m0 * N + q.n0; ga.bias = nullptr;
    ga.Mi = (int)q.rows; ga.Nj = (int)q.cols; ga.K = (int)K; ga.ldI = (int)M; ga.ldJ = (int)N; ga.ldD = (int)N;
    ga.I_bytes = (unsigned)(((uint64_t)K * M - q.m0) * 4); ga.J_bytes = (unsigned)(((uint64_t)K * N - q.n0) * 4); ga.D_bytes = (unsigned)((((uint64_t)q.rows - 1) * N + q.cols) * 4);
    ga.tiles_i = (int)((q.rows + p.cfg.BI - 1) / p.cfg.BI); ga.tiles_j = (int)((q.cols + p.cfg.BJ - 1) / p.cfg.BJ); ga.splitk = 1;
    launch(host, k, ga, p.cfg);
    grid += (uint32_t)ga.tiles_i * ga.tiles_j;
    if (first) { nk->last_launch.kernel = p.kname; nk->last_launch.cfg = p.cfg; nk->last_launch.block = p.cfg.threads(); first = false; }
  }
  nk->last_launch.grid = grid; nk->last_launch.flops = 2.0 * M * N * K; nk->last_launch.algo_bytes = 4.0 * ((double)K * M + (double)K * N + (double)M * N);
}

void native_kernels_t::sgemm(float const *a, float const *b, float *c, uint32_t M, uint32_t N, uint32_t K, bool bf16, bool half) {
  if (!M || !N) return;
  size_t const esz = half ? 2 : 4;   // half: a / b / c stored as IEEE half, fp32 math (the reference's 16-bit-storage sgemm, src/cnn_codegen.cc:440-449)
  if (half && bf16) unsup_err("hip_sgemm_bf16: half-typed tensors are not supported (bf16 OPERANDS are made from float tensors)");
  if (!K) { hip_err_chk(hipMemsetAsync(c, 0, (size_t)M * N * esz, host->nh_stream()), "hipMemsetAsync"); return; }
  if (M > 0x7fffffffu || N > 0x7fffffffu || K > 0x7fffffffu) unsup_err("hip_sgemm: dims exceed int32");
  string tile_for;   // experiments: BODAHIP_SGEMM_TILE_FOR="4096=256x128x8x3x4x1;5120=128x128x8x3x4x1": the tile of the square sgemm of that size (no two-level split)
  if (char const *e = getenv("BODAHIP_SGEMM_TILE_FOR")) {
    string const key = std::to_string(M) + "="; string const v = e; size_t const at = (";" + v).find(";" + key);
    if (M == N && N == K && at != string::npos) { size_t const b = at + key.size(), en = v.find(';', b); tile_for = v.substr(b, en == string::npos ? string::npos : en - b); }
  }
  // Round 5: where 256 x 128 tiles of the staging-wave kernel deal out over the CUs in (nearly) whole rounds they run ahead of 256 x 256 -- same kernel, 4 x 2 multiplying
  // waves of 64 x 64; measured in the layer sequence of sgemm-ops-full, three alternating repetitions on one box (tools/sgemm_policy_ab.sh, TF/s): 4096^3 141.2 -> 142.7,
  // 8192^3 143.6 -> 144.9, 12288^3 144.2 -> 145.5, and 10240^3 (3200 tiles = 12.5 rounds, against 6 rounds of 256 x 256 + a tail launch) 136.5 -> 139.7; the whole list
  // 139.6 -> 141.2.  Where the last round is emptier (5120 / 6144 / 7168: 0.78-0.90 full) the two-level split below stays ahead.  Bit-identical either way.
  if (!bf16 && !half && tune_of(impl, "sgemm_tile").empty() && tile_for.empty() && M % 4 == 0 && N % 4 == 0) {
    std::vector<sgemm_part_t> parts;
    if (!parse_parts_env(M, N, K, parts)) parts = plan_sgemm_parts(M, N, K, host->nh_num_cus());
    if (!parts.empty()) { sgemm_parts(this, impl, host, a, b, c, M, N, K, parts); return; }
  }
  if (!bf16 && !half && tune_of(impl, "sgemm_tile").empty() && tile_for.empty()) tile_for = sgemm_wide_tile(M, N, K, host->nh_num_cus());
  if (!bf16 && !half && tune_of(impl, "sgemm_tile").empty() && tile_for.empty()) {
    sgemm_split_t const sp = plan_sgemm_split(M, N, K, host->nh_num_cus());
    if (sp.m_main) {
      if ((uint64_t)K * M * 4 > 0x80000000ull || (uint64_t)K * N * 4 > 0x80000000ull || (uint64_t)M * N * 4 >= 0x7ffffff0ull) unsup_err("hip_sgemm: operands / c of 2 GiB or more are not supported (32-bit buffer offsets)");
      uint32_t grid = 0;
      for (int part = 0; part < 2; ++part) {
        uint32_t const m0 = part ? sp.m_main : 0, rows = part ? M - sp.m_main : sp.m_main;
        plan_t const p = plan_sgemm(rows, N, K, host->nh_num_cus(), part ? sp.tail_tile : string(kBigTile), false);
        kernel_t &k = get_kernel(impl, host, p);
        gemm_args_t ga; memset(&ga, 0, sizeof(ga));
        ga.I = a + m0; ga.J = b; ga.D = c + (size_t)m0 * N; ga.bias = nullptr;
        ga.Mi = (int)rows; ga.Nj = (int)N; ga.K = (int)K; ga.ldI = (int)M; ga.ldJ = (int)N; ga.ldD = (int)N;
        ga.I_bytes = (unsigned)(((uint64_t)K * M - m0) * 4); ga.J_bytes = (unsigned)((uint64_t)K * N * 4); ga.D_bytes = (unsigned)((uint64_t)rows * N * 4);
        ga.tiles_i = (int)((rows + p.cfg.BI - 1) / p.cfg.BI); ga.tiles_j = (int)((N + p.cfg.BJ - 1) / p.cfg.BJ); ga.splitk = 1;
        launch(host, k, ga, p.cfg);
        grid += (uint32_t)ga.tiles_i * ga.tiles_j;
        if (!part) { last_launch.kernel = p.kname; last_launch.cfg = p.cfg; last_launch.block = p.cfg.threads(); }
      }
      last_launch.grid = grid; last_launch.flops = 2.0 * M * N * K; last_launch.algo_bytes = 4.0 * ((double)K * M + (double)K * N + (double)M * N);
      return;
    }
  }
  plan_t p = plan_sgemm(M, N, K, host->nh_num_cus(), tile_for.empty() ? tune_of(impl, "sgemm_tile") : tile_for, bf16, 1, !half);
  if (half) {
    if (p.cfg.SPLITK > 1) unsup_err("hip_sgemm: split-K tiles are not supported for half-typed tensors");
    p.kname = "bodahip_sgemm_f16s"; p.defs.push_back("-DHALF=1");
  }
  tile_cfg_t const &cfg = p.cfg;
  kernel_t &k = get_kernel(impl, host, p);
  gemm_args_t ga; memset(&ga, 0, sizeof(ga));
  ga.I = a; ga.J = b; ga.D = c; ga.bias = nullptr;
  ga.Mi = (int)M; ga.Nj = (int)N; ga.K = (int)K; ga.ldI = (int)M; ga.ldJ = (int)N; ga.ldD = (int)N;
  if ((uint64_t)K * M * 4 > 0x80000000ull || (uint64_t)K * N * 4 > 0x80000000ull) unsup_err("hip_sgemm: operands larger than 2 GiB are not supported (32-bit buffer offsets)");
  ga.I_bytes = (unsigned)((uint64_t)K * M * esz); ga.J_bytes = (unsigned)((uint64_t)K * N * esz);
  // (outputs share the operands' 2 GiB limit: the epilogue masks lanes past the last column with byte offset 0x80000000, which the
  //  buffer range check only drops while the output itself ends below that offset)
  if ((uint64_t)M * N * 4 >= 0x7ffffff0ull) unsup_err("hip_sgemm: c of 2 GiB or more is not supported (32-bit store offsets, masked lanes use offset 2^31)");
  ga.D_bytes = (unsigned)((uint64_t)M * N * esz);
  ga.tiles_i = (int)((M + cfg.BI - 1) / cfg.BI); ga.tiles_j = (int)((N + cfg.BJ - 1) / cfg.BJ);
  setup_splitk(impl, host, ga, cfg, (size_t)M * N);
  launch(host, k, ga, cfg);
  if (cfg.SPLITK > 1) reduce_splitk(impl, host, ga, (long)M * N, false, false, 1, 1);
  last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)ga.tiles_i * ga.tiles_j * cfg.SPLITK; last_launch.block = cfg.threads();
  last_launch.flops = 2.0 * M * N * K; last_launch.algo_bytes = (double)esz * ((double)K * M + (double)K * N + (double)M * N);
}

// patch16 launch: filters re-laid-out once per call into F'[group][tap][out_chan][8] bf16 (scratch at ws_off), then the patch kernel
static void launch_patch16(native_kernels_t::impl_t *impl, native_host_t *host, plan_t const &p, kernel_t &k, gemm_args_t &ga, float const *filts,
                           conv_geom_t const &g, size_t ws_off) {
  int const taps = g.KH * g.KW, ncg = (g.C + 7) / 8;
  size_t const fbytes = (size_t)ncg * taps * g.OC * 16;
  if (fbytes >= 0x7ffffff0ull) unsup_err("hip_conv_bf16: re-laid-out filters of 2 GiB or more");
  if (impl->ws_bytes < ws_off + fbytes) ensure_ws(impl, host, ws_off + fbytes);
  plan_t fp; fp.patch16 = true; fp.bf16 = true; fp.kname = "bodahip_filt_bf16"; fp.defs = {"-DFILT_ONLY=1"};
  kernel_t &fk = get_kernel(impl, host, fp);
  gemm_args_t fa; memset(&fa, 0, sizeof(fa));
  fa.I = filts; fa.D = (float *)((char *)impl->ws + ws_off); fa.Mi = g.OC; fa.C = g.C; fa.K = taps;
  void *fparams[] = {&fa};
  long const nchunks = (long)ncg * taps * g.OC;
  hip_err_chk(host->nh_launch(fk.func, (uint32_t)((nchunks + 255) / 256), 1, 256, fparams), "hipModuleLaunchKernel(filt_bf16)");
  ga.I = (float const *)((char *)impl->ws + ws_off); ga.I_bytes = (unsigned)fbytes;
  void *params[] = {&ga};
  hip_err_chk(host->nh_launch(k.func, (uint32_t)(ga.tiles_i * ga.tiles_j), 1, (uint32_t)p.cfg.threads(), params), "hipModuleLaunchKernel(conv_patch_bf16)");
}

// bf16 conv1-type layers (stride s in both axes, few input channels): space-to-depth front end + the patch kernel (kernels/conv_patch_bf16.hip)
struct s2d_args_t { float const *src; float *dst; int B, C, H, W, OC, KH, KW; int S, C2, H2, W2; int oy, ox; int filt; int mode, c2_lo; };
static bool s2d_geom(conv_geom_t const &g, conv_geom_t &g2, int &pry, int &prx) {
  if (getenv("BODAHIP_NO_S2D")) return false;
  int const s = g.SY;
  if (!(s >= 2 && s <= 4 && g.SX == s && g.KH > s && g.KW > s && g.C * s * s <= 64 && g.C <= 8)) return false;
  pry = (g.PY + s - 1) / s * s; prx = (g.PX + s - 1) / s * s;
  int const khb = (g.KH + (pry - g.PY) + s - 1) / s, kwb = (g.KW + (prx - g.PX) + s - 1) / s;
  if (khb * kwb > 16) return false;
  g2 = g; g2.C = (g.C * s * s + 7) / 8 * 8; if (g2.C < 16) g2.C = 16;
  g2.KH = khb; g2.KW = kwb; g2.SY = 1; g2.SX = 1; g2.PY = 0; g2.PX = 0; g2.H = g.OH + khb - 1; g2.W = g.OW + kwb - 1;
  return true;
}

// ---- F(2x2,3x3) Winograd path (kernels/winograd_f32.hip): opt-in, tune key conv_algo = "winograd" -----------------------------------
struct wino_args_t { // must match kernels/winograd_f32.hip
  float const *in; float const *filts; float const *bias; float *out;
  float *U; float *V; float *M;
  int B0, Bc;
  int C, H, W, OC, OH, OW;
  int TH, TW, Tc;
  int PY, PX, relu;
  int out_ctot, out_coff;
};
// conv_algo = "winograd_all": every 3x3 / stride-1 convolution; "winograd": only where it measured ahead of the direct kernel -- with
// fewer than 96 input channels the transform-domain sgemms are too short (K = in_chan) and the streaming transforms dominate
// (ResNet res2 64->64 @56x56: 228 vs 179 us direct; res3 128->128 @28x28: 145 vs 172; res4 256->256 @14x14: 108 vs 185)
static bool winograd_applies(conv_geom_t const &g, string const &algo) {
  if (!(g.KH == 3 && g.KW == 3 && g.SY == 1 && g.SX == 1)) return false;
  return algo == "winograd_all" || (algo == "winograd" && g.C >= 96 && g.OC >= 64);
}
// images per chunk of the three-kernel Winograd pipeline (see conv_winograd)
static long wino_chunk_imgs(conv_geom_t const &g) {
  int const tpi = ((g.OH + 1) / 2) * ((g.OW + 1) / 2);
  size_t const per_img = (size_t)16 * (g.C + g.OC) * tpi * 4;
  size_t chunk_mb = 1024; if (char const *e = getenv("BODAHIP_WINO_CHUNK_MB")) chunk_mb = (size_t)std::max(1, atoi(e));
  long Bc = std::max<long>(1, std::min<long>(g.B, (long)((chunk_mb << 20) / per_img)));
  if (Bc >= 4) Bc &= ~3l;
  return Bc;
}
void native_kernels_t::conv_winograd(float const *filts, float const *biases, float const *in, float *out, conv_geom_t const &g, int out_ctot, int out_coff) {
  if (!impl->wino_mod) {
    if (host->nh_capturing()) rt_err("graph capture: Winograd transform kernels are not compiled yet -- run the call list once before capturing it");
    string log;
    vect_string wdefs; if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) wdefs.push_back(tok); } // (experiments)
    std::vector<char> code = hiprtc_compile(k_src_winograd_f32, "bodahip_winograd", host->nh_arch(), wdefs, &log, true);
    hip_err_chk(hipModuleLoadData(&impl->wino_mod, code.data()), "hipModuleLoadData(winograd)");
    hip_err_chk(hipModuleGetFunction(&impl->wino_filt, impl->wino_mod, "bodahip_wino_filt"), "hipModuleGetFunction(wino_filt)");
    hip_err_chk(hipModuleGetFunction(&impl->wino_in, impl->wino_mod, "bodahip_wino_in"), "hipModuleGetFunction(wino_in)");
    hip_err_chk(hipModuleGetFunction(&impl->wino_out, impl->wino_mod, "bodahip_wino_out"), "hipModuleGetFunction(wino_out)");
    hip_err_chk(hipModuleGetFunction(&impl->wino_fused, impl->wino_mod, "bodahip_wino_fused"), "hipModuleGetFunction(wino_fused)");
    hip_err_chk(hipModuleGetFunction(&impl->wino_filt_t, impl->wino_mod, "bodahip_wino_filt_t"), "hipModuleGetFunction(wino_filt_t)");
  }
  int const TH = (g.OH + 1) / 2, TW = (g.OW + 1) / 2, tpi = TH * TW;
  // Fused kernel (bodahip_wino_fused): transforms inside the MFMA kernel, only U goes through memory; same results bit for bit as the
  // three-kernel pipeline below.  Opt-in (BODAHIP_WINO_FUSED=1): on MI355X it measures level with the pipeline, not ahead of it -- AlexNet
  // conv3 / 4 / 5 at 256 images 456 / 645 / 514 us against 474 / 668 / 489 (direct kernel 578 / 855 / 661); lists with winograd_all:
  // AlexNet 136.7 vs 137.0 TF/s, NiN 127.1 vs 130.5, GoogLeNet@64 66.3 vs 64.0, ResNet-50@64 90.7 vs 90.4.  With all staging work taken
  // out (ablations, -DWABLATE) the bare MFMA loop of conv4 already takes 529 us: 1176 workgroups over 256 CUs are 5 rounds for 4.6 rounds
  // of work, one barrier per 8 channels costs ~0.5 us of drained matrix pipe per stage, 14x14 tiles cover a 13x13 plane -- while the batched
  // transform-domain sgemm of the pipeline runs 256x256 tiles at 120-130 TF/s.  Kept, tested and measurable; not the default.
  if (g.OC % 4 == 0 && getenv("BODAHIP_WINO_FUSED") && (uint64_t)g.B * g.C * g.H * g.W * 4 < 0x3ff00000ull && (uint64_t)16 * g.C * g.OC * 4 < 0x7ffffff0ull &&
      (uint64_t)g.B * tpi < 0x7fffffc0ull) {
    size_t const nU = (size_t)16 * g.C * g.OC;
    ensure_ws(impl, host, nU * sizeof(float));
    wino_args_t wa; memset(&wa, 0, sizeof(wa));
    wa.in = in; wa.filts = filts; wa.bias = biases; wa.out = out; wa.U = (float *)impl->ws;
    wa.C = g.C; wa.H = g.H; wa.W = g.W; wa.OC = g.OC; wa.OH = g.OH; wa.OW = g.OW; wa.TH = TH; wa.TW = TW; wa.PY = g.PY; wa.PX = g.PX; wa.relu = g.relu ? 1 : 0;
    wa.out_ctot = out_ctot; wa.out_coff = out_coff; wa.B0 = 0; wa.Bc = g.B; wa.Tc = (int)((long)g.B * tpi);
    void *wparams[] = {&wa};
    hip_err_chk(host->nh_launch(impl->wino_filt_t, (uint32_t)(((long)g.C * g.OC + 255) / 256), 1, 256, wparams), "hipModuleLaunchKernel(wino_filt_t)");
    uint32_t const grid = (uint32_t)((g.OC + 63) / 64) * (uint32_t)((wa.Tc + 63) / 64);
    hip_err_chk(host->nh_launch(impl->wino_fused, grid, 1, 512, wparams), "hipModuleLaunchKernel(wino_fused)");
    tile_cfg_t fc; fc.BI = 64; fc.BJ = 64; fc.BK = 8; fc.WI = 2; fc.WJ = 4; fc.MINW = 1; fc.MT = 32; fc.PF = 1; fc.SPLITK = 1;
    long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * 9;
    last_launch.kernel = "bodahip_conv_winograd_fused_f32"; last_launch.cfg = fc; last_launch.grid = grid; last_launch.block = 512;
    last_launch.flops = 2.0 * Nj * g.OC * Kt; // effective flops, as the reference credits any fast algorithm (src/latex-util.H:116-133)
    last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
    return;
  }
  // images per chunk: bounds the scratch (V + M <= 1 GiB; BODAHIP_WINO_CHUNK_MB overrides); a multiple of 4 keeps the sgemm's N on its
  // vector-load path.  Measured: (a) chunks small enough to keep V and M inside the 256 MB Infinity Cache (96 / 192 MB) LOSE 5 % to
  // one big batched sgemm (AlexNet conv4 B=256: 713 / 715 vs 674 us) -- the shorter sgemms cost more than the HBM round trip saves;
  // (b) a three-stream pipeline over chunks (input transform k+1 | sgemm k | output transform k-1, event-linked) is slower still
  // (2 / 4 / 8 chunks: 722 / 741 / 808 us): each cross-stream event wait costs more than the ~40 us of transform it would hide.
  long const Bc = wino_chunk_imgs(g);
  long const Tc_max = Bc * tpi;
  if ((uint64_t)g.C * Tc_max * 4 >= 0x7ffffff0ull || (uint64_t)g.OC * Tc_max * 4 >= 0x7ffffff0ull) unsup_err("hip_conv (winograd): transformed planes of 2 GiB or more");
  size_t const nU = (size_t)16 * g.C * g.OC, nV = (size_t)16 * g.C * Tc_max, nM = (size_t)16 * g.OC * Tc_max;
  ensure_ws(impl, host, (nU + nV + nM) * sizeof(float));
  wino_args_t wa; memset(&wa, 0, sizeof(wa));
  wa.in = in; wa.filts = filts; wa.bias = biases; wa.out = out;
  wa.U = (float *)impl->ws; wa.V = wa.U + nU; wa.M = wa.V + nV;
  wa.C = g.C; wa.H = g.H; wa.W = g.W; wa.OC = g.OC; wa.OH = g.OH; wa.OW = g.OW; wa.TH = TH; wa.TW = TW; wa.PY = g.PY; wa.PX = g.PX; wa.relu = g.relu ? 1 : 0;
  wa.out_ctot = out_ctot; wa.out_coff = out_coff;
  void *wparams[] = {&wa};
  auto launch1 = [&](hipFunction_t f, long n, char const *what) {
    hip_err_chk(host->nh_launch(f, (uint32_t)((n + 255) / 256), 1, 256, wparams), what);
  };
  launch1(impl->wino_filt, (long)g.C * g.OC, "hipModuleLaunchKernel(wino_filt)");
  string const tile = tune_of(impl, "sgemm_tile");
  tile_cfg_t last_cfg; uint32_t last_grid = 0;
  for (long b0 = 0; b0 < g.B; b0 += Bc) {
    long const bc = std::min<long>(Bc, g.B - b0), Tc = bc * tpi;
    wa.B0 = (int)b0; wa.Bc = (int)bc; wa.Tc = (int)Tc;
    launch1(impl->wino_in, (long)g.C * Tc, "hipModuleLaunchKernel(wino_in)");
    plan_t const p = plan_sgemm((uint32_t)g.OC, (uint32_t)Tc, (uint32_t)g.C, host->nh_num_cus(), tile, false, 16);
    if (p.cfg.SPLITK > 1) unsup_err("hip_conv (winograd): split-K sgemm tiles are not supported for the batched transform-domain sgemm");
    kernel_t &k = get_kernel(impl, host, p);
    gemm_args_t ga; memset(&ga, 0, sizeof(ga));
    ga.I = wa.U; ga.J = wa.V; ga.D = wa.M;
    ga.Mi = g.OC; ga.Nj = (int)Tc; ga.K = g.C; ga.ldI = g.OC; ga.ldJ = (int)Tc; ga.ldD = (int)Tc;
    ga.I_bytes = (unsigned)((uint64_t)g.C * g.OC * 4); ga.J_bytes = (unsigned)((uint64_t)g.C * Tc * 4); ga.D_bytes = (unsigned)((uint64_t)g.OC * Tc * 4);
    ga.bsI = (long)g.C * g.OC; ga.bsJ = (long)g.C * Tc; ga.bsD = (long)g.OC * Tc;
    ga.tiles_i = (g.OC + p.cfg.BI - 1) / p.cfg.BI; ga.tiles_j = (int)((Tc + p.cfg.BJ - 1) / p.cfg.BJ); ga.splitk = 1;
    void *gparams[] = {&ga};
    hip_err_chk(host->nh_launch(k.func, (uint32_t)(ga.tiles_i * ga.tiles_j), 16, (uint32_t)p.cfg.threads(), gparams),
                "hipModuleLaunchKernel(winograd sgemm)");
    launch1(impl->wino_out, (long)g.OC * Tc, "hipModuleLaunchKernel(wino_out)");
    last_cfg = p.cfg; last_grid = (uint32_t)(ga.tiles_i * ga.tiles_j * 16);
  }
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * 9;
  last_launch.kernel = "bodahip_conv_winograd_f32"; last_launch.cfg = last_cfg; last_launch.grid = last_grid; last_launch.block = last_cfg.threads();
  last_launch.flops = 2.0 * Nj * g.OC * Kt; // effective flops, as the reference credits any fast algorithm (src/latex-util.H:116-133)
  last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
}

void native_kernels_t::conv(float const *filts, float const *biases, float const *in, float *out, conv_geom_t const &g, bool bf16, int out_ctot, int out_coff, char const *algo, float const *filts_km) {
  if (out_ctot <= 0) { out_ctot = g.OC; out_coff = 0; }
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
  if (!Nj || !g.OC) return;
  if (Nj > 0x7fffffffl || Kt > 0x7fffffffl) unsup_err("hip_conv: dims exceed int32");
  bool const exact = tune_of(impl, "exact") != "0";
  // tolerance mode: 3x3 / stride-1 layers take the F(2x2,3x3) path where it measured ahead of the direct kernel unless conv_algo says otherwise
  // (the reference holds Winograd results to mrd < 2e-3, src/rtc_prof.cc:317-319,436)
  string const conv_algo = algo ? string(algo) : (tune_of(impl, "conv_algo").empty() && !exact ? string("winograd") : tune_of(impl, "conv_algo"));
  if (!bf16 && !g.pooled() && winograd_applies(g, conv_algo) && tune_of(impl, "conv_tile").empty()) {
    conv_winograd(filts, biases, in, out, g, out_ctot, out_coff); return;
  }
  if (g.pooled() && bf16) unsup_err("hip_conv: fused pooling is an fp32 form");
  if (bf16 && tune_of(impl, "conv_tile").empty()) {
    conv_geom_t g2; int pry = 0, prx = 0; plan_t p2;
    if (s2d_geom(g, g2, pry, prx) && plan_patch_bf16(g2, host->nh_num_cus(), p2)) {
      size_t const in2 = (size_t)g.B * g2.C * g2.H * g2.W, f2 = (size_t)g.OC * g2.C * g2.KH * g2.KW;
      if (in2 * 4 >= 0x7ffffff0ull) unsup_err("hip_conv_bf16: space-to-depth input of 2 GiB or more");
      size_t const off_in2 = 0, off_f2 = (in2 * 4 + 255) & ~size_t(255), off_fp = (off_f2 + f2 * 4 + 255) & ~size_t(255);
      size_t const fpb = (size_t)(g2.C / 8) * g2.KH * g2.KW * g.OC * 16;
      ensure_ws(impl, host, off_fp + fpb);
      plan_t sp; sp.patch16 = true; sp.bf16 = true; sp.kname = "bodahip_s2d"; sp.defs = {"-DS2D_ONLY=1"};
      kernel_t &sk = get_kernel(impl, host, sp);
      s2d_args_t sa; memset(&sa, 0, sizeof(sa));
      sa.B = g.B; sa.C = g.C; sa.H = g.H; sa.W = g.W; sa.OC = g.OC; sa.KH = g.KH; sa.KW = g.KW; sa.S = g.SY; sa.C2 = g2.C;
      void *sparams[] = {&sa};
      sa.src = in; sa.dst = (float *)((char *)impl->ws + off_in2); sa.H2 = g2.H; sa.W2 = g2.W; sa.oy = pry; sa.ox = prx; sa.filt = 0;
      sa.mode = 1; sa.c2_lo = 0;
      long const n1 = (long)g.B * g.C * g2.H * g.SY * g2.W;
      hip_err_chk(host->nh_launch(sk.func, (uint32_t)((n1 + 255) / 256), 1, 256, sparams), "hipModuleLaunchKernel(s2d in)");
      sa.mode = 0; sa.c2_lo = g.C * g.SY * g.SY;
      if (sa.c2_lo < g2.C) { // zero pad channels up to a multiple of 8
        long const n0 = (long)g.B * (g2.C - sa.c2_lo) * g2.H * g2.W;
        hip_err_chk(host->nh_launch(sk.func, (uint32_t)((n0 + 255) / 256), 1, 256, sparams), "hipModuleLaunchKernel(s2d pad)");
      }
      sa.c2_lo = 0;
      sa.src = filts; sa.dst = (float *)((char *)impl->ws + off_f2); sa.H2 = g2.KH; sa.W2 = g2.KW; sa.oy = pry - g.PY; sa.ox = prx - g.PX; sa.filt = 1;
      hip_err_chk(host->nh_launch(sk.func, (uint32_t)((f2 + 255) / 256), 1, 256, sparams), "hipModuleLaunchKernel(s2d filts)");
      kernel_t &k2 = get_kernel(impl, host, p2);
      gemm_args_t ga; memset(&ga, 0, sizeof(ga));
      ga.J = (float const *)((char *)impl->ws + off_in2); ga.D = out; ga.bias = biases;
      ga.Mi = g.OC; ga.Nj = (int)Nj; ga.K = g2.C * g2.KH * g2.KW; ga.C = g2.C; ga.H = g2.H; ga.W = g2.W; ga.OH = g.OH; ga.OW = g.OW;
      ga.J_bytes = (unsigned)(in2 * 4);
      uint64_t const out_bytes = (uint64_t)Nj * out_ctot * 4;
      if (out_bytes >= 0x7ffffff0ull) unsup_err("hip_conv: out of 2 GiB or more is not supported (32-bit store offsets, masked lanes use offset 2^31)");
      ga.D_bytes = (unsigned)out_bytes; ga.out_ctot = out_ctot; ga.out_coff = out_coff;
      ga.tiles_i = (g.OC + p2.cfg.BI - 1) / p2.cfg.BI; ga.tiles_j = (int)((Nj + p2.cfg.BJ - 1) / p2.cfg.BJ); ga.splitk = 1;
      launch_patch16(impl, host, p2, k2, ga, (float const *)((char *)impl->ws + off_f2), g2, off_fp);
      last_launch.kernel = "bodahip_conv_patch_bf16(s2d)"; last_launch.cfg = p2.cfg; last_launch.grid = (uint32_t)(ga.tiles_i * ga.tiles_j); last_launch.block = p2.cfg.threads();
      last_launch.flops = 2.0 * Nj * g.OC * Kt;
      last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
      return;
    }
  }
  plan_t const p = plan_conv(g, host->nh_num_cus(), tune_of(impl, "conv_tile"), bf16, tune_of(impl, "k1_stream"), out_ctot == g.OC, exact);
  tile_cfg_t const &cfg = p.cfg;
  kernel_t &k = get_kernel(impl, host, p);
  if (p.nhwc) {   // (fp32 here: the k-contiguous shapes through the LDS-DMA kernel, plan_ipconv_dma)
    gemm_args_t na; memset(&na, 0, sizeof(na));
    uint64_t const ib = (uint64_t)g.B * Kt * 4, fb = (uint64_t)g.OC * Kt * 4, ob = (uint64_t)Nj * out_ctot * 4;
    if (ib >= 0x7ffffff0ull || fb >= 0x7ffffff0ull || ob >= 0x7ffffff0ull) unsup_err("hip_conv: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
    na.I = filts; na.J = in; na.D = out; na.bias = biases; na.Mi = g.OC; na.Nj = (int)Nj; na.K = (int)Kt; na.C = (int)Kt; na.H = 1; na.W = 1; na.OH = 1; na.OW = 1;
    na.I_bytes = (unsigned)fb; na.J_bytes = (unsigned)ib; na.D_bytes = (unsigned)ob; na.out_ctot = out_ctot; na.out_coff = out_coff; na.splitk = 1;
    na.tiles_i = (g.OC + cfg.BI - 1) / cfg.BI; na.tiles_j = (int)((Nj + cfg.BJ - 1) / cfg.BJ);
    void *nparams[] = {&na};
    hip_err_chk(host->nh_launch(k.func, (uint32_t)(na.tiles_i * na.tiles_j), 1, (uint32_t)cfg.threads(), nparams), "hipModuleLaunchKernel(conv_nhwc_f32)");
    last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)(na.tiles_i * na.tiles_j); last_launch.block = cfg.threads();
    last_launch.flops = 2.0 * Nj * g.OC * Kt;
    last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
    return;
  }
  gemm_args_t ga; memset(&ga, 0, sizeof(ga));
  uint64_t const in_bytes = (uint64_t)g.B * g.C * (g.pooled() ? (uint64_t)g.UH * g.UW : (uint64_t)g.H * g.W) * 4, f_bytes = (uint64_t)g.OC * Kt * 4;   // (fused pooling: the tensor read is the pooling's input)
  if (in_bytes >= 0x7ffffff0ull || f_bytes >= 0x7ffffff0ull) unsup_err("hip_conv: in / filts of 2 GiB or more are not supported (32-bit buffer offsets)");
  if (g.pooled() && !(p.patch && !p.rdec && cfg.PF == 1 && cfg.SW == 0 && cfg.SPLITK == 1)) unsup_err("hip_conv: fused pooling needs the LDS-patch form of the kernel with one K tile in flight (plan: " + cfg.str() + ")");
  ga.I = filts; ga.J = in; ga.D = out; ga.bias = biases;
  ga.Mi = g.OC; ga.Nj = (int)Nj; ga.K = (int)Kt; ga.ldI = (int)Kt; ga.ldJ = p.ipconv ? (int)Kt : 0; ga.ldD = g.OH * g.OW;
  ga.C = p.rdec ? g.C * g.KH : g.C; ga.H = g.H; ga.W = g.W; ga.OH = g.OH; ga.OW = g.OW;   // (row-decimated patch: the kernel's "channels" are the C * KH row sets)
  ga.I_bytes = (unsigned)f_bytes; ga.J_bytes = (unsigned)in_bytes;
  uint64_t const out_bytes = (uint64_t)Nj * out_ctot * 4;
  if (cfg.SPLITK > 1 && out_ctot != g.OC) unsup_err("hip_conv: split-K tiles cannot write a channel slice of a wider output");
  ga.out_ctot = out_ctot; ga.out_coff = out_coff;
  if (out_bytes >= 0x7ffffff0ull) unsup_err("hip_conv: out of 2 GiB or more is not supported (32-bit store offsets, masked lanes use offset 2^31)");
  ga.D_bytes = (unsigned)out_bytes;
  if (p.rows) { ktab_t const kt = get_rtab(impl, host, g); ga.ktab = kt.d; ga.ktab_n = kt.n; }
  else if (!p.ipconv && !p.k1 && !p.patch && !p.stream && !p.patch16) { ktab_t const kt = get_ktab(impl, host, g); ga.ktab = kt.d; ga.ktab_n = kt.n; }
  ga.tiles_i = (g.OC + cfg.BI - 1) / cfg.BI; ga.tiles_j = (int)((Nj + cfg.BJ - 1) / cfg.BJ);
  if (p.patch16) { launch_patch16(impl, host, p, k, ga, filts, g, 0); 
    last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)(ga.tiles_i * ga.tiles_j); last_launch.block = cfg.threads();
    last_launch.flops = 2.0 * Nj * g.OC * Kt;
    last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
    return;
  }
  if (p.fc) {
    void *params[] = {&ga};
    hip_err_chk(host->nh_launch(k.func, (uint32_t)(ga.tiles_i * ga.tiles_j), 1, (uint32_t)cfg.threads(), params), "hipModuleLaunchKernel(fc_f32)");
    last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)(ga.tiles_i * ga.tiles_j); last_launch.block = cfg.threads();
    last_launch.flops = 2.0 * Nj * g.OC * Kt;
    last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
    return;
  }
  if (p.stream) {
    // persistent workgroups: as many as fit the chip at the kernel's occupancy, trimmed to the smallest count with the same number of
    // super-blocks per workgroup (an even deal); workgroup w of an out_chan tile takes super-blocks w, w + kt_per, ...
    long const slots = std::max(1l, (long)host->nh_num_cus() * std::max(1, cfg.MINW * 4 / (cfg.WI * cfg.WJ)) / ga.tiles_i);
    if (p.quad) ga.tiles_j = (int)(((long)g.B * ((g.OH * g.OW + 127) / 128) + cfg.WJ - 1) / cfg.WJ);   // super-blocks of WJ 128-pel blocks, blocks never straddle images
    long const per = (ga.tiles_j + slots - 1) / slots;
    ga.kt_per = (int)((ga.tiles_j + per - 1) / per); ga.splitk = 1;
    if (p.quad && ga.kt_per >= 8) ga.kt_per = (ga.kt_per + 7) / 8 * 8;   // a multiple of 8 workgroups: the kernel then gives each XCD a contiguous range of super-blocks
    void *params[] = {&ga};
    hip_err_chk(host->nh_launch(k.func, (uint32_t)(ga.kt_per * ga.tiles_i), 1, (uint32_t)cfg.threads(), params), "hipModuleLaunchKernel(k1_stream)");
    last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)(ga.kt_per * ga.tiles_i); last_launch.block = cfg.threads();
    last_launch.flops = 2.0 * Nj * g.OC * Kt;
    last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
    return;
  }
  uint32_t kho_grid = 0, tail_grid = 0;
  if (cfg.KHO > 1 && !p.cbig) { setup_kho(impl, host, ga, cfg, out); kho_grid = launch_kho(host, k, ga, cfg); }
  else {
    setup_splitk(impl, host, ga, cfg, (size_t)Nj * g.OC);
    size_t ts_off = 0;
    if (p.cbig && std::find(p.defs.begin(), p.defs.end(), string("-DI_VW=0")) != p.defs.end()) {   // the staging-wave kernel reads its filters k-major: transposed into the scratch first (part of the call)
      long const mi4 = ((long)g.OC + 3) / 4 * 4, kp = filts_km ? Kt + 128 : (Kt + cfg.BK - 1) / cfg.BK * cfg.BK;
      uint64_t const xb = (uint64_t)kp * mi4 * 4;
      if (xb >= 0x7ffffff0ull) unsup_err("hip_conv: filts of 2 GiB or more are not supported (32-bit buffer offsets)");
      if (filts_km) {   // the caller holds the k-major copy (hip_conv_filts_kmajor, made once: a net's weights do not change between forward passes -- src/rtc_fwd.cc:229-243 transposes its filters at set-up, too)
        ga.I = filts_km; ga.ldI = (int)mi4; ga.I_bytes = (unsigned)xb;
        if (getenv("BODAHIP_CBIG_TSTAMP")) ensure_ws(impl, host, (size_t)ga.tiles_i * ga.tiles_j * 128);
      } else {
      ts_off = (xb + 255) & ~size_t(255);
      ensure_ws(impl, host, ts_off + (getenv("BODAHIP_CBIG_TSTAMP") ? (size_t)ga.tiles_i * ga.tiles_j * 128 : 0));
      plan_t xp; xp.cbig = true; xp.kname = "bodahip_conv_big_xpose"; xp.defs = {"-DXPOSE_ONLY=1"};
      kernel_t &xk = get_kernel(impl, host, xp);
      float const *src = filts; float *dst = (float *)impl->ws; int Mi = g.OC, Mi4 = (int)mi4, Kk = (int)Kt, Kp = (int)kp;
      void *xparams[] = {&src, &dst, &Mi, &Mi4, &Kk, &Kp};
      hip_err_chk(host->nh_launch(xk.func, (uint32_t)((kp + 31) / 32), (uint32_t)((mi4 + 31) / 32), 256, xparams), "hipModuleLaunchKernel(conv_big_xpose)");
      ga.I = (float const *)impl->ws; ga.ldI = (int)mi4; ga.I_bytes = (unsigned)xb;
      }
    }
    char const *const tstamp = p.cbig ? getenv("BODAHIP_CBIG_TSTAMP") : nullptr;   // experiment hook (tools/cbig_timeline.py): kernel built with -DTSTAMP=1 leaves 16 clock stamps per workgroup in the scratch; appended to the named file
    size_t const ts_bytes = (size_t)ga.tiles_i * ga.tiles_j * 128;
    bool const ts_late = tstamp && strlen(tstamp) > 5 && !strcmp(tstamp + strlen(tstamp) - 5, ":late");   // no synchronisation, no copy per launch: the sequence runs undisturbed
    if (tstamp) { ensure_ws(impl, host, ts_off + ts_bytes); ga.ws = (float *)((char *)impl->ws + ts_off); if (!ts_late) hip_err_chk(hipMemsetAsync(ga.ws, 0, ts_bytes, host->nh_stream()), "hipMemsetAsync(tstamp)"); }
    if (p.cbig && p.split_pels > 0) {   // two-level tiling along the pels: this plan's tiles over the first split_pels pels, the tail plan's over the rest
      ga.tiles_j = (int)(p.split_pels / cfg.BJ);
      launch(host, k, ga, cfg);
      plan_t tp; tp.cbig = true; tp.kname = p.kname; tp.cfg = p.tail_cfg; tp.defs = p.tail_defs; tp.patch = p.patch; tp.k1 = p.k1; tp.rdec = p.rdec;
      kernel_t &k2 = get_kernel(impl, host, tp);
      gemm_args_t g2 = ga; g2.tiles_i = (g.OC + tp.cfg.BI - 1) / tp.cfg.BI; g2.tiles_j = (int)((Nj - p.split_pels + tp.cfg.BJ - 1) / tp.cfg.BJ); g2.bsJ = p.split_pels; g2.ws = nullptr;
      launch(host, k2, g2, tp.cfg);
      tail_grid = (uint32_t)g2.tiles_i * g2.tiles_j;
    } else
    launch(host, k, ga, cfg);
    if (ts_late) { impl->ts_off = ts_off; impl->ts_bytes = ts_bytes; impl->ts_hdr = "launch " + cfg.str() + " grid " + std::to_string(ga.tiles_i * ga.tiles_j); }
    else if (tstamp) {
      std::vector<unsigned long long> h(ts_bytes / 8);
      hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize(tstamp)");
      hip_err_chk(hipMemcpy(h.data(), ga.ws, ts_bytes, hipMemcpyDeviceToHost), "hipMemcpy(tstamp)");
      if (FILE *f = fopen(tstamp, "a")) { fprintf(f, "launch %s grid %d\n", cfg.str().c_str(), ga.tiles_i * ga.tiles_j);
        for (size_t w = 0; w < h.size() / 16; ++w) { for (int e = 0; e < 16; ++e) fprintf(f, "%llu ", h[w * 16 + e]); fprintf(f, "\n"); } fclose(f); }
    }
    if (cfg.SPLITK > 1) reduce_splitk(impl, host, ga, Nj * g.OC, true, g.relu, g.OH * g.OW, g.OC);
  }
  last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (kho_grid ? kho_grid : (uint32_t)ga.tiles_i * ga.tiles_j * cfg.SPLITK) + tail_grid; last_launch.block = cfg.threads();
  last_launch.flops = 2.0 * Nj * g.OC * Kt;
  last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * (g.pooled() ? (double)g.UH * g.UW : (double)g.H * g.W) + (double)Nj * g.OC + (double)g.OC * Kt + g.OC);
}


// ---- hip_conv_k1_chain: two 1x1 / stride-1 / unpadded fp32 convolutions back to back as ONE launch (kernels/k1_quad_f32.hip -DCHAIN=1): the first one's output
// tile stays in the accumulator registers and becomes the second one's MFMA operand after a half exchange; both filter images are resident in LDS.  NiN's
// cccp1 -> cccp2 (96 -> 96 -> 96 on 55 x 55): the 24 flop/B layers of the net become one 48 flop/B pass, the intermediate tensor's write + read are gone.
// g = the FIRST convolution's geometry (g.OC = the intermediate channels, g.relu its ReLU); oc2 / relu2 = the second one's.  Covered: at most 96 intermediate
// channels (one accumulator set of three 32-row blocks: 192 registers + 64 for the second convolution's block), at most 128 out_chans, both filter images in LDS.
static bool plan_k1_chain(conv_geom_t const &g, int oc2, bool relu2, plan_t &p) {
  if (!(g.KH == 1 && g.KW == 1 && g.SY == 1 && g.SX == 1 && g.PY == 0 && g.PX == 0)) return false;
  if (g.OC < 1 || g.OC > 96 || oc2 < 1 || oc2 > 128 || g.C < 1 || g.OH * g.OW < 4) return false;
  int const OCB = (g.OC + 31) / 32, OCB2 = (oc2 + 31) / 32, ksteps = (g.C + 1) / 2;
  long const kp = (long)ksteps * 2, kp2 = (long)(g.OC + 1) / 2 * 2;
  long const lds = 4 * (kp * ((OCB * 32) | 1) + OCB * 32 + kp2 * ((OCB2 * 32) | 1) + OCB2 * 32);
  if (lds > 160 * 1024) return false;
  int RING = 8; while (ksteps % RING) --RING;
  p = plan_t(); p.stream = true; p.quad = true; p.kname = "bodahip_k1_chain_f32";
  p.cfg.BI = OCB * 32; p.cfg.BJ = 4 * 128; p.cfg.BK = g.C; p.cfg.WI = 1; p.cfg.WJ = 4; p.cfg.MINW = 1; p.cfg.SPLITK = 1; p.cfg.MT = 32; p.cfg.PF = RING;
  p.defs = {"-DKC=" + std::to_string(g.C), "-DHW=" + std::to_string(g.OH * g.OW), "-DWJ=4", "-DOCB=" + std::to_string(OCB), "-DRING=" + std::to_string(RING), "-DMINW=1",
            string("-DRELU=") + (g.relu ? "1" : "0"), "-DEDGE_OC=1", "-DCHAIN=1", "-DMID=" + std::to_string(g.OC), "-DOCB2=" + std::to_string(OCB2),
            string("-DRELU2=") + (relu2 ? "1" : "0"), string("-DEDGE_OC2=") + ((oc2 % 32) ? "1" : "0")};
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return true;
}

void native_kernels_t::conv_k1_chain(float const *filts, float const *biases, float const *filts2, float const *biases2, float const *in, float *out, float *mid,
                                     conv_geom_t const &g, int oc2, bool relu2, int out_ctot, int out_coff) {
  if (out_ctot <= 0) { out_ctot = oc2; out_coff = 0; }
  long const Nj = (long)g.B * g.OH * g.OW;
  if (!Nj || !g.OC || !oc2) return;
  plan_t p;
  if (!plan_k1_chain(g, oc2, relu2, p)) unsup_err("hip_conv_k1_chain: two 1x1 / stride-1 / unpadded convolutions with at most 96 intermediate channels, at most 128 out_chans and filters that fit the LDS");
  uint64_t const in_bytes = (uint64_t)g.B * g.C * g.H * g.W * 4, mid_bytes = (uint64_t)Nj * g.OC * 4, out_bytes = (uint64_t)Nj * out_ctot * 4;
  if (in_bytes >= 0x7ffffff0ull || mid_bytes >= 0x7ffffff0ull || out_bytes >= 0x7ffffff0ull) unsup_err("hip_conv_k1_chain: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
  kernel_t &k = get_kernel(impl, host, p);
  tile_cfg_t const &cfg = p.cfg;
  chain_args_t ga; memset((void *)&ga, 0, sizeof(ga));
  ga.I = filts; ga.J = in; ga.D = out; ga.bias = biases; ga.Mi = g.OC; ga.Nj = (int)Nj; ga.K = g.C; ga.ldI = g.C; ga.ldD = g.OH * g.OW;
  ga.C = g.C; ga.H = g.H; ga.W = g.W; ga.OH = g.OH; ga.OW = g.OW;
  ga.I_bytes = (unsigned)((uint64_t)g.OC * g.C * 4); ga.J_bytes = (unsigned)in_bytes; ga.D_bytes = (unsigned)out_bytes; ga.out_ctot = out_ctot; ga.out_coff = out_coff;
  ga.I2 = filts2; ga.bias2 = biases2; ga.Dmid = mid; ga.M2 = oc2; ga.I2_bytes = (unsigned)((uint64_t)oc2 * g.OC * 4); ga.Dmid_bytes = mid ? (unsigned)mid_bytes : 0u;
  // persistent workgroups, one per CU (four waves, one per SIMD), an even deal of super-blocks (four 128-pel blocks of one image each), XCD-contiguous ranges
  ga.tiles_i = 1; ga.tiles_j = (int)(((long)g.B * ((g.OH * g.OW + 127) / 128) + cfg.WJ - 1) / cfg.WJ);
  long const slots = std::max(1l, (long)host->nh_num_cus());
  long const per = (ga.tiles_j + slots - 1) / slots;
  ga.kt_per = (int)((ga.tiles_j + per - 1) / per); ga.splitk = 1;
  if (ga.kt_per >= 8) ga.kt_per = (ga.kt_per + 7) / 8 * 8;
  void *params[] = {&ga};
  hip_err_chk(host->nh_launch(k.func, (uint32_t)ga.kt_per, 1, (uint32_t)cfg.threads(), params), "hipModuleLaunchKernel(k1_chain)");
  last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)ga.kt_per; last_launch.block = cfg.threads();
  last_launch.flops = 2.0 * Nj * ((double)g.OC * g.C + (double)oc2 * g.OC);
  last_launch.algo_bytes = 4.0 * ((double)g.B * g.C * g.H * g.W + (double)Nj * oc2 + (double)g.OC * g.C + g.OC + (double)oc2 * g.OC + oc2);
}

struct grp_args_t { // must match kernels/conv_nhwc_bf16.hip
  int n; int oc0[4]; int noc[4];
  void *D[4]; unsigned D_bytes[4]; int ctot[4]; int coff[4];
};
// ---- hip_conv_nhwc_multi: several INDEPENDENT channels-last convolutions as one launch (kernels/conv_nhwc_multi_bf16.hip) ------------------------------------
struct multi_prob_t { // must match prob_t of kernels/conv_nhwc_multi_bf16.hip (128 bytes)
  void const *I; void const *J; void *D; float const *bias;
  unsigned I_bytes, J_bytes, D_bytes; int Mi;
  int Nj, CIN, KH, KW;
  int SY, SX, PY, PX;
  int CH, CW, COH, COW;
  int kCG, kKC, nK, relu;
  int out_ctot, out_coff, tiles_i, tiles_j;
};
static_assert(sizeof(multi_prob_t) == 128, "multi_prob_t must stay 128 bytes (kernel-side prob_t)");
struct multi_tile_t { int prob, tile_i, tile_j, pad; };
struct multi_args_t { multi_prob_t const *probs; multi_tile_t const *tiles; int n_tiles; int n_probs; };

// One tile shape for the whole launch (the kernel is specialised on it, not on any member's geometry).  tile: "BIxBJxBKxWIxWJ[xMINW[x1[x32[xNBUF]]]]" or "".
// Default 64 x 128 x 64, 2 x 2 waves, a ring of TWO (48 KB of LDS: three workgroups, i.e. three waves per SIMD, per CU) -- the members are small (that is why they
// are here), mostly narrow in out_chan (16-256) and short in K (3-13 steps), so what hides their latency is co-resident workgroups, not a deeper ring.  Measured on
// MI355X (tools/r4f.sh: the 44 implicit-GEMM members of the GoogLeNet list at 64 images as one launch, us): 64x128x64 ring 2 / three per CU 205 | 64x128x32 ring 4 222 |
// 64x256x32 ring 3 224 | 64x64x64 266 | 128x128x64 ring 3 326 (the members one by one: 533, of which 260 are launch floors).
static plan_t plan_conv_nhwc_multi(std::vector<conv_geom_t> const &gs, string const &tile, bool out_f32) {
  (void)gs;
  plan_t p; p.nhwc = true; p.nhwc_multi = true; p.bf16 = true; p.kname = "bodahip_conv_nhwc_multi_bf16";
  tile_cfg_t c; c.MT = 32; c.SPLITK = 1; c.PF = 2; c.BI = 64; c.BJ = 128; c.BK = 64; c.WI = 2; c.WJ = 2; c.MINW = 3;
  if (!tile.empty()) {
    int nbuf = 3;
    if (!parse_tile(tile, c)) rt_err("bad conv_tile '" + tile + "'");
    { int nf = 1; for (char ch : tile) if (ch == 'x' || ch == ':') ++nf; if (nf >= 9) nbuf = c.PF; }
    c.MT = 32; c.SPLITK = 1; c.PF = nbuf;
  }
  int const cpr = c.BK / 8, nw = c.WI * c.WJ;
  bool ok = (c.BK == 32 || c.BK == 64) && c.BI > 0 && c.BJ > 0 && c.WI > 0 && c.WJ > 0 && c.threads() <= 1024 && (c.BI % (c.WI * 32) == 0) && (c.BJ % (c.WJ * 32) == 0) &&
            ((c.BI * cpr) % 64 == 0) && ((c.BJ * cpr) % 64 == 0) && ((c.BI * cpr / 64) % nw == 0) && ((c.BJ * cpr / 64) % nw == 0) &&
            (c.BI / (c.WI * 32)) * (c.BJ / (c.WJ * 32)) * 16 <= 256 && c.MINW >= 1 && c.PF >= 2 && c.PF <= 4;
  long const lds = std::max<long>((long)c.PF * (c.BI + c.BJ) * c.BK * 2, out_f32 ? 0 : (long)c.BJ * (c.BI * 2 + 16));
  if (!ok || lds > 160 * 1024) unsup_err("hip_conv_nhwc_multi: unsupported tile configuration " + c.str());
  p.cfg = c;
  p.defs = {"-DBI=" + std::to_string(c.BI), "-DBJ=" + std::to_string(c.BJ), "-DBK=" + std::to_string(c.BK), "-DWI=" + std::to_string(c.WI), "-DWJ=" + std::to_string(c.WJ),
            "-DMINW=" + std::to_string(c.MINW), string("-DOUT_F32=") + (out_f32 ? "1" : "0"), "-DNBUF=" + std::to_string(c.PF)};
  if (char const *e = getenv("BODAHIP_EXTRA_DEFS")) { std::istringstream is(e); string tok; while (is >> tok) p.defs.push_back(tok); }
  return p;
}

void native_kernels_t::conv_nhwc_multi(int n, multi_member_t const *ms, bool out_f32) {
  if (n < 1 || n > 256) unsup_err("hip_conv_nhwc_multi: 1..256 members");
  std::vector<conv_geom_t> gs; for (int m = 0; m < n; ++m) gs.push_back(ms[m].g);
  plan_t const p = plan_conv_nhwc_multi(gs, tune_of(impl, "conv_tile"), out_f32);
  tile_cfg_t const &cfg = p.cfg;
  std::vector<multi_prob_t> probs((size_t)n); std::vector<multi_tile_t> tiles;
  double flops = 0, bytes = 0;
  for (int m = 0; m < n; ++m) {
    conv_geom_t const &g = ms[m].g; multi_prob_t &q = probs[(size_t)m]; memset(&q, 0, sizeof(q));
    if (g.C % 8) unsup_err("hip_conv_nhwc_multi: in_chan of a channels-last bf16 tensor must be a multiple of 8");
    if (g.H >= 32768 || g.W >= 32768) unsup_err("hip_conv_nhwc_multi: planes of 32768 rows / columns or more are not supported");
    long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
    if (Nj < 1 || g.OC < 1) rt_err("hip_conv_nhwc_multi: empty member");
    if (Nj > 0x7fffffffl || Kt > 0x7fffffffl) unsup_err("hip_conv_nhwc_multi: dims exceed int32");
    int const ctot = ms[m].out_ctot > 0 ? ms[m].out_ctot : g.OC, coff = ms[m].out_ctot > 0 ? ms[m].out_coff : 0;
    uint64_t const in_bytes = (uint64_t)g.B * g.C * g.H * g.W * 2, f_bytes = (uint64_t)g.OC * Kt * 2, out_bytes = (uint64_t)Nj * ctot * (out_f32 ? 4 : 2);
    if (in_bytes >= 0x7ffffff0ull || f_bytes >= 0x7ffffff0ull || out_bytes >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc_multi: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
    q.I = ms[m].filts; q.J = ms[m].in; q.D = ms[m].out; q.bias = ms[m].biases;
    q.I_bytes = (unsigned)f_bytes; q.J_bytes = (unsigned)in_bytes; q.D_bytes = (unsigned)out_bytes; q.Mi = g.OC; q.Nj = (int)Nj;
    q.CIN = g.C; q.KH = g.KH; q.KW = g.KW; q.SY = g.SY; q.SX = g.SX; q.PY = g.PY; q.PX = g.PX; q.CH = g.H; q.CW = g.W; q.COH = g.OH; q.COW = g.OW;
    q.kCG = g.C / 8; q.kKC = q.kCG * g.KH * g.KW; q.nK = (q.kKC + cfg.BK / 8 - 1) / (cfg.BK / 8); q.relu = g.relu ? 1 : 0;
    q.out_ctot = ctot; q.out_coff = coff; q.tiles_i = (g.OC + cfg.BI - 1) / cfg.BI; q.tiles_j = (int)((Nj + cfg.BJ - 1) / cfg.BJ);
    // this member's tiles, XCD-aware: consecutive workgroups go to consecutive XCDs, so the eight of a round are eight different pel tiles of one out_chan tile and
    // an XCD walks the out_chan tiles of "its" pel tiles back to back -- an input tile is fetched into one L2
    for (int tj0 = 0; tj0 < q.tiles_j; tj0 += 8)
      for (int ti = 0; ti < q.tiles_i; ++ti)
        for (int tj = tj0; tj < std::min(q.tiles_j, tj0 + 8); ++tj) tiles.push_back(multi_tile_t{m, ti, tj, q.nK});
    flops += 2.0 * Nj * g.OC * Kt;
    bytes += 2.0 * ((double)g.B * g.C * g.H * g.W + (double)g.OC * Kt) + (out_f32 ? 4.0 : 2.0) * (double)Nj * g.OC + 4.0 * g.OC;
  }
  // longest tiles first (the hardware hands workgroups to CUs as CUs free up: a longest-processing-time-first schedule); the sort is stable, so members of equal
  // length keep their order and their XCD-aware tile order
  std::stable_sort(tiles.begin(), tiles.end(), [](multi_tile_t const &x, multi_tile_t const &y) { return x.pad > y.pad; });
  if (tiles.size() > 0x7fffffffull) unsup_err("hip_conv_nhwc_multi: too many tiles");
  // the descriptor table and the tile list live in device memory, one copy per distinct call (pointers included): built on the first call, reused after
  size_t const pb = probs.size() * sizeof(multi_prob_t), tb = tiles.size() * sizeof(multi_tile_t), tb_off = (pb + 255) & ~size_t(255);
  string key = "multi:" + cfg.str();
  { uint64_t h = 1469598103934665603ull; auto mix = [&](void const *d, size_t nb) { for (size_t i = 0; i < nb; ++i) { h ^= ((unsigned char const *)d)[i]; h *= 1099511628211ull; } };
    mix(probs.data(), pb); mix(tiles.data(), tb); key += ":" + std::to_string(h) + ":" + std::to_string(pb + tb); }
  auto it = impl->ktabs.find(key);
  if (it == impl->ktabs.end()) {
    if (host->nh_capturing()) rt_err("graph capture: the descriptor table of this hip_conv_nhwc_multi call is not on the device yet -- run the call list once before capturing it");
    void *dev = nullptr;
    hip_err_chk(hipMalloc(&dev, tb_off + tb), "hipMalloc(multi table)");
    hip_err_chk(hipMemcpyAsync(dev, probs.data(), pb, hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(multi probs)");
    hip_err_chk(hipMemcpyAsync((char *)dev + tb_off, tiles.data(), tb, hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(multi tiles)");
    hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize(multi table)");   // (the host vectors die with this call)
    it = impl->ktabs.emplace(key, dev).first;
  }
  kernel_t &k = get_kernel(impl, host, p);
  multi_args_t ma; ma.probs = (multi_prob_t const *)it->second; ma.tiles = (multi_tile_t const *)((char *)it->second + tb_off); ma.n_tiles = (int)tiles.size(); ma.n_probs = n;
  void *params[] = {&ma};
  hip_err_chk(host->nh_launch(k.func, (uint32_t)tiles.size(), 1, (uint32_t)cfg.threads(), params), "hipModuleLaunchKernel(conv_nhwc_multi_bf16)");
  last_launch.kernel = p.kname + "(x" + std::to_string(n) + ")"; last_launch.cfg = cfg; last_launch.grid = (uint32_t)tiles.size(); last_launch.block = cfg.threads();
  last_launch.flops = flops; last_launch.algo_bytes = bytes;
}

// ---- hip_conv_nhwc_set: a few INDEPENDENT channels-last convolutions, each on ITS OWN specialised kernel code, as one launch ---------------------------------------
// The counterpart of conv_nhwc_multi for the members that deserve their specialisation (an inception module's 3x3 / 5x5 / pool-projection convolutions: three launches
// of 100-200 tiles each on 256 CUs, which a dependency-wired hipGraph does not overlap -- a cross-branch edge costs about what such a kernel takes): the kernel
// sources are instantiated once per distinct member plan inside one translation unit (BODAHIP_AS_MEMBER: kernels/conv_nhwc_bf16.hip, conv_nhwc_patch_bf16.hip become
// __device__ functions), a wrapper kernel maps its workgroup to (member, tile) and calls the member's code.  Same code, same arguments, same tile -> same bits as the
// member's own launch.  Members that cannot join (another workgroup size, K slices) are launched on their own by the same call.
struct set_member_plan_t { plan_t p; gemm_args_t ga; grp_args_t q; long tiles; int variant; double tile_cost; };
static char const *const k_set_macros[] = {"BI", "BJ", "BK", "WI", "WJ", "MINW", "CIN", "KH", "KW", "SY", "SX", "PY", "PX", "CH", "CW", "COH", "COW", "RELU", "OUT_F32", "NBUF", "CG",
                                           "ADIRECT", "PF", "BPF", "WPITCH", "DBUF", "ABLATE", "POOL", "GROUP_I", "IN_F32", "SPLITK", "KSL", "INTERLEAVE", "GROUPS", "KNAME", "BODAHIP_BID"};
static string set_kernel_source(std::vector<plan_t const *> const &variants, int threads, int minw) {
  std::ostringstream o;
  o << "// generated by native_kernels.cc (conv_nhwc_set): " << variants.size() << " member specialisations in one kernel\n";
  o << "#define BODAHIP_AS_MEMBER 1\n#define BODAHIP_ARGS_DEFINED 1\n";
  o << "struct gemm_args_t { float const *I; float const *J; float *D; float const *bias; int Mi, Nj, K; int ldI, ldJ, ldD; int C, H, W, OH, OW; int tiles_i, tiles_j; int splitk, kt_per;\n"
       "  float *ws; long ws_slab; unsigned I_bytes, J_bytes; unsigned D_bytes; int out_ctot, out_coff; int const *ktab; int ktab_n; long bsI, bsJ, bsD; };\n"
       "struct grp_args_t { int n; int oc0[4]; int noc[4]; void *D[4]; unsigned D_bytes[4]; int ctot[4]; int coff[4]; };\n";
  for (size_t v = 0; v < variants.size(); ++v) {
    plan_t const &p = *variants[v];
    for (string const &d : p.defs) {   // "-DNAME=value"
      size_t const eq = d.find('=');
      if (d.compare(0, 2, "-D") != 0 || eq == string::npos) rt_err("conv_nhwc_set: unexpected kernel option '" + d + "'");
      o << "#define " << d.substr(2, eq - 2) << " " << d.substr(eq + 1) << "\n";
    }
    o << "#define KNAME run\nnamespace member_v" << v << " {\n" << (p.nhwc_patch ? k_src_conv_nhwc_patch_bf16 : k_src_conv_nhwc_bf16) << "\n}\n";
    for (char const *m : k_set_macros) o << "#undef " << m << "\n";
  }
  o << "struct set_args_t { gemm_args_t const *m; int const *ends; int const *variant; grp_args_t const *g; int n; };\n";
  // the largest LDS need of the members, as a constant expression
  o << "namespace { constexpr int set_max(int a, int b) { return a > b ? a : b; }\nconstexpr int kSmemAll = ";
  for (size_t v = 0; v < variants.size(); ++v) o << "set_max(member_v" << v << "::member_smem_bytes, ";
  o << "16"; for (size_t v = 0; v < variants.size(); ++v) o << ")"; o << "; }\n";
  for (size_t v = 0; v < variants.size(); ++v) o << "static_assert(member_v" << v << "::member_threads == " << threads << ", \"members of a set share one workgroup size\");\n";
  o << "extern \"C\" __global__ __launch_bounds__(" << threads << ", " << minw << ") void bodahip_conv_nhwc_set(set_args_t const a) {\n"
       "  __shared__ __attribute__((aligned(1024))) char smem[kSmemAll];\n"
       "  int const bid = blockIdx.x;\n"
       "  int k = 0; while (k + 1 < a.n && bid >= __builtin_amdgcn_readfirstlane(a.ends[k])) ++k;      // (workgroup-uniform; a set has a handful of members)\n"
       "  int const local = bid - (k ? __builtin_amdgcn_readfirstlane(a.ends[k - 1]) : 0);\n"
       "  int const var = __builtin_amdgcn_readfirstlane(a.variant[k]);\n"
       "  gemm_args_t p;\n"
       "  { int const *src = reinterpret_cast<int const *>(a.m + k); int *dst = reinterpret_cast<int *>(&p);\n"
       "#pragma unroll\n"
       "    for (int i = 0; i < (int)(sizeof(gemm_args_t) / 4); ++i) dst[i] = __builtin_amdgcn_readfirstlane(src[i]); }\n"
       "  switch (var) {\n";
  for (size_t v = 0; v < variants.size(); ++v) o << "    case " << v << ": member_v" << v << "::run(p, a.g[k], local, smem); break;\n";
  o << "    default: break;\n  }\n}\n";
  return o.str();
}

static gemm_args_t nhwc_member_args(native_kernels_t::multi_member_t const &mm, tile_cfg_t const &cfg, bool out_f32, char const *what) {
  conv_geom_t const &g = mm.g;
  long const Nj = (long)g.B * g.OH * g.OW, Kt = mm.pool ? (long)g.C : (long)g.C * g.KH * g.KW;
  int const ctot = mm.out_ctot > 0 ? mm.out_ctot : g.OC, coff = mm.out_ctot > 0 ? mm.out_coff : 0;
  uint64_t const in_bytes = (uint64_t)g.B * g.C * g.H * g.W * 2, f_bytes = (uint64_t)g.OC * Kt * 2, out_bytes = (uint64_t)Nj * ctot * (out_f32 ? 4 : 2);
  if (Nj > 0x7fffffffl || Kt > 0x7fffffffl || in_bytes >= 0x7ffffff0ull || f_bytes >= 0x7ffffff0ull || out_bytes >= 0x7ffffff0ull) unsup_err(string(what) + ": tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
  gemm_args_t ga; memset(&ga, 0, sizeof(ga));
  ga.I = (float const *)mm.filts; ga.J = (float const *)mm.in; ga.D = (float *)mm.out; ga.bias = mm.biases;
  ga.Mi = g.OC; ga.Nj = (int)Nj; ga.K = (int)Kt; ga.C = g.C; ga.H = g.H; ga.W = g.W; ga.OH = g.OH; ga.OW = g.OW;
  ga.I_bytes = (unsigned)f_bytes; ga.J_bytes = (unsigned)in_bytes; ga.D_bytes = (unsigned)out_bytes; ga.out_ctot = ctot; ga.out_coff = coff; ga.splitk = 1;
  ga.tiles_i = (g.OC + cfg.BI - 1) / cfg.BI; ga.tiles_j = (int)((Nj + cfg.BJ - 1) / cfg.BJ);
  return ga;
}

// What decides a set member's plan and the wrapper kernel -- ONE routine for run() and prebuild(), so that the ahead-of-time build compiles exactly the translation unit
// the first run would: a member's plan (its own launch's plan; K slices only in their in-launch form), then the members that share the wrapper (256 threads) longest
// tile first, variants numbered in that order, a lone 256-thread member launched on its own kernel instead.
struct set_member_in_t { conv_geom_t g; bool patch_filts, pool; int grp_pad; };
static plan_t plan_set_member(set_member_in_t const &mi, int num_cus, bool out_f32) {
  if (mi.grp_pad > 0) return plan_conv_nhwc(mi.g, num_cus, string(), out_f32, mi.grp_pad);
  return mi.patch_filts ? plan_conv_nhwc_patch(mi.g, num_cus, string(), out_f32, mi.pool) : plan_conv_nhwc(mi.g, num_cus, string(), out_f32, 0, /*allow_split=*/getenv("BODAHIP_NHWC_SPLITK2") == nullptr);
}
static double set_tile_cost(set_member_in_t const &mi, plan_t const &p) {
  return (double)p.cfg.BI * p.cfg.BJ * (double)mi.g.C * (mi.pool ? 2 : mi.g.KH * mi.g.KW) / std::max(1, p.ksl ? p.cfg.SPLITK : 1);
}
struct set_layout_t { std::vector<int> in_set, alone, variant_of; std::vector<plan_t const *> variants; std::vector<string> vkeys; int minw = 8; string skey; };
static set_layout_t layout_set(std::vector<plan_t> const &plans, std::vector<double> const &tile_cost) {
  set_layout_t L; L.variant_of.assign(plans.size(), -1);
  for (size_t m = 0; m < plans.size(); ++m) (plans[m].cfg.threads() == 256 ? L.in_set : L.alone).push_back((int)m);
  if (L.in_set.size() < 2) { L.alone.insert(L.alone.end(), L.in_set.begin(), L.in_set.end()); L.in_set.clear(); }
  // longest tiles first: the dispatcher hands workgroups out in grid order
  std::stable_sort(L.in_set.begin(), L.in_set.end(), [&](int x, int y) { return tile_cost[(size_t)x] > tile_cost[(size_t)y]; });
  for (int m : L.in_set) {
    string key = plans[(size_t)m].kname; for (auto const &d : plans[(size_t)m].defs) key += " " + d;
    size_t v = 0; while (v < L.vkeys.size() && L.vkeys[v] != key) ++v;
    if (v == L.vkeys.size()) { L.vkeys.push_back(key); L.variants.push_back(&plans[(size_t)m]); }
    L.variant_of[(size_t)m] = (int)v; L.minw = std::min(L.minw, plans[(size_t)m].cfg.MINW);
  }
  L.skey = "set:"; for (auto const &vk : L.vkeys) L.skey += "[" + vk + "]";
  return L;
}

void native_kernels_t::conv_nhwc_set(int n, multi_member_t const *ms, bool const *patch_filts, bool out_f32) {
  impl->call_ws_hold = 0;   // (a hold left behind by an earlier call that threw)
  if (n < 1 || n > 16) unsup_err("hip_conv_nhwc_set: 1..16 members");
  std::vector<set_member_plan_t> mp((size_t)n);
  std::vector<plan_t> plans((size_t)n); std::vector<double> costs((size_t)n);
  for (int m = 0; m < n; ++m) {
    conv_geom_t const &g = ms[m].g;
    if (!((long)g.B * g.OH * g.OW) || !g.OC) rt_err("hip_conv_nhwc_set: empty member");
    set_member_plan_t &q = mp[(size_t)m];
    memset(&q.q, 0, sizeof(q.q));
    set_member_in_t const mi{g, patch_filts[m], ms[m].pool, ms[m].grp_n > 0 ? ms[m].grp_pad : 0};
    if (ms[m].grp_n > 0) {   // a horizontally fused member: the GROUPS form of the implicit-GEMM kernel, its members' destinations in q
      if (patch_filts[m] || ms[m].pool) rt_err("hip_conv_nhwc_set: a fused (grp) member takes out_chan:y:x:in_chan filters");
      q.p = plan_set_member(mi, host->nh_num_cus(), out_f32);
      multi_member_t mm = ms[m]; mm.out_ctot = 0; mm.out_coff = 0;
      q.ga = nhwc_member_args(mm, q.p.cfg, out_f32, "hip_conv_nhwc_set"); q.ga.D = nullptr; q.ga.D_bytes = 0;
      long const Njg = (long)g.B * g.OH * g.OW; int tot = 0;
      q.q.n = ms[m].grp_n;
      for (int j = 0; j < ms[m].grp_n; ++j) {
        q.q.oc0[j] = tot; q.q.noc[j] = ms[m].grp_noc[j]; tot += (ms[m].grp_noc[j] + ms[m].grp_pad - 1) / ms[m].grp_pad * ms[m].grp_pad;
        uint64_t const ob = (uint64_t)Njg * ms[m].grp_ctot[j] * (out_f32 ? 4 : 2);
        if (ob >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc_set: out of 2 GiB or more");
        q.q.D[j] = ms[m].grp_out[j]; q.q.D_bytes[j] = (unsigned)ob; q.q.ctot[j] = ms[m].grp_ctot[j]; q.q.coff[j] = ms[m].grp_coff[j];
      }
      if (tot != g.OC) rt_err("hip_conv_nhwc_set: a fused member's filts hold " + std::to_string(g.OC) + " out_chans, its members need " + std::to_string(tot));
    } else {
      q.p = plan_set_member(mi, host->nh_num_cus(), out_f32);
      q.ga = nhwc_member_args(ms[m], q.p.cfg, out_f32, "hip_conv_nhwc_set");
    }
    if (q.p.ksl) {   // K slices reduced inside the launch: the member's grid is tiles x slices, its workspace its own
      long const nk = q.p.nhwc_patch ? ((long)(g.C / 8) + q.p.cg - 1) / q.p.cg : ((long)(g.C / 8) * g.KH * g.KW + q.p.cfg.BK / 8 - 1) / (q.p.cfg.BK / 8);
      if (!impl->call_ws_hold) { call_ws_make_room(impl, host, size_t(1) << 30); impl->call_ws_hold = 1; }   // (room for this launch's members first; then none of them may go while the others are set up)
      try { setup_ksl(impl, host, q.ga, q.p.cfg, nk, ms[m].grp_n > 0 ? ms[m].grp_out[0] : ms[m].out, "hip_conv_nhwc_set"); } catch (...) { impl->call_ws_hold = 0; throw; }
    }
    q.tiles = (long)q.ga.tiles_i * q.ga.tiles_j * std::max(1, q.ga.splitk);
    q.tile_cost = set_tile_cost(mi, q.p);
    plans[(size_t)m] = q.p; costs[(size_t)m] = q.tile_cost;
  }
  impl->call_ws_hold = 0;
  set_layout_t const L = layout_set(plans, costs);
  std::vector<int> const &in_set = L.in_set, &alone = L.alone;
  for (int m = 0; m < n; ++m) mp[(size_t)m].variant = L.variant_of[(size_t)m];
  double flops = 0, bytes = 0;
  for (int m = 0; m < n; ++m) { conv_geom_t g = ms[m].g; double const Nj = (double)g.B * g.OH * g.OW, Kt = ms[m].pool ? (double)g.C : (double)g.C * g.KH * g.KW;
    if (ms[m].grp_n > 0) { int roc = 0; for (int j = 0; j < ms[m].grp_n; ++j) roc += ms[m].grp_noc[j]; g.OC = roc; }   // (a fused member's own out_chans: zero padding rows are not credit)
    flops += 2.0 * Nj * g.OC * Kt; bytes += 2.0 * ((double)g.B * g.C * g.H * g.W + (double)g.OC * Kt) + (out_f32 ? 4.0 : 2.0) * Nj * g.OC + 4.0 * g.OC; }
  for (int m : alone) {   // members with another workgroup size (or a lone 256-thread member): their own launch, the plan they would have taken anyway
    kernel_t &k = get_kernel(impl, host, mp[(size_t)m].p);
    void *params[] = {&mp[(size_t)m].ga, &mp[(size_t)m].q};     // (the second argument is read by the GROUPS form only)
    hip_err_chk(host->nh_launch(k.func, (uint32_t)mp[(size_t)m].tiles, 1, (uint32_t)mp[(size_t)m].p.cfg.threads(), params), "hipModuleLaunchKernel(conv_nhwc_set, lone member)");
  }
  if (!in_set.empty()) {
    string const &skey = L.skey;
    auto kit = impl->kernels.find(skey);
    if (kit == impl->kernels.end()) {
      if (host->nh_capturing()) rt_err("graph capture: this hip_conv_nhwc_set kernel is not compiled yet -- run the call list once before capturing it");
      string log;
      std::vector<char> code = hiprtc_compile(set_kernel_source(L.variants, 256, std::max(1, L.minw)), "bodahip_conv_nhwc_set", host->nh_arch(), vect_string(), &log, true);
      kernel_t k;
      hip_err_chk(hipModuleLoadData(&k.mod, code.data()), "hipModuleLoadData(conv_nhwc_set)");
      hip_err_chk(hipModuleGetFunction(&k.func, k.mod, "bodahip_conv_nhwc_set"), "hipModuleGetFunction(conv_nhwc_set)");
      kit = impl->kernels.emplace(skey, k).first;
    }
    // member table (arguments, grid ends, variant ids) in device memory: one copy per distinct call
    size_t const ns = in_set.size();
    std::vector<gemm_args_t> args; std::vector<grp_args_t> gargs; std::vector<int> ends, vars; long tot = 0;
    for (int m : in_set) { args.push_back(mp[(size_t)m].ga); gargs.push_back(mp[(size_t)m].q); tot += mp[(size_t)m].tiles; ends.push_back((int)tot); vars.push_back(mp[(size_t)m].variant); }
    if (tot > 0x7fffffffl) unsup_err("hip_conv_nhwc_set: too many tiles");
    size_t const ab = ns * sizeof(gemm_args_t), eo = (ab + 255) & ~size_t(255), vo = eo + ((ns * 4 + 255) & ~size_t(255)), go = vo + ((ns * 4 + 255) & ~size_t(255)), total = go + ns * sizeof(grp_args_t);
    string tkey = "settab:";
    { uint64_t h = 1469598103934665603ull; auto mix = [&](void const *d, size_t nb) { for (size_t i = 0; i < nb; ++i) { h ^= ((unsigned char const *)d)[i]; h *= 1099511628211ull; } };
      mix(args.data(), ab); mix(gargs.data(), ns * sizeof(grp_args_t)); mix(ends.data(), ns * 4); mix(vars.data(), ns * 4); mix(skey.data(), skey.size()); tkey += std::to_string(h) + ":" + std::to_string(total); }
    auto it = impl->ktabs.find(tkey);
    if (it == impl->ktabs.end()) {
      if (host->nh_capturing()) rt_err("graph capture: the member table of this hip_conv_nhwc_set call is not on the device yet -- run the call list once before capturing it");
      void *dev = nullptr;
      hip_err_chk(hipMalloc(&dev, total), "hipMalloc(set table)");
      hip_err_chk(hipMemcpyAsync(dev, args.data(), ab, hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(set args)");
      hip_err_chk(hipMemcpyAsync((char *)dev + eo, ends.data(), ns * 4, hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(set ends)");
      hip_err_chk(hipMemcpyAsync((char *)dev + vo, vars.data(), ns * 4, hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(set variants)");
      hip_err_chk(hipMemcpyAsync((char *)dev + go, gargs.data(), ns * sizeof(grp_args_t), hipMemcpyHostToDevice, host->nh_stream()), "hipMemcpyAsync(set grp args)");
      hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize(set table)");
      it = impl->ktabs.emplace(tkey, dev).first;
    }
    struct { gemm_args_t const *m; int const *ends; int const *variant; grp_args_t const *g; int n; } sa;
    sa.m = (gemm_args_t const *)it->second; sa.ends = (int const *)((char *)it->second + eo); sa.variant = (int const *)((char *)it->second + vo);
    sa.g = (grp_args_t const *)((char *)it->second + go); sa.n = (int)ns;
    void *params[] = {&sa};
    hip_err_chk(host->nh_launch(kit->second.func, (uint32_t)tot, 1, 256, params), "hipModuleLaunchKernel(conv_nhwc_set)");
    last_launch.cfg = mp[(size_t)in_set[0]].p.cfg; last_launch.grid = (uint32_t)tot; last_launch.block = 256;
  } else { last_launch.cfg = mp[0].p.cfg; last_launch.grid = (uint32_t)mp[0].tiles; last_launch.block = (uint32_t)mp[0].p.cfg.threads(); }
  last_launch.kernel = "bodahip_conv_nhwc_set(x" + std::to_string(in_set.size()) + (alone.empty() ? string() : ("+" + std::to_string(alone.size()))) + ")";
  last_launch.flops = flops; last_launch.algo_bytes = bytes;
}

// Horizontally fused channels-last convolutions (hip_conv_nhwc_grp): n <= 4 members that read the same `in` with the same kernel geometry; filts / biases hold
// the members stacked along out_chan, member m at rows [m_oc0, m_oc0 + noc[m]) with m_oc0 = sum of the earlier members' out_chans each rounded up to `pad`.
void native_kernels_t::conv_nhwc_grp(void const *filts, float const *biases, void const *in, conv_geom_t const &g, bool out_f32, int n, int const *noc, void *const *outs,
                                     int const *ctot, int const *coff, int pad) {
  if (n < 1 || n > 4 || pad < 32 || pad % 32) unsup_err("hip_conv_nhwc_grp: 1..4 members, padding a multiple of 32 out_chans");
  long const Nj = (long)g.B * g.OH * g.OW, Kt = (long)g.C * g.KH * g.KW;
  if (!Nj) return;
  if (Nj > 0x7fffffffl || Kt > 0x7fffffffl) unsup_err("hip_conv_nhwc_grp: dims exceed int32");
  grp_args_t q; memset(&q, 0, sizeof(q)); q.n = n;
  int tot = 0; double real_oc = 0;
  for (int m = 0; m < n; ++m) {
    if (noc[m] < 1) rt_err("hip_conv_nhwc_grp: empty member");
    q.oc0[m] = tot; q.noc[m] = noc[m]; tot += (noc[m] + pad - 1) / pad * pad; real_oc += noc[m];
    uint64_t const ob = (uint64_t)Nj * ctot[m] * (out_f32 ? 4 : 2);
    if (ob >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc_grp: out of 2 GiB or more");
    q.D[m] = outs[m]; q.D_bytes[m] = (unsigned)ob; q.ctot[m] = ctot[m]; q.coff[m] = coff[m];
  }
  if (tot != g.OC) rt_err("hip_conv_nhwc_grp: filts hold " + std::to_string(g.OC) + " out_chans, the members (padded to " + std::to_string(pad) + ") need " + std::to_string(tot));
  plan_t const p = plan_conv_nhwc(g, host->nh_num_cus(), tune_of(impl, "conv_tile"), out_f32, pad);
  tile_cfg_t const &cfg = p.cfg;
  kernel_t &k = get_kernel(impl, host, p);
  gemm_args_t ga; memset(&ga, 0, sizeof(ga));
  uint64_t const in_bytes = (uint64_t)g.B * g.C * g.H * g.W * 2, f_bytes = (uint64_t)g.OC * Kt * 2;
  if (in_bytes >= 0x7ffffff0ull || f_bytes >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc_grp: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
  ga.I = (float const *)filts; ga.J = (float const *)in; ga.D = nullptr; ga.bias = biases;
  ga.Mi = g.OC; ga.Nj = (int)Nj; ga.K = (int)Kt; ga.C = g.C; ga.H = g.H; ga.W = g.W; ga.OH = g.OH; ga.OW = g.OW;
  ga.I_bytes = (unsigned)f_bytes; ga.J_bytes = (unsigned)in_bytes; ga.splitk = 1;
  ga.tiles_i = (g.OC + cfg.BI - 1) / cfg.BI; ga.tiles_j = (int)((Nj + cfg.BJ - 1) / cfg.BJ);
  if (p.ksl) setup_ksl(impl, host, ga, cfg, ((long)(g.C / 8) * g.KH * g.KW + cfg.BK / 8 - 1) / (cfg.BK / 8), outs[0], "hip_conv_nhwc_grp");
  void *params[] = {&ga, &q};
  hip_err_chk(host->nh_launch(k.func, (uint32_t)(ga.tiles_i * ga.tiles_j * ga.splitk), 1, (uint32_t)cfg.threads(), params), "hipModuleLaunchKernel(conv_nhwc_bf16, fused)");
  last_launch.kernel = p.kname + "(x" + std::to_string(n) + ")"; last_launch.cfg = cfg; last_launch.grid = (uint32_t)(ga.tiles_i * ga.tiles_j * ga.splitk); last_launch.block = cfg.threads();
  last_launch.flops = 2.0 * Nj * real_oc * Kt;   // (the members' own out_chans: zero padding rows are work done, not credit)
  last_launch.algo_bytes = 2.0 * ((double)g.B * g.C * g.H * g.W + real_oc * Kt) + (out_f32 ? 4.0 : 2.0) * (double)Nj * real_oc + 4.0 * real_oc;
}

void native_kernels_t::conv_nhwc(void const *filts, float const *biases, void const *in, void *out, conv_geom_t const &g, bool out_f32, int out_ctot, int out_coff, bool patch_filts, bool pool) {
  if (out_ctot <= 0) { out_ctot = g.OC; out_coff = 0; }
  if (pool && !patch_filts) rt_err("hip_conv_nhwc: fused pooling needs the patch form of filts");
  long const Nj = (long)g.B * g.OH * g.OW, Kt = pool ? (long)g.C : (long)g.C * g.KH * g.KW;
  if (!Nj || !g.OC) return;
  if (Nj > 0x7fffffffl || Kt > 0x7fffffffl) unsup_err("hip_conv_nhwc: dims exceed int32");
  if (patch_filts && !pool && !out_f32 && rows_auto(g, host->nh_num_cus(), tune_of(impl, "conv_tile"))) { conv_nhwc_rows(filts, biases, in, out, g, post_ops_t(), out_ctot, out_coff); return; }   // (output-bound stems)
  plan_t const p = patch_filts ? plan_conv_nhwc_patch(g, host->nh_num_cus(), tune_of(impl, "conv_tile"), out_f32, pool) : plan_conv_nhwc(g, host->nh_num_cus(), tune_of(impl, "conv_tile"), out_f32);
  tile_cfg_t const &cfg = p.cfg;
  kernel_t &k = get_kernel(impl, host, p);
  gemm_args_t ga; memset(&ga, 0, sizeof(ga));
  uint64_t const in_bytes = (uint64_t)g.B * g.C * g.H * g.W * 2, f_bytes = (uint64_t)g.OC * Kt * 2, out_bytes = (uint64_t)Nj * out_ctot * (out_f32 ? 4 : 2);
  if (in_bytes >= 0x7ffffff0ull || f_bytes >= 0x7ffffff0ull || out_bytes >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc: tensors of 2 GiB or more are not supported (32-bit buffer offsets)");
  ga.I = (float const *)filts; ga.J = (float const *)in; ga.D = (float *)out; ga.bias = biases;
  ga.Mi = g.OC; ga.Nj = (int)Nj; ga.K = (int)Kt; ga.C = g.C; ga.H = g.H; ga.W = g.W; ga.OH = g.OH; ga.OW = g.OW;
  ga.I_bytes = (unsigned)f_bytes; ga.J_bytes = (unsigned)in_bytes; ga.D_bytes = (unsigned)out_bytes;
  ga.out_ctot = out_ctot; ga.out_coff = out_coff; ga.splitk = 1;
  ga.tiles_i = (g.OC + cfg.BI - 1) / cfg.BI; ga.tiles_j = (int)((Nj + cfg.BJ - 1) / cfg.BJ);
  if (p.ksl) {
    long const nk = p.nhwc_patch ? ((long)(g.C / 8) + p.cg - 1) / p.cg : ((long)(g.C / 8) * g.KH * g.KW + cfg.BK / 8 - 1) / (cfg.BK / 8);
    setup_ksl(impl, host, ga, cfg, nk, out, "hip_conv_nhwc");
  } else if (cfg.SPLITK > 1) {
    long const nk = ((long)(g.C / 8) * g.KH * g.KW + cfg.BK / 8 - 1) / (cfg.BK / 8);
    size_t const slab = ((size_t)Nj * g.OC + 3) & ~size_t(3);
    if (slab * 4 >= 0x7ffffff0ull) unsup_err("hip_conv_nhwc: split-K slab of 2 GiB or more");
    ensure_ws(impl, host, slab * (size_t)cfg.SPLITK * sizeof(float));
    ga.splitk = cfg.SPLITK; ga.kt_per = (int)((nk + cfg.SPLITK - 1) / cfg.SPLITK); ga.ws = (float *)impl->ws; ga.ws_slab = (long)slab;
  }
  void *params[] = {&ga};
  hip_err_chk(host->nh_launch(k.func, (uint32_t)(ga.tiles_i * ga.tiles_j * ga.splitk), 1, (uint32_t)cfg.threads(), params), "hipModuleLaunchKernel(conv_nhwc_bf16)");
  if (cfg.SPLITK > 1 && !p.ksl) {
    plan_t rp; rp.nhwc = true; rp.bf16 = true; rp.kname = "bodahip_nhwc_splitk_reduce";
    rp.defs = {"-DREDUCE_ONLY=1", string("-DRELU=") + (g.relu ? "1" : "0"), string("-DOUT_F32=") + (out_f32 ? "1" : "0")};
    kernel_t &rk = get_kernel(impl, host, rp);
    bool const v4 = (g.OC % 4 == 0) && (((out_ctot | out_coff) & 3) == 0);
    long const n = v4 ? Nj * g.OC / 4 : Nj * g.OC;
    hip_err_chk(host->nh_launch(rk.func, (uint32_t)((n + 255) / 256), 1, 256, params), "hipModuleLaunchKernel(nhwc_splitk_reduce)");
  }
  last_launch.kernel = p.kname; last_launch.cfg = cfg; last_launch.grid = (uint32_t)(ga.tiles_i * ga.tiles_j * ga.splitk); last_launch.block = cfg.threads();
  last_launch.flops = 2.0 * Nj * g.OC * Kt;   // (as stored: zero pad channels of a conv1-type layer count as work done, not as credit -- bench.py credits the op's own 2MNK)
  last_launch.algo_bytes = 2.0 * ((double)g.B * g.C * g.H * g.W + (double)g.OC * Kt) + (out_f32 ? 4.0 : 2.0) * (double)Nj * g.OC + 4.0 * g.OC;
}

static conv_geom_t geom_from_dims(dims_t const &f, dims_t const &in, dims_t const &out, dims_t const &stride, dims_t const &in_pad, bool relu) {
  conv_geom_t g;
  g.B = in.dsz("img"); g.C = in.dsz("chan"); g.H = in.dsz("y"); g.W = in.dsz("x");
  g.OC = f.dsz("out_chan"); g.KH = f.dsz("y"); g.KW = f.dsz("x");
  g.SY = stride.dsz("y"); g.SX = stride.dsz("x"); g.PY = in_pad.dsz("y"); g.PX = in_pad.dsz("x");
  g.OH = out.dsz("y"); g.OW = out.dsz("x"); g.relu = relu;
  return g;
}

// Max pooling fused in front of a 1x1 convolution (annotation: uint32 nhwc_pool[<sfx>] = 1, REF-style dims pool_sz[<sfx>] / pool_pad[<sfx>] carried by the op): the
// function's `in` is the POOLING's input; the geometry handed to the patch kernel takes the pooling's window and padding (stride 1), the filters stay 1x1.
static bool apply_pool_window(op_base_t const &op, string const &sfx, conv_geom_t &g, char const *what) {
  if (!op.has("nhwc_pool" + sfx) || !op.get_u32("nhwc_pool" + sfx)) return false;
  dims_t const &ks = op.get_dims("pool_sz" + sfx), &pp = op.get_dims("pool_pad" + sfx);
  if (!(g.KH == 1 && g.KW == 1 && g.SY == 1 && g.SX == 1 && g.PY == 0 && g.PX == 0)) rt_err(string(what) + ": fused pooling needs a 1x1 / stride-1 / unpadded convolution");
  g.KH = (int)ks.dsz("y"); g.KW = (int)ks.dsz("x"); g.PY = (int)pp.dsz("y"); g.PX = (int)pp.dsz("x");
  if (g.KH * g.KW < 2 || g.KH * g.KW > 25) unsup_err(string(what) + ": fused pooling windows of 2..25 positions");
  return true;
}

// fp32 hip_conv with a max pooling fused in front (annotation: uint32 hip_pool = 1, dims pool_sz / pool_stride carried by the op; boda_amd/conv_pipe.py): `in` is the
// POOLING's input.  g arrives with H / W = that tensor's planes; they become the pooled plane.  Only windows that tile the plane exactly (no pooling pad, no clipped window).
static bool apply_f32_pool(op_base_t const &op, conv_geom_t &g, char const *what) {
  if (!op.has("hip_pool") || !op.get_u32("hip_pool")) return false;
  dims_t const &ks = op.get_dims("pool_sz"), &st = op.get_dims("pool_stride");
  g.PKH = (int)ks.dsz("y"); g.PKW = (int)ks.dsz("x"); g.PSY = (int)st.dsz("y"); g.PSX = (int)st.dsz("x"); g.UH = g.H; g.UW = g.W;
  if (g.PKH < 1 || g.PKW < 1 || g.PKH > 3 || g.PKW > 3 || g.PKH * g.PKW < 2 || g.PSY < 1 || g.PSX < 1 || g.UH < g.PKH || g.UW < g.PKW) unsup_err(string(what) + ": fused pooling takes windows of 2..9 positions, at most 3 x 3");
  if ((g.UH - g.PKH) % g.PSY || (g.UW - g.PKW) % g.PSX) unsup_err(string(what) + ": fused pooling needs windows that tile the plane exactly (no clipped last window)");
  g.H = (g.UH - g.PKH) / g.PSY + 1; g.W = (g.UW - g.PKW) / g.PSX + 1;
  return true;
}

// AOT: compile (into the on-disk code-object cache) the specialisation that run() would pick for `op`.  No device needed.
// With arch == "" nothing is compiled and *plan_out receives "<kernel> <tile> <-D options>": the planner's decision (host-logic tests).
size_t native_kernels_t::prebuild(op_base_t const &op, string const &arch, int num_cus, string const &tile_arg, string *plan_out) {
  string const &t = op.get_type();
  // a tile that travels with the function (str_val hip_tile: per-op tuned tiles, see tile_override_t) is what run() would use
  auto const ht = op.str_vals.find("hip_tile");
  string const tile = (tile_arg.empty() && ht != op.str_vals.end()) ? ht->second : tile_arg;
  plan_t p; string log, s2d;
  bool const bf16 = op.has_func_name() && (op.get_func_name() == "hip_sgemm_bf16" || op.get_func_name() == "hip_conv_bf16");
  if (t == "sgemm") {
    dims_t const &a = op.get_dims("a"), &b = op.get_dims("b");
    string const wide = (!bf16 && tile.empty() && a.tn != "half") ? sgemm_wide_tile(a.dsz("M"), b.dsz("N"), a.dsz("K"), num_cus) : string();
    sgemm_split_t sp; if (!bf16 && tile.empty() && a.tn != "half" && wide.empty()) sp = plan_sgemm_split(a.dsz("M"), b.dsz("N"), a.dsz("K"), num_cus);
    std::vector<sgemm_part_t> parts; if (!bf16 && tile.empty() && a.tn != "half") parts = plan_sgemm_parts(a.dsz("M"), b.dsz("N"), a.dsz("K"), num_cus);
    if (!parts.empty()) {   // guillotine decomposition: every part's plan is compiled, the last one reported in full
      s2d = "parts=" + std::to_string(parts.size());
      for (size_t i = 0; i < parts.size(); ++i) { sgemm_part_t const &q = parts[i];
        p = plan_sgemm(q.rows, q.cols, a.dsz("K"), num_cus, q.tile, false);
        s2d += " [" + std::to_string(q.m0) + "+" + std::to_string(q.rows) + "," + std::to_string(q.n0) + "+" + std::to_string(q.cols) + "]:" + p.cfg.str();
        if (i + 1 < parts.size() && !arch.empty()) compile_plan(p, arch, &log); }
      s2d += " last:";
    }
    else if (!wide.empty()) p = plan_sgemm(a.dsz("M"), b.dsz("N"), a.dsz("K"), num_cus, wide, false);
    else if (sp.m_main) {   // two-level tiling: the large tile over the first m_main rows (reported), small tiles over the rest
      plan_t const tp = plan_sgemm(a.dsz("M") - sp.m_main, b.dsz("N"), a.dsz("K"), num_cus, sp.tail_tile, false);
      p = plan_sgemm(sp.m_main, b.dsz("N"), a.dsz("K"), num_cus, kBigTile, false);
      s2d = "rows<" + std::to_string(sp.m_main) + ":" + p.cfg.str() + "+rest:";
      if (!arch.empty()) compile_plan(p, arch, &log);
      p = tp;
    } else p = plan_sgemm(a.dsz("M"), b.dsz("N"), a.dsz("K"), num_cus, tile, bf16);
    if (a.tn == "half") { s2d.clear(); p = plan_sgemm(a.dsz("M"), b.dsz("N"), a.dsz("K"), num_cus, tile, false, 1, false); p.kname = "bodahip_sgemm_f16s"; p.defs.push_back("-DHALF=1"); }
  }
  else if (t == "Convolution") {
    bool const relu = op.has("conv_has_relu") ? (op.get_u32("conv_has_relu") != 0) : true;
    bool const multi = op.has_func_name() && op.get_func_name() == "hip_conv_nhwc_multi";
    if (op.has_func_name() && op.get_func_name() == "hip_conv_nhwc_set") {   // the wrapper kernel of the members' specialisations (and the kernels of members that stay alone)
      int const n = (int)op.get_dims("multi").dsz("n"); bool const out_f32 = op.get_dims(op.has("out_0") ? "out_0" : "out_0_0").tn == "float";
      std::vector<plan_t> plans; std::vector<double> costs; size_t bytes = 0; string desc;
      for (int m = 0; m < n; ++m) { string const sfx = "_" + std::to_string(m);
        bool const relu_m = op.has("relu_mask") ? (((op.get_u32("relu_mask") >> m) & 1u) != 0) : relu;   // (per member, fused members included)
        set_member_in_t mi; memset(&mi, 0, sizeof(mi));
        if (op.has("grp" + sfx)) {   // a horizontally fused member
          dims_t const &grp = op.get_dims("grp" + sfx);
          mi.g = geom_from_dims(op.get_dims("filts" + sfx), op.get_dims("in" + sfx), op.get_dims("out_0" + sfx), op.get_dims("stride" + sfx), op.get_dims("in_pad" + sfx), relu_m);
          mi.grp_pad = (int)grp.dims(grp.sz() - 1);
        } else {
          dims_t f = op.get_dims("filts" + sfx); mi.patch_filts = f.sz() == 5;
          if (mi.patch_filts) f = dims_t({f.dims(3), f.dims(1), f.dims(2), f.dims(0) * 8}, {"out_chan", "y", "x", "in_chan"}, f.tn);
          mi.g = geom_from_dims(f, op.get_dims("in" + sfx), op.get_dims("out" + sfx), op.get_dims("stride" + sfx), op.get_dims("in_pad" + sfx), relu_m);
          mi.pool = apply_pool_window(op, sfx, mi.g, "hip_conv_nhwc_set");
        }
        plans.push_back(plan_set_member(mi, num_cus, out_f32)); costs.push_back(set_tile_cost(mi, plans.back())); }
      set_layout_t const L = layout_set(plans, costs);   // (the ordering, variant numbering and lone-member rule of conv_nhwc_set)
      for (int m : L.alone) { plan_t const &q = plans[(size_t)m]; if (!arch.empty()) bytes += compile_plan(q, arch, &log).size(); desc += " alone:" + q.kname + ":" + q.cfg.str(); }
      for (plan_t const *q : L.variants) desc += " " + q->kname + ":" + q->cfg.str();
      if (plan_out) *plan_out = "bodahip_conv_nhwc_set variants=" + std::to_string(L.variants.size()) + desc;
      if (arch.empty()) return 0;
      if (L.variants.size() >= 1) bytes += hiprtc_compile(set_kernel_source(L.variants, 256, std::max(1, L.minw)), "bodahip_conv_nhwc_set", arch, vect_string(), &log, true).size();
      return bytes;
    }
    conv_geom_t g; memset(&g, 0, sizeof(g));
    if (!multi) g = geom_from_dims(op.get_dims("filts"), op.get_dims("in"), op.get_dims(op.has("out") ? "out" : "out_0"), op.get_dims("stride"), op.get_dims("in_pad"), relu);
    if (!multi) (void)apply_f32_pool(op, g, "prebuild");   // (hip_conv with a pooling fused in front: `in` is the pooling's input)
    conv_geom_t g2; int pry = 0, prx = 0;
    if (multi) {
      int const n = (int)op.get_dims("multi").dsz("n"); std::vector<conv_geom_t> gs;
      for (int m = 0; m < n; ++m) { string const sfx = "_" + std::to_string(m);
        gs.push_back(geom_from_dims(op.get_dims("filts" + sfx), op.get_dims("in" + sfx), op.get_dims("out" + sfx), op.get_dims("stride" + sfx), op.get_dims("in_pad" + sfx), relu)); }
      p = plan_conv_nhwc_multi(gs, tile, op.get_dims("out_0").tn == "float");
    }
    else if (op.has_func_name() && op.get_func_name() == "hip_conv_nhwc" && op.get_dims("filts").sz() == 5) {
      conv_geom_t gp = g; bool pool = false;
      if (op.has("nhwc_pool") && op.get_u32("nhwc_pool")) {   // (filts are in_grp:1:1:out_chan:8: geom_from_dims read in_grp / 1 as out_chan / y -- rebuild from the logical dims)
        dims_t const &f5 = op.get_dims("filts");
        dims_t const fl({f5.dims(3), f5.dims(1), f5.dims(2), f5.dims(0) * 8}, {"out_chan", "y", "x", "in_chan"}, f5.tn);
        gp = geom_from_dims(fl, op.get_dims("in"), op.get_dims("out"), op.get_dims("stride"), op.get_dims("in_pad"), relu);
        pool = apply_pool_window(op, string(), gp, "hip_conv_nhwc");
      }
      post_ops_t post; string why;
      if (apply_post_ops(op, gp, post, "prebuild")) { if (pool || !plan_conv_nhwc_rows(gp, post, num_cus, p, &why)) unsup_err("hip_conv_nhwc (rolling-rows form): " + (pool ? string("no pooling in front") : why)); }
      else if (!pool && op.get_dims("out").tn != "float" && rows_auto(gp, num_cus, tile)) plan_conv_nhwc_rows(gp, post_ops_t(), num_cus, p);
      else p = plan_conv_nhwc_patch(gp, num_cus, tile, op.get_dims("out").tn == "float", pool);
    }
    else if (op.has_func_name() && op.get_func_name() == "hip_conv_k1_chain") {
      if (!plan_k1_chain(g, (int)op.get_dims("filts2").dsz("out_chan"), op.get_u32("conv_has_relu2") != 0, p)) unsup_err("prebuild: hip_conv_k1_chain does not cover this pair of convolutions");
    }
    else if (op.has_func_name() && op.get_func_name() == "hip_conv_nhwc") p = plan_conv_nhwc(g, num_cus, tile, op.get_dims("out").tn == "float");
    else if (op.has_func_name() && op.get_func_name() == "hip_conv_nhwc_grp") { dims_t const &grp = op.get_dims("grp"); p = plan_conv_nhwc(g, num_cus, tile, op.get_dims("out_0").tn == "float", (int)grp.dims(grp.sz() - 1)); }
    else if (bf16 && tile.empty() && s2d_geom(g, g2, pry, prx) && plan_patch_bf16(g2, num_cus, p)) { // conv1-type layers: space-to-depth front end (see conv())
      s2d = "s2d(" + std::to_string(g2.C) + "x" + std::to_string(g2.H) + "x" + std::to_string(g2.W) + ",k" + std::to_string(g2.KH) + "x" + std::to_string(g2.KW) + ")+";
      if (!arch.empty()) { plan_t sp; sp.patch16 = true; sp.bf16 = true; sp.kname = "bodahip_s2d"; sp.defs = {"-DS2D_ONLY=1"}; compile_plan(sp, arch, &log); }
    } else {
      auto xe = op.str_vals.find("hip_exact"); bool const exact = !(xe != op.str_vals.end() && xe->second == "0");
      // the same resolution as conv(): in tolerance mode (and with no conv_algo / tile given) the 3x3 / stride-1 layers take the F(2x2,3x3) pipeline -- what is
      // compiled ahead of time and reported is then ITS kernels (the transforms' module and the batched transform-domain sgemm of every chunk size)
      if (!bf16 && !exact && tile.empty() && winograd_applies(g, "winograd")) {
        int const tpi = ((g.OH + 1) / 2) * ((g.OW + 1) / 2); long const Bc = wino_chunk_imgs(g);
        s2d = "winograd(F2x2,3x3)+";
        if (!arch.empty()) hiprtc_compile(k_src_winograd_f32, "bodahip_winograd", arch, vect_string(), &log, true);
        long const rem = g.B % Bc;
        if (rem) { plan_t const rp = plan_sgemm((uint32_t)g.OC, (uint32_t)(rem * tpi), (uint32_t)g.C, num_cus, string(), false, 16); if (!arch.empty()) compile_plan(rp, arch, &log); }
        p = plan_sgemm((uint32_t)g.OC, (uint32_t)(std::min<long>(Bc, g.B) * tpi), (uint32_t)g.C, num_cus, string(), false, 16);
      } else { char const *k1e = getenv("BODAHIP_K1_STREAM"); p = plan_conv(g, num_cus, tile, bf16, k1e ? string(k1e) : string(), true, exact); }   // (the env var a backend instance reads its k1_stream tune from)
    }
  } else rt_err("prebuild: op type '" + t + "' has no native kernel");
  if (plan_out) { *plan_out = s2d + p.kname + " " + p.cfg.str(); for (auto const &d : p.defs) *plan_out += " " + d;
    if (p.split_pels > 0) *plan_out += " pels<" + std::to_string(p.split_pels) + "+rest:" + p.tail_cfg.str(); }
  if (arch.empty()) return 0;
  size_t const n = compile_plan(p, arch, &log).size();
  if (p.cbig && p.split_pels > 0) { plan_t tp; tp.cbig = true; tp.kname = p.kname; tp.cfg = p.tail_cfg; tp.defs = p.tail_defs; compile_plan(tp, arch, &log); }
  if (p.cbig) { plan_t xp; xp.cbig = true; xp.kname = "bodahip_conv_big_xpose"; xp.defs = {"-DXPOSE_ONLY=1"}; compile_plan(xp, arch, &log); }   // (the filter transposition that runs in front of it)
  if (p.patch16) { plan_t fp; fp.patch16 = true; fp.bf16 = true; fp.kname = "bodahip_filt_bf16"; fp.defs = {"-DFILT_ONLY=1"}; compile_plan(fp, arch, &log); }
  if (p.ksl) {   // (K slices reduced inside the launch: no second kernel)
  } else if (p.cfg.SPLITK > 1 && p.nhwc) {
    plan_t rp; rp.nhwc = true; rp.bf16 = true; rp.kname = "bodahip_nhwc_splitk_reduce";
    bool const relu = op.has("conv_has_relu") ? (op.get_u32("conv_has_relu") != 0) : true;
    rp.defs = {"-DREDUCE_ONLY=1", string("-DRELU=") + (relu ? "1" : "0"), string("-DOUT_F32=") + ((op.get_dims("out").tn == "float") ? "1" : "0")};
    compile_plan(rp, arch, &log);
  } else if (p.cfg.SPLITK > 1) { // the matching second-pass kernel
    plan_t r; r.kname = "bodahip_splitk_reduce"; bool const epi = (t == "Convolution");
    bool const relu = epi && (op.has("conv_has_relu") ? (op.get_u32("conv_has_relu") != 0) : true);
    r.defs = {"-DREDUCE_ONLY=1", string("-DRED_EPI=") + (epi ? "1" : "0"), string("-DRED_RELU=") + (relu ? "1" : "0")};
    compile_plan(r, arch, &log);
  }
  return n;
}

static kernel_t &get_kernel(native_kernels_t::impl_t *impl, native_host_t *host, plan_t const &p) {
  string key = p.kname; for (auto const &d : p.defs) key += " " + d;
  auto it = impl->kernels.find(key);
  if (it != impl->kernels.end()) return it->second;
  if (host->nh_capturing()) rt_err("graph capture: native kernel '" + key + "' is not specialised yet -- run the call list once before capturing it");
  string log;
  std::vector<char> code = compile_plan(p, host->nh_arch(), &log);
  kernel_t k;
  hip_err_chk(hipModuleLoadData(&k.mod, code.data()), "hipModuleLoadData(native)");
  hip_err_chk(hipModuleGetFunction(&k.func, k.mod, p.kname.c_str()), "hipModuleGetFunction(native)");
  return impl->kernels.emplace(key, k).first->second;
}

static string var_of(map_str_rtc_arg_t const &am, string const &an) {
  auto i = am.find(an);
  if (i == am.end()) rt_err("native hip function: arg '" + an + "' not found in arg_map for call.");
  if (!i->second.is_valid() || !i->second.is_var()) rt_err("native hip function: arg '" + an + "' must be a var");
  return i->second.n;
}
static void need_float(dims_t const &d, char const *an) {
  if (d.tn != "float") unsup_err(string("native hip kernels: arg '") + an + "' has type " + d.tn + "; only float storage is supported");
}

// a function may carry its own tile (str_val hip_tile of the annotated op: per-layer tuned tiles, the op_tune_t-per-op analogue of the
// reference's wisdom files); it overrides the backend-wide tune for that call only
struct tile_override_t {
  native_kernels_t::impl_t *impl; char const *key; bool active = false, had = false; string old;
  tile_override_t(native_kernels_t::impl_t *impl_, char const *key_, op_base_t const &op) : impl(impl_), key(key_) {
    auto it = op.str_vals.find("hip_tile");
    if (it == op.str_vals.end() || it->second.empty()) return;
    active = true; auto t = impl->tune.find(key); had = (t != impl->tune.end()); if (had) old = t->second;
    impl->tune[key] = it->second;
  }
  ~tile_override_t() { if (!active) return; if (had) impl->tune[key] = old; else impl->tune.erase(key); }
};

// str_val hip_exact of the annotated op (op_tune hip_exact=0): tolerance mode for this function's calls only
struct exact_override_t {
  native_kernels_t::impl_t *impl; bool active = false, had = false; string old;
  exact_override_t(native_kernels_t::impl_t *impl_, op_base_t const &op) : impl(impl_) {
    auto it = op.str_vals.find("hip_exact");
    if (it == op.str_vals.end() || it->second.empty()) return;
    if (it->second != "0" && it->second != "1") rt_err("hip_exact must be 0 | 1, got '" + it->second + "'");
    active = true; auto t = impl->tune.find("exact"); had = (t != impl->tune.end()); if (had) old = t->second;
    impl->tune["exact"] = it->second;
  }
  ~exact_override_t() { if (!active) return; if (had) impl->tune["exact"] = old; else impl->tune.erase("exact"); }
};

void native_kernels_t::run(rtc_func_info_t const &fi, map_str_rtc_arg_t const &am) {
  string const &fn = fi.op.get_func_name();
  exact_override_t const xov(impl, fi.op);
  bool const bf16 = (fn == "hip_sgemm_bf16" || fn == "hip_conv_bf16");
  if (fn == "hip_sgemm" || fn == "cublas_sgemm" || fn == "hip_sgemm_bf16") {
    string const an = var_of(am, "a"), bn = var_of(am, "b"), cn = var_of(am, "c");
    dims_t const a = host->nh_var_dims(an), b = host->nh_var_dims(bn), c = host->nh_var_dims(cn);
    // storage type: float, or all three `half` (16-bit storage, fp32 math: the reference's sgemm with __tn__=half dims, test/sgemm-ops-debug-half.txt)
    bool const half = (a.tn == "half" && b.tn == "half" && c.tn == "half");
    if (!half) { need_float(a, "a"); need_float(b, "b"); need_float(c, "c"); }
    uint32_t const M = a.dsz("M"), K = a.dsz("K"), N = b.dsz("N");
    // same consistency checks as culibs_wrap_t::sgemm (src/culibs-wrap.cc:218-225); a is K:M, b is K:N, c is M:N
    assert_st(a.sz() == 2 && b.sz() == 2 && c.sz() == 2);
    assert_st(a.names(0) == "K" && a.names(1) == "M" && b.names(0) == "K" && b.names(1) == "N" && c.names(0) == "M" && c.names(1) == "N");
    assert_st(b.dsz("K") == K); assert_st(c.dsz("M") == M); assert_st(c.dsz("N") == N);
    tile_override_t const tov(impl, "sgemm_tile", fi.op);
    sgemm((float const *)host->nh_var_ptr(an), (float const *)host->nh_var_ptr(bn), (float *)host->nh_var_ptr(cn), M, N, K, bf16, half);
    return;
  }
  if (fn == "hip_conv_nhwc_grp") {
    // horizontally fused channels-last convolutions: filts / biases stacked and padded (REF `grp`: dims m0..m{n-1} = the members' out_chans, pad = padding granularity),
    // outputs out_0 .. out_{n-1} (vars), optional by-value out_chan_off_<m> (member m writes a channel slice of a wider tensor)
    string const fnm = var_of(am, "filts"), bnm = var_of(am, "biases"), inm = var_of(am, "in");
    dims_t const f = host->nh_var_dims(fnm), bi = host->nh_var_dims(bnm), in = host->nh_var_dims(inm);
    need_float(bi, "biases");
    if (f.tn != "bfloat16" || in.tn != "bfloat16") unsup_err("hip_conv_nhwc_grp: filts / in must have type bfloat16");
    assert_st(f.sz() == 4 && in.sz() == 4 && bi.sz() == 1);
    if (!(f.names(0) == "out_chan" && f.names(1) == "y" && f.names(2) == "x" && f.names(3) == "in_chan")) rt_err("hip_conv_nhwc_grp: filts must be out_chan:y:x:in_chan, got " + f.pretty_str());
    auto si = am.find("stride"), pi = am.find("in_pad"), gi = am.find("grp");
    if (si == am.end() || pi == am.end() || gi == am.end()) rt_err("hip_conv_nhwc_grp: 'stride', 'in_pad' and 'grp' REF args are required");
    dims_t const stride = si->second.get_dims(host->nh_rtc()), in_pad = pi->second.get_dims(host->nh_rtc()), grp = gi->second.get_dims(host->nh_rtc());
    int const n = (int)grp.sz() - 1;
    if (n < 1 || n > 4 || grp.names(n) != "pad") rt_err("hip_conv_nhwc_grp: grp must be (m0=..,..,pad=..) with 1..4 members");
    int noc[4], ctot[4], coff[4]; void *outs[4]; bool out_f32 = false; dims_t out0;
    for (int m = 0; m < n; ++m) {
      string const onm = var_of(am, "out_" + std::to_string(m));
      dims_t const out = host->nh_var_dims(onm);
      if (out.tn != "bfloat16" && out.tn != "float") unsup_err("hip_conv_nhwc_grp: outputs must have type bfloat16 or float");
      if (!(out.sz() == 4 && out.names(0) == "img" && out.names(1) == "y" && out.names(2) == "x" && out.names(3) == "chan")) rt_err("hip_conv_nhwc_grp: outputs must be img:y:x:chan, got " + out.pretty_str());
      if (m == 0) { out0 = out; out_f32 = (out.tn == "float"); }
      else if (out.tn != out0.tn || out.dsz("img") != out0.dsz("img") || out.dsz("y") != out0.dsz("y") || out.dsz("x") != out0.dsz("x")) rt_err("hip_conv_nhwc_grp: the members' outputs must agree in type and map size");
      noc[m] = (int)grp.dims(m); ctot[m] = (int)out.dsz("chan"); coff[m] = 0; outs[m] = host->nh_var_ptr(onm);
      auto oi = am.find("out_chan_off_" + std::to_string(m));
      if (oi != am.end()) {
        if (oi->second.is_var() || !oi->second.v || !oi->second.v->rp_elems()) rt_err("hip_conv_nhwc_grp: out_chan_off_<m> must be a by-value uint32");
        coff[m] = (int)*(uint32_t const *)oi->second.v->rp_elems();
      } else if (ctot[m] != noc[m]) rt_err("hip_conv_nhwc_grp: member " + std::to_string(m) + " writes a wider tensor: out_chan_off_" + std::to_string(m) + " is required");
      if (coff[m] < 0 || coff[m] + noc[m] > ctot[m]) rt_err("hip_conv_nhwc_grp: out_chan_off + out_chans exceeds the channels of out_" + std::to_string(m));
    }
    conv_geom_t g = geom_from_dims(f, in, out0, stride, in_pad, fi.op.get_u32("conv_has_relu") != 0);
    if (f.dsz("in_chan") != (uint32_t)g.C || bi.dsz("out_chan") != (uint32_t)g.OC) rt_err("hip_conv_nhwc_grp: inconsistent filts / biases / in dims");
    if (!g.SY || !g.SX) rt_err("hip_conv_nhwc_grp: zero stride");
    if ((g.H + 2 * g.PY - g.KH) / g.SY + 1 != g.OH || (g.W + 2 * g.PX - g.KW) / g.SX + 1 != g.OW || out0.dsz("img") != (uint32_t)g.B) rt_err("hip_conv_nhwc_grp: out dims do not match in/filts/stride/in_pad");
    tile_override_t const tov(impl, "conv_tile", fi.op);
    conv_nhwc_grp(host->nh_var_ptr(fnm), (float const *)host->nh_var_ptr(bnm), host->nh_var_ptr(inm), g, out_f32, n, noc, outs, ctot, coff, (int)grp.dims(n));
    return;
  }
  if (fn == "hip_conv_nhwc_multi" || fn == "hip_conv_nhwc_set") {
    bool const is_set = (fn == "hip_conv_nhwc_set");   // (a set's members keep their own specialised kernels -- implicit-GEMM or input-patch form, by the dims of their filts)
    // n independent channels-last convolutions (REF `multi`: dims n = the member count), member m: vars filts_<m> (out_chan:y:x:in_chan) biases_<m> in_<m> out_<m>,
    // REFs stride_<m> in_pad_<m>, optional by-value out_chan_off_<m>; ReLU: conv_has_relu for all, or bit m of the optional uint32 relu_mask
    auto mi = am.find("multi");
    if (mi == am.end()) rt_err("hip_conv_nhwc_multi: the REF arg 'multi' (dims n=<members>) is required");
    int const n = (int)mi->second.get_dims(host->nh_rtc()).dsz("n");
    if (n < 1 || n > 256) unsup_err("hip_conv_nhwc_multi: 1..256 members");
    bool const relu_all = fi.op.get_u32("conv_has_relu") != 0; bool const has_mask = fi.op.has("relu_mask"); uint32_t const mask = has_mask ? fi.op.get_u32("relu_mask") : 0u;
    if (has_mask && n > 32) unsup_err("hip_conv_nhwc_multi: relu_mask covers 32 members");
    std::vector<native_kernels_t::multi_member_t> ms((size_t)n); string out_tn; std::vector<char> patch_f((size_t)n, 0);
    for (int m = 0; m < n; ++m) {
      string const sfx = "_" + std::to_string(m);
      if (is_set && am.find("grp" + sfx) != am.end()) {   // a horizontally fused member (the args of hip_conv_nhwc_grp, every name with the member's suffix)
        string const fnm = var_of(am, "filts" + sfx), bnm = var_of(am, "biases" + sfx), inm = var_of(am, "in" + sfx);
        dims_t const f = host->nh_var_dims(fnm), bi = host->nh_var_dims(bnm), in = host->nh_var_dims(inm);
        need_float(bi, "biases");
        if (f.tn != "bfloat16" || in.tn != "bfloat16") unsup_err("hip_conv_nhwc_set: filts / in must have type bfloat16");
        if (!(f.sz() == 4 && f.names(0) == "out_chan" && f.names(3) == "in_chan" && in.sz() == 4)) rt_err("hip_conv_nhwc_set: a fused member's filts must be out_chan:y:x:in_chan");
        auto si = am.find("stride" + sfx), pi = am.find("in_pad" + sfx);
        if (si == am.end() || pi == am.end()) rt_err("hip_conv_nhwc_set: 'stride_<m>' and 'in_pad_<m>' REF args are required");
        dims_t const stride = si->second.get_dims(host->nh_rtc()), in_pad = pi->second.get_dims(host->nh_rtc()), grp = am.find("grp" + sfx)->second.get_dims(host->nh_rtc());
        int const gn = (int)grp.sz() - 1;
        if (gn < 1 || gn > 4 || grp.names(gn) != "pad") rt_err("hip_conv_nhwc_set: grp_<m> must be (m0=..,..,pad=..) with 1..4 members");
        native_kernels_t::multi_member_t &mm = ms[(size_t)m];
        mm.grp_n = gn; mm.grp_pad = (int)grp.dims(gn); dims_t out0;
        for (int j = 0; j < gn; ++j) {
          string const onm = var_of(am, "out_" + std::to_string(j) + sfx);
          dims_t const out = host->nh_var_dims(onm);
          if (out.tn != "bfloat16" && out.tn != "float") unsup_err("hip_conv_nhwc_set: outputs must have type bfloat16 or float");
          if (j == 0) out0 = out;
          if (m == 0 && j == 0) out_tn = out.tn; else if (out.tn != out_tn) rt_err("hip_conv_nhwc_set: the members' outputs must have one type");
          mm.grp_noc[j] = (int)grp.dims(j); mm.grp_ctot[j] = (int)out.dsz("chan"); mm.grp_coff[j] = 0; mm.grp_out[j] = host->nh_var_ptr(onm);
          auto oi = am.find("out_chan_off_" + std::to_string(j) + sfx);
          if (oi != am.end()) {
            if (oi->second.is_var() || !oi->second.v || !oi->second.v->rp_elems()) rt_err("hip_conv_nhwc_set: out_chan_off must be a by-value uint32");
            mm.grp_coff[j] = (int)*(uint32_t const *)oi->second.v->rp_elems();
          } else if (mm.grp_ctot[j] != mm.grp_noc[j]) rt_err("hip_conv_nhwc_set: a fused member writes a wider tensor: out_chan_off is required");
          if (mm.grp_coff[j] < 0 || mm.grp_coff[j] + mm.grp_noc[j] > mm.grp_ctot[j]) rt_err("hip_conv_nhwc_set: out_chan_off + out_chans exceeds the channels of the output");
        }
        mm.g = geom_from_dims(f, in, out0, stride, in_pad, has_mask ? ((mask >> m) & 1u) != 0 : relu_all);
        if (f.dsz("in_chan") != (uint32_t)mm.g.C || bi.dsz("out_chan") != (uint32_t)mm.g.OC) rt_err("hip_conv_nhwc_set: inconsistent filts / biases / in dims of a fused member");
        if (!mm.g.SY || !mm.g.SX || (mm.g.H + 2 * mm.g.PY - mm.g.KH) / mm.g.SY + 1 != mm.g.OH || (mm.g.W + 2 * mm.g.PX - mm.g.KW) / mm.g.SX + 1 != mm.g.OW) rt_err("hip_conv_nhwc_set: out dims of a fused member do not match");
        mm.filts = host->nh_var_ptr(fnm); mm.biases = (float const *)host->nh_var_ptr(bnm); mm.in = host->nh_var_ptr(inm); mm.out = nullptr; mm.out_ctot = 0; mm.out_coff = 0;
        continue;
      }
      string const fnm = var_of(am, "filts" + sfx), bnm = var_of(am, "biases" + sfx), inm = var_of(am, "in" + sfx), onm = var_of(am, "out" + sfx);
      dims_t f = host->nh_var_dims(fnm); dims_t const bi = host->nh_var_dims(bnm), in = host->nh_var_dims(inm), out = host->nh_var_dims(onm);
      if (is_set && f.sz() == 5) {   // the input-patch form F'[in_grp][ky][kx][out_chan][8]
        if (!(f.names(0) == "in_grp" && f.names(1) == "y" && f.names(2) == "x" && f.names(3) == "out_chan" && f.names(4) == "in_chan8" && f.dims(4) == 8)) rt_err("hip_conv_nhwc_set: 5-d filts must be in_grp:y:x:out_chan:in_chan8(=8), got " + f.pretty_str());
        f = dims_t({f.dims(3), f.dims(1), f.dims(2), f.dims(0) * 8}, {"out_chan", "y", "x", "in_chan"}, f.tn); patch_f[(size_t)m] = 1;
      }
      need_float(bi, "biases");
      if (f.tn != "bfloat16" || in.tn != "bfloat16") unsup_err("hip_conv_nhwc_multi: filts / in must have type bfloat16");
      if (out.tn != "bfloat16" && out.tn != "float") unsup_err("hip_conv_nhwc_multi: out must have type bfloat16 or float");
      if (m == 0) out_tn = out.tn; else if (out.tn != out_tn) rt_err("hip_conv_nhwc_multi: the members' outputs must have one type");
      assert_st(f.sz() == 4 && in.sz() == 4 && out.sz() == 4 && bi.sz() == 1);
      if (!(f.names(0) == "out_chan" && f.names(1) == "y" && f.names(2) == "x" && f.names(3) == "in_chan")) rt_err("hip_conv_nhwc_multi: filts must be out_chan:y:x:in_chan, got " + f.pretty_str());
      for (dims_t const *d : {&in, &out}) if (!(d->names(0) == "img" && d->names(1) == "y" && d->names(2) == "x" && d->names(3) == "chan")) rt_err("hip_conv_nhwc_multi: in / out must be img:y:x:chan, got " + d->pretty_str());
      auto si = am.find("stride" + sfx), pi = am.find("in_pad" + sfx);
      if (si == am.end() || pi == am.end()) rt_err("hip_conv_nhwc_multi: 'stride_<m>' and 'in_pad_<m>' REF args are required");
      dims_t const stride = si->second.get_dims(host->nh_rtc()), in_pad = pi->second.get_dims(host->nh_rtc());
      assert_st(stride.sz() == 2); assert_st(in_pad.sz() == 2);
      native_kernels_t::multi_member_t &mm = ms[(size_t)m];
      mm.g = geom_from_dims(f, in, out, stride, in_pad, has_mask ? ((mask >> m) & 1u) != 0 : relu_all);
      if (is_set) { mm.pool = apply_pool_window(fi.op, sfx, mm.g, "hip_conv_nhwc_set"); if (mm.pool && !patch_f[(size_t)m]) rt_err("hip_conv_nhwc_set: fused pooling needs the patch form of filts"); }
      conv_geom_t const &g = mm.g;
      if (f.dsz("in_chan") != (uint32_t)g.C) rt_err("hip_conv_nhwc_multi: filts.in_chan != in.chan (member " + std::to_string(m) + ")");
      mm.out_ctot = 0; mm.out_coff = 0;
      auto oi = am.find("out_chan_off" + sfx);
      if (oi != am.end()) {
        if (oi->second.is_var() || !oi->second.v || !oi->second.v->rp_elems()) rt_err("hip_conv_nhwc_multi: out_chan_off_<m> must be a by-value uint32");
        mm.out_coff = (int)*(uint32_t const *)oi->second.v->rp_elems(); mm.out_ctot = (int)out.dsz("chan");
        if (mm.out_coff < 0 || mm.out_coff + g.OC > mm.out_ctot) rt_err("hip_conv_nhwc_multi: out_chan_off + out_chan exceeds the channels of out");
      }
      if (bi.dsz("out_chan") != (uint32_t)g.OC || (!mm.out_ctot && out.dsz("chan") != (uint32_t)g.OC) || out.dsz("img") != (uint32_t)g.B) rt_err("hip_conv_nhwc_multi: inconsistent biases / out dims (member " + std::to_string(m) + ")");
      if (!g.SY || !g.SX) rt_err("hip_conv_nhwc_multi: zero stride");
      if ((g.H + 2 * g.PY - g.KH) / g.SY + 1 != g.OH || (g.W + 2 * g.PX - g.KW) / g.SX + 1 != g.OW) rt_err("hip_conv_nhwc_multi: out dims do not match in / filts / stride / in_pad (member " + std::to_string(m) + ")");
      mm.filts = host->nh_var_ptr(fnm); mm.biases = (float const *)host->nh_var_ptr(bnm); mm.in = host->nh_var_ptr(inm); mm.out = host->nh_var_ptr(onm);
    }
    if (is_set) {
      std::vector<char> pf(patch_f); bool pfb[16]; if (n > 16) unsup_err("hip_conv_nhwc_set: 1..16 members");
      for (int m = 0; m < n; ++m) pfb[m] = pf[(size_t)m] != 0;
      conv_nhwc_set(n, ms.data(), pfb, out_tn == "float");
      return;
    }
    tile_override_t const tov(impl, "conv_tile", fi.op);
    conv_nhwc_multi(n, ms.data(), out_tn == "float");
    return;
  }
  if (fn == "hip_conv_nhwc") {
    // channels-last bf16 tensors: filts out_chan:y:x:in_chan, in / out img:y:x:chan (out bf16 or float), biases float
    string const fnm = var_of(am, "filts"), bnm = var_of(am, "biases"), inm = var_of(am, "in"), onm = var_of(am, "out");
    dims_t f = host->nh_var_dims(fnm); dims_t const bi = host->nh_var_dims(bnm), in = host->nh_var_dims(inm), out = host->nh_var_dims(onm);
    need_float(bi, "biases");
    if (f.tn != "bfloat16" || in.tn != "bfloat16") unsup_err("hip_conv_nhwc: filts / in must have type bfloat16 (got " + f.tn + " / " + in.tn + ")");
    if (out.tn != "bfloat16" && out.tn != "float") unsup_err("hip_conv_nhwc: out must have type bfloat16 or float (got " + out.tn + ")");
    assert_st((f.sz() == 4 || f.sz() == 5) && in.sz() == 4 && out.sz() == 4 && bi.sz() == 1);
    // filts: out_chan:y:x:in_chan (implicit-GEMM kernel) or in_grp:y:x:out_chan:in_chan8 (LDS input-patch kernel; in_chan8 = 8): the layout the function was
    // annotated with decides the kernel
    bool const patch_filts = (f.sz() == 5);
    if (patch_filts) {
      if (!(f.names(0) == "in_grp" && f.names(1) == "y" && f.names(2) == "x" && f.names(3) == "out_chan" && f.names(4) == "in_chan8" && f.dims(4) == 8)) rt_err("hip_conv_nhwc: 5-d filts must be in_grp:y:x:out_chan:in_chan8(=8), got " + f.pretty_str());
      f = dims_t({f.dims(3), f.dims(1), f.dims(2), f.dims(0) * 8}, {"out_chan", "y", "x", "in_chan"}, f.tn);   // (the logical filter dims)
    }
    if (!(f.names(0) == "out_chan" && f.names(1) == "y" && f.names(2) == "x" && f.names(3) == "in_chan")) rt_err("hip_conv_nhwc: filts must be out_chan:y:x:in_chan, got " + f.pretty_str());
    for (dims_t const *d : {&in, &out}) if (!(d->names(0) == "img" && d->names(1) == "y" && d->names(2) == "x" && d->names(3) == "chan")) rt_err("hip_conv_nhwc: in / out must be img:y:x:chan, got " + d->pretty_str());
    auto si = am.find("stride"), pi = am.find("in_pad");
    if (si == am.end() || pi == am.end()) rt_err("hip_conv_nhwc: 'stride' and 'in_pad' REF args are required");
    dims_t const stride = si->second.get_dims(host->nh_rtc()), in_pad = pi->second.get_dims(host->nh_rtc());
    assert_st(stride.sz() == 2); assert_st(in_pad.sz() == 2);
    conv_geom_t g = geom_from_dims(f, in, out, stride, in_pad, fi.op.get_u32("conv_has_relu") != 0);
    if (f.dsz("in_chan") != (uint32_t)g.C) rt_err("hip_conv_nhwc: filts.in_chan != in.chan");
    bool const pool = apply_pool_window(fi.op, string(), g, "hip_conv_nhwc");
    if (pool && !patch_filts) rt_err("hip_conv_nhwc: fused pooling needs the in_grp:y:x:out_chan:in_chan8 form of filts");
    post_ops_t post; bool const has_post = apply_post_ops(fi.op, g, post, "hip_conv_nhwc");   // (out = the pooled tensor; g.OH x g.OW are the convolution's own planes from here on)
    if (has_post && (!patch_filts || pool || out.tn != "bfloat16")) rt_err("hip_conv_nhwc: a pooling fused behind the convolution needs the in_grp:y:x:out_chan:in_chan8 form of filts, a bfloat16 out and no pooling in front");
    int out_ctot = 0, out_coff = 0;
    auto oi = am.find("out_chan_off");
    if (oi != am.end()) {
      if (oi->second.is_var() || !oi->second.v || !oi->second.v->rp_elems()) rt_err("hip_conv_nhwc: out_chan_off must be a by-value uint32");
      out_coff = (int)*(uint32_t const *)oi->second.v->rp_elems(); out_ctot = (int)out.dsz("chan");
      if (out_coff < 0 || out_coff + g.OC > out_ctot) rt_err("hip_conv_nhwc: out_chan_off + out_chan exceeds the channels of out");
    }
    if (bi.dsz("out_chan") != (uint32_t)g.OC || (!out_ctot && out.dsz("chan") != (uint32_t)g.OC) || out.dsz("img") != (uint32_t)g.B) rt_err("hip_conv_nhwc: inconsistent biases/out dims");
    if (!g.SY || !g.SX) rt_err("hip_conv_nhwc: zero stride");
    if ((g.H + 2 * g.PY - g.KH) / g.SY + 1 != g.OH || (g.W + 2 * g.PX - g.KW) / g.SX + 1 != g.OW) rt_err("hip_conv_nhwc: out dims do not match in/filts/stride/in_pad");
    if (has_post) {
      if ((int)out.dsz("y") != post.POH || (int)out.dsz("x") != post.POW) rt_err("hip_conv_nhwc: out dims do not match the fused pooling");
      conv_nhwc_rows(host->nh_var_ptr(fnm), (float const *)host->nh_var_ptr(bnm), host->nh_var_ptr(inm), host->nh_var_ptr(onm), g, post, out_ctot, out_coff);
      return;
    }
    tile_override_t const tov(impl, "conv_tile", fi.op);
    conv_nhwc(host->nh_var_ptr(fnm), (float const *)host->nh_var_ptr(bnm), host->nh_var_ptr(inm), host->nh_var_ptr(onm), g, out.tn == "float", out_ctot, out_coff, patch_filts, pool);
    return;
  }
  if (fn == "hip_conv_k1_chain") {
    // two chained 1x1 convolutions (see conv_k1_chain): vars filts / biases (first conv), filts2 / biases2 (second), in, out, optionally mid (the first conv's output,
    // written as well); REFs stride / in_pad (of both: 1x1 / stride 1 / no padding); uint32 conv_has_relu / conv_has_relu2; optional by-value out_chan_off
    string const fnm = var_of(am, "filts"), bnm = var_of(am, "biases"), f2nm = var_of(am, "filts2"), b2nm = var_of(am, "biases2"), inm = var_of(am, "in"), onm = var_of(am, "out");
    dims_t const f = host->nh_var_dims(fnm), bi = host->nh_var_dims(bnm), f2 = host->nh_var_dims(f2nm), b2 = host->nh_var_dims(b2nm), in = host->nh_var_dims(inm), out = host->nh_var_dims(onm);
    need_float(f, "filts"); need_float(bi, "biases"); need_float(f2, "filts2"); need_float(b2, "biases2"); need_float(in, "in"); need_float(out, "out");
    auto si = am.find("stride"), pi = am.find("in_pad");
    if (si == am.end() || pi == am.end()) rt_err("hip_conv_k1_chain: 'stride' and 'in_pad' REF args are required");
    dims_t const stride = si->second.get_dims(host->nh_rtc()), in_pad = pi->second.get_dims(host->nh_rtc());
    assert_st(stride.sz() == 2); assert_st(in_pad.sz() == 2);
    assert_st(f.sz() == 4 && f2.sz() == 4 && in.sz() == 4 && out.sz() == 4 && bi.sz() == 1 && b2.sz() == 1);
    conv_geom_t g = geom_from_dims(f, in, out, stride, in_pad, fi.op.get_u32("conv_has_relu") != 0);
    bool const relu2 = fi.op.get_u32("conv_has_relu2") != 0;
    int const oc2 = (int)f2.dsz("out_chan");
    if (f.dsz("in_chan") != (uint32_t)g.C || f2.dsz("in_chan") != (uint32_t)g.OC) rt_err("hip_conv_k1_chain: filts.in_chan != in.chan or filts2.in_chan != filts.out_chan");
    if (f2.dsz("y") != 1 || f2.dsz("x") != 1 || g.KH != 1 || g.KW != 1 || g.SY != 1 || g.SX != 1 || g.PY || g.PX) unsup_err("hip_conv_k1_chain: both convolutions must be 1x1 / stride 1 / unpadded");
    int out_ctot = 0, out_coff = 0;
    auto oi = am.find("out_chan_off");
    if (oi != am.end()) {
      if (oi->second.is_var() || !oi->second.v || !oi->second.v->rp_elems()) rt_err("hip_conv_k1_chain: out_chan_off must be a by-value uint32");
      out_coff = (int)*(uint32_t const *)oi->second.v->rp_elems(); out_ctot = (int)out.dsz("chan");
      if (out_coff < 0 || out_coff + oc2 > out_ctot) rt_err("hip_conv_k1_chain: out_chan_off + out_chan exceeds the channels of out");
    }
    if (bi.dsz("out_chan") != (uint32_t)g.OC || b2.dsz("out_chan") != (uint32_t)oc2 || (!out_ctot && out.dsz("chan") != (uint32_t)oc2) || out.dsz("img") != (uint32_t)g.B) rt_err("hip_conv_k1_chain: inconsistent biases / out dims");
    if (g.OH != g.H || g.OW != g.W) rt_err("hip_conv_k1_chain: out dims do not match in");
    float *mid = nullptr;
    auto mi = am.find("mid");
    if (mi != am.end()) {
      string const mnm = var_of(am, "mid"); dims_t const md = host->nh_var_dims(mnm); need_float(md, "mid");
      if (md.sz() != 4 || md.dsz("img") != (uint32_t)g.B || md.dsz("chan") != (uint32_t)g.OC || md.dsz("y") != (uint32_t)g.OH || md.dsz("x") != (uint32_t)g.OW) rt_err("hip_conv_k1_chain: mid must be img:chan:y:x of the first convolution's output");
      mid = (float *)host->nh_var_ptr(mnm);
    }
    conv_k1_chain((float const *)host->nh_var_ptr(fnm), (float const *)host->nh_var_ptr(bnm), (float const *)host->nh_var_ptr(f2nm), (float const *)host->nh_var_ptr(b2nm),
                  (float const *)host->nh_var_ptr(inm), (float *)host->nh_var_ptr(onm), mid, g, oc2, relu2, out_ctot, out_coff);
    return;
  }
  if (fn == "hip_conv" || fn == "cudnn_conv" || fn == "hip_conv_bf16" || fn == "hip_conv_winograd") {
    string const fnm = var_of(am, "filts"), bnm = var_of(am, "biases"), inm = var_of(am, "in"), onm = var_of(am, "out");
    dims_t const f = host->nh_var_dims(fnm), bi = host->nh_var_dims(bnm), in = host->nh_var_dims(inm), out = host->nh_var_dims(onm);
    need_float(f, "filts"); need_float(bi, "biases"); need_float(in, "in"); need_float(out, "out");
    auto si = am.find("stride"), pi = am.find("in_pad");
    if (si == am.end() || pi == am.end()) rt_err("hip_conv: 'stride' and 'in_pad' REF args are required");
    dims_t const stride = si->second.get_dims(host->nh_rtc()), in_pad = pi->second.get_dims(host->nh_rtc());
    assert_st(stride.sz() == 2); assert_st(in_pad.sz() == 2);
    assert_st(f.sz() == 4 && in.sz() == 4 && out.sz() == 4 && bi.sz() == 1);
    conv_geom_t g = geom_from_dims(f, in, out, stride, in_pad, fi.op.get_u32("conv_has_relu") != 0);
    if (f.dsz("in_chan") != (uint32_t)g.C) rt_err("hip_conv: filts.in_chan != in.chan");
    if (apply_f32_pool(fi.op, g, "hip_conv") && (fn != "hip_conv" || bf16)) unsup_err("fused pooling (hip_pool): the fp32 hip_conv function only");
    // optional by-value arg out_chan_off: `out` is then a wider tensor (an inception module's Concat output) and this conv writes
    // channels [out_chan_off, out_chan_off + out_chan) of it -- the channel-offset copy of src/rtc_fwd.cc:267-280 folded into the store
    int out_ctot = 0, out_coff = 0;
    auto oi = am.find("out_chan_off");
    if (oi != am.end()) {
      if (oi->second.is_var() || !oi->second.v || !oi->second.v->rp_elems()) rt_err("hip_conv: out_chan_off must be a by-value uint32");
      out_coff = (int)*(uint32_t const *)oi->second.v->rp_elems(); out_ctot = (int)out.dsz("chan");
      if (out_coff < 0 || out_coff + g.OC > out_ctot) rt_err("hip_conv: out_chan_off + out_chan exceeds the channels of out");
    }
    if (bi.dsz("out_chan") != (uint32_t)g.OC || (!out_ctot && out.dsz("chan") != (uint32_t)g.OC) || out.dsz("img") != (uint32_t)g.B) rt_err("hip_conv: inconsistent biases/out dims");
    if (!g.SY || !g.SX) rt_err("hip_conv: zero stride");
    // out = (in + 2*pad - k)/stride + 1, floor (src/conv_util.cc:167-173)
    if ((g.H + 2 * g.PY - g.KH) / g.SY + 1 != g.OH || (g.W + 2 * g.PX - g.KW) / g.SX + 1 != g.OW) rt_err("hip_conv: out dims do not match in/filts/stride/in_pad");
    // optional var arg filts_km (round 6): the k-major copy of filts that hip_conv_filts_kmajor wrote -- [K + 128][out_chan padded to 4], zero rows behind K.  A plan that
    // reads its filters k-major (the staging-wave kernel) then skips the transposition it would run in front of every call; every other plan ignores it
    float const *km = nullptr;
    auto ki = am.find("filts_km");
    if (ki != am.end() && fn == "hip_conv") {
      if (!ki->second.is_var()) rt_err("hip_conv: filts_km must be a var");
      dims_t const kd = host->nh_var_dims(ki->second.n); need_float(kd, "filts_km");
      long const Ktot = (long)g.C * g.KH * g.KW, mi4 = ((long)g.OC + 3) / 4 * 4;
      if (kd.sz() != 2 || (long)kd.dims(0) != Ktot + 128 || (long)kd.dims(1) != mi4) rt_err("hip_conv: filts_km must be [K + 128][out_chan padded to a multiple of 4] floats");
      km = (float const *)host->nh_var_ptr(ki->second.n);
    }
    tile_override_t const tov(impl, "conv_tile", fi.op);
    conv((float const *)host->nh_var_ptr(fnm), (float const *)host->nh_var_ptr(bnm), (float const *)host->nh_var_ptr(inm), (float *)host->nh_var_ptr(onm), g, bf16, out_ctot, out_coff,
         (fn == "hip_conv_winograd") ? "winograd_all" : nullptr, km); // hip_conv_winograd: the F(2x2,3x3) path for this function (3x3 / stride 1; others: direct)
    return;
  }
  if (fn == "hip_conv_filts_kmajor") {   // filts (out_chan:in_chan:y:x) -> filts_km ([K + 128][out_chan padded to 4], zeros in the padding): see filts_km above
    string const fnm = var_of(am, "filts"), knm = var_of(am, "filts_km");
    dims_t const f = host->nh_var_dims(fnm), kd = host->nh_var_dims(knm);
    need_float(f, "filts"); need_float(kd, "filts_km"); assert_st(f.sz() == 4);
    long const OC = f.dims(0), Ktot = (long)f.dims(1) * f.dims(2) * f.dims(3), mi4 = (OC + 3) / 4 * 4, kp = Ktot + 128;
    if (kd.sz() != 2 || (long)kd.dims(0) != kp || (long)kd.dims(1) != mi4) rt_err("hip_conv_filts_kmajor: filts_km must be [K + 128][out_chan padded to a multiple of 4] floats");
    if ((uint64_t)kp * mi4 * 4 >= 0x7ffffff0ull) unsup_err("hip_conv_filts_kmajor: filts of 2 GiB or more");
    plan_t xp; xp.cbig = true; xp.kname = "bodahip_conv_big_xpose"; xp.defs = {"-DXPOSE_ONLY=1"};
    kernel_t &xk = get_kernel(impl, host, xp);
    float const *src = (float const *)host->nh_var_ptr(fnm); float *dst = (float *)host->nh_var_ptr(knm); int Mi = (int)OC, Mi4 = (int)mi4, Kk = (int)Ktot, Kp = (int)kp;
    void *xparams[] = {&src, &dst, &Mi, &Mi4, &Kk, &Kp};
    hip_err_chk(host->nh_launch(xk.func, (uint32_t)((kp + 31) / 32), (uint32_t)((mi4 + 31) / 32), 256, xparams), "hipModuleLaunchKernel(conv_big_xpose)");
    last_launch.kernel = "bodahip_conv_big_xpose"; last_launch.grid = (uint32_t)(((kp + 31) / 32) * ((mi4 + 31) / 32)); last_launch.block = 256; last_launch.flops = 0; last_launch.algo_bytes = 8.0 * OC * Ktot;
    return;
  }
  rt_err("unknown/unhandled native hip function: " + fn);
}

} // namespace bodahip

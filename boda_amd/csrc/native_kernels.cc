// native_kernels.cc -- host side of the hand-written kernels: hiprtc specialisation and the code-object table, launch helpers and workspaces, and the fp32 entry points
// (sgemm, conv, Winograd, the 1x1 chain).  Tile / variant selection: native_plan.cc; channels-last bf16 entry points: native_nhwc.cc; run() / prebuild(): native_run.cc.
// See native_kernels.h for the contract and the reference precedent.
#include "native_internal.h"

namespace bodahip {

// kernel template sources, embedded at build time from kernels/*.hip (see build.py: kernels_embed.inc)
#include "kernels_embed.inc"

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


void launch(native_host_t *host, kernel_t &k, gemm_args_t &a, tile_cfg_t const &c) {
  void *params[] = {&a};
  uint32_t const grid = (uint32_t)a.tiles_i * (uint32_t)a.tiles_j * (uint32_t)std::max(1, a.splitk);
  hip_err_chk(host->nh_launch(k.func, grid, 1, (uint32_t)c.threads(), params), "hipModuleLaunchKernel(native)");
}


std::vector<char> compile_plan(plan_t const &p, string const &arch, string *log) {
  vect_string opts = p.defs; opts.push_back("-DKNAME=" + p.kname);
  return hiprtc_compile(p.nhwc_rows ? k_src_conv_nhwc_rows_bf16 : p.nhwc_multi ? k_src_conv_nhwc_multi_bf16 : p.nhwc_patch ? k_src_conv_nhwc_patch_bf16 : p.nhwc ? k_src_conv_nhwc_bf16 : p.patch16 ? k_src_conv_patch_bf16 : (p.cbig ? k_src_conv_big_f32 : p.big ? k_src_sgemm_big_f32 : p.fc ? k_src_fc_f32 : p.stream ? (p.quad ? k_src_k1_quad_f32 : k_src_k1_stream_f32) : (p.bf16 ? k_src_gemm_conv_bf16 : k_src_gemm_conv_f32)), p.kname, arch, opts, log, true);
}

// grow-only scratch shared by the split-K slabs and the Winograd-domain tensors (like the reference's cudnn scratch var)
void ensure_ws(native_kernels_t::impl_t *impl, native_host_t *host, size_t need) {
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
void call_ws_make_room(native_kernels_t::impl_t *impl, native_host_t *host, size_t need) {
  static size_t const cap = (getenv("BODAHIP_CALL_WS_MB") ? (size_t)atol(getenv("BODAHIP_CALL_WS_MB")) : 2048) << 20;
  if (impl->call_ws_bytes + need <= cap || impl->call_ws_bytes == 0 || impl->call_ws_hold > 0 || host->nh_capturing() || host->nh_live_graphs() > 0) return;
  hip_err_chk(hipStreamSynchronize(host->nh_stream()), "hipStreamSynchronize");
  for (auto it = impl->ktabs.begin(); it != impl->ktabs.end();) {
    if (it->first.compare(0, 4, "ksl:") == 0 || it->first.compare(0, 4, "kho:") == 0) { (void)hipFree(it->second); it = impl->ktabs.erase(it); } else ++it;
  }
  impl->call_ws_bytes = 0;
}
void setup_ksl(native_kernels_t::impl_t *impl, native_host_t *host, gemm_args_t &ga, tile_cfg_t const &cfg, long nk, void const *key_ptr, char const *what) {
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

string tune_of(native_kernels_t::impl_t *impl, char const *key) { auto t = impl->tune.find(key); return (t == impl->tune.end()) ? string() : t->second; }

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
bool s2d_geom(conv_geom_t const &g, conv_geom_t &g2, int &pry, int &prx) {
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
bool winograd_applies(conv_geom_t const &g, string const &algo) {
  if (!(g.KH == 3 && g.KW == 3 && g.SY == 1 && g.SX == 1)) return false;
  return algo == "winograd_all" || (algo == "winograd" && g.C >= 96 && g.OC >= 64);
}
// images per chunk of the three-kernel Winograd pipeline (see conv_winograd)
long wino_chunk_imgs(conv_geom_t const &g) {
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
    size_t const ts_main = (size_t)ga.tiles_i * ga.tiles_j * 128;   // (with a tail launch: room for its workgroups' stamps behind the main launch's -- its kernel is built with -DTSTAMP=1, too)
    size_t const ts_bytes = ts_main + ((p.cbig && p.split_pels > 0) ? (size_t)((g.OC + p.tail_cfg.BI - 1) / p.tail_cfg.BI) * (size_t)((Nj - p.split_pels + p.tail_cfg.BJ - 1) / p.tail_cfg.BJ) * 128 : 0);
    bool const ts_late = tstamp && strlen(tstamp) > 5 && !strcmp(tstamp + strlen(tstamp) - 5, ":late");   // no synchronisation, no copy per launch: the sequence runs undisturbed
    if (tstamp) { ensure_ws(impl, host, ts_off + ts_bytes); ga.ws = (float *)((char *)impl->ws + ts_off); if (!ts_late) hip_err_chk(hipMemsetAsync(ga.ws, 0, ts_bytes, host->nh_stream()), "hipMemsetAsync(tstamp)"); }
    if (p.cbig && p.split_pels > 0) {   // two-level tiling along the pels: this plan's tiles over the first split_pels pels, the tail plan's over the rest
      ga.tiles_j = (int)(p.split_pels / cfg.BJ);
      launch(host, k, ga, cfg);
      plan_t tp; tp.cbig = true; tp.kname = p.kname; tp.cfg = p.tail_cfg; tp.defs = p.tail_defs; tp.patch = p.patch; tp.k1 = p.k1; tp.rdec = p.rdec;
      kernel_t &k2 = get_kernel(impl, host, tp);
      gemm_args_t g2 = ga; g2.tiles_i = (g.OC + tp.cfg.BI - 1) / tp.cfg.BI; g2.tiles_j = (int)((Nj - p.split_pels + tp.cfg.BJ - 1) / tp.cfg.BJ); g2.bsJ = p.split_pels; g2.ws = tstamp ? (float *)((char *)ga.ws + ts_main) : nullptr;
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
bool plan_k1_chain(conv_geom_t const &g, int oc2, bool relu2, plan_t &p) {
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


kernel_t &get_kernel(native_kernels_t::impl_t *impl, native_host_t *host, plan_t const &p) {
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


} // namespace bodahip

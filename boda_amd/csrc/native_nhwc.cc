// native_nhwc.cc -- the channels-last bf16 entry points (config 5): rolling-rows stems, sibling groups, multi-problem launches, level sets, hip_conv_nhwc.
#include "native_internal.h"

namespace bodahip {

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
plan_t plan_conv_nhwc_multi(std::vector<conv_geom_t> const &gs, string const &tile, bool out_f32) {
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
string set_kernel_source(std::vector<plan_t const *> const &variants, int threads, int minw) {
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
    o << "#define KNAME run\nnamespace member_v" << v << " {\n" << (p.nhwc_patch ? k_src_conv_nhwc_patch_bf16_ptr : k_src_conv_nhwc_bf16_ptr) << "\n}\n";
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
plan_t plan_set_member(set_member_in_t const &mi, int num_cus, bool out_f32) {
  if (mi.grp_pad > 0) return plan_conv_nhwc(mi.g, num_cus, string(), out_f32, mi.grp_pad);
  return mi.patch_filts ? plan_conv_nhwc_patch(mi.g, num_cus, string(), out_f32, mi.pool) : plan_conv_nhwc(mi.g, num_cus, string(), out_f32, 0, /*allow_split=*/getenv("BODAHIP_NHWC_SPLITK2") == nullptr);
}
double set_tile_cost(set_member_in_t const &mi, plan_t const &p) {
  return (double)p.cfg.BI * p.cfg.BJ * (double)mi.g.C * (mi.pool ? 2 : mi.g.KH * mi.g.KW) / std::max(1, p.ksl ? p.cfg.SPLITK : 1);
}
set_layout_t layout_set(std::vector<plan_t> const &plans, std::vector<double> const &tile_cost) {
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


} // namespace bodahip

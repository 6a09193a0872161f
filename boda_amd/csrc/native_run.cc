// native_run.cc -- the rtc function interface of the native kernels: run() (argument checks, dispatch) and prebuild() / explain_plan (the same plans, ahead of time).
#include "native_internal.h"

namespace bodahip {

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
        if (!arch.empty()) hiprtc_compile(k_src_winograd_f32_ptr, "bodahip_winograd", arch, vect_string(), &log, true);
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


// hip_multi.cc -- hip_multi_compute_t: N GPUs behind ONE rtc_compute_t ("multi-device must stay invisible behind the
// interface: one logical var <-> N shards", one host thread, one in-order stream per device, no collective on the data path).
//
// The reference's backends drive a single device (src/nvrtc_util.cc device 0); its callers (conv_pipe_fwd_t, ops-prof, rtc_test) know vars by
// name only (src/rtc_compute.H:48-80) -- which is exactly the surface a multi-device backend can hide behind:
//   * a var with a dim named `img` (its leading dim) or `M` is SHARDED along it into N contiguous chunks (floor splits; a chunk may be empty):
//     every op of the hot path is independent per image (conv, pool, ReLU, across-channel LRN), sgemm c = a^T b per row of c / column of a;
//     every other var (filts, biases, sgemm b) is REPLICATED;
//   * copy_nda_to_var scatters (sharded: each device receives its slice straight from the host buffer; sgemm `a`, K:M, is split along its
//     second dim: packed per shard) or broadcasts (replicated: one H2D to device 0, then a peer-to-peer fan-out over xGMI,
//     hipMemcpyPeerAsync, one copy per link; devices without peer access get their own H2D); copy_var_to_nda gathers shards into the slices
//     of the caller's buffer (replicated vars are read from device 0);
//   * run(): native functions (hip_sgemm / hip_conv / hip_conv_nhwc and aliases: shapes come from the vars at run time) are enqueued on every
//     device's stream in turn from the one host thread -- the devices then work concurrently -- and get ONE call id.  Generated CUCL functions
//     have their sizes baked in or passed by value for the WHOLE tensor.  On replicated vars they run as they are, on every device (replicas
//     stay equal).  On vars sharded along their leading dim they run too, when they are per-element functions: the function's own index
//     declaration -- the `// CUCL IX GLOB_ID_1D <arg> [use_dims=...]` line every reference template carries into its generated source
//     (src/rtc_func_gen.cc:227-246; test/rtc/{pool,lrn,relu,copy,gen_data_*}.cucl) -- says that its 1-D ids enumerate the elements of <arg>,
//     leading dim first.  Device i then launches only the ids of its own images, [b_i, e_i) x ids-per-image: GLOB_ID_1D starts at b_i x
//     ids-per-image (threads past the shard's last id see U32_MAX and leave through the function's own range test), every sharded var is passed as (shard pointer - b_i x bytes-per-image), and all
//     sizes stay those of the whole tensor -- each thread computes exactly what it computes on one device (incl. the flat-index hash of
//     gen_data), on memory its device owns.  Functions that use the workgroup (LOC_ID_1D, GRP_ID_1D, LOCSHAR_MEM, BARRIER_SYNC), that declare
//     no GLOB_ID_1D index, or that take a var sharded along its second dim (sgemm `a`) are refused with unsup_err on sharded vars;
//   * get_dur(b, e) = the longest of the devices' durations; finish_and_sync() waits for all.
// The per-GPU process model of bench.py / boda_amd/shard.py (torch.distributed, RCCL weight broadcast) stays: this class is for callers that
// want one process -- an unmodified Boda with --rtc='(be=hip,devices=...)'.  Devices may repeat (e.g. {0, 0}): the same GPU then holds several
// shards, which is how the sharding logic is tested with the HIP kernels on a one-GPU box (tests/test_gpu_multi.py).
#include "rtc_types.h"
#include "native_kernels.h"

#include <hip/hip_runtime.h>
#include <sstream>

namespace bodahip {

void *hip_compute_stream(rtc_compute_t *rtc);
void hip_compute_graph_begin(rtc_compute_t *rtc);
void hip_compute_graph_begin_drained(rtc_compute_t *rtc);
void hip_compute_graph_abort(rtc_compute_t *rtc);
uint32_t hip_compute_graph_end(rtc_compute_t *rtc);
uint32_t hip_compute_graph_launch(rtc_compute_t *rtc, uint32_t id);
uint32_t hip_compute_graph_num_calls(rtc_compute_t *rtc, uint32_t id);
void hip_compute_graph_destroy(rtc_compute_t *rtc, uint32_t id);
uint32_t hip_compute_graph_end_deps(rtc_compute_t *rtc, uint32_t n, uint32_t const *ptr, uint32_t const *idx);
void hip_compute_set_shard_aware(rtc_compute_t *rtc);
uint32_t hip_compute_run_shard(rtc_compute_t *rtc, rtc_func_call_t const &rfc, uint32_t blks, uint32_t gid_off, uint32_t gid_last, std::map<string, int64_t> const &var_bias);

struct multi_var_t { dims_t dims; int shard_dim = -1; };   // logical dims; index of the sharded dim (-1: replicated)

// what a generated function's source says about its index space
struct gen_func_t {
  bool has_ix = false, uses_group = false;
  string ix_arg; vect_string use_dims;
  string n_arg;            // `n=<arg>`: the ids enumerate <by-value uint32 arg> items in all, batch-major over <ix_arg>'s leading `img` (kernels that walk 16-byte chunks or
                           // several outputs per thread: ids per image = that count / images -- the dims of <ix_arg> alone cannot say it)
  bool wave_local = false; // `wave_local`: LOC_ID_1D is used only as the lane number, for shuffles between consecutive ids (a shard's first id need not start a wave)
  bool grp_ix = false;     // `// CUCL IX GRP_ID_1D <arg> n=<threads>`: a WORKGROUP function (LDS, barriers) whose groups enumerate (img, ...) batch-major -- n / tpb groups in
                           // all, the same number per image: a shard is a contiguous run of whole workgroups (the LRN -> Pooling pass through LDS, boda_amd/nhwc.py)
  // `// CUCL SHARD2 <var arg> size=<by-value arg> off=<by-value arg>`: the function walks a tensor sharded along its SECOND dim (sgemm a, K:M -- every device holds
  // K x M_i packed); per device the two by-value uint32 arguments become the shard's size and first index (gen_data_sgemm_a: M, m_off; its M_glob stays)
  string s2_arg, s2_size, s2_off;
};
static bool ident_char(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }
static bool has_token(string const &s, char const *tok) {
  size_t const n = strlen(tok);
  for (size_t p = s.find(tok); p != string::npos; p = s.find(tok, p + 1))
    if ((p == 0 || !ident_char(s[p - 1])) && (p + n >= s.size() || !ident_char(s[p + n]))) return true;
  return false;
}
static gen_func_t scan_gen_func(string const &all_src, string const &func_name) {
  gen_func_t g;
  // the text of this kernel: from its name (followed by an argument list) to the next kernel of the batch
  size_t at = string::npos;
  for (size_t p = all_src.find(func_name); p != string::npos; p = all_src.find(func_name, p + 1)) {
    size_t q = p + func_name.size();
    if (p && ident_char(all_src[p - 1])) continue;
    while (q < all_src.size() && (all_src[q] == ' ' || all_src[q] == '\t' || all_src[q] == '\n')) ++q;
    if (q < all_src.size() && all_src[q] == '(') { at = p; break; }
  }
  if (at == string::npos) return g;
  size_t const end = all_src.find("CUCL_GLOBAL_KERNEL", at);
  string const body = all_src.substr(at, (end == string::npos) ? string::npos : end - at);
  for (char const *t : {"LOC_ID_1D", "GRP_ID_1D", "LOC_SZ_1D", "LOCSHAR_MEM", "BARRIER_SYNC"}) if (has_token(body, t)) g.uses_group = true;
  char const *const key = "// CUCL IX ";
  for (size_t p = body.find(key); p != string::npos; p = body.find(key, p + 1)) {
    size_t const eol = body.find('\n', p);
    std::istringstream is(body.substr(p + strlen(key), (eol == string::npos) ? string::npos : eol - p - strlen(key)));
    string ix, arg, opt; is >> ix >> arg;
    if (ix != "GLOB_ID_1D" && ix != "GRP_ID_1D") { g.uses_group = true; continue; }
    if (g.has_ix) { g.has_ix = false; g.uses_group = true; g.grp_ix = false; return g; }   // (two declarations of the same index: not a form we know)
    g.has_ix = !arg.empty(); g.ix_arg = arg; g.grp_ix = (ix == "GRP_ID_1D");
    while (is >> opt) {
      if (startswith(opt, "use_dims=")) {
        string cur; for (char c : opt.substr(9)) { if (c == ':') { if (!cur.empty()) g.use_dims.push_back(cur); cur.clear(); } else cur.push_back(c); }
        if (!cur.empty()) g.use_dims.push_back(cur);
      } else if (startswith(opt, "n=")) g.n_arg = opt.substr(2);
      else if (opt == "wave_local") g.wave_local = true;
    }
  }
  if (g.grp_ix && g.n_arg.empty()) { g.has_ix = false; g.grp_ix = false; }   // (a group index needs its count)
  {
    char const *const k2 = "// CUCL SHARD2 ";
    size_t const p2 = body.find(k2);
    if (p2 != string::npos) {
      size_t const eol = body.find('\n', p2);
      std::istringstream is(body.substr(p2 + strlen(k2), (eol == string::npos) ? string::npos : eol - p2 - strlen(k2)));
      string opt; is >> g.s2_arg;
      while (is >> opt) { if (startswith(opt, "size=")) g.s2_size = opt.substr(5); else if (startswith(opt, "off=")) g.s2_off = opt.substr(4); }
      if (g.s2_size.empty() || g.s2_off.empty()) g.s2_arg.clear();
    }
  }
  if (g.wave_local && g.has_ix) {   // only LOC_ID_1D (the lane number) is excused; real workgroup cooperation is not
    g.uses_group = false;
    for (char const *t : {"GRP_ID_1D", "LOC_SZ_1D", "LOCSHAR_MEM", "BARRIER_SYNC"}) if (has_token(body, t)) g.uses_group = true;
  }
  return g;
}

struct hip_multi_compute_t : public rtc_compute_t {
  std::vector<int> devs;
  std::vector<p_rtc_compute_t> subs;
  std::map<string, multi_var_t> vis;
  std::map<string, bool> func_native;
  std::map<string, gen_func_t> func_gen;       // generated functions: their index declaration
  std::vector<hipEvent_t> peer_evs;            // per device: marks the end of its last peer copy out of device 0
  static constexpr uint32_t kNoCall = 0xffffffffu;   // per-device call id of a call that launched nothing there (an empty shard)
  std::vector<std::vector<uint32_t>> calls;   // multi call id -> per-device call ids
  std::vector<char> peer_ok;                   // device i reachable from device 0 by hipMemcpyPeerAsync
  bool init_done = false;
  // hipGraph capture over all devices: every device's stream captures its share of the calls (one graph per device), a replay launches them all and gets ONE call id
  static constexpr uint32_t kCapturedCallId = 0xfffffffeu;   // (what run() returns while capturing, as the single-device backend)
  bool capturing = false, cap_skipped = false;               // cap_skipped: some captured call launched nothing on some device (an empty shard)
  struct mgraph_t { std::vector<uint32_t> ids; uint32_t n_calls = 0; bool live = false; };
  std::vector<mgraph_t> mgraphs;
  void graph_begin() {
    assert_st(init_done);
    if (capturing) rt_err("graph_begin: a capture is already in progress");
    for (auto &s : subs) s->finish_and_sync();      // every stream drained BEFORE the first capture opens: a synchronising call is not allowed from this thread afterwards
    size_t opened = 0;
    try { for (; opened < n(); ++opened) hip_compute_graph_begin_drained(subs[opened].get()); }
    catch (...) { for (size_t i = 0; i < opened; ++i) hip_compute_graph_abort(subs[i].get()); throw; }
    capturing = true; cap_skipped = false;
  }
  uint32_t graph_end_common(uint32_t n_calls, uint32_t const *dep_ptr, uint32_t const *dep_idx) {
    if (!capturing) rt_err("graph_end: no capture in progress");
    capturing = false;
    mgraph_t g; g.live = true;
    bool const with_deps = dep_ptr != nullptr;
    string err;
    std::vector<std::pair<size_t, uint32_t>> made;   // (device, graph id) of every graph instantiated so far: destroyed again if any device fails
    for (size_t d = 0; d < n(); ++d) {   // every device's capture is closed whatever happens on another
      rtc_compute_t *const s = subs[d].get();
      try {
        uint32_t id;
        if (with_deps && cap_skipped) { id = hip_compute_graph_end(s); made.emplace_back(d, id); err = "graph_end_deps: a captured call launched nothing on some device (a batch smaller than the device count): dependencies cannot be attributed"; continue; }
        id = with_deps ? hip_compute_graph_end_deps(s, n_calls, dep_ptr, dep_idx) : hip_compute_graph_end(s);
        made.emplace_back(d, id); g.ids.push_back(id);
      } catch (std::exception const &e) { err = e.what(); }
    }
    if (!err.empty()) {
      for (auto const &m : made) { try { hip_compute_graph_destroy(subs[m.first].get(), m.second); } catch (...) {} }
      rt_err(err);
    }
    g.n_calls = 0;   // devices with an empty shard capture fewer calls: report the largest count
    for (size_t d = 0; d < n(); ++d) g.n_calls = std::max(g.n_calls, hip_compute_graph_num_calls(subs[d].get(), g.ids[d]));
    mgraphs.push_back(g);
    return (uint32_t)mgraphs.size() - 1;
  }
  mgraph_t &get_graph(uint32_t id) { if (id >= mgraphs.size() || !mgraphs[id].live) rt_err("invalid graph id " + std::to_string(id)); return mgraphs[id]; }
  uint32_t graph_launch(uint32_t id) {
    if (capturing) rt_err("graph_launch during capture");
    mgraph_t &g = get_graph(id);
    std::vector<uint32_t> ids;
    for (size_t i = 0; i < n(); ++i) ids.push_back(hip_compute_graph_launch(subs[i].get(), g.ids[i]));   // enqueued on every device in turn; the devices replay concurrently
    calls.push_back(ids);
    return (uint32_t)calls.size() - 1;
  }
  void graph_destroy(uint32_t id) { mgraph_t &g = get_graph(id); for (size_t i = 0; i < n(); ++i) hip_compute_graph_destroy(subs[i].get(), g.ids[i]); g.live = false; }

  explicit hip_multi_compute_t(std::vector<int> const &devs_) : devs(devs_) {
    be = "hip";
    if (devs.empty()) rt_err("hip multi-device backend: empty device list");
    for (int d : devs) subs.push_back(make_hip_compute(d));
  }
  ~hip_multi_compute_t() override { for (hipEvent_t e : peer_evs) if (e) (void)hipEventDestroy(e); }
  size_t n() const { return subs.size(); }

  void init() override {
    assert_st(!init_done);
    for (auto &s : subs) { s->gen_src = gen_src; s->gen_src_output_dir = gen_src_output_dir; s->init(); hip_compute_set_shard_aware(s.get()); }
    peer_ok.assign(n(), 0); peer_evs.assign(n(), nullptr);
    bool const force_peer = getenv("BODAHIP_FORCE_PEER") != nullptr;   // (tests: take the device-to-device fan-out between shards of ONE GPU too)
    for (size_t i = 1; i < n(); ++i) {
      if (devs[i] == devs[0]) { peer_ok[i] = force_peer ? 1 : 0; continue; }
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devs[i], devs[0]) == hipSuccess && can) {
        (void)hipSetDevice(devs[i]);
        hipError_t const e = hipDeviceEnablePeerAccess(devs[0], 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) peer_ok[i] = 1;
        (void)hipGetLastError();
      }
    }
    init_done = true;
  }
  string get_plat_tag() override { assert_st(init_done); return subs[0]->get_plat_tag() + "*" + std::to_string(n()); }

  // ---- sharding rule
  static int shard_dim_of(dims_t const &d) {
    for (uint32_t i = 0; i < d.sz(); ++i) if (d.names(i) == "img") return (i == 0) ? 0 : -1;   // (an `img` dim that is not leading: not a batch of this path)
    for (uint32_t i = 0; i < d.sz(); ++i) if (d.names(i) == "M" && i <= 1) return (int)i;
    return -1;
  }
  uint32_t chunk_begin(uint32_t total, size_t i) const { return (uint32_t)((uint64_t)total * i / n()); }
  dims_t shard_dims(multi_var_t const &v, size_t i) const {
    if (v.shard_dim < 0) return v.dims;
    std::vector<uint32_t> sz; vect_string nm;
    for (uint32_t k = 0; k < v.dims.sz(); ++k) { sz.push_back(v.dims.dims(k)); nm.push_back(v.dims.names(k)); }
    uint32_t const tot = sz[v.shard_dim];
    sz[v.shard_dim] = chunk_begin(tot, i + 1) - chunk_begin(tot, i);
    return dims_t(sz, nm, v.dims.tn);
  }

  // ---- vars
  void create_var_with_dims(string const &vn, dims_t const &dims) override {
    assert_st(init_done);
    if (vis.count(vn)) rt_err("create_var_with_dims: var '" + vn + "' already exists");
    multi_var_t v; v.dims = dims; v.shard_dim = shard_dim_of(dims);
    for (size_t i = 0; i < n(); ++i) subs[i]->create_var_with_dims(vn, shard_dims(v, i));
    vis.emplace(vn, v);
  }
  void create_var_with_dims_as_reshaped_view_of_var(string const &vn, dims_t const &dims, string const &src_vn) override {
    multi_var_t const &src = must_find(vis, src_vn);
    rtc_reshape_check(dims, src.dims);
    if (vis.count(vn)) rt_err("create_var_with_dims_as_reshaped_view_of_var: var '" + vn + "' already exists");
    multi_var_t v; v.dims = dims; v.shard_dim = shard_dim_of(dims);
    // a view must cut the same bytes per device as its source: both replicated, or both sharded on a leading dim of equal size
    bool const ok = (v.shard_dim < 0 && src.shard_dim < 0) || (v.shard_dim == 0 && src.shard_dim == 0 && dims.dims(0) == src.dims.dims(0));
    if (!ok) unsup_err("multi-device backend: view '" + vn + "' of '" + src_vn + "' would not shard like its source");
    for (size_t i = 0; i < n(); ++i) subs[i]->create_var_with_dims_as_reshaped_view_of_var(vn, shard_dims(v, i), src_vn);
    vis.emplace(vn, v);
  }
  void release_var(string const &vn) override { must_find(vis, vn); for (auto &s : subs) s->release_var(vn); vis.erase(vn); }
  dims_t get_var_dims(string const &vn) override { return must_find(vis, vn).dims; }
  void set_var_to_zero(string const &vn) override { must_find(vis, vn); for (auto &s : subs) s->set_var_to_zero(vn); }

  void copy_nda_to_var(string const &vn, p_nda_t const &nda) override {
    multi_var_t const &v = must_find(vis, vn);
    if (!(v.dims == nda->dims)) rt_err("copy_nda_to_var: dims mismatch for var '" + vn + "': var " + v.dims.pretty_str() + " nda " + nda->dims.pretty_str());
    char *const host = (char *)nda->rp_elems();
    uint64_t const tsz = v.dims.tsz();
    if (v.shard_dim < 0) {   // replicated: H2D once, then peer fan-out from device 0 (one copy per xGMI link), H2D where there is no peer path
      subs[0]->copy_nda_to_var(vn, nda);
      bool any_peer = false; for (size_t i = 1; i < n(); ++i) any_peer = any_peer || peer_ok[i];
      if (any_peer) subs[0]->finish_and_sync();
      void *const src = subs[0]->get_var_raw_native_pointer(vn)->rp_elems();
      bool copied = false;
      for (size_t i = 1; i < n(); ++i) {
        if (peer_ok[i] && v.dims.bytes_sz()) {
          void *const dst = subs[i]->get_var_raw_native_pointer(vn)->rp_elems();
          hipStream_t const si = (hipStream_t)hip_compute_stream(subs[i].get());
          hip_err_chk(hipSetDevice(devs[i]), "hipSetDevice");
          hip_err_chk(hipMemcpyPeerAsync(dst, devs[i], src, devs[0], v.dims.bytes_sz(), si), "hipMemcpyPeerAsync");
          // the copy READS device 0's buffer on device i's stream: device 0's stream must not overwrite that buffer (an in-place function,
          // set_var_to_zero, the next upload) before the copy has read it
          if (!peer_evs[i]) hip_err_chk(hipEventCreateWithFlags(&peer_evs[i], hipEventDisableTiming), "hipEventCreateWithFlags");
          hip_err_chk(hipEventRecord(peer_evs[i], si), "hipEventRecord(peer copy)");
          copied = true;
        } else subs[i]->copy_nda_to_var(vn, nda);
      }
      if (copied) {
        hip_err_chk(hipSetDevice(devs[0]), "hipSetDevice");
        for (size_t i = 1; i < n(); ++i) if (peer_ok[i] && v.dims.bytes_sz())
          hip_err_chk(hipStreamWaitEvent((hipStream_t)hip_compute_stream(subs[0].get()), peer_evs[i], 0), "hipStreamWaitEvent(peer copy)");
      }
      return;
    }
    if (v.shard_dim == 0) {  // leading dim: each shard is a contiguous slice of the host buffer
      uint64_t const row = v.dims.dims(0) ? v.dims.dims_prod() / v.dims.dims(0) * tsz : 0;
      for (size_t i = 0; i < n(); ++i) {
        dims_t const sd = shard_dims(v, i);
        if (sd.dims_prod()) subs[i]->copy_nda_to_var(vn, std::make_shared<nda_t>(sd, host + (uint64_t)chunk_begin(v.dims.dims(0), i) * row));
      }
      return;
    }
    // second dim (sgemm a, K:M): columns [m0, m1) of every row -> packed per shard
    uint32_t const R = v.dims.dims(0), Ctot = v.dims.dims(1);
    uint64_t const inner = (v.dims.dims_prod() / std::max<uint64_t>(1, (uint64_t)R * Ctot)) * tsz;
    for (size_t i = 0; i < n(); ++i) {
      dims_t const sd = shard_dims(v, i);
      uint32_t const c0 = chunk_begin(Ctot, i), cn = sd.dims(1);
      if (!sd.dims_prod()) continue;
      std::vector<char> pack((size_t)R * cn * inner);
      for (uint32_t r = 0; r < R; ++r) memcpy(pack.data() + (size_t)r * cn * inner, host + ((size_t)r * Ctot + c0) * inner, (size_t)cn * inner);
      subs[i]->copy_nda_to_var(vn, std::make_shared<nda_t>(sd, pack.data()));
      subs[i]->finish_and_sync();   // (the packed buffer dies here)
    }
  }
  void copy_var_to_nda(p_nda_t const &nda, string const &vn) override {
    multi_var_t const &v = must_find(vis, vn);
    if (!(v.dims == nda->dims)) rt_err("copy_var_to_nda: dims mismatch for var '" + vn + "': var " + v.dims.pretty_str() + " nda " + nda->dims.pretty_str());
    char *const host = (char *)nda->rp_elems();
    uint64_t const tsz = v.dims.tsz();
    if (v.shard_dim < 0) { subs[0]->copy_var_to_nda(nda, vn); return; }
    if (v.shard_dim == 0) {
      uint64_t const row = v.dims.dims(0) ? v.dims.dims_prod() / v.dims.dims(0) * tsz : 0;
      for (size_t i = 0; i < n(); ++i) {
        dims_t const sd = shard_dims(v, i);
        if (sd.dims_prod()) subs[i]->copy_var_to_nda(std::make_shared<nda_t>(sd, host + (uint64_t)chunk_begin(v.dims.dims(0), i) * row), vn);
      }
      return;
    }
    uint32_t const R = v.dims.dims(0), Ctot = v.dims.dims(1);
    uint64_t const inner = (v.dims.dims_prod() / std::max<uint64_t>(1, (uint64_t)R * Ctot)) * tsz;
    for (size_t i = 0; i < n(); ++i) {
      dims_t const sd = shard_dims(v, i);
      uint32_t const c0 = chunk_begin(Ctot, i), cn = sd.dims(1);
      if (!sd.dims_prod()) continue;
      std::vector<char> pack((size_t)R * cn * inner);
      subs[i]->copy_var_to_nda(std::make_shared<nda_t>(sd, pack.data()), vn);
      for (uint32_t r = 0; r < R; ++r) memcpy(host + ((size_t)r * Ctot + c0) * inner, pack.data() + (size_t)r * cn * inner, (size_t)cn * inner);
    }
  }
  p_nda_t get_var_raw_native_pointer(string const &vn) override {
    multi_var_t const &v = must_find(vis, vn);
    if (v.shard_dim >= 0) rt_err("multi-device backend: var '" + vn + "' is sharded over " + std::to_string(n()) + " devices and has no single device pointer");
    return subs[0]->get_var_raw_native_pointer(vn);
  }

  // ---- functions
  void compile(vect_rtc_func_info_t const &func_infos, rtc_compile_opts_t const &opts) override {
    assert_st(init_done);
    for (auto const &fi : func_infos) if (func_native.count(fi.func_name)) rt_err("compile: function '" + fi.func_name + "' already exists");
    for (auto &s : subs) s->compile(func_infos, opts);
    string all_src; for (auto const &fi : func_infos) all_src += fi.func_src;
    for (auto const &fi : func_infos) {
      bool const nat = native_kernels_t::is_native_func_name(fi.op.has_func_name() ? fi.op.get_func_name() : string());
      func_native[fi.func_name] = nat;
      if (!nat) func_gen[fi.func_name] = scan_gen_func(all_src, fi.func_name);
    }
  }
  void release_func(string const &fn) override { must_find(func_native, fn); for (auto &s : subs) s->release_func(fn); func_native.erase(fn); func_gen.erase(fn); }
  void release_all_funcs() override { for (auto &s : subs) s->release_all_funcs(); func_native.clear(); func_gen.clear(); }

  uint32_t run(rtc_func_call_t const &rfc) override {
    assert_st(init_done);
    auto fit = func_native.find(rfc.rtc_func_name);
    if (fit == func_native.end()) rt_err("run: unknown function '" + rfc.rtc_func_name + "' (not compiled, or released)");
    if (!fit->second) {
      bool sharded = false;
      for (auto const &kv : rfc.arg_map) if (kv.second.is_valid() && kv.second.is_var() && must_find(vis, kv.second.n).shard_dim >= 0) sharded = true;
      if (sharded) return run_generated_on_shards(rfc);
    }
    std::vector<uint32_t> ids;
    for (auto &s : subs) ids.push_back(s->run(rfc));   // enqueue on every device's stream in turn; the devices then run concurrently
    if (capturing) return kCapturedCallId;             // (recorded into every device's graph: no call id of its own)
    calls.push_back(ids);
    return (uint32_t)calls.size() - 1;
  }
  // a per-element generated function over vars sharded along their leading dim: see the header of this file
  uint32_t run_generated_on_shards(rtc_func_call_t const &rfc) {
    string const &fn = rfc.rtc_func_name;
    gen_func_t const &g = must_find(func_gen, fn);
    string const why = "multi-device backend: generated function '" + fn + "' takes sharded vars but ";
    if (!g.s2_arg.empty()) {   // a tensor sharded along its second dim, walked by a function that takes the shard's size and first index by value
      auto vi = rfc.arg_map.find(g.s2_arg);
      if (vi != rfc.arg_map.end() && vi->second.is_valid() && vi->second.is_var() && must_find(vis, vi->second.n).shard_dim == 1) return run_generated_on_dim2_shards(rfc, g);
    }
    if (g.uses_group && !g.grp_ix) unsup_err(why + "uses the workgroup (LOC_ID_1D / GRP_ID_1D / LOCSHAR_MEM / BARRIER_SYNC) without declaring a group index (`// CUCL IX GRP_ID_1D <arg> n=<threads>`): only per-element functions and such workgroup functions run on shards");
    if (!g.has_ix) unsup_err(why + "its source declares no `// CUCL IX GLOB_ID_1D <arg>` index: the backend cannot tell which ids belong to which image");
    uint32_t T = 0;
    for (auto const &kv : rfc.arg_map) {
      if (!kv.second.is_valid() || !kv.second.is_var()) continue;
      multi_var_t const &v = must_find(vis, kv.second.n);
      if (v.shard_dim < 0) continue;
      if (v.shard_dim != 0) unsup_err(why + "var '" + kv.second.n + "' is sharded along its second dim (sgemm a, K:M)");
      if (T && v.dims.dims(0) != T) unsup_err(why + "with different batch sizes (" + std::to_string(T) + " and " + std::to_string(v.dims.dims(0)) + ")");
      T = v.dims.dims(0);
    }
    if (!T) { if (capturing) { cap_skipped = true; return kCapturedCallId; } calls.push_back(std::vector<uint32_t>(n(), kNoCall)); return (uint32_t)calls.size() - 1; }   // (an empty batch: nothing to run anywhere)
    auto ai = rfc.arg_map.find(g.ix_arg);
    if (ai == rfc.arg_map.end() || !ai->second.is_valid()) rt_err(why + "binds no argument named '" + g.ix_arg + "', the one its index is declared over");
    dims_t const ixd = ai->second.is_var() ? must_find(vis, ai->second.n).dims : ai->second.v->dims;
    uint64_t W = 1; bool lead_ok = false;
    if (!g.n_arg.empty()) {   // ids = <n_arg> items, batch-major
      auto ni = rfc.arg_map.find(g.n_arg);
      if (ni == rfc.arg_map.end() || !ni->second.is_valid() || ni->second.is_var() || !ni->second.v->rp_elems() || ni->second.v->dims.tsz() != 4) rt_err(why + "its index count '" + g.n_arg + "' is not a by-value 32-bit argument of the call");
      uint64_t const tot = *(uint32_t const *)ni->second.v->rp_elems();
      lead_ok = ixd.sz() > 0 && ixd.names(0) == "img" && ixd.dims(0) == T && tot % T == 0;
      W = tot / T;
    } else {
      vect_string names; std::vector<uint32_t> sizes;
      if (g.use_dims.empty()) for (uint32_t k = 0; k < ixd.sz(); ++k) { names.push_back(ixd.names(k)); sizes.push_back(ixd.dims(k)); }
      else for (auto const &u : g.use_dims) { names.push_back(u); sizes.push_back(ixd.dsz(u)); }
      lead_ok = !names.empty() && names[0] == "img" && sizes[0] == T;
      for (size_t k = 1; k < sizes.size(); ++k) W *= sizes[k];
    }
    if (!lead_ok) unsup_err(why + "its index over '" + g.ix_arg + "' " + ixd.pretty_str() + " does not lead with the sharded batch dim img=" + std::to_string(T));
    if (W * T >= 0xffffffffull || !W) unsup_err(why + "its index space does not fit 32 bits");
    if (!rfc.tpb) rt_err("boda/rtc: can't launch kernel; tpb is zero: rtc_func_name=" + fn);
    if (g.grp_ix && (W % rfc.tpb)) unsup_err(why + "its " + std::to_string(W) + " ids per image are not whole workgroups of " + std::to_string(rfc.tpb));
    std::vector<uint32_t> ids;
    for (size_t i = 0; i < n(); ++i) {
      uint32_t const b = chunk_begin(T, i), e = chunk_begin(T, i + 1);
      if (e == b) { ids.push_back(kNoCall); if (capturing) cap_skipped = true; continue; }
      std::map<string, int64_t> bias;
      for (auto const &kv : rfc.arg_map) {
        if (!kv.second.is_valid() || !kv.second.is_var()) continue;
        multi_var_t const &v = must_find(vis, kv.second.n);
        if (v.shard_dim == 0) bias[kv.second.n] = -(int64_t)((uint64_t)b * (v.dims.dims_prod() / T) * v.dims.tsz());
      }
      uint64_t const work = (uint64_t)(e - b) * W;
      // (a workgroup function: whole groups only, so nothing is cut at the shard's end -- and its threads must not see U32_MAX ids before their barriers)
      ids.push_back(hip_compute_run_shard(subs[i].get(), rfc, (uint32_t)((work + rfc.tpb - 1) / rfc.tpb), (uint32_t)(b * W), g.grp_ix ? 0xffffffffu : (uint32_t)(e * W - 1), bias));
    }
    if (capturing) return kCapturedCallId;
    calls.push_back(ids);
    return (uint32_t)calls.size() - 1;
  }
  // sgemm `a` (K:M, split along M: device i holds K x M_i packed) filled by a function that takes the extent and the first index of what it walks by value
  uint32_t run_generated_on_dim2_shards(rtc_func_call_t const &rfc, gen_func_t const &g) {
    string const &fn = rfc.rtc_func_name;
    string const why = "multi-device backend: generated function '" + fn + "' (second-dim shards of '" + g.s2_arg + "') ";
    for (auto const &kv : rfc.arg_map)
      if (kv.first != g.s2_arg && kv.second.is_valid() && kv.second.is_var() && must_find(vis, kv.second.n).shard_dim >= 0) unsup_err(why + "takes another sharded var '" + kv.second.n + "'");
    auto si = rfc.arg_map.find(g.s2_size), oi = rfc.arg_map.find(g.s2_off);
    auto u32_ok = [](std::map<string, rtc_arg_t>::const_iterator it, std::map<string, rtc_arg_t> const &m) { return it != m.end() && it->second.is_valid() && !it->second.is_var() && it->second.v->rp_elems() && it->second.v->dims.tsz() == 4; };
    if (!u32_ok(si, rfc.arg_map) || !u32_ok(oi, rfc.arg_map)) rt_err(why + "binds no by-value 32-bit '" + g.s2_size + "' / '" + g.s2_off + "'");
    multi_var_t const &v = must_find(vis, rfc.arg_map.find(g.s2_arg)->second.n);
    uint32_t const M = v.dims.dims(1), K = v.dims.dims(0);
    if (*(uint32_t const *)si->second.v->rp_elems() != M) rt_err(why + "was called with " + g.s2_size + " = " + std::to_string(*(uint32_t const *)si->second.v->rp_elems()) + " for a var of " + std::to_string(M) + " columns");
    uint32_t const off0 = *(uint32_t const *)oi->second.v->rp_elems();
    if (!rfc.tpb) rt_err("boda/rtc: can't launch kernel; tpb is zero: rtc_func_name=" + fn);
    std::vector<uint32_t> ids;
    for (size_t i = 0; i < n(); ++i) {
      uint32_t const b = chunk_begin(M, i), e = chunk_begin(M, i + 1);
      if (e == b) { ids.push_back(kNoCall); if (capturing) cap_skipped = true; continue; }
      rtc_func_call_t sub = rfc;
      auto put = [&](string const &an, uint32_t val) { sub.arg_map[an] = rtc_arg_t(make_scalar_nda<uint32_t>(val)); };   // (a fresh by-value nda: the caller's is shared)
      put(g.s2_size, e - b); put(g.s2_off, off0 + b);
      uint64_t const work = (uint64_t)K * (e - b);
      sub.blks = (uint32_t)((work + rfc.tpb - 1) / rfc.tpb);
      ids.push_back(subs[i]->run(sub));
    }
    if (capturing) return kCapturedCallId;
    calls.push_back(ids);
    return (uint32_t)calls.size() - 1;
  }
  void finish_and_sync() override {
    if (capturing) { capturing = false; for (auto &s : subs) hip_compute_graph_abort(s.get()); rt_err("finish_and_sync during graph capture"); }
    for (auto &s : subs) s->finish_and_sync();
  }
  void release_per_call_id_data() override { for (auto &s : subs) s->release_per_call_id_data(); calls.clear(); }
  float get_dur(uint32_t const &b, uint32_t const &e) override {
    if (b >= calls.size() || e >= calls.size()) rt_err("invalid call_id");
    float ms = 0.f;
    for (size_t i = 0; i < n(); ++i) if (calls[b][i] != kNoCall && calls[e][i] != kNoCall) ms = std::max(ms, subs[i]->get_dur(calls[b][i], calls[e][i]));
    return ms;
  }
  void profile_start() override { subs[0]->profile_start(); }
  void profile_stop() override { subs[0]->profile_stop(); }
};

p_rtc_compute_t make_hip_multi_compute(std::vector<int> const &device_ordinals) { return std::make_shared<hip_multi_compute_t>(device_ordinals); }
rtc_compute_t *hip_multi_sub(rtc_compute_t *rtc, uint32_t i) {   // device i's own backend (rtc itself for a single-device backend)
  hip_multi_compute_t *m = dynamic_cast<hip_multi_compute_t *>(rtc);
  if (!m) return rtc;
  if (i >= m->n()) rt_err("multi-device backend: device index out of range");
  return m->subs[i].get();
}
// graph entry points of the C ABI: a multi-device backend captures / replays on all its devices, a single-device one as before
void hip_any_graph_begin(rtc_compute_t *rtc) { if (auto *m = dynamic_cast<hip_multi_compute_t *>(rtc)) m->graph_begin(); else hip_compute_graph_begin(rtc); }
uint32_t hip_any_graph_end(rtc_compute_t *rtc) { if (auto *m = dynamic_cast<hip_multi_compute_t *>(rtc)) return m->graph_end_common(0, nullptr, nullptr); return hip_compute_graph_end(rtc); }
uint32_t hip_any_graph_end_deps(rtc_compute_t *rtc, uint32_t n, uint32_t const *ptr, uint32_t const *idx) {
  if (auto *m = dynamic_cast<hip_multi_compute_t *>(rtc)) return m->graph_end_common(n, ptr, idx); return hip_compute_graph_end_deps(rtc, n, ptr, idx); }
uint32_t hip_any_graph_launch(rtc_compute_t *rtc, uint32_t id) { if (auto *m = dynamic_cast<hip_multi_compute_t *>(rtc)) return m->graph_launch(id); return hip_compute_graph_launch(rtc, id); }
uint32_t hip_any_graph_num_calls(rtc_compute_t *rtc, uint32_t id) { if (auto *m = dynamic_cast<hip_multi_compute_t *>(rtc)) return m->get_graph(id).n_calls; return hip_compute_graph_num_calls(rtc, id); }
void hip_any_graph_destroy(rtc_compute_t *rtc, uint32_t id) { if (auto *m = dynamic_cast<hip_multi_compute_t *>(rtc)) m->graph_destroy(id); else hip_compute_graph_destroy(rtc, id); }
uint32_t hip_multi_num_devices(rtc_compute_t *rtc) { hip_multi_compute_t *m = dynamic_cast<hip_multi_compute_t *>(rtc); return m ? (uint32_t)m->n() : 1u; }

} // namespace bodahip

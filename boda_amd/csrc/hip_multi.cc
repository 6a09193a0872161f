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
//     device's stream in turn from the one host thread -- the devices then work concurrently -- and get ONE call id; generated CUCL functions
//     have their sizes baked in or passed by value for the whole tensor, so they run (on every device, keeping replicas equal) only when all
//     their var arguments are replicated, and are refused with unsup_err otherwise;
//   * get_dur(b, e) = the longest of the devices' durations; finish_and_sync() waits for all.
// The per-GPU process model of bench.py / boda_amd/shard.py (torch.distributed, RCCL weight broadcast) stays: this class is for callers that
// want one process -- an unmodified Boda with --rtc='(be=hip,devices=...)'.  Devices may repeat (e.g. {0, 0}): the same GPU then holds several
// shards, which is how the sharding logic is tested with the HIP kernels on a one-GPU box (tests/test_gpu_multi.py).
#include "rtc_types.h"
#include "native_kernels.h"

#include <hip/hip_runtime.h>

namespace bodahip {

void *hip_compute_stream(rtc_compute_t *rtc);

struct multi_var_t { dims_t dims; int shard_dim = -1; };   // logical dims; index of the sharded dim (-1: replicated)

struct hip_multi_compute_t : public rtc_compute_t {
  std::vector<int> devs;
  std::vector<p_rtc_compute_t> subs;
  std::map<string, multi_var_t> vis;
  std::map<string, bool> func_native;
  std::vector<std::vector<uint32_t>> calls;   // multi call id -> per-device call ids
  std::vector<char> peer_ok;                   // device i reachable from device 0 by hipMemcpyPeerAsync
  bool init_done = false;

  explicit hip_multi_compute_t(std::vector<int> const &devs_) : devs(devs_) {
    be = "hip";
    if (devs.empty()) rt_err("hip multi-device backend: empty device list");
    for (int d : devs) subs.push_back(make_hip_compute(d));
  }
  size_t n() const { return subs.size(); }

  void init() override {
    assert_st(!init_done);
    for (auto &s : subs) { s->gen_src = gen_src; s->gen_src_output_dir = gen_src_output_dir; s->init(); }
    peer_ok.assign(n(), 0);
    for (size_t i = 1; i < n(); ++i) {
      if (devs[i] == devs[0]) continue;
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
      for (size_t i = 1; i < n(); ++i) {
        if (peer_ok[i] && v.dims.bytes_sz()) {
          void *const dst = subs[i]->get_var_raw_native_pointer(vn)->rp_elems();
          hip_err_chk(hipSetDevice(devs[i]), "hipSetDevice");
          hip_err_chk(hipMemcpyPeerAsync(dst, devs[i], src, devs[0], v.dims.bytes_sz(), (hipStream_t)hip_compute_stream(subs[i].get())), "hipMemcpyPeerAsync");
        } else subs[i]->copy_nda_to_var(vn, nda);
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
    for (auto const &fi : func_infos) func_native[fi.func_name] = native_kernels_t::is_native_func_name(fi.op.has_func_name() ? fi.op.get_func_name() : string());
  }
  void release_func(string const &fn) override { must_find(func_native, fn); for (auto &s : subs) s->release_func(fn); func_native.erase(fn); }
  void release_all_funcs() override { for (auto &s : subs) s->release_all_funcs(); func_native.clear(); }

  uint32_t run(rtc_func_call_t const &rfc) override {
    assert_st(init_done);
    auto fit = func_native.find(rfc.rtc_func_name);
    if (fit == func_native.end()) rt_err("run: unknown function '" + rfc.rtc_func_name + "' (not compiled, or released)");
    if (!fit->second) {   // generated CUCL source: sizes are baked in / passed for the whole tensor -> only on replicated vars
      for (auto const &kv : rfc.arg_map) if (kv.second.is_valid() && kv.second.is_var() && must_find(vis, kv.second.n).shard_dim >= 0)
        unsup_err("multi-device backend: generated function '" + rfc.rtc_func_name + "' takes the sharded var '" + kv.second.n +
                  "'; only the native functions (hip_sgemm / hip_conv ...) run on sharded vars");
    }
    std::vector<uint32_t> ids;
    for (auto &s : subs) ids.push_back(s->run(rfc));   // enqueue on every device's stream in turn; the devices then run concurrently
    calls.push_back(ids);
    return (uint32_t)calls.size() - 1;
  }
  void finish_and_sync() override { for (auto &s : subs) s->finish_and_sync(); }
  void release_per_call_id_data() override { for (auto &s : subs) s->release_per_call_id_data(); calls.clear(); }
  float get_dur(uint32_t const &b, uint32_t const &e) override {
    if (b >= calls.size() || e >= calls.size()) rt_err("invalid call_id");
    float ms = 0.f;
    for (size_t i = 0; i < n(); ++i) ms = std::max(ms, subs[i]->get_dur(calls[b][i], calls[e][i]));
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
uint32_t hip_multi_num_devices(rtc_compute_t *rtc) { hip_multi_compute_t *m = dynamic_cast<hip_multi_compute_t *>(rtc); return m ? (uint32_t)m->n() : 1u; }

} // namespace bodahip

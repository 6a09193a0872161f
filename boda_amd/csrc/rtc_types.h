// rtc_types.h -- value types and the abstract backend interface of the rtc_compute layer.
//
// Mirrors (same names, same field meaning, same error behaviour) the reference's plugin boundary so that a
// backend written against this header is a drop-in for Boda's own callers:
//   dims_t / dim_t            src/boda_base.H:424-450,498-690      (row-major named dims + type name)
//   nda_t                     src/boda_base.H:751-810              (dims + optional raw element pointer)
//   op_base_t                 src/op_base.H:9-43, src/op_base.cc   (str_vals + nda_vals, ordering)
//   rtc_compile_opts_t, rtc_func_info_t, rtc_arg_t, rtc_func_call_t, rtc_compute_t
//                             src/rtc_compute.H:9-127
//   rt_err / unsup_err        src/boda_base.H:98,105               (fatal vs. "unsupported, caller may record")
// Written from the interface's behaviour; this is not a copy of those headers (no NESI, no boost).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace bodahip {

using std::string;
typedef std::vector<string> vect_string;

// ---- errors -------------------------------------------------------------------------------------------------------
struct rt_exception : public std::runtime_error { explicit rt_exception(string const &m) : std::runtime_error(m) {} };
struct unsup_exception : public std::runtime_error { explicit unsup_exception(string const &m) : std::runtime_error(m) {} };
[[noreturn]] inline void rt_err(string const &m) { throw rt_exception("error: " + m); }
[[noreturn]] inline void unsup_err(string const &m) { throw unsup_exception("error: " + m); }
#define assert_st(x) do { if (!(x)) { ::bodahip::rt_err(string("assertion failed: " #x " at ") + __FILE__ + ":" + std::to_string(__LINE__)); } } while (0)

template <typename M> typename M::mapped_type &must_find(M &m, typename M::key_type const &k) {
  auto i = m.find(k); if (i == m.end()) { rt_err("missing key '" + string(k) + "'"); } return i->second; }
template <typename M> typename M::mapped_type const &must_find(M const &m, typename M::key_type const &k) {
  auto i = m.find(k); if (i == m.end()) { rt_err("missing key '" + string(k) + "'"); } return i->second; }
template <typename M, typename V> void must_insert(M &m, typename M::key_type const &k, V &&v) {
  if (!m.emplace(k, std::forward<V>(v)).second) { rt_err("duplicate key '" + string(k) + "'"); } }
template <typename M> void must_erase(M &m, typename M::key_type const &k) {
  if (!m.erase(k)) { rt_err("tried to erase missing key '" + string(k) + "'"); } }
inline bool startswith(string const &s, string const &p) { return s.size() >= p.size() && !s.compare(0, p.size(), p); }

// ---- dims_t -------------------------------------------------------------------------------------------------------
inline uint64_t tn_size(string const &tn) {
  if (tn == "none") return 0; if (tn == "half") return 2; if (tn == "bfloat16") return 2; if (tn == "float") return 4; if (tn == "double") return 8;
  if (tn == "int32_t") return 4; if (tn == "uint32_t") return 4; if (tn == "uint16_t") return 2; if (tn == "uint8_t") return 1;
  rt_err("unknown type name '" + tn + "'");
}
struct dim_t {
  uint32_t sz = 0, stride = 0; string name;
  bool operator==(dim_t const &o) const { return sz == o.sz && stride == o.stride && name == o.name; }
  bool operator<(dim_t const &o) const {
    if (sz != o.sz) return sz < o.sz; if (stride != o.stride) return stride < o.stride; return name < o.name; }
};
struct dims_t : public std::vector<dim_t> {
  string tn;                 // "float", "none", ...
  uint64_t strides_sz = 0;   // total element count (no padding support on this path)
  dims_t() {}
  dims_t(std::vector<uint32_t> const &szs, vect_string const &names_, string const &tn_) : tn(tn_) {
    assert_st(szs.size() == names_.size());
    for (size_t i = 0; i < szs.size(); ++i) { dim_t d; d.sz = szs[i]; d.name = names_[i]; push_back(d); }
    calc_strides();
  }
  void add_dims(string const &n, uint32_t sz) { dim_t d; d.sz = sz; d.name = n; push_back(d); }
  void calc_strides() { strides_sz = 1; for (size_t d = size(); d-- > 0;) { (*this)[d].stride = (uint32_t)strides_sz; strides_sz *= (*this)[d].sz; } }
  uint32_t sz() const { return (uint32_t)size(); }
  uint32_t dims(uint32_t i) const { return at(i).sz; }
  string const &names(uint32_t i) const { return at(i).name; }
  uint32_t strides(uint32_t i) const { return at(i).stride; }
  dim_t const *get_dim_by_name(string const &n) const { for (auto const &d : *this) { if (d.name == n) return &d; } return nullptr; }
  uint32_t dsz(string const &n) const { dim_t const *d = get_dim_by_name(n); if (!d) rt_err("dim not found:" + n); return d->sz; }
  uint32_t dstride(string const &n) const { dim_t const *d = get_dim_by_name(n); if (!d) rt_err("dim not found:" + n); return d->stride; }
  uint64_t dims_prod() const { uint64_t r = 1; for (auto const &d : *this) r *= d.sz; return r; }
  uint64_t tsz() const { return tn_size(tn); }
  uint64_t bytes_sz() const { return tsz() * strides_sz; }
  bool operator==(dims_t const &o) const { return tn == o.tn && static_cast<std::vector<dim_t> const &>(*this) == static_cast<std::vector<dim_t> const &>(o); }
  bool operator!=(dims_t const &o) const { return !(*this == o); }
  bool operator<(dims_t const &o) const {
    return (tn == o.tn) ? (static_cast<std::vector<dim_t> const &>(*this) < static_cast<std::vector<dim_t> const &>(o)) : (tn < o.tn); }
  string pretty_str() const { string r = "DIMS["; for (size_t i = 0; i < size(); ++i) { if (i) r += ":"; r += at(i).name + "=" + std::to_string(at(i).sz); } return r + "]"; }
};

// ---- nda_t --------------------------------------------------------------------------------------------------------
// dims + raw pointer.  Owning (host buffer, 32-byte aligned as the reference's, src/boda_base.H:784) or non-owning
// (e.g. get_var_raw_native_pointer(): dims + device pointer), or data-less (REF args: the information is the dims).
struct nda_t {
  dims_t dims;
  void *rp = nullptr;
  std::shared_ptr<void> owned;
  nda_t() {}
  explicit nda_t(dims_t const &d) : dims(d) {
    size_t const b = (size_t)d.bytes_sz();
    if (b) { void *p = nullptr; if (posix_memalign(&p, 32, (b + 31) & ~size_t(31))) rt_err("nda_t: host allocation failed"); memset(p, 0, b); owned.reset(p, free); rp = p; }
  }
  nda_t(dims_t const &d, void *rp_) : dims(d), rp(rp_) {}
  void *rp_elems() const { return rp; }
  uint64_t elems_sz() const { return dims.strides_sz; }
};
typedef std::shared_ptr<nda_t> p_nda_t;
typedef std::map<string, p_nda_t> map_str_p_nda_t;
inline p_nda_t make_dims_nda(dims_t const &d) { return std::make_shared<nda_t>(d, nullptr); }
template <typename T> inline char const *tn_of();
template <> inline char const *tn_of<float>() { return "float"; }
template <> inline char const *tn_of<uint32_t>() { return "uint32_t"; }
template <> inline char const *tn_of<int32_t>() { return "int32_t"; }
template <typename T> p_nda_t make_scalar_nda(T const &v) {
  dims_t d; d.tn = tn_of<T>(); d.calc_strides(); p_nda_t r = std::make_shared<nda_t>(d); *static_cast<T *>(r->rp) = v; return r; }

// ---- op_base_t ----------------------------------------------------------------------------------------------------
struct op_base_t {
  std::map<string, string> str_vals;
  map_str_p_nda_t nda_vals;
  bool has(string const &an) const { return nda_vals.count(an) != 0; }
  void set(string const &an, p_nda_t const &n) { must_insert(nda_vals, an, n); }
  void set_dims(string const &an, dims_t const &d) { set(an, make_dims_nda(d)); }
  p_nda_t const &get(string const &an) const { return must_find(nda_vals, an); }
  dims_t const &get_dims(string const &an) const { return get(an)->dims; }
  string const &get_str(string const &an) const { return must_find(str_vals, an); }
  uint32_t get_u32(string const &an) const {
    p_nda_t const &n = get(an); if (n->dims.tn != "uint32_t" || n->dims.sz() != 0 || !n->rp) rt_err("op: '" + an + "' is not a uint32_t scalar");
    return *static_cast<uint32_t const *>(n->rp); }
  void set_u32(string const &an, uint32_t v) { set(an, make_scalar_nda(v)); }
  bool has_type() const { return str_vals.count("type") != 0; }
  string const &get_type() const { return must_find(str_vals, "type"); }
  bool has_func_name() const { return str_vals.count("func_name") != 0; }
  string const &get_func_name() const { return must_find(str_vals, "func_name"); }
  void set_func_name(string const &f) { must_insert(str_vals, "func_name", f); }
};
typedef std::shared_ptr<op_base_t> p_op_base_t;

// ---- rtc layer ----------------------------------------------------------------------------------------------------
struct rtc_compile_opts_t {
  uint32_t show_compile_log = 0, enable_lineinfo = 0, show_func_attrs = 0, show_rtc_calls = 0;
};
struct rtc_func_info_t {
  string func_name;      // name of the extern "C" kernel inside func_src; also the handle used by run()
  string func_src;       // CUCL-dialect source text
  vect_string arg_names; // kernel parameter order
  op_base_t op;          // the (annotated) op this function was generated for; op.func_name selects native kernels
};
typedef std::vector<rtc_func_info_t> vect_rtc_func_info_t;

struct rtc_compute_t;
// an argument is either the name of a var (device pointer is passed) or a value (raw bytes passed by value; a
// value with null data is a REF / optional argument: a null pointer is passed and only its dims carry information).
struct rtc_arg_t {
  string n; p_nda_t v;
  rtc_arg_t() {}
  rtc_arg_t(string const &n_) : n(n_) {}
  rtc_arg_t(char const *n_) : n(n_) {}
  rtc_arg_t(p_nda_t const &v_) : v(v_) {}
  bool is_valid() const { return bool(v) != (!n.empty()); }
  bool is_var() const { assert_st(is_valid()); return !n.empty(); }
  bool is_nda() const { assert_st(is_valid()); return bool(v); }
  string const &get_var() const { assert_st(is_var()); return n; }
  p_nda_t const &get_nda() const { assert_st(is_nda()); return v; }
  inline dims_t get_dims(rtc_compute_t &rtc) const;
};
typedef std::map<string, rtc_arg_t> map_str_rtc_arg_t;
struct rtc_func_call_t {
  string rtc_func_name;
  map_str_rtc_arg_t arg_map;
  uint32_t tpb = 0, blks = 0;
};

struct rtc_compute_t {
  string be;                                   // back-end id ("hip")
  uint32_t gen_src = 0;                        // if 1, dump generated sources / code objects before load
  string gen_src_output_dir = "rtc-gen-src";
  virtual ~rtc_compute_t() {}

  virtual void init() = 0;
  virtual string get_plat_tag() = 0;
  virtual void create_var_with_dims(string const &vn, dims_t const &dims) = 0;
  virtual void create_var_with_dims_as_reshaped_view_of_var(string const &vn, dims_t const &dims, string const &src_vn) = 0;
  virtual void release_var(string const &vn) = 0;
  virtual dims_t get_var_dims(string const &vn) = 0;
  virtual void set_var_to_zero(string const &vn) = 0;
  virtual void compile(vect_rtc_func_info_t const &func_infos, rtc_compile_opts_t const &opts) = 0;
  virtual void release_func(string const &func_name) = 0;
  virtual uint32_t run(rtc_func_call_t const &rfc) = 0;
  virtual void finish_and_sync() = 0;
  virtual void release_per_call_id_data() = 0;
  virtual void release_all_funcs() = 0;
  virtual float get_dur(uint32_t const &b, uint32_t const &e) = 0; // ms, start of call b to end of call e
  virtual void profile_start() = 0;
  virtual void profile_stop() = 0;
  virtual void copy_var_to_nda(p_nda_t const &nda, string const &vn) = 0;
  virtual p_nda_t get_var_raw_native_pointer(string const &vn) = 0;
  virtual void copy_nda_to_var(string const &vn, p_nda_t const &nda) = 0;

  // non-virtual conveniences layered on the above (src/rtc_compute.cc:43-97)
  void create_var_from_nda(p_nda_t const &nda, string const &vn) { create_var_with_dims(vn, nda->dims); copy_nda_to_var(vn, nda); }
  p_nda_t create_nda_from_var(string const &vn) { p_nda_t r = std::make_shared<nda_t>(get_var_dims(vn)); copy_var_to_nda(r, vn); return r; }
  void init_var_from_vect_float(string const &vn, std::vector<float> const &v) {
    p_nda_t nda = std::make_shared<nda_t>(dims_t({uint32_t(v.size())}, {"v"}, "float"), (void *)v.data());
    create_var_with_dims(vn, nda->dims); copy_nda_to_var(vn, nda); }
  void set_vect_float_from_var(std::vector<float> &v, string const &vn) {
    dims_t d = get_var_dims(vn); assert_st(d.sz() == 1); assert_st(v.size() == d.dims(0));
    copy_var_to_nda(std::make_shared<nda_t>(d, (void *)v.data()), vn); }
  void copy_ndas_to_vars(vect_string const &names, map_str_p_nda_t const &ndas) { for (auto const &n : names) copy_nda_to_var(n, must_find(ndas, n)); }
  void copy_vars_to_ndas(vect_string const &names, map_str_p_nda_t &ndas) {
    for (auto const &n : names) { auto i = ndas.find(n); if (i != ndas.end()) copy_var_to_nda(i->second, n); else ndas[n] = create_nda_from_var(n); } }
};
typedef std::shared_ptr<rtc_compute_t> p_rtc_compute_t;
inline dims_t rtc_arg_t::get_dims(rtc_compute_t &rtc) const { return is_nda() ? v->dims : rtc.get_var_dims(n); }

// shared-by-backends checks (src/rtc_compute.cc:21-41)
inline void rtc_launch_check_blks_and_tpb(string const &fn, uint64_t blks, uint64_t tpb) {
  if (!((blks > 0) && (tpb > 0))) {
    rt_err("boda/rtc: can't launch kernel; blks or tpb is zero: rtc_func_name=" + fn + " blks=" + std::to_string(blks) + " tpb=" +
           std::to_string(tpb) + "; perhaps is a culibs stub function that should not have been attempted to be run?"); }
}
inline void rtc_reshape_check(dims_t const &dims, dims_t const &src_dims) {
  if (dims.tn != src_dims.tn) rt_err("invalid reshape; types don't match: dims.tn=" + dims.tn + " src_vi.tn=" + src_dims.tn);
  if (dims.dims_prod() != src_dims.dims_prod())
    rt_err("invalid reshape; types match but sizes don't: dims.dims_prod()=" + std::to_string(dims.dims_prod()) +
           " src_dims.dims_prod()=" + std::to_string(src_dims.dims_prod()));
}

// op line text -> op_base_t (lexp grammar src/lexp.cc; nda text form src/nesi.cc:720-785).  lexp.cc
op_base_t parse_op_lexp(string const &s);
string op_to_str(op_base_t const &op);

// factory for the MI355X backend (hip_compute.cc)
p_rtc_compute_t make_hip_compute(int device_ordinal);
// factory for be=cpu, the host-cores backend behind the same contract (cpu_compute.cc; the CPU baseline of SURVEY.md section 8d)
p_rtc_compute_t make_cpu_compute();

} // namespace bodahip

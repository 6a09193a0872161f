// lexp.cc -- one-line op descriptions -> op_base_t (and back).
// Grammar restated from the reference's behaviour (src/lexp.cc): value = leaf | '(' name '=' value {',' name '=' value} ')';
// '\' escapes the next character.  nda text form (src/nesi.cc:720-785, printer src/boda_base.cc:403-440):
// fields tn / dims=(name=sz,...) / v=a:b:c ; tn defaults to float when dims are present.
// The legacy '(type=T,dims_vals=(...),str_vals=(out_chans=N))' form used by test/sgemm-ops-*.txt is accepted too.
#include "rtc_types.h"
#include <cstdlib>
#include <utility>

namespace bodahip {
namespace {
struct lexp_t { bool leaf = true; string s; std::vector<std::pair<string, lexp_t>> kids; };

struct parser_t {
  string const &s; size_t pos = 0;
  explicit parser_t(string const &s_) : s(s_) {}
  lexp_t value() {
    lexp_t r;
    if (pos < s.size() && s[pos] == '(') {
      r.leaf = false; ++pos;
      if (pos < s.size() && s[pos] == ')') { ++pos; return r; }
      while (true) {
        string name;
        while (pos < s.size() && s[pos] != '=') {
          char c = s[pos];
          if (c == '(' || c == ')' || c == ',') rt_err("lexp: invalid char in name at offset " + std::to_string(pos));
          if (c == '\\') { ++pos; if (pos >= s.size()) rt_err("lexp: dangling escape"); c = s[pos]; }
          name.push_back(c); ++pos;
        }
        if (pos >= s.size()) rt_err("lexp: unexpected end of input in name");
        ++pos; // '='
        lexp_t v = value();
        r.kids.emplace_back(name, std::move(v));
        if (pos >= s.size()) rt_err("lexp: unexpected end of input in list");
        if (s[pos] == ',') { ++pos; continue; }
        if (s[pos] == ')') { ++pos; return r; }
        rt_err("lexp: expected ',' or ')' at offset " + std::to_string(pos));
      }
    }
    while (pos < s.size() && s[pos] != ',' && s[pos] != ')') {
      char c = s[pos];
      if (c == '(') rt_err("lexp: unexpected '(' in leaf at offset " + std::to_string(pos));
      if (c == '\\') { ++pos; if (pos >= s.size()) rt_err("lexp: dangling escape"); c = s[pos]; }
      r.s.push_back(c); ++pos;
    }
    return r;
  }
};

lexp_t const *find_kid(lexp_t const &l, string const &k) { for (auto const &kv : l.kids) { if (kv.first == k) return &kv.second; } return nullptr; }
string leaf_str(lexp_t const &l, char const *what) { if (!l.leaf) rt_err(string("lexp: expected leaf for ") + what); return l.s; }
uint32_t to_u32(string const &s) {
  char *e = nullptr; unsigned long v = strtoul(s.c_str(), &e, 10);
  if (s.empty() || *e) rt_err("lexp: bad unsigned integer '" + s + "'"); return (uint32_t)v; }

dims_t parse_dims(lexp_t const &l, string tn) {
  if (l.leaf && !l.s.empty()) rt_err("nda: dims must be a list");
  dims_t d;
  for (auto const &kv : l.kids) {
    if (kv.first == "__tn__") { tn = leaf_str(kv.second, "__tn__"); continue; }
    d.add_dims(kv.first, to_u32(leaf_str(kv.second, "dim size")));
  }
  d.tn = tn; tn_size(tn); d.calc_strides();
  return d;
}

p_nda_t parse_nda(lexp_t const &l) {
  if (l.leaf) rt_err("nda: expected list");
  for (auto const &kv : l.kids) { if (kv.first != "tn" && kv.first != "dims" && kv.first != "v") rt_err("nda: unknown field '" + kv.first + "'"); }
  lexp_t const *ltn = find_kid(l, "tn"), *ldims = find_kid(l, "dims"), *lv = find_kid(l, "v");
  string tn = ltn ? leaf_str(*ltn, "tn") : (ldims ? "float" : "");
  if (tn.empty()) rt_err("nda: scalar without tn");
  dims_t d; d.tn = tn; tn_size(tn); d.calc_strides();
  if (ldims) d = parse_dims(*ldims, tn);
  if (!lv) return make_dims_nda(d);
  // values: colon or space separated
  string const vs = leaf_str(*lv, "v");
  std::vector<string> toks; string cur;
  for (char c : vs) { if (c == ':' || c == ' ') { if (!cur.empty()) { toks.push_back(cur); cur.clear(); } } else cur.push_back(c); }
  if (!cur.empty()) toks.push_back(cur);
  if (toks.size() != d.dims_prod()) rt_err("nda: expected " + std::to_string(d.dims_prod()) + " values, got " + std::to_string(toks.size()));
  p_nda_t r = std::make_shared<nda_t>(d);
  for (size_t i = 0; i < toks.size(); ++i) {
    if (d.tn == "float") static_cast<float *>(r->rp)[i] = strtof(toks[i].c_str(), nullptr);
    else if (d.tn == "double") static_cast<double *>(r->rp)[i] = strtod(toks[i].c_str(), nullptr);
    else if (d.tn == "uint32_t") static_cast<uint32_t *>(r->rp)[i] = to_u32(toks[i]);
    else if (d.tn == "int32_t") static_cast<int32_t *>(r->rp)[i] = (int32_t)strtol(toks[i].c_str(), nullptr, 10);
    else rt_err("nda: values of type '" + d.tn + "' are not supported in op text");
  }
  return r;
}
} // namespace

op_base_t parse_op_lexp(string const &s_in) {
  string s = s_in;
  while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
  parser_t p(s);
  lexp_t l = p.value();
  if (p.pos != s.size()) rt_err("lexp: trailing characters at offset " + std::to_string(p.pos));
  if (l.leaf) rt_err("op: expected a list");
  op_base_t op;
  bool const legacy = find_kid(l, "dims_vals") || find_kid(l, "type");
  for (auto const &kv : l.kids) {
    if (legacy) {
      if (kv.first == "type") op.str_vals["type"] = leaf_str(kv.second, "type");
      else if (kv.first == "dims_vals") {
        for (auto const &dv : kv.second.kids) {
          bool const none_tn = (dv.first == "kern_sz" || dv.first == "stride" || dv.first == "in_pad");
          op.set_dims(dv.first, parse_dims(dv.second, none_tn ? "none" : "float"));
        }
      } else if (kv.first == "str_vals") {
        for (auto const &sv : kv.second.kids) {
          if (sv.first == "out_chans") op.set_u32("out_chans", to_u32(leaf_str(sv.second, "out_chans")));
          else must_insert(op.str_vals, sv.first, leaf_str(sv.second, "str_val"));
        }
      } else rt_err("op(legacy): unknown field '" + kv.first + "'");
    } else {
      if (kv.first == "str_vals") { for (auto const &sv : kv.second.kids) must_insert(op.str_vals, sv.first, leaf_str(sv.second, "str_val")); }
      else if (kv.first == "nda_vals") { for (auto const &nv : kv.second.kids) op.set(nv.first, parse_nda(nv.second)); }
      else rt_err("op: unknown field '" + kv.first + "'");
    }
  }
  return op;
}

string op_to_str(op_base_t const &op) {
  string r = "(str_vals=(";
  bool first = true;
  for (auto const &kv : op.str_vals) { if (!first) r += ","; first = false; r += kv.first + "=" + kv.second; }
  r += "),nda_vals=(";
  first = true;
  for (auto const &kv : op.nda_vals) {
    if (!first) r += ","; first = false;
    nda_t const &n = *kv.second;
    r += kv.first + "=(";
    bool f2 = true;
    bool const scalar = (n.dims.sz() == 0);
    if (scalar || n.dims.tn != "float") { r += "tn=" + n.dims.tn; f2 = false; }
    if (!scalar) {
      if (!f2) r += ","; f2 = false; r += "dims=(";
      for (uint32_t i = 0; i < n.dims.sz(); ++i) { if (i) r += ","; r += n.dims.names(i) + "=" + std::to_string(n.dims.dims(i)); }
      r += ")";
    }
    if (n.rp && n.dims.dims_prod() <= 16) {
      if (!f2) r += ","; r += "v=";
      for (uint64_t i = 0; i < n.dims.dims_prod(); ++i) {
        if (i) r += ":";
        if (n.dims.tn == "uint32_t") r += std::to_string(static_cast<uint32_t const *>(n.rp)[i]);
        else if (n.dims.tn == "int32_t") r += std::to_string(static_cast<int32_t const *>(n.rp)[i]);
        else if (n.dims.tn == "float") r += std::to_string(static_cast<float const *>(n.rp)[i]);
        else r += "?";
      }
    }
    r += ")";
  }
  return r + "))";
}
} // namespace bodahip

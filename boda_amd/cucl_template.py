"""CUCL template instantiation -- the part of Boda's code generator that every *generic* (non-convolution-variant) kernel goes
through, restated from `src/rtc_func_gen.cc` so that a Boda checkout's `test/rtc/*.cucl` templates can be turned into the
source text + arg list + launch geometry that `be=hip` compiles (first part).

A template is C-like source with
  * `%(name)` template variables,
  * one magic comment per kernel argument:  `<decl> // CUCL IN|OUT|INOUT|REF[_DYN] dim:dim:... [alt spec ...]`  (`:` alone = scalar),
  * index declarations:                     `// CUCL IX <ix_var> <arg> [use_dims=a:b:c]`,
  * `// CUCL INCLUDE file.h`.
Instantiating it for an op (`op_base_t`: `func_name` + named ndas) defines (`rtc_call_gen_t::init`, `src/rtc_func_gen.cc:346-421`)
  rtc_func_name | <arg>_tn | <arg>_<dim>_dim | <arg>_<dim>_stride | <arg>_dims_prod   (`insert_nda_dims_sz`, `:214-225`)
  <scalar by-value arg> -> its C constant when the op carries a value, else the argument's own name
  <ix>_<dim> = ((ix/stride)%dim) | <ix>_<dim>_nomod | <ix>_dims_prod                     (`insert_nda_ix_exprs`, `:227-246`)
  tpb | blks | warp_sz
and the launch geometry from the special index names (`:7-23`): GLOB_ID_1D -> tpb 256, blks = ceil(prod/tpb); GRP_ID_1D -> blks;
LOC_ID_1D -> tpb.  `_DYN` arguments keep their dims out of the generated source: `<arg>_<dim>_dim` etc. become references to extra
trailing `int32_t cai__<arg>_<dim>_{dim,stride}` / `cai__<arg>_dims_prod` by-value arguments (declared through the template's
`%(cucl_arg_info_decls)`), an index over such an argument likewise, and values and launch geometry are supplied per call
(`Instance.call_args`; `add_dyn_nda_dims_sz`, `src/rtc_func_gen.cc:429-469`, `rcg_func_call_t::run`, `:496-584`).  `_multi`
arguments (`float_multi const * const ins`: a pack of op[`ins_num`] arguments `ins_0` .. declared through `%(ins_decl)`, `src/rtc_func_gen.cc:24-41,
143-151,388-391`) expand per op; custom code generation is a hook (`custom`), restated for the reference's sgemm / conv variants and `reduce` in
`oracle/cnn_codegen.py`.
"""
from __future__ import annotations
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .op import Dims, Nda, Op, RtErr, UnsupErr

DEFAULT_TPB = 256  # rtc_call_geom_t::get_default_tpb (src/rtc_func_gen.H)


@dataclass
class ArgDecl:
    vn: str
    tn: str            # C type name; "" = any (declared as %(vn_tn)); "none" for void
    loi: int           # levels of indirection: 0 = by value, 1 = pointer
    io_type: str       # IN | OUT | INOUT | REF
    ok_dims: List[Tuple[str, ...]]   # acceptable dim-name lists
    dyn: bool = False  # _DYN: dims arrive per call through cai__* arguments
    multi: bool = False  # type <tn>_multi: stands for op[<vn>_num] arguments <vn>_0 .. <vn>_{n-1} (arg_decl_t::set_vn_tn, src/rtc_func_gen.cc:24-41)


@dataclass
class IxDecl:
    ix_vn: str
    arg_vn: str
    use_dims: List[str] = field(default_factory=list)


@dataclass
class Template:
    name: str
    text: str
    arg_decls: List[ArgDecl]
    ix_decls: List[IxDecl]


@dataclass
class Instance:
    """What rtc_codegen_t hands to rtc_compute_t::compile / ::run for one generated function."""
    func_name: str
    src: str
    arg_names: List[str]   # regular args in declaration order, then the cai__* args of dynamic dims
    tpb: int
    blks: int              # 0 when the geometry depends on dynamic dims (see call_args)
    dyn_vars: List[Tuple[str, str, Tuple[str, ...]]] = field(default_factory=list)   # (cai prefix, source arg, use_dims)

    def call_args(self, dims_of: Dict[str, Dims]) -> Tuple[Dict[str, int], int, int]:
        """Per-call part of a function with dynamic dims: the cai__* argument values from the actual dims of the source args,
        and the launch geometry (a dynamic GLOB_ID_1D index sets blks = ceil(prod / tpb)).  -> (cai values, tpb, blks)"""
        vals: Dict[str, int] = {}
        tpb, blks = self.tpb, self.blks
        for pre, src, use in self.dyn_vars:
            d = dims_of[src]
            names, sizes = list(d.names), list(d.sizes)
            if use:
                sel = [names.index(u) for u in use]
                names, sizes = [names[i] for i in sel], [sizes[i] for i in sel]
            st = _strides(tuple(sizes)); prod = 1
            for n_, s_, sd in zip(names, sizes, st):
                vals[f"cai__{pre}_{n_}_dim"] = s_; vals[f"cai__{pre}_{n_}_stride"] = sd; prod *= s_
            vals[f"cai__{pre}_dims_prod"] = prod
            if pre == "GLOB_ID_1D":
                tpb = tpb or DEFAULT_TPB; blks = -(-prod // tpb)
            elif pre == "GRP_ID_1D":
                blks = prod
            elif pre == "LOC_ID_1D":
                tpb = prod
        return vals, tpb, blks


def _parse_arg_decl(line: str) -> Tuple[str, str, int]:
    """`GASQ float const * const in, // ...` -> (vn, tn, loi)   (arg_decl_t::arg_parse, src/rtc_func_gen.cc:30-38)"""
    decl = line.split("//")[0].strip().rstrip(" ,);{")
    toks = decl.split()
    if not toks:
        raise RtErr("invalid CUCL io var decl; no var name found:" + line)
    vn, loi, tn = toks[-1], 0, ""
    for t in reversed(toks[:-1]):
        if t == "*":
            loi += 1; continue
        if t == "const":
            continue
        tn = t; break
    return vn, tn, loi


def parse_template(name: str, text: str, include_dir: Optional[str] = None) -> Template:
    out_lines: List[str] = []
    arg_decls: List[ArgDecl] = []
    ix_decls: List[IxDecl] = []
    for ln_no, line in enumerate(text.split("\n"), 1):
        out_lines.append(line)
        if "//" not in line:
            continue
        parts = line.split("//", 1)[1].split()
        if not parts or parts[0] != "CUCL":
            continue
        try:
            if len(parts) < 2:
                raise RtErr("invalid CUCL magic comment. missing directive after CUCL.")
            cd = parts[1]
            dyn = cd.endswith("_DYN")
            if dyn:
                cd = cd[:-4]
            if cd == "IX":
                if dyn:
                    raise RtErr("invalid use of _DYN suffix on CUCL IX decl")
                if len(parts) < 4:
                    raise RtErr("invalid CUCL IX decl; missing ix_name and/or arg_name.")
                ix = IxDecl(parts[2], parts[3])
                for opt in parts[4:]:
                    kv = opt.split("=")
                    if len(kv) != 2 or kv[0] != "use_dims":
                        raise RtErr(f"invalid CUCL IX decl option '{opt}'. known opts: use_dims")
                    ix.use_dims = kv[1].split(":")
                ix_decls.append(ix)
            elif cd in ("IN", "INOUT", "OUT", "REF"):
                if len(parts) < 3:
                    raise RtErr("invalid CUCL IN/INOUT/OUT annotation; missing dims spec.")
                vn, tn, loi = _parse_arg_decl(line)
                if dyn and loi == 0:
                    raise RtErr("invalid CUCL io var decl; by-value arguments must not be DYN")
                multi = tn.endswith("_multi")
                if multi:
                    tn = tn[:-len("_multi")]
                if not tn:
                    raise RtErr("invalid CUCL io var decl; no var type found.")
                if loi > 1:
                    raise RtErr("invalid CUCL io var decl; should be exactly zero or one level-of-indirection/*.")
                if loi == 0 and cd == "REF":
                    raise RtErr("invalid CUCL io var decl; REF args must not be by-value (since no value(s) will be passed)")
                if tn == f"%({vn}_tn)":
                    tn = ""
                if tn == "void":
                    tn = "none"
                specs = [tuple() if sp == ":" else tuple(sp.split(":")) for sp in parts[2:]]
                for sp in specs:
                    if any(not d for d in sp):
                        raise RtErr("invalid (currently forbidden/unused) empty dim name in nda_spec")
                arg_decls.append(ArgDecl(vn, tn, loi, cd, specs, dyn, multi))
            elif cd == "INCLUDE":
                if len(parts) != 3:
                    raise RtErr("invalid CUCL INCLUDE decl; must be exactly CUCL INCLUDE filename.h.")
                if include_dir is None:
                    raise RtErr(f"CUCL INCLUDE {parts[2]}: no include directory given")
                with open(os.path.join(include_dir, parts[2])) as f:
                    out_lines.append(f.read())
            else:
                raise RtErr(f"invalid CUCL directive '{cd}'.")
        except RtErr as e:
            raise type(e)(f"Error parsing CUCL template {name} on line {ln_no}:\n--> {line}\n{e}") from None
    return Template(name, "\n".join(out_lines), arg_decls, ix_decls)


def _scalar_const(nda: Nda) -> str:
    """get_scalar_c_const_str (src/boda_base.cc:422-441): integers print plain (a template may paste them into identifiers -- `float%(vw)`), floats
    as %#.9g with an f suffix."""
    v = nda.v[0] if nda.v is not None else None
    if nda.tn in ("float", "half"):
        return "%#.9gf" % float(v)
    if nda.tn == "double":
        return repr(float(v))
    return str(int(v))


def _strides(sizes: Tuple[int, ...]) -> List[int]:
    st, acc = [], 1
    for s in reversed(sizes):
        st.append(acc); acc *= s
    return list(reversed(st))


class CallGen:
    """What a custom code-generation hook sees of the call being generated (the public face of rtc_call_gen_t, src/rtc_func_gen.H:120-170):
    the op, the launch geometry fixed by the template's index declarations, `set(var, val)` for template variables and `line(section, text)`
    for bulk sections, plus insert_nda_ix_exprs for index expressions over the dims of a declared IX."""

    def __init__(self, op: Op, tsvs: Dict[str, str], all_ix_dims: Dict[str, Tuple[Tuple[str, ...], Tuple[int, ...]]], tpb: int, blks: int):
        self.op, self.tsvs, self.all_ix_dims, self.tpb, self.blks = op, tsvs, all_ix_dims, tpb, blks
        self.cgs: Dict[str, List[str]] = {}
        self.multi_args: Dict[str, List[str]] = {}   # _multi declaration -> the argument names it expands to for this op

    def get_arg_dims_by_name(self, vn: str) -> Dims:
        if vn not in self.op.nda_vals:
            raise RtErr(f"referenced arg '{vn}' not present in dims_vals")
        return self.op.get_dims(vn)

    def set(self, var: str, val: str) -> None:
        if var in self.tsvs:
            raise RtErr(f"template variable '{var}' defined twice")
        self.tsvs[var] = val

    def line(self, sec: str, text: str) -> None:
        self.cgs.setdefault(sec, [f"// begin {sec}"]).append("   " + text)

    def insert_nda_ix_exprs(self, ix_vn: str, ix_dims: Tuple[Tuple[str, ...], Tuple[int, ...]], ix_expr: str = "") -> None:
        """src/rtc_func_gen.cc:220-239: <ix>_<dim> = ((expr / stride) %% size), the outermost dim left to overflow."""
        names, sizes = ix_dims
        e = ix_expr or ix_vn
        st = _strides(tuple(sizes)); prod = 1
        for i, (n, sz, sd) in enumerate(zip(names, sizes, st)):
            v = f"({e}/{sd})" if sd > 1 else e
            self.set(f"{ix_vn}_{n}_nomod", v)
            if i:
                v = f"({v}%%{sz})" if sz > 1 else "0"
            self.set(f"{ix_vn}_{n}", v)
            prod *= sz
        self.set(f"{ix_vn}_dims_prod", str(prod))


def instantiate(t: Template, op: Op, gen_fn: str, custom=None) -> Instance:
    """rtc_call_gen_t::init + instantiate_template for a fully static op.  `custom(CallGen, template name)`: the custom_codegen_t hook, called
    after the index declarations and before the arguments are processed, as the reference does (src/rtc_func_gen.cc:386)."""
    tsvs: Dict[str, str] = {"rtc_func_name": gen_fn}
    all_ix_dims: Dict[str, Tuple[Tuple[str, ...], Tuple[int, ...]]] = {}
    tpb = int(op.nda_vals["tpb"].v[0]) if "tpb" in op.nda_vals and op.nda_vals["tpb"].v is not None else 0
    blks = 0
    errs: List[str] = []

    def arg_dims(vn: str, tag: str) -> Dims:
        if vn not in op.nda_vals:
            raise RtErr(f"referenced {tag} arg '{vn}' not present in dims_vals")
        return op.get_dims(vn)

    def put(k: str, v: str) -> None:
        if k in tsvs:
            raise RtErr(f"template variable '{k}' defined twice")
        tsvs[k] = v

    dyn_args = {ad.vn for ad in t.arg_decls if ad.dyn}
    dyn_vars: List[Tuple[str, str, Tuple[str, ...]]] = []
    cai_names: List[str] = []
    cai_decls: List[str] = []

    def add_dyn(pre: str, names, add_refs: bool) -> None:
        for n in names:
            for kind in ("dim", "stride"):
                cn = f"cai__{pre}_{n}_{kind}"
                cai_names.append(cn); cai_decls.append(f"   ,int32_t {cn}")
                if add_refs:
                    put(f"{pre}_{n}_{kind}", cn)
        cn = f"cai__{pre}_dims_prod"
        cai_names.append(cn); cai_decls.append(f"   ,int32_t {cn}")
        if add_refs:
            put(f"{pre}_dims_prod", cn)

    for ix in t.ix_decls:
        if ix.arg_vn in dyn_args:      # index over an argument whose dims are only known per call (insert_nda_dyn_ix_exprs)
            d = arg_dims(ix.arg_vn, "IX")
            names = list(ix.use_dims) if ix.use_dims else list(d.names)
            for u in names:
                if u not in d.names:
                    raise RtErr(f"specified use_dim '{u}' not found in target arg's dims")
            if ix.ix_vn == ix.arg_vn:
                raise RtErr("CUCL IX over a dynamic arg must not share its name")
            dyn_vars.append((ix.ix_vn, ix.arg_vn, tuple(ix.use_dims)))
            add_dyn(ix.ix_vn, names, False)
            for i, n in enumerate(names):
                v = f"({ix.ix_vn}/cai__{ix.ix_vn}_{n}_stride)"
                put(f"{ix.ix_vn}_{n}_nomod", v)
                if i:
                    v = f"({v}%cai__{ix.ix_vn}_{n}_dim)"
                put(f"{ix.ix_vn}_{n}", v)
            put(f"{ix.ix_vn}_dims_prod", f"cai__{ix.ix_vn}_dims_prod")
            if ix.ix_vn == "GLOB_ID_1D":
                tpb = tpb or DEFAULT_TPB
            continue
        d = arg_dims(ix.arg_vn, "IX")
        names, sizes = list(d.names), list(d.sizes)
        if ix.use_dims:
            sel = []
            for u in ix.use_dims:
                if u not in names:
                    raise RtErr(f"specified use_dim '{u}' not found in target arg's dims")
                sel.append(names.index(u))
            names, sizes = [names[i] for i in sel], [sizes[i] for i in sel]
        if not names or any(s == 0 for s in sizes):
            raise UnsupErr(f"CUCL template {t.name}: IX {ix.ix_vn} over dynamically sized arg")
        all_ix_dims[ix.ix_vn] = (tuple(names), tuple(sizes))
        st = _strides(tuple(sizes)); prod = 1
        for s in sizes:
            prod *= s
        for i, (n, s, sd) in enumerate(zip(names, sizes, st)):
            v = f"({ix.ix_vn}/{sd})" if sd > 1 else ix.ix_vn
            put(f"{ix.ix_vn}_{n}_nomod", v)
            if i:
                v = f"({v}%{s})" if s > 1 else "0"   # the outermost dim is left to overflow (src/rtc_func_gen.cc:233-241)
            put(f"{ix.ix_vn}_{n}", v)
        put(f"{ix.ix_vn}_dims_prod", str(prod))
        if ix.ix_vn == "GLOB_ID_1D":
            tpb = tpb or DEFAULT_TPB
            if blks:
                raise RtErr("CUCL error: GLOB_ID_1D IX encoutered after setting blks (some other way)")
            blks = -(-prod // tpb)
        elif ix.ix_vn == "GRP_ID_1D":
            if blks:
                raise RtErr("CUCL error: GRP_ID_1D IX encoutered after setting blks (some other way)")
            blks = prod
        elif ix.ix_vn == "LOC_ID_1D":
            if tpb:
                raise RtErr("CUCL error: LOC_ID_1D IX encoutered after setting tpb (some other way)")
            tpb = prod

    # a _multi declaration stands for op[<vn>_num] arguments <vn>_0 .. <vn>_{n-1} (vect_arg_decl_t::multi_iter, src/rtc_func_gen.H:69-86)
    expanded: List[Tuple[ArgDecl, str]] = []
    for ad in t.arg_decls:
        if not ad.multi:
            expanded.append((ad, ad.vn)); continue
        num_vn = ad.vn + "_num"
        if num_vn not in op.nda_vals or op.nda_vals[num_vn].v is None:
            errs.append(f"multi arg '{ad.vn}' in template is missing required num field '{num_vn}' in op; "); continue
        for mix in range(int(op.nda_vals[num_vn].v[0])):
            expanded.append((ad, f"{ad.vn}_{mix}"))
    cg = CallGen(op, tsvs, all_ix_dims, tpb, blks)
    cg.multi_args = {ad.vn: [vn for a2, vn in expanded if a2 is ad] for ad in t.arg_decls if ad.multi}
    if custom is not None:
        custom(cg, t.name)
    arg_names: List[str] = []
    for ad, vn_x in expanded:
        arg_names.append(vn_x)
        if ad.multi:      # the declaration line of this member of the pack (src/rtc_func_gen.cc:391)
            cg.line(ad.vn + "_decl", f"GASQ {ad.tn} const * const {vn_x},")
        ad = ad if not ad.multi else ArgDecl(vn_x, ad.tn, ad.loi, ad.io_type, ad.ok_dims, ad.dyn, True)
        if ad.vn not in op.nda_vals:
            errs.append(f"referenced {ad.io_type} arg '{ad.vn}' not present in dims_vals; "); continue
        nda = op.nda_vals[ad.vn]; d = nda.dims if nda.dims is not None else Dims((), (), nda.tn)
        if not any(tuple(d.names) == sp for sp in ad.ok_dims):
            errs.append(f"call arg '{ad.vn}' incompatible with decl arg (dim count mismatch or dim name mismatch: {tuple(d.names)} vs {ad.ok_dims}); ")
            continue
        if ad.tn and ad.tn != "none" and nda.tn not in (ad.tn, "none") and d.names:
            errs.append(f"call arg '{ad.vn}' has type {nda.tn}, template wants {ad.tn}; ")
        if ad.loi == 0 and d.dims_prod() != 1:
            errs.append(f"call arg '{ad.vn}' incompatible with decl arg (by-value arguments must be scalar); "); continue
        put(f"{ad.vn}_tn", nda.tn)
        if ad.dyn:
            dyn_vars.append((ad.vn, ad.vn, ()))
            add_dyn(ad.vn, d.names, True)
            continue
        dims_only = False      # (insert_nda_dims_sz leaves strides out only for dims without them; the dims an annotation builds all carry strides -- conv_simd reads %(in_pels_x_stride) of a REF)
        st = _strides(tuple(d.sizes))
        for n, s, sd in zip(d.names, d.sizes, st):
            put(f"{ad.vn}_{n}_dim", str(s))
            if not dims_only:
                put(f"{ad.vn}_{n}_stride", str(sd))
        if not dims_only:
            put(f"{ad.vn}_dims_prod", str(d.dims_prod()))
        if ad.loi == 0:
            put(ad.vn, _scalar_const(nda) if nda.v is not None else ad.vn)
    if errs:
        raise RtErr(f"RTC template function instantiation argument error: {t.name}: " + "".join(errs))
    if "tpb" not in tsvs:
        tsvs["tpb"] = str(tpb)
    tsvs["blks"] = str(blks); tsvs["warp_sz"] = "UNKNOWN"
    if cai_decls:
        tsvs["cucl_arg_info_decls"] = "// begin cucl_arg_info_decls\n" + "\n".join(cai_decls) + "\n"
    elif "cucl_arg_info_decls" not in tsvs:
        tsvs["cucl_arg_info_decls"] = ""

    if True:                # terminate and emit the bulk sections (src/rtc_func_gen.cc:478-481)
        for sec, lines in cg.cgs.items():
            if sec in tsvs:
                raise RtErr(f"template variable '{sec}' defined twice")
            tsvs[sec] = "\n".join(lines) + f"\n    // end {sec}\n"

    def expand(text: str, depth: int = 0) -> str:
        """%(name) -> the (itself expanded) value, %% -> % : values may refer to other template variables (generated lines do)."""
        if depth > 16:
            raise RtErr(f"CUCL template {t.name}: template variables nest deeper than 16 levels (a cycle?)")
        out: List[str] = []; i = 0; n = len(text)
        while i < n:
            c = text[i]
            if c != "%" or i + 1 >= n:
                out.append(c); i += 1; continue
            if text[i + 1] == "%":
                out.append("%"); i += 2; continue
            if text[i + 1] == "(":
                j = text.find(")", i + 2)
                k = text[i + 2:j] if j > 0 else ""
                if j > 0 and re.fullmatch(r"[A-Za-z0-9_]+", k):
                    if k not in tsvs:
                        raise RtErr(f"CUCL template {t.name}: unknown template variable %({k})")
                    out.append(expand(tsvs[k], depth + 1)); i = j + 1; continue
            out.append(c); i += 1
        return "".join(out)
    src = expand(t.text)
    if not tpb:
        raise RtErr(f"CUCL template {t.name}: launch geometry not determined (no GLOB_ID_1D / LOC_ID_1D index)")
    return Instance(gen_fn, src, arg_names + cai_names, tpb, blks, dyn_vars)


def load_template(rtc_dir: str, name: str) -> Template:
    """A Boda checkout's test/rtc/<name>.cucl (rtc_template_t::init, src/rtc_func_gen.cc:47-63)."""
    with open(os.path.join(rtc_dir, name + ".cucl")) as f:
        return parse_template(name, f.read(), include_dir=rtc_dir)

"""Op descriptions: dims_t / nda_t text forms, lexp grammar, op_base_t.

Restates (behaviour only) the reference's
  * lexp grammar                      src/lexp.cc (value = leaf | '(' name '=' value {',' ...} ')', '\\' escapes)
  * nda / dims text forms             src/nesi.cc:661-785, printer src/boda_base.cc:403-440
  * op_base_t {str_vals, nda_vals}    src/op_base.H:9-43, ordering src/op_base.cc:16-23
  * legacy '(type=T,dims_vals=(...))' form still used by test/sgemm-ops-{micro,tiny,small,full}.txt
  * Convolution / sgemm arg tables    src/conv_util.cc:25-35
"""
from __future__ import annotations
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple, Union

TYPE_SIZES = {"none": 0, "half": 2, "bfloat16": 2, "float": 4, "double": 8, "int32_t": 4, "uint32_t": 4, "uint16_t": 2, "uint8_t": 1}


class RtErr(RuntimeError):
    """rt_err: fatal error (src/boda_base.H:98)."""


class UnsupErr(RuntimeError):
    """unsup_err: 'this configuration is unsupported'; callers may catch and record (src/boda_base.H:105)."""


# ------------------------------------------------------------------------------------------------
# lexp
# ------------------------------------------------------------------------------------------------
Lexp = Union[str, List[Tuple[str, "Lexp"]]]


def parse_lexp(s: str) -> Lexp:
    """Parse one lexp.  Leaf -> str; list -> [(name, value), ...] (order preserved)."""
    pos = 0
    n = len(s)

    def parse_value() -> Lexp:
        nonlocal pos
        if pos < n and s[pos] == "(":
            pos += 1
            items: List[Tuple[str, Lexp]] = []
            if pos < n and s[pos] == ")":
                pos += 1
                return items
            while True:
                name = []
                while pos < n and s[pos] != "=":
                    if s[pos] in "(),":
                        raise RtErr(f"lexp: invalid char {s[pos]!r} in name at {pos}: {s[:pos+1]!r}")
                    if s[pos] == "\\":
                        pos += 1
                    name.append(s[pos])
                    pos += 1
                if pos >= n:
                    raise RtErr("lexp: unexpected end in name")
                pos += 1  # '='
                val = parse_value()
                items.append(("".join(name), val))
                if pos >= n:
                    raise RtErr("lexp: unexpected end of input in list")
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    return items
                raise RtErr(f"lexp: expected ',' or ')' at {pos}")
        leaf = []
        while pos < n and s[pos] not in ",)":
            if s[pos] == "(":
                raise RtErr(f"lexp: unexpected '(' in leaf at {pos}")
            if s[pos] == "\\":
                pos += 1
            leaf.append(s[pos])
            pos += 1
        return "".join(leaf)

    v = parse_value()
    if pos != n:
        raise RtErr(f"lexp: trailing characters at {pos}: {s[pos:]!r}")
    return v


def _kv(l: Lexp) -> Dict[str, Lexp]:
    if isinstance(l, str):
        raise RtErr(f"lexp: expected list, got leaf {l!r}")
    d: Dict[str, Lexp] = {}
    for k, v in l:
        if k in d:
            raise RtErr(f"lexp: duplicate key {k!r}")
        d[k] = v
    return d


# ------------------------------------------------------------------------------------------------
# dims_t / nda_t
# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Dims:
    """Row-major named dims + element type name (src/boda_base.H:498-690). Unpadded strides only."""
    names: Tuple[str, ...] = ()
    sizes: Tuple[int, ...] = ()
    tn: str = "float"

    def __post_init__(self):
        if len(self.names) != len(self.sizes):
            raise RtErr("dims: names/sizes length mismatch")
        if self.tn not in TYPE_SIZES:
            raise RtErr(f"dims: unknown type name {self.tn!r}")

    @staticmethod
    def make(tn: str = "float", **kw: int) -> "Dims":
        return Dims(tuple(kw.keys()), tuple(int(v) for v in kw.values()), tn)

    def dsz(self, name: str) -> int:
        try:
            return self.sizes[self.names.index(name)]
        except ValueError:
            raise RtErr("dim not found:" + name)

    def has(self, name: str) -> bool:
        return name in self.names

    @property
    def strides(self) -> Tuple[int, ...]:
        st, acc = [], 1
        for sz in reversed(self.sizes):
            st.append(acc)
            acc *= sz
        return tuple(reversed(st))

    def dstride(self, name: str) -> int:
        return self.strides[self.names.index(name)]

    def dims_prod(self) -> int:
        p = 1
        for s in self.sizes:
            p *= s
        return p

    def bytes_sz(self) -> int:
        return self.dims_prod() * TYPE_SIZES[self.tn]

    def is_zeros(self) -> bool:
        return all(s == 0 for s in self.sizes)

    def with_tn(self, tn: str) -> "Dims":
        return Dims(self.names, self.sizes, tn)

    def param_str(self) -> str:
        return "(" + ",".join(f"{n}={s}" for n, s in zip(self.names, self.sizes)) + ")"

    def pretty(self) -> str:
        return "DIMS[" + ":".join(f"{n}={s}" for n, s in zip(self.names, self.sizes)) + "]"


@dataclass
class Nda:
    """nda_t restricted to what op descriptions need: dims (+ optional small value list)."""
    dims: Optional[Dims] = None  # None => scalar
    tn: str = "float"
    v: Optional[Tuple[Union[int, float], ...]] = None

    def key(self):
        return (self.tn, self.dims.names if self.dims else (), self.dims.sizes if self.dims else (), self.v or ())

    def scalar(self):
        if self.v is None or len(self.v) != 1:
            raise RtErr("nda: not a scalar-with-value")
        return self.v[0]

    def to_str(self) -> str:
        """Canonical printer: `tn` only when dims absent or tn != float (src/boda_base.cc:421-440)."""
        parts = []
        if self.dims is None or self.tn != "float":
            parts.append(f"tn={self.tn}")
        if self.dims is not None:
            parts.append("dims=" + self.dims.param_str())
        if self.v is not None:
            parts.append("v=" + ":".join(_fmt_val(x, self.tn) for x in self.v))
        return "(" + ",".join(parts) + ")"


def _fmt_val(x, tn: str) -> str:
    if tn in ("float", "double", "half"):
        return repr(float(x)) if float(x) != int(float(x)) else str(int(float(x)))
    return str(int(x))


def _parse_dims(l: Lexp, tn: str) -> Tuple[Dims, str]:
    names, sizes = [], []
    for k, v in (l if not isinstance(l, str) else []):
        if k == "__tn__":
            tn = str(v)
            continue
        names.append(k)
        sizes.append(int(v))
    return Dims(tuple(names), tuple(sizes), tn), tn


def parse_nda(l: Lexp) -> Nda:
    d = _kv(l)
    unknown = set(d) - {"tn", "dims", "v"}
    if unknown:
        raise RtErr(f"nda: unknown fields {sorted(unknown)}")
    has_dims = "dims" in d
    tn = str(d["tn"]) if "tn" in d else ("float" if has_dims else None)
    if tn is None:
        raise RtErr("nda: scalar without tn")
    dims = None
    if has_dims:
        dims, tn = _parse_dims(d["dims"], tn)
    vals = None
    if "v" in d:
        toks = [t for t in str(d["v"]).replace(" ", ":").split(":") if t]
        conv = float if tn in ("float", "double", "half") else int
        vals = tuple(conv(t) for t in toks)
        n_expect = dims.dims_prod() if dims is not None else 1
        if len(vals) != n_expect:
            raise RtErr(f"nda: expected {n_expect} values, got {len(vals)}")
    return Nda(dims=dims, tn=tn, v=vals)


# ------------------------------------------------------------------------------------------------
# op_base_t
# ------------------------------------------------------------------------------------------------
# (type) -> (bottom/input arg names, top/output arg names, required non-tensor fields); src/conv_util.cc:25-35
OP_INFO = {
    "Convolution": (("in", "filts", "biases"), ("out",), ("kern_sz", "stride", "in_pad", "out_chans")),
    "sgemm": (("a", "b"), ("c",), ()),
}


@dataclass
class Op:
    str_vals: Dict[str, str] = field(default_factory=dict)
    nda_vals: Dict[str, Nda] = field(default_factory=dict)

    # -- op_base_t convenience accessors (src/op_base.cc:25-51); same names
    def has(self, an: str) -> bool:
        return an in self.nda_vals

    def get(self, an: str) -> Nda:
        if an not in self.nda_vals:
            raise RtErr(f"op: missing nda_val {an!r}")
        return self.nda_vals[an]

    def get_dims(self, an: str) -> Dims:
        d = self.get(an).dims
        if d is None:
            raise RtErr(f"op: {an!r} is a scalar, has no dims")
        return d

    def set_dims(self, an: str, dims: Dims) -> None:
        if an in self.nda_vals:
            raise RtErr(f"op: {an!r} already set")  # must_insert
        self.nda_vals[an] = Nda(dims=dims, tn=dims.tn)

    def reset_dims(self, an: str, dims: Dims) -> None:
        self.get(an)
        self.nda_vals[an] = Nda(dims=dims, tn=dims.tn)

    def get_u32(self, an: str) -> int:
        return int(self.get(an).scalar())

    def set_u32(self, an: str, v: int) -> None:
        if an in self.nda_vals:
            raise RtErr(f"op: {an!r} already set")
        self.nda_vals[an] = Nda(dims=None, tn="uint32_t", v=(int(v),))

    def get_str(self, k: str) -> str:
        if k not in self.str_vals:
            raise RtErr(f"op: missing str_val {k!r}")
        return self.str_vals[k]

    def get_type(self) -> str:
        return self.get_str("type")

    def has_func_name(self) -> bool:
        return "func_name" in self.str_vals

    def get_func_name(self) -> str:
        return self.get_str("func_name")

    def set_func_name(self, fn: str) -> None:
        if "func_name" in self.str_vals:
            raise RtErr("op: func_name already set")
        self.str_vals["func_name"] = fn

    def copy(self) -> "Op":
        return Op(dict(self.str_vals), {k: Nda(v.dims, v.tn, v.v) for k, v in self.nda_vals.items()})

    def sort_key(self):
        """op_base_t::operator< : str_vals map, then nda_vals map, lexicographic (src/op_base.cc:16-23)."""
        return (tuple(sorted(self.str_vals.items())), tuple((k, self.nda_vals[k].key()) for k in sorted(self.nda_vals)))

    def __eq__(self, o):
        return isinstance(o, Op) and self.sort_key() == o.sort_key()

    def to_str(self) -> str:
        """Canonical one-line form; std::map order (sorted keys), as NESI prints it."""
        sv = ",".join(f"{k}={self.str_vals[k]}" for k in sorted(self.str_vals))
        nv = ",".join(f"{k}={self.nda_vals[k].to_str()}" for k in sorted(self.nda_vals))
        return f"(str_vals=({sv}),nda_vals=({nv}))"

    # -- shape helpers for the two op types on the hot path
    def conv_geom(self) -> dict:
        """Named geometry of a Convolution; validates out = (in + 2*pad - k)/stride + 1 (src/conv_util.cc:167-173)."""
        i, f, o = self.get_dims("in"), self.get_dims("filts"), self.get_dims("out")
        st, pad = self.get_dims("stride"), self.get_dims("in_pad")
        g = dict(B=i.dsz("img"), C=i.dsz("chan"), H=i.dsz("y"), W=i.dsz("x"),
                 OC=f.dsz("out_chan"), KH=f.dsz("y"), KW=f.dsz("x"),
                 SY=st.dsz("y"), SX=st.dsz("x"), PY=pad.dsz("y"), PX=pad.dsz("x"),
                 OH=o.dsz("y"), OW=o.dsz("x"))
        if self.has("hip_pool") and self.get_u32("hip_pool"):   # a max pooling fused in front (cnn_op.fuse_f32_pool): `in` is the POOLING's input, H x W the pooled plane
            ks, ps = self.get_dims("pool_sz"), self.get_dims("pool_stride")
            g.update(UH=g["H"], UW=g["W"], PKH=ks.dsz("y"), PKW=ks.dsz("x"), PSY=ps.dsz("y"), PSX=ps.dsz("x"))
            g["H"] = (g["UH"] - g["PKH"]) // g["PSY"] + 1; g["W"] = (g["UW"] - g["PKW"]) // g["PSX"] + 1
        f_in = f.dsz("in_grp") * f.dsz("in_chan8") if f.has("in_grp") else f.dsz("in_chan")   # (filts in the input-patch kernel's in_grp:y:x:out_chan:in_chan8 form, boda_amd/nhwc.py)
        if f_in != g["C"]:
            raise RtErr("conv: filts.in_chan != in.chan (groups are not on this path)")
        if o.dsz("img") != g["B"] or o.dsz("chan") != g["OC"]:
            raise RtErr("conv: out dims inconsistent with in/filts")
        for hw, k, s, p, oo in (("H", "KH", "SY", "PY", "OH"), ("W", "KW", "SX", "PX", "OW")):
            if (g[hw] + 2 * g[p] - g[k]) // g[s] + 1 != g[oo]:
                raise RtErr(f"conv: out {oo}={g[oo]} != ({hw}+2*{p}-{k})/{s}+1")
        return g

    def sgemm_geom(self) -> dict:
        a, b, c = self.get_dims("a"), self.get_dims("b"), self.get_dims("c")
        g = dict(M=a.dsz("M"), K=a.dsz("K"), N=b.dsz("N"))
        if b.dsz("K") != g["K"] or c.dsz("M") != g["M"] or c.dsz("N") != g["N"]:
            raise RtErr("sgemm: inconsistent a/b/c dims")
        return g

    def flops(self) -> int:
        """2*M*N*K with the reference's accounting (src/latex-util.H:116-120,126-133)."""
        if self.get_type() == "sgemm":
            g = self.sgemm_geom()
            return 2 * g["M"] * g["N"] * g["K"]
        g = self.conv_geom()
        return 2 * (g["B"] * g["OH"] * g["OW"]) * g["OC"] * (g["C"] * g["KH"] * g["KW"])

    def algo_bytes(self) -> int:
        """4*(in+out+filts+biases) resp. 4*(a+b+c) (src/latex-util.H:119,133)."""
        ins, outs, _ = OP_INFO[self.get_type()]
        return sum(self.get_dims(a).bytes_sz() for a in ins + outs)


def parse_op(line: str) -> Op:
    """Parse one op line in either the current or the legacy form."""
    d = _kv(parse_lexp(line.strip()))
    op = Op()
    if "dims_vals" in d or "type" in d:  # legacy form
        unknown = set(d) - {"type", "dims_vals", "str_vals"}
        if unknown:
            raise RtErr(f"op(legacy): unknown fields {sorted(unknown)}")
        op.str_vals["type"] = str(d["type"])
        none_dims = {"kern_sz", "stride", "in_pad"}
        for k, v in _kv(d.get("dims_vals", [])).items():
            tn = "none" if k in none_dims else "float"
            dims, tn = _parse_dims(v, tn)
            op.nda_vals[k] = Nda(dims=dims, tn=tn)
        for k, v in _kv(d.get("str_vals", [])).items():
            if k == "out_chans":  # became a uint32 nda in the current form
                op.nda_vals[k] = Nda(dims=None, tn="uint32_t", v=(int(v),))
            else:
                op.str_vals[k] = str(v)
    else:
        unknown = set(d) - {"str_vals", "nda_vals"}
        if unknown:
            raise RtErr(f"op: unknown fields {sorted(unknown)}")
        for k, v in _kv(d.get("str_vals", [])).items():
            op.str_vals[k] = str(v)
        for k, v in _kv(d.get("nda_vals", [])).items():
            op.nda_vals[k] = parse_nda(v)
    t = op.str_vals.get("type")
    if t in OP_INFO and "func_name" not in op.str_vals:  # annotated ops carry variant-specific layouts: not validated
        ins, outs, req = OP_INFO[t]
        for an in ins + outs + req:
            if an not in op.nda_vals:
                raise RtErr(f"op: {t} is missing required field {an!r}")
        if t == "Convolution":
            op.conv_geom()
        else:
            op.sgemm_geom()
    return op


def read_ops(path: str) -> List[Op]:
    with open(path) as f:
        return [parse_op(l) for l in f if l.strip()]


def data_path(*parts: str) -> str:
    """Path of a shape-data file the package ships (boda_amd/data/): the op lists / net records of the BASELINE workloads.  Test fixtures
    (reference-held digests, the reference's other op lists) live under tests/golden/ and are never read by product code."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", *parts)

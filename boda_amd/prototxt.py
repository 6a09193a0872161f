"""Caffe prototxt (text format) -> layer list -> per-layer Convolution ops with inferred shapes.

Restates what the reference's net reader needs for this path (src/caffepb.cc:166-326, src/conv_util.cc:405-529):
TEST-phase layers only; Convolution out = (in + 2*pad - k)/stride + 1 (floor); Pooling uses ceil and `global_pooling`;
Concat sums channels; ReLU / LRN / Dropout / BatchNorm / Scale / Eltwise keep the shape; InnerProduct is a convolution
whose kernel is the whole input; Accuracy / Softmax* / loss layers are ignored.  Every Convolution gets a bias input
(the reference always attaches `<name>_biases`, src/caffepb.cc:227-228, even when the prototxt says bias_term: false).
Only the subset of the text format that net definitions use is parsed: `key: value` and `key { ... }`.
"""
from __future__ import annotations
import re
from typing import Dict, List, Tuple, Union

from .op import Op, RtErr, parse_op

Node = Dict[str, list]


def parse(text: str) -> Node:
    toks = re.findall(r'"[^"]*"|[{}]|[^\s{}:]+:?', re.sub(r"#.*", "", text))
    pos = 0

    def block() -> Node:
        nonlocal pos
        d: Node = {}
        while pos < len(toks) and toks[pos] != "}":
            key = toks[pos]; pos += 1
            if key.endswith(":"):
                key = key[:-1]
                val = toks[pos]; pos += 1
                d.setdefault(key, []).append(val.strip('"'))
            else:
                if toks[pos] != "{":
                    raise RtErr(f"prototxt: expected '{{' after {key!r}")
                pos += 1
                d.setdefault(key, []).append(block())
                if pos >= len(toks) or toks[pos] != "}":
                    raise RtErr("prototxt: unbalanced braces")
                pos += 1
        return d
    root = block()
    if pos != len(toks):
        raise RtErr("prototxt: trailing tokens")
    return root


def _one(d: Node, k: str, default=None):
    v = d.get(k)
    return v[0] if v else default


def _is_test_phase(layer: Node) -> bool:
    for inc in layer.get("include", []):
        if _one(inc, "phase") == "TRAIN":
            return False
    return True


def conv_ops(text: str, batch: int, in_chw: Tuple[int, int, int] = (3, 224, 224)) -> List[Tuple[str, Op]]:
    """-> [(layer name, Convolution op at `batch`)] in definition order, shapes inferred through the whole net."""
    root = parse(text)
    layers = root.get("layer", []) or root.get("layers", [])
    shapes: Dict[str, Tuple[int, int, int]] = {}
    if "input" in root:
        dims = [int(x) for x in root.get("input_dim", [])]
        shapes[_one(root, "input")] = (dims[1], dims[2], dims[3]) if len(dims) == 4 else in_chw
    out: List[Tuple[str, Op]] = []
    for L in layers:
        if not _is_test_phase(L):
            continue
        t = str(_one(L, "type")).upper().replace("_", "")
        bots, tops, name = L.get("bottom", []), L.get("top", []), _one(L, "name")
        if t == "DATA":
            cs = int(_one(_one(L, "transform_param", {}), "crop_size", in_chw[1]))
            shapes[tops[0]] = (in_chw[0], cs, cs)
            continue
        if t in ("ACCURACY", "SOFTMAX", "SOFTMAXWITHLOSS", "SOFTMAXLOSS"):
            continue
        if not bots or bots[0] not in shapes:
            raise RtErr(f"prototxt: layer {name!r} reads unknown blob {bots[:1]}")
        C, H, W = shapes[bots[0]]
        if t in ("CONVOLUTION", "INNERPRODUCT"):
            if t == "CONVOLUTION":
                cp = _one(L, "convolution_param", {})
                oc = int(_one(cp, "num_output")); k = int(_one(cp, "kernel_size", 1)); s = int(_one(cp, "stride", 1)); p = int(_one(cp, "pad", 0))
                if int(_one(cp, "group", 1)) != 1:
                    raise RtErr(f"prototxt: grouped convolution {name!r} is not on this path")
                kh = kw = k
            else:
                oc = int(_one(_one(L, "inner_product_param", {}), "num_output")); kh, kw, s, p = H, W, 1, 0
            oh = (H + 2 * p - kh) // s + 1; ow = (W + 2 * p - kw) // s + 1
            out.append((name, parse_op(
                f"(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan={oc})),filts=(dims=(out_chan={oc},in_chan={C},y={kh},x={kw})),"
                f"in=(dims=(img={batch},chan={C},y={H},x={W})),in_pad=(tn=none,dims=(y={p},x={p})),kern_sz=(tn=none,dims=(y={kh},x={kw})),"
                f"out=(dims=(img={batch},chan={oc},y={oh},x={ow})),out_chans=(tn=uint32_t,v={oc}),stride=(tn=none,dims=(y={s},x={s}))))")))
            shapes[tops[0]] = (oc, oh, ow)
        elif t == "POOLING":
            pp = _one(L, "pooling_param", {})
            if str(_one(pp, "global_pooling", "false")).lower() == "true":
                shapes[tops[0]] = (C, 1, 1)
            else:
                k = int(_one(pp, "kernel_size")); s = int(_one(pp, "stride", 1)); p = int(_one(pp, "pad", 0))
                osz = lambda i: 1 if i + 2 * p < k else -(-(i + 2 * p - k) // s) + 1
                shapes[tops[0]] = (C, osz(H), osz(W))
        elif t == "CONCAT":
            cs = [shapes[b] for b in bots]
            if any(c[1:] != cs[0][1:] for c in cs):
                raise RtErr(f"prototxt: concat {name!r} of mismatched spatial sizes")
            shapes[tops[0]] = (sum(c[0] for c in cs), H, W)
        elif t in ("RELU", "LRN", "DROPOUT", "BATCHNORM", "SCALE", "ELTWISE", "SPLIT"):
            for tp in tops:
                shapes[tp] = (C, H, W)
        else:
            raise RtErr(f"prototxt: unhandled layer type {t!r} ({name!r})")
    return out


def conv_bottoms(text: str) -> List[Tuple[str, str]]:
    """-> [(layer name, bottom blob)] of the TEST-phase Convolution / InnerProduct layers in definition order (the order of conv_ops): which convolutions read the
    same blob -- sibling convolutions that can run as one fused launch -- is a fact of the net graph, not of the layer shapes."""
    root = parse(text)
    out: List[Tuple[str, str]] = []
    for L in root.get("layer", []) or root.get("layers", []):
        if _is_test_phase(L) and str(_one(L, "type")).upper().replace("_", "") in ("CONVOLUTION", "INNERPRODUCT"):
            out.append((str(_one(L, "name")), str(L.get("bottom", [""])[0])))
    return out


def pipe_spec(text: str, in_chw: Tuple[int, int, int] = (3, 224, 224)) -> List[str]:
    """-> the TEST-phase forward ops of a net definition as compact one-line records (boda_amd.conv_pipe.pipe_from_spec reads
    them back): what the reference's reader keeps of each layer for conv_pipe_t (src/caffepb.cc:166-326).
      input C H W | conv tag bot top oc kh kw sy sx py px | pool tag bot top kh kw sy sx py px avg global
      lrn tag bot top local_size alpha beta k | relu tag bot top | drop tag bot top | concat tag top bot1,bot2,..."""
    root = parse(text)
    layers = root.get("layer", []) or root.get("layers", [])
    out: List[str] = []
    have_input = False
    for L in layers:
        if not _is_test_phase(L):
            continue
        t = str(_one(L, "type")).upper().replace("_", "")
        bots, tops, name = L.get("bottom", []), L.get("top", []), _one(L, "name")
        if t == "DATA":
            cs = int(_one(_one(L, "transform_param", {}), "crop_size", in_chw[1]))
            out.append(f"input {tops[0]} {in_chw[0]} {cs} {cs}"); have_input = True
        elif t in ("ACCURACY", "SOFTMAX", "SOFTMAXWITHLOSS", "SOFTMAXLOSS"):
            continue
        elif t == "CONVOLUTION":
            cp = _one(L, "convolution_param", {})
            if int(_one(cp, "group", 1)) != 1:
                raise RtErr(f"prototxt: grouped convolution {name!r} is not on this path")
            k, s_, p_ = int(_one(cp, "kernel_size", 1)), int(_one(cp, "stride", 1)), int(_one(cp, "pad", 0))
            out.append(f"conv {name} {bots[0]} {tops[0]} {int(_one(cp, 'num_output'))} {k} {k} {s_} {s_} {p_} {p_}")
        elif t == "INNERPRODUCT":
            out.append(f"conv {name} {bots[0]} {tops[0]} {int(_one(_one(L, 'inner_product_param', {}), 'num_output'))} 0 0 1 1 0 0")  # kernel = whole input
        elif t == "POOLING":
            pp = _one(L, "pooling_param", {})
            glob = int(str(_one(pp, "global_pooling", "false")).lower() == "true")
            k = 0 if glob else int(_one(pp, "kernel_size")); s_ = int(_one(pp, "stride", 1)); p_ = int(_one(pp, "pad", 0))
            avg = int(str(_one(pp, "pool", "MAX")).upper() == "AVE")
            out.append(f"pool {name} {bots[0]} {tops[0]} {k} {k} {s_} {s_} {p_} {p_} {avg} {glob}")
        elif t == "LRN":
            lp = _one(L, "lrn_param", {})
            out.append(f"lrn {name} {bots[0]} {tops[0]} {int(_one(lp, 'local_size', 5))} {float(_one(lp, 'alpha', 1.0))} {float(_one(lp, 'beta', 0.75))} {float(_one(lp, 'k', 1.0))}")
        elif t == "RELU":
            out.append(f"relu {name} {bots[0]} {tops[0]}")
        elif t == "DROPOUT":
            out.append(f"drop {name} {bots[0]} {tops[0]}")
        elif t == "CONCAT":
            out.append(f"concat {name} {tops[0]} {','.join(bots)}")
        else:
            raise RtErr(f"prototxt: layer type {t!r} ({name!r}) has no forward op on this path")
    if not have_input and "input" in root:
        dims = [int(x) for x in root.get("input_dim", [])]
        c, h, w = (dims[1], dims[2], dims[3]) if len(dims) == 4 else in_chw
        out.insert(0, f"input {_one(root, 'input')} {c} {h} {w}")
    return out

"""Full-net forward over be=hip: a `conv_pipe_t` (graph) and the `has_conv_fwd_t(mode=rtc)` driver.

Restates the behaviour of the reference's net-level path (not its code):
  * conv_pipe_t / shape inference      src/conv_util.H:169-243, src/conv_util.cc:167-204,405-505
      conv out = (in + 2*pad - k)/stride + 1 (floor); pooling uses the Caffe CEIL convention; no kern_sz = global pooling
  * prototxt -> pipe conventions       src/caffepb.cc:166-326 : each Convolution gets inputs <name>_filts / <name>_biases;
      Dropout is a no-op at inference; Accuracy / Softmax* layers are ignored; ReLU is in-place
  * conv_pipe_fwd_t::{init,gen_op,run_fwd}  src/rtc_fwd.cc:263-577 : ops are annotated, a ReLU that immediately follows a
      conv in place is FUSED into it (conv_has_relu=1, the ReLU emits no call, :486-493,266); one call per remaining op in
      topological order; params are uploaded once; run_fwd = set inputs -> run all calls -> get outputs; the duration
      is get_dur(first call, last call); an optional per-call profile (python assignments, :560-572)
Convolutions run on the native side door (hip_conv); pooling / LRN / un-fused ReLU are this project's own CUCL-dialect
sources (semantics of test/rtc/{pool,lrn,relu}.cucl) and go through the backend's generic hiprtc path.
The reference ships no trained weights (nets/ holds prototxts only), so params default to its deterministic
gen_data mode-5 pattern generated on the device; callers may pass real arrays instead.
"""
from __future__ import annotations
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import os

import numpy as np

from . import gen_data as gd
from .cnn_op import FILTS_KMAJOR_FUNC, K1_CHAIN_FUNC, NATIVE_ARGS, OpTune, add_codegen_annotations, annotate_k1_chain, f32_pool_fusable, fuse_f32_pool, k1_chain_applies
from .cucl_template import instantiate, parse_template
from .op import Dims, Nda, Op, RtErr, UnsupErr
from .rtc import HipCompute, RtcArg, RtcCompileOpts, RtcFuncCall, RtcFuncInfo


# ------------------------------------------------------------------------------------------------
# graph
# ------------------------------------------------------------------------------------------------
@dataclass
class PipeOp:
    tag: str
    type: str                      # Convolution | Pooling | ReLU | LRN | Dropout | Concat
    bot: str
    top: str
    out_chans: int = 0
    kern_sz: Optional[Tuple[int, int]] = None   # (y, x); None for Pooling = global pooling
    stride: Tuple[int, int] = (1, 1)
    in_pad: Tuple[int, int] = (0, 0)
    avg_pool: int = 0
    lrn: Tuple[int, float, float, float] = (5, 1.0, 0.75, 1.0)  # local_size, alpha, beta, k (src/conv_util.cc:42-48)
    bots: Tuple[str, ...] = ()     # Concat: all inputs in channel order (bot == bots[0])

    @property
    def in_place(self) -> bool:
        return self.bot == self.top


class ConvPipe:
    """Linear-chain-or-DAG of ops over named nodes; ops are kept in topological (definition) order."""

    def __init__(self, name: str, in_node: str, in_dims: Dims):
        self.name, self.in_node = name, in_node
        self.nodes: Dict[str, Dims] = {in_node: in_dims}
        self.ops: List[PipeOp] = []
        self.params: Dict[str, Dims] = {}

    def add(self, op: PipeOp) -> "ConvPipe":
        if op.bot not in self.nodes:
            raise RtErr(f"pipe: op {op.tag} reads unknown node {op.bot!r}")
        d = self.nodes[op.bot]
        B, C, H, W = d.dsz("img"), d.dsz("chan"), d.dsz("y"), d.dsz("x")
        if op.type == "Convolution":
            if tuple(op.kern_sz) == (0, 0):   # InnerProduct: a convolution whose kernel is the whole input (src/caffepb.cc:240-262)
                op.kern_sz = (H, W)
            kh, kw = op.kern_sz
            oh = (H + 2 * op.in_pad[0] - kh) // op.stride[0] + 1; ow = (W + 2 * op.in_pad[1] - kw) // op.stride[1] + 1
            if oh < 1 or ow < 1:
                raise RtErr(f"pipe: conv {op.tag}: padded input too small")
            out = Dims.make("float", img=B, chan=op.out_chans, y=oh, x=ow)
            self.params[op.tag + "_filts"] = Dims.make("float", out_chan=op.out_chans, in_chan=C, y=kh, x=kw)
            self.params[op.tag + "_biases"] = Dims.make("float", out_chan=op.out_chans)
        elif op.type == "Pooling":
            if op.kern_sz is None:   # global pooling
                op.kern_sz, op.stride, op.in_pad = (H, W), (1, 1), (0, 0)
            # Caffe CEIL convention; when EITHER padded dim is smaller than the kernel the reference yields (1,1) for both
            # (pad_in_sz.both_dims_ge, src/conv_util.cc:201-203)
            piy, pix = H + 2 * op.in_pad[0], W + 2 * op.in_pad[1]
            if piy < op.kern_sz[0] or pix < op.kern_sz[1]:
                oy = ox = 1
            else:
                oy = -(-(piy - op.kern_sz[0]) // op.stride[0]) + 1; ox = -(-(pix - op.kern_sz[1]) // op.stride[1]) + 1
            out = Dims.make("float", img=B, chan=C, y=oy, x=ox)
        elif op.type in ("ReLU", "LRN", "Dropout"):
            out = d
        elif op.type == "Concat":          # channel concatenation of same-sized maps (src/conv_util.cc:437-446)
            ds = []
            for b in op.bots:
                if b not in self.nodes:
                    raise RtErr(f"pipe: concat {op.tag} reads unknown node {b!r}")
                ds.append(self.nodes[b])
            if any((x.dsz("img"), x.dsz("y"), x.dsz("x")) != (B, H, W) for x in ds):
                raise RtErr(f"pipe: concat {op.tag} of mismatched sizes")
            out = Dims.make("float", img=B, chan=sum(x.dsz("chan") for x in ds), y=H, x=W)
        else:
            raise UnsupErr(f"pipe: op type {op.type!r} has no forward kernel in this backend")
        if op.top in self.nodes and not op.in_place:
            raise RtErr(f"pipe: node {op.top!r} written twice")
        self.nodes[op.top] = out
        self.ops.append(op)
        return self

    def conv_op(self, op: PipeOp) -> Op:
        """The op_base_t line of a Convolution of this pipe (same text form as the ops-prof op lists)."""
        i, o = self.nodes[op.bot], self.nodes[op.top]
        none = lambda y, x: Nda(Dims(("y", "x"), (y, x), "none"), "none")
        return Op({"type": "Convolution"}, {"in": Nda(i), "out": Nda(o), "filts": Nda(self.params[op.tag + "_filts"]),
                                            "biases": Nda(self.params[op.tag + "_biases"]), "kern_sz": none(*op.kern_sz),
                                            "stride": none(*op.stride), "in_pad": none(*op.in_pad),
                                            "out_chans": Nda(None, "uint32_t", (op.out_chans,))})

    def conv_flops(self) -> int:
        return sum(self.conv_op(o).flops() for o in self.ops if o.type == "Convolution")

    def out_node(self) -> str:
        return self.ops[-1].top


class DryRtc:
    """A backend that records instead of running: ConvPipeFwd.init() on it yields the functions a net compiles (`.infos`: every RtcFuncInfo, native and generated)
    without a device -- what `__graft_entry__.build()` pre-specialises so that a fresh GPU box starts warm (sibling groups, level sets, fused poolings included)."""

    def __init__(self):
        self.infos: List[RtcFuncInfo] = []; self._dims: Dict[str, Dims] = {}
        self._gen_data_compiled = True; self._fwd_funcs_compiled = True

    def compile(self, infos, *a, **k):
        self.infos += list(infos)

    def create_var_with_dims(self, vn, dims):
        self._dims[vn] = dims

    def get_var_dims(self, vn):
        return self._dims[vn]

    def copy_var_to_nda(self, vn, dims=None):
        d = dims or self._dims[vn]
        return np.zeros(d.sizes if d.sizes else (1,), dtype={"float": np.float32, "bfloat16": np.uint16}.get(d.tn, np.float32))

    def get_plat_tag(self):
        return "dry"

    def run(self, rfc):
        return 0

    def __getattr__(self, name):       # everything else (release / sync / copies to the device): nothing to do
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: None


def sibling_runs(ops: List[PipeOp], members: List[PipeOp], fused=frozenset()) -> List[List[PipeOp]]:
    """Convolutions that read the same node with the same geometry (`members`, definition order) split into runs that may be emitted as ONE call at
    the run's first member: a later member joins only if no op between the two rewrites the shared bottom -- an in-place op on it that was not fused
    into its producer (X -> convA; in-place ReLU on X; X -> convB: convB must read the rectified data), or a second writer of the node."""
    pos = {o.tag: i for i, o in enumerate(ops)}
    def writes(o: PipeOp, node: str) -> bool:
        return o.tag not in fused and ((o.in_place and o.type != "Dropout" and node in (o.bots or (o.bot,))) or o.top == node)
    runs: List[List[PipeOp]] = []
    for o in members:
        if runs and not any(writes(q, o.bot) for q in ops[pos[runs[-1][0].tag] + 1:pos[o.tag]]):
            runs[-1].append(o)
        else:
            runs.append([o])
    return runs


def _conv(p, tag, bot, oc, k, s=1, pad=0):
    p.add(PipeOp(tag, "Convolution", bot, tag, out_chans=oc, kern_sz=(k, k), stride=(s, s), in_pad=(pad, pad)))
    p.add(PipeOp("relu_" + tag, "ReLU", tag, tag))
    return tag


def nin_imagenet(batch: int, in_hw: int = 227) -> ConvPipe:
    """nets/nin_imagenet/train_val.prototxt (TEST phase): conv1 cccp1 cccp2 pool0 conv2 cccp3 cccp4 pool2 conv3 cccp5 cccp6
    pool3 drop conv4 cccp7 cccp8 pool4(avg 6x6); every conv is followed by an in-place ReLU."""
    p = ConvPipe("nin_imagenet", "data", Dims.make("float", img=batch, chan=3, y=in_hw, x=in_hw))
    n = _conv(p, "conv1", "data", 96, 11, 4); n = _conv(p, "cccp1", n, 96, 1); n = _conv(p, "cccp2", n, 96, 1)
    p.add(PipeOp("pool0", "Pooling", n, "pool0", kern_sz=(3, 3), stride=(2, 2)))
    n = _conv(p, "conv2", "pool0", 256, 5, 1, 2); n = _conv(p, "cccp3", n, 256, 1); n = _conv(p, "cccp4", n, 256, 1)
    p.add(PipeOp("pool2", "Pooling", n, "pool2", kern_sz=(3, 3), stride=(2, 2)))
    n = _conv(p, "conv3", "pool2", 384, 3, 1, 1); n = _conv(p, "cccp5", n, 384, 1); n = _conv(p, "cccp6", n, 384, 1)
    p.add(PipeOp("pool3", "Pooling", n, "pool3", kern_sz=(3, 3), stride=(2, 2)))
    p.add(PipeOp("drop", "Dropout", "pool3", "pool3"))
    n = _conv(p, "conv4", "pool3", 1024, 3, 1, 1); n = _conv(p, "cccp7", n, 1024, 1); n = _conv(p, "cccp8", n, 1000, 1)
    p.add(PipeOp("pool4", "Pooling", n, "pool4", kern_sz=(6, 6), stride=(1, 1), avg_pool=1))
    return p


def alexnet_ng_conv(batch: int, in_hw: int = 227) -> ConvPipe:
    """nets/alexnet_ng_conv (fc layers as convolutions, no groups): conv1 norm1 pool1 conv2 norm2 pool2 conv3 conv4 conv5 pool5
    fc6 drop fc7 drop fc8."""
    p = ConvPipe("alexnet_ng_conv", "data", Dims.make("float", img=batch, chan=3, y=in_hw, x=in_hw))
    lrn = (5, 1e-4, 0.75, 1.0)
    n = _conv(p, "conv1", "data", 96, 11, 4)
    p.add(PipeOp("norm1", "LRN", n, "norm1", lrn=lrn)); p.add(PipeOp("pool1", "Pooling", "norm1", "pool1", kern_sz=(3, 3), stride=(2, 2)))
    n = _conv(p, "conv2", "pool1", 256, 5, 1, 2)
    p.add(PipeOp("norm2", "LRN", n, "norm2", lrn=lrn)); p.add(PipeOp("pool2", "Pooling", "norm2", "pool2", kern_sz=(3, 3), stride=(2, 2)))
    n = _conv(p, "conv3", "pool2", 384, 3, 1, 1); n = _conv(p, "conv4", n, 384, 3, 1, 1); n = _conv(p, "conv5", n, 256, 3, 1, 1)
    p.add(PipeOp("pool5", "Pooling", n, "pool5", kern_sz=(3, 3), stride=(2, 2)))
    n = _conv(p, "fc6", "pool5", 4096, 6); p.add(PipeOp("drop6", "Dropout", n, n))
    n = _conv(p, "fc7", n, 4096, 1); p.add(PipeOp("drop7", "Dropout", n, n))
    p.add(PipeOp("fc8", "Convolution", n, "fc8", out_chans=1000, kern_sz=(1, 1)))
    return p


def pipe_from_spec(name: str, lines: Sequence[str], batch: int) -> ConvPipe:
    """ConvPipe from the one-line op records boda_amd.prototxt.pipe_spec writes (shape data: boda_amd/data/nets/*.txt)."""
    p: Optional[ConvPipe] = None
    for ln in lines:
        f = ln.split()
        if not f or f[0].startswith("#"):
            continue
        if f[0] == "input":
            p = ConvPipe(name, f[1], Dims.make("float", img=batch, chan=int(f[2]), y=int(f[3]), x=int(f[4])))
            continue
        if p is None:
            raise RtErr("pipe spec: no input line before the first op")
        k = f[0]
        if k == "conv":
            oc, kh, kw, sy, sx, py, px = (int(x) for x in f[4:11])
            p.add(PipeOp(f[1], "Convolution", f[2], f[3], out_chans=oc, kern_sz=(kh, kw), stride=(sy, sx), in_pad=(py, px)))
        elif k == "pool":
            kh, kw, sy, sx, py, px, avg, glob = (int(x) for x in f[4:12])
            p.add(PipeOp(f[1], "Pooling", f[2], f[3], kern_sz=None if glob else (kh, kw), stride=(sy, sx), in_pad=(py, px), avg_pool=avg))
        elif k == "lrn":
            p.add(PipeOp(f[1], "LRN", f[2], f[3], lrn=(int(f[4]), float(f[5]), float(f[6]), float(f[7]))))
        elif k == "relu":
            p.add(PipeOp(f[1], "ReLU", f[2], f[3]))
        elif k == "drop":
            p.add(PipeOp(f[1], "Dropout", f[2], f[3]))
        elif k == "concat":
            bots = tuple(f[3].split(","))
            p.add(PipeOp(f[1], "Concat", bots[0], f[2], bots=bots))
        else:
            raise RtErr(f"pipe spec: unknown record {k!r}")
    if p is None:
        raise RtErr("pipe spec: empty")
    return p


def googlenet_conv(batch: int) -> ConvPipe:
    """nets/googlenet_conv (TEST phase, incl. the two auxiliary heads): 64 convs, 9 inception Concats, 2 LRN, 16 pools; the op
    records are shape data written by boda_amd/data/make_net_ops.py with this project's prototxt reader."""
    from .op import data_path
    fn = data_path("nets", "googlenet_conv.txt")
    with open(fn) as f:
        return pipe_from_spec("googlenet_conv", f.read().splitlines(), batch)


# ------------------------------------------------------------------------------------------------
# non-conv forward kernels (CUCL dialect, dims as by-value args)
# ------------------------------------------------------------------------------------------------
FWD_SRC = """
// Concat: copy one input into its channel range of the output (semantics of test/rtc/copy.cucl; src/rtc_fwd.cc:267-280)
CUCL_GLOBAL_KERNEL void fwd_copy( GASQ float const * const in, GASQ float * const out, uint32_t const n_in, uint32_t const chw_in,
                                  uint32_t const chw_out, uint32_t const off_out ) {
  // CUCL IX GLOB_ID_1D in
  if( GLOB_ID_1D >= n_in ) { return; }
  uint32_t const img = GLOB_ID_1D / chw_in;
  out[img*chw_out + off_out + ( GLOB_ID_1D - img*chw_in )] = in[GLOB_ID_1D];
}
// stand-alone ReLU (one that could not be fused into its conv): non-positive values, -0.0 included, become +0.0; NaN passes through
CUCL_GLOBAL_KERNEL void fwd_relu( GASQ float * const inout, uint32_t const n ) {
  // CUCL IX GLOB_ID_1D inout
  uint32_t const i = GLOB_ID_1D;
  if( i < n ) {
    float const v = inout[i];
    if( v <= 0.0f ) { inout[i] = 0.0f; }
  }
}
"""
# Pooling as a CUCL *template* (boda_amd/cucl_template.py): one generated function per distinct pooling geometry, window size /
# stride / padding / plane sizes as constants, so the window loops unroll and the index arithmetic folds -- how the reference
# specialises its own pool kernel (test/rtc/pool.cucl + src/rtc_func_gen.cc).  The window is clipped to the plane once per output
# (padding pixels never take part, for max or for average); taps are visited column by column, the order the reference sums an
# average in, so averages come out bit-identical to it; the divisor is the clipped window's area.
POOL_TEMPLATE = """
CUCL_GLOBAL_KERNEL void %(rtc_func_name)( GASQ float const * const in, // CUCL IN img:chan:y:x
                                          uint32_t const avg_pool, // CUCL IN :
                                          GASQ void const * const kern_sz, // CUCL REF y:x
                                          GASQ void const * const stride, // CUCL REF y:x
                                          GASQ void const * const in_pad, // CUCL REF y:x
                                          GASQ float * const out ) // CUCL OUT img:chan:y:x
{
  // CUCL IX GLOB_ID_1D out
  if( GLOB_ID_1D >= %(GLOB_ID_1D_dims_prod) ) { return; }
  int32_t const y0 = (int32_t)%(GLOB_ID_1D_y)*%(stride_y_dim) - %(in_pad_y_dim);   // window origin in the plane (may lie in the padding)
  int32_t const x0 = (int32_t)%(GLOB_ID_1D_x)*%(stride_x_dim) - %(in_pad_x_dim);
  int32_t const ty_lo = ( y0 < 0 ) ? -y0 : 0;                                       // taps [t_lo, t_hi) fall inside the plane
  int32_t const tx_lo = ( x0 < 0 ) ? -x0 : 0;
  int32_t const ty_hi = ( %(in_y_dim) - y0 < %(kern_sz_y_dim) ) ? %(in_y_dim) - y0 : %(kern_sz_y_dim);
  int32_t const tx_hi = ( %(in_x_dim) - x0 < %(kern_sz_x_dim) ) ? %(in_x_dim) - x0 : %(kern_sz_x_dim);
  int32_t const w0 = %(GLOB_ID_1D_img_nomod)*%(in_img_stride) + %(GLOB_ID_1D_chan)*%(in_chan_stride) + y0*%(in_y_stride) + x0;
  float acc = %(avg_pool) ? 0.0f : -FLT_MAX;
#if BODAHIP_POOL_UNCOND
  // every tap is loaded, unconditionally, from the nearest tap INSIDE the clipped window (so the loads are independent and all in flight together): a duplicate
  // does not change a maximum, an average adds only the taps that are inside -- same values, same order of additions.  Needs a non-empty window.
  float v[%(kern_sz_x_dim)][%(kern_sz_y_dim)];
#pragma unroll
  for( int32_t tx = 0; tx != %(kern_sz_x_dim); ++tx ) {
#pragma unroll
    for( int32_t ty = 0; ty != %(kern_sz_y_dim); ++ty ) {
      int32_t const cy = ( ty < ty_lo ) ? ty_lo : ( ( ty >= ty_hi ) ? ty_hi - 1 : ty ), cx = ( tx < tx_lo ) ? tx_lo : ( ( tx >= tx_hi ) ? tx_hi - 1 : tx );
      v[tx][ty] = in[w0 + cy*%(in_y_stride) + cx];
    }
  }
#pragma unroll
  for( int32_t tx = 0; tx != %(kern_sz_x_dim); ++tx ) {
#pragma unroll
    for( int32_t ty = 0; ty != %(kern_sz_y_dim); ++ty ) {
      #pragma clang fp reassociate(off) contract(off)      // (fast-math build: an average keeps the written order of its additions)
      bool const ok = ( tx >= tx_lo ) && ( tx < tx_hi ) && ( ty >= ty_lo ) && ( ty < ty_hi );
      if( %(avg_pool) ) { if( ok ) { acc = acc + v[tx][ty]; } } else { acc = ( v[tx][ty] > acc ) ? v[tx][ty] : acc; }
    }
  }
#else
  for( int32_t tx = 0; tx != %(kern_sz_x_dim); ++tx ) {
    if( tx < tx_lo || tx >= tx_hi ) { continue; }
    for( int32_t ty = 0; ty != %(kern_sz_y_dim); ++ty ) {
      if( ty < ty_lo || ty >= ty_hi ) { continue; }
      float const v = in[w0 + ty*%(in_y_stride) + tx];
      acc = %(avg_pool) ? ( acc + v ) : ( ( v > acc ) ? v : acc );
    }
  }
#endif
  int32_t const n_y = ( ty_hi > ty_lo ) ? ty_hi - ty_lo : 0;
  int32_t const n_x = ( tx_hi > tx_lo ) ? tx_hi - tx_lo : 0;
  if( %(avg_pool) ) { acc /= (float)( n_y*n_x ); }
  out[GLOB_ID_1D] = acc;
}
"""
# Across-channel LRN as a template: local_size, channel count and strides are constants.  The window of the last local_size inputs
# lives in a register shift line (newest last); the sum of its squares is carried from channel to channel -- add the entering
# square, then take away the leaving one, the update order of the reference's caffe-matching path (test/rtc/lrn.cucl:35-50), which
# the oracle restates -- and   out[c] = in[c] * ( k + sumsq * alpha / local_size ) ^ -beta   once the window is centred on c.
LRN_TEMPLATE = """
CUCL_GLOBAL_KERNEL void %(rtc_func_name)( float const alpha, // CUCL IN :
                                          float const beta, // CUCL IN :
                                          float const k, // CUCL IN :
                                          uint32_t const local_size, // CUCL IN :
                                          GASQ float const * const in, // CUCL IN img:chan:y:x
                                          GASQ void const * const work, // CUCL REF img:cblk:cblk_sz:y:x
                                          GASQ float * const out ) // CUCL OUT img:chan:y:x
{
  // one thread per (img, block of cblk_sz channels, y, x): it slides the window over its block plus a halo of local_size/2 channels
  // on either side (a few loads repeated per block buys cblk x the threads of a one-thread-per-pixel form)
  // CUCL IX GLOB_ID_1D work use_dims=img:cblk:y:x
  if( GLOB_ID_1D >= %(GLOB_ID_1D_dims_prod) ) { return; }
  int32_t const kHalf = %(local_size) / 2;
  int32_t const kLast = %(local_size) - 1;
  int32_t const pel = %(GLOB_ID_1D_img_nomod)*%(in_img_stride) + %(GLOB_ID_1D_y)*%(in_y_stride) + %(GLOB_ID_1D_x);
  int32_t const c_first = (int32_t)%(GLOB_ID_1D_cblk)*%(work_cblk_sz_dim);      // first channel this thread writes
  float const per_elem = %(alpha) / (float)%(local_size);
  float line[%(local_size)];
  for( int32_t i = 0; i <= kLast; ++i ) { line[i] = 0.0f; }
  float sumsq = 0.0f;
  for( int32_t step = 0; step != %(work_cblk_sz_dim) + 2*kHalf; ++step ) {
    int32_t const c_in = c_first - kHalf + step;          // channel entering the window (zero outside the tensor)
    int32_t const c_out = c_in - kHalf;                   // channel the window is centred on after this step
    float const entering = ( c_in >= 0 && c_in < %(in_chan_dim) ) ? in[pel + c_in*%(in_chan_stride)] : 0.0f;
    float const leaving = line[0];
    for( int32_t i = 0; i != kLast; ++i ) { line[i] = line[i+1]; }
    line[kLast] = entering;
    {
      // the carried sum loses digits if the compiler regroups it (CUCL sources build with fast-math): keep the written order
      #pragma clang fp reassociate(off) contract(off)
      sumsq = ( sumsq + entering*entering ) - leaving*leaving;
    }
    if( c_out >= c_first && c_out < %(in_chan_dim) ) {
#if BODAHIP_LRN_FASTPOW
      out[pel + c_out*%(in_chan_stride)] = line[kLast - kHalf] * __builtin_amdgcn_exp2f( -%(beta) * __builtin_amdgcn_logf( %(k) + sumsq*per_elem ) );   // (k > 0, alpha >= 0: the argument is positive)
#else
      out[pel + c_out*%(in_chan_stride)] = line[kLast - kHalf] * powf( %(k) + sumsq*per_elem, -%(beta) );
#endif
    }
  }
}
"""
_POOL_T = parse_template("pool", POOL_TEMPLATE)
_LRN_T = parse_template("lrn", LRN_TEMPLATE)
FWD_FUNCS = {"fwd_copy": ["in", "out", "n_in", "chw_in", "chw_out", "off_out"],
             "fwd_relu": ["inout", "n"]}
_TPB = 256
_u32 = lambda v: RtcArg.scalar(int(v), "uint32_t")
_f32 = lambda v: RtcArg.scalar(float(v), "float")


@dataclass
class FwdCall:
    tag: str
    rfc: RtcFuncCall
    func: str
    flops: int = 0
    call_id: int = -1


class ConvPipeFwd:
    """`has_conv_fwd_t` with mode=rtc over an rtc backend (src/has_conv_fwd.H:16-25, src/rtc_fwd.cc:43-577)."""
    mode = "rtc"

    def __init__(self, rtc: HipCompute, op_tune: Optional[OpTune] = None, per_call_fn: str = "", enable_double_run: bool = False, fuse_siblings: bool = True, fuse_levels: bool = True, fuse_pools: bool = True, sets_take_groups: bool = True,
                 spec_fwd: bool = True, fuse_pool_lrn="pool_first", fuse_k1_chains: bool = True, fuse_f32_pools: Optional[bool] = None, fuse_post: bool = True, filts_kmajor_once: bool = True):
        self.rtc, self.op_tune = rtc, op_tune or OpTune()
        # fp32 nets (round 6): a convolution whose plan reads its filters k-major gets that copy made once per set of weights (hip_conv_filts_kmajor, refresh_group_params())
        # instead of in front of every call -- seven launches less per NiN pass.  Bit-identical (the same transposition kernel, the same bytes); BODAHIP_FILTS_KM_ONCE=off: per call
        self.filts_kmajor_once = filts_kmajor_once and os.environ.get("BODAHIP_FILTS_KM_ONCE") != "off"
        self._km_params: List[RtcFuncCall] = []
        # channels-last bf16 nets (round 5): a convolution on the rolling-rows kernel (csrc/kernels/conv_nhwc_rows_bf16.hip: the 7x7/2 stems) takes the max pooling that alone
        # reads it, and the LRN that alone reads that, into its launch -- its rows are pooled out of an LDS ring, its own output (four times the pooled tensor) is never
        # written or read.  GoogLeNet: conv1 + pool1 + norm1 = one call.  Bit-identical to the calls run apart; the skipped nodes are materialised on demand.
        self.fuse_post = fuse_post and os.environ.get("BODAHIP_NO_NHWC_POST_FUSION") is None
        self.fused_post: Dict[str, Tuple[str, Optional[str]]] = {}     # convolution tag -> (pooling tag, LRN tag or None)
        self._lazy_pre: Dict[str, List[str]] = {}                        # lazy node -> the lazy nodes its call reads (materialised first)
        # fp32 nets: a max pooling read by one LDS-patch convolution is formed inside that convolution's patch loads (see init()).  Built and exact, but OFF by default: it
        # measured slower (same-box A/B, profiles/r05_probe_f32_pool_fusion.txt).  None: the env switch BODAHIP_F32_POOL_FUSION=1 decides
        self.fuse_f32_pools = (os.environ.get("BODAHIP_F32_POOL_FUSION") == "1") if fuse_f32_pools is None else bool(fuse_f32_pools)
        # fp32 nets: a 1x1 convolution whose output is read by ONE other 1x1 convolution only (NiN's cccp1 -> cccp2) runs with it as one hip_conv_k1_chain launch: the
        # intermediate tensor stays in the accumulator registers (kernels/k1_quad_f32.hip -DCHAIN=1), its write + read are gone.  Bit-identical; the first
        # convolution's node is materialised on demand.  The reference chains them through memory (src/rtc_fwd.cc:495-503)
        self.fuse_k1_chains = fuse_k1_chains
        self.k1_chains: List[Tuple[str, str]] = []     # (first conv tag, second conv tag) of each fused pair
        self.spec_fwd = spec_fwd     # channels-last nets: pool / LRN kernels specialised per geometry (False: the generic kernels with run-time geometry)
        # channels-last bf16 nets: convolutions that read the SAME node with the same kernel geometry (an inception module's 1x1 / 3x3-reduce / 5x5-reduce
        # convs) run as one hip_conv_nhwc_grp launch -- input read once, the members' tiles in one grid, two launches fewer per module; same bits
        self.fuse_siblings = fuse_siblings
        self.fuse_levels = fuse_levels       # channels-last nets: the independent convolutions that fill one Concat (an inception module's 3x3 / 5x5 / pool projection) as ONE hip_conv_nhwc_set launch
        self.level_sets: List[Tuple[str, ...]] = []
        self.sets_take_groups = sets_take_groups
        self.fuse_pools = fuse_pools         # channels-last nets: a stride-1 max pooling whose only reader is a 1x1 convolution is taken into that convolution (its input must be non-negative).
        # Measured on GoogLeNet at 64 images (tools/r4k.sh, r4l.sh): by itself it LOSES -- nine launches and 0.10 ms of pooling time go, but the window maximum (nine LDS
        # reads + 32 packed max per B fragment) makes the convolution a 12-24 us launch of its own at the module's first level: 78.9 -> 75.1 k img/s.  Together with
        # sets_take_groups -- the pool projection then shares ONE launch with the module's sibling group, both read the module's input -- it wins: 79.8 -> 83.0 k img/s
        self.fused_pools: Dict[str, str] = {}  # pooling tag -> the convolution that took it
        # channels-last nets: a max pooling and an across-channel LRN that follow each other (either order, the first one's output read by nothing else) run as ONE pass
        # over the tensor (nhwc.POOL_LRN_SPEC_SRC): each of them runs at the HBM roof by itself, the fused call saves the intermediate's write + read.  Bit-identical.
        # "pool_first" (default since round 4c): only Pooling -> LRN pairs -- a pure saving of the intermediate's write + read; True: LRN -> Pooling pairs too (their fused kernel evaluates the
        # LRN once per window position and is compute-bound: whether it still pays differs BY BOX -- same-box A/Bs on four MI355X boxes, AlexNet-net at 256 images with / without its two
        # LRN-first pairs fused: 309 / 325 k img/s, 328 / 319 k, 338 / 324 k, 296 / 311 k, 310 / 322 k; GoogLeNet-net 83.3 / 84.0, 82.9 / 81.4, 85.8 / 82.8, 81.0 / 81.6, 83.1 / 83.5 k); False: none
        self.fuse_pool_lrn = fuse_pool_lrn
        self.fused_pool_lrn: Dict[str, Tuple[str, bool]] = {}   # tag of the FIRST op of a pair -> (tag of the second, lrn_first)
        self.lds_pool_lrn = set()    # tags of the first op (an LRN) of the pairs that run through the LDS kernel (nhwc.LRN_POOL_LDS_SRC)
        self._lazy: Dict[str, FwdCall] = {}    # nodes no call of the forward pass writes any more (a fused pooling's output): the call that materialises one when it is asked for
        self.groups: List[Tuple[str, ...]] = []      # tags of the members of each fused call
        self.per_call_fn, self.enable_double_run = per_call_fn, enable_double_run
        self.fwd_calls: List[FwdCall] = []
        self.op_param_names: List[str] = []
        self.cp: Optional[ConvPipe] = None
        self.compute_dur_ms = float("nan")
        self._vars: List[str] = []
        self._funcs: List[str] = []
        self.concat_elim = True      # convs write straight into their channel range of a Concat output where legal
        # op_tune (hip_dtype=bf16, hip_layout=nhwc): EVERY node lives as a channels-last bf16 tensor (channels padded to a multiple of 8), convs
        # run hip_conv_nhwc, the other ops their channels-last kernels (boda_amd/nhwc.py); the caller's input is transposed by the first call of
        # the pass, filters once at init, outputs when they are fetched -- the reference keeps k1conv / tconv intermediates in the consumer's
        # transposed format the same way (src/rtc_fwd.cc:229-243,495-503)
        self.nhwc = (self.op_tune.hip_dtype == "bf16" and self.op_tune.hip_layout == "nhwc")
        self.in_var = ""             # the var that takes the caller's input (reference layout)

    # -- init: annotate, fuse, create vars, generate calls, upload params
    def init(self, cp: ConvPipe, op_params: Optional[Dict[str, np.ndarray]] = None, gen_mode: int = 5) -> None:
        rtc = self.rtc
        self.cp = cp
        try:    # the device's CU count for the planner questions asked below (a recording backend has none: the MI355X's 256)
            info = rtc.get_device_info()
            self._num_cus = int(info["num_cus"]) if isinstance(info, dict) and info.get("num_cus") else 256
        except Exception:
            self._num_cus = 256
        if not getattr(rtc, "_gen_data_compiled", False):
            rtc.compile(gd.func_infos()); rtc._gen_data_compiled = True
        if not getattr(rtc, "_fwd_funcs_compiled", False):
            infos = [RtcFuncInfo(fn, FWD_SRC if i == 0 else "", args, Op({"type": "fwd", "func_name": fn}, {})) for i, (fn, args) in enumerate(FWD_FUNCS.items())]
            rtc.compile(infos); rtc._fwd_funcs_compiled = True
        # ReLU fusion: a ReLU that directly follows a conv, in place on its output (src/rtc_fwd.cc:486-493)
        fused = set()
        has_relu: Dict[str, int] = {}
        for i, op in enumerate(cp.ops):
            if op.type == "Convolution":
                nxt = cp.ops[i + 1] if i + 1 < len(cp.ops) else None
                hr = int(nxt is not None and nxt.type == "ReLU" and nxt.in_place and nxt.bot == op.top)
                has_relu[op.tag] = hr
                if hr:
                    fused.add(nxt.tag)
        # Concat elimination: a Concat input that is produced by a convolution and read by nothing but the Concat is never
        # materialised -- the conv writes its channel range of the Concat output directly (hip_conv's out_chan_off), which removes
        # the channel-offset copy the reference makes per input (src/rtc_fwd.cc:267-280) and one tensor round trip through HBM
        self.slices: Dict[str, Tuple[str, int, int]] = {}      # node -> (concat output node, first channel, channels)
        if self.concat_elim:
            readers: Dict[str, int] = {}
            for o in cp.ops:
                if not o.in_place:
                    for b in (o.bots or (o.bot,)):
                        readers[b] = readers.get(b, 0) + 1
            conv_tops = {o.top for o in cp.ops if o.type == "Convolution"}
            # a top that an un-fused in-place op (a ReLU that is not the conv's immediate successor, an in-place LRN) still works on
            # must exist as a var of its own: it is not eliminated
            inplace_targets = {o.top for o in cp.ops if o.in_place and o.tag not in fused and o.type != "Dropout"}
            for o in cp.ops:
                if o.type != "Concat":
                    continue
                c_done = 0
                for b in o.bots:
                    ch = cp.nodes[b].dsz("chan")
                    if b in conv_tops and b not in inplace_targets and readers.get(b, 0) == 1 and o.bots.count(b) == 1:
                        self.slices[b] = (o.top, c_done, ch)
                    c_done += ch
        # vars: the source node, then one per op output (in-place ops and Dropout reuse their input var)
        alias: Dict[str, str] = {}
        def vn(node: str) -> str:
            return alias.get(node, node)
        if self.nhwc:
            from . import nhwc as _nhwc
            _nhwc.ensure_fwd_compiled(rtc)
            for o in cp.ops:
                if o.type == "Concat" and any(cp.nodes[b].dsz("chan") % 8 for b in o.bots):
                    raise UnsupErr(f"channels-last bf16 nets: Concat {o.tag} of inputs whose channel counts are not multiples of 8")
        # annotated convs up front (channels-last nets: a conv1-type layer that is the ONLY reader of the net's input runs space-to-depth, the
        # regrouping done by the input's layout pass; everywhere else intermediates are plain img:y:x:chan)
        import dataclasses
        annos: Dict[str, Op] = {}
        in_readers = [o for o in cp.ops if cp.in_node in (o.bots or (o.bot,))]
        for o in cp.ops:
            if o.type == "Convolution":
                s2d_ok = self.nhwc and len(in_readers) == 1 and in_readers[0] is o
                annos[o.tag] = add_codegen_annotations(cp.conv_op(o), dataclasses.replace(self.op_tune, hip_s2d=int(s2d_ok)) if self.nhwc else self.op_tune)
        in_anno = annos[in_readers[0].tag] if (self.nhwc and len(in_readers) == 1 and in_readers[0].type == "Convolution") else None
        # 1x1 -> 1x1 chains (fp32 nets; see fuse_k1_chains): first conv's output (after its fused ReLU) read by the second conv and nothing else, both plain hip_conv
        # functions without a tile of their own, the pair covered by the chain kernel, and a layer the streaming kernels are measured ahead on (long pel axis)
        chain_first: Dict[str, PipeOp] = {}    # tag of the second conv -> the first conv
        chain_lazy = set()                      # tags of first convs
        if (not self.nhwc) and self.fuse_k1_chains and os.environ.get("BODAHIP_NO_K1_CHAIN") is None:
            rd_k1: Dict[str, List[PipeOp]] = {}
            for o in cp.ops:
                if o.tag not in fused:
                    for b in (o.bots or (o.bot,)):
                        rd_k1.setdefault(b, []).append(o)
            busy_tops = {o.top for o in cp.ops if o.in_place and o.tag not in fused and o.type != "Dropout"}
            for a_ in cp.ops:
                rd = rd_k1.get(a_.top, [])
                if a_.type != "Convolution" or a_.tag in chain_first or len(rd) != 1 or rd[0].type != "Convolution" or rd[0].tag in chain_first or a_.top in self.slices or a_.top in busy_tops or a_.bot in busy_tops:
                    continue
                b_ = rd[0]; aa, ab = annos[a_.tag], annos[b_.tag]
                if aa.get_func_name() != "hip_conv" or ab.get_func_name() != "hip_conv" or "hip_tile" in aa.str_vals or "hip_tile" in ab.str_vals or not k1_chain_applies(aa, ab):
                    continue
                ga = aa.conv_geom()
                if self.fuse_k1_chains != "all" and (ga["OH"] * ga["OW"] < 512 or ga["B"] * ga["OH"] * ga["OW"] < 150000):   # ("all": every pair the kernel covers -- tests)
                    continue
                chain_first[b_.tag] = a_; chain_lazy.add(a_.tag); self.k1_chains.append((a_.tag, b_.tag))
        # pooling -> 1x1 convolution pairs (channels-last nets; an inception module's pool -> pool projection): the pooling is taken into the convolution where that is
        # legal -- max, stride 1, the convolution its only reader, and a NON-NEGATIVE input (the kernel pads the window with zeros and orders bf16 patterns as integers:
        # right exactly for values >= 0).  Non-negative nodes: outputs of a conv with fused ReLU / a ReLU / a pooling, LRN or Dropout of such / a Concat of such.
        conv_in: Dict[str, str] = {}    # convolution tag -> the node it reads instead of its bottom
        if self.nhwc and self.fuse_pools:
            # decided IN OP ORDER: a pooling is fusable when its input is non-negative at the pooling's own position (a node rectified by a LATER in-place ReLU does not
            # count), and every writer redefines its top's state (a convolution without ReLU makes it unknown again)
            state: Dict[str, bool] = {}
            pool_in_nonneg = set()      # tags of poolings whose input was non-negative when they read it
            for o in cp.ops:
                if o.type == "Pooling" and state.get(o.bot, False):
                    pool_in_nonneg.add(o.tag)
                if (o.type == "Convolution" and has_relu[o.tag]) or o.type == "ReLU":
                    nn = True
                elif o.type in ("Pooling", "Dropout"):
                    nn = state.get(o.bot, False)
                elif o.type == "LRN":
                    nn = state.get(o.bot, False) and o.lrn[3] > 0 and o.lrn[1] >= 0
                elif o.type == "Concat":
                    nn = all(state.get(b, False) for b in o.bots)
                else:
                    nn = False
                state[o.top] = nn
            op_index = {o.tag: i for i, o in enumerate(cp.ops)}
            n_readers: Dict[str, List[PipeOp]] = {}
            for o in cp.ops:
                if o.tag not in fused:
                    for b in (o.bots or (o.bot,)):
                        n_readers.setdefault(b, []).append(o)
            for o in cp.ops:
                rd = n_readers.get(o.top, [])
                if not (o.type == "Pooling" and not o.in_place and o.tag in pool_in_nonneg and len(rd) == 1 and rd[0].type == "Convolution" and _nhwc.multi_eligible(annos[rd[0].tag])):
                    continue
                q = rd[0]; pin = cp.nodes[o.bot]
                # the fused kernel reads the pooling's input at the CONVOLUTION's position: nobody may rewrite it in between
                if any(w.top == o.bot and w.tag not in fused for w in cp.ops[op_index[o.tag] + 1:op_index[q.tag]]):
                    continue
                if pin.dsz("chan") % 8 or not _nhwc.pool_fusable(cp.conv_op(q).conv_geom(), (pin.dsz("y"), pin.dsz("x")), o.kern_sz, o.stride, o.in_pad, bool(o.avg_pool)):
                    continue
                _nhwc.fuse_pool(annos[q.tag], pin, tuple(o.kern_sz), tuple(o.in_pad))
                self.fused_pools[o.tag] = q.tag; conv_in[q.tag] = o.bot
        # fp32 nets (round 5; the fusion clause of SURVEY section 8 F2 on the path config 4 runs): a max pooling whose only reader is a convolution that takes the LDS-patch
        # form is taken INTO that convolution -- a patch element becomes the window maximum, formed while the patch is staged (kernels/gemm_conv_f32.hip, PKH): the pooled
        # tensor's write + read and a launch are gone.  Exact (a true float maximum, the convolution's own fma chains): no condition on the values.  NiN: pool0 -> conv2,
        # pool2 -> conv3 (conv4's 64 x 64 tile keeps two K tiles in flight, which the fused form does not have: pool3 stays).  The pooling's node is materialised on demand.
        # MEASURED SLOWER, hence opt-in (fuse_f32_pools): every one of the convolution's OC tiles forms the window maxima of its patch again (9 loads per element, x OC/BI
        # tiles), which costs the convolutions more than the pooling launches took -- NiN-net at 128 images 113.0 -> 112.2 TF/s, AlexNet-net at 128 111.7 -> 110.3 TF/s
        if (not self.nhwc) and self.fuse_f32_pools and self.op_tune.hip_dtype == "":
            from .rtc import explain_plan
            rd_p: Dict[str, List[PipeOp]] = {}
            for o in cp.ops:
                if o.tag not in fused:
                    for b in (o.bots or (o.bot,)):
                        rd_p.setdefault(b, []).append(o)
            op_index = {o.tag: i for i, o in enumerate(cp.ops)}
            for o in cp.ops:
                rd = rd_p.get(o.top, [])
                if not (o.type == "Pooling" and not o.in_place and len(rd) == 1 and rd[0].type == "Convolution" and o.top not in self.slices):
                    continue
                q = rd[0]; qa = annos[q.tag]
                if q.tag in chain_first or q.tag in chain_lazy or not f32_pool_fusable(qa, cp.nodes[o.bot], tuple(o.kern_sz), tuple(o.stride), tuple(o.in_pad), bool(o.avg_pool)):
                    continue
                if any(w.top == o.bot and w.tag not in fused for w in cp.ops[op_index[o.tag] + 1:op_index[q.tag]]):
                    continue      # (the fused kernel reads the pooling's input at the convolution's position: nobody may rewrite it in between)
                plan0 = explain_plan(qa, getattr(self, "_num_cus", 256)).split()
                if "-DJ_MODE=7" not in plan0 or (("_p" in plan0[1]) and ("_big" not in plan0[1]) and os.environ.get("BODAHIP_F32_POOL_ALL") is None) or "-DRDEC=1" in plan0:
                    continue      # (not the patch form / a plan with several K tiles in flight; a staging-wave plan -- round 6 -- gives way to the fused form's own patch kernel)
                fa = qa.copy(); fuse_f32_pool(fa, cp.nodes[o.bot], tuple(o.kern_sz), tuple(o.stride))
                try:
                    explain_plan(fa, getattr(self, "_num_cus", 256))
                except UnsupErr:
                    continue
                annos[q.tag] = fa; self.fused_pools[o.tag] = q.tag; conv_in[q.tag] = o.bot
        # convolution -> pooling [-> LRN] inside the convolution's launch (channels-last bf16 nets): see fuse_post
        post_of: Dict[str, str] = {}       # tag of a pooling / LRN taken into a convolution -> that convolution's tag
        if self.nhwc and self.fuse_post and self.op_tune.hip_dtype == "bf16":
            from .rtc import explain_plan
            rd_q: Dict[str, List[PipeOp]] = {}
            for o in cp.ops:
                if o.tag not in fused:
                    for b in (o.bots or (o.bot,)):
                        rd_q.setdefault(b, []).append(o)
            for q in cp.ops:
                if not (q.type == "Convolution" and has_relu[q.tag] and q.top not in self.slices and q.tag not in self.fused_pools.values()):
                    continue
                rd = rd_q.get(q.top, [])
                if len(rd) != 1 or rd[0].type != "Pooling" or rd[0].in_place or rd[0].top in self.slices or rd[0].tag in self.fused_pools or q.top == cp.out_node():
                    continue
                pool = rd[0]; rd2 = rd_q.get(pool.top, [])
                lrn = rd2[0] if (len(rd2) == 1 and rd2[0].type == "LRN" and not rd2[0].in_place and pool.top != cp.out_node()) else None
                for with_lrn in ((lrn, None) if lrn is not None else (None,)):
                    lp = tuple(with_lrn.lrn) if with_lrn is not None else None
                    if not _nhwc.post_fusable(annos[q.tag], pool.kern_sz, pool.stride, pool.in_pad, bool(pool.avg_pool), lp):
                        continue
                    fa = annos[q.tag].copy(); fa.nda_vals["conv_has_relu"].v = (1,)
                    _nhwc.fuse_post(fa, cp.nodes[pool.top], tuple(pool.kern_sz), tuple(pool.stride), tuple(pool.in_pad), lp)
                    try:
                        explain_plan(fa, getattr(self, "_num_cus", 256))
                    except (UnsupErr, RtErr):
                        continue
                    self._post_plain = getattr(self, "_post_plain", {}); self._post_plain[q.tag] = annos[q.tag]
                    annos[q.tag] = fa; self.fused_post[q.tag] = (pool.tag, with_lrn.tag if with_lrn is not None else None)
                    post_of[pool.tag] = q.tag
                    if with_lrn is not None:
                        post_of[with_lrn.tag] = q.tag
                    break
        # pooling <-> LRN pairs (channels-last nets, specialised kernels): see fuse_pool_lrn
        pl_second: Dict[str, str] = {}     # tag of the second op of a pair -> tag of the first
        if self.nhwc and self.spec_fwd and self.fuse_pool_lrn:
            rd_all: Dict[str, List[PipeOp]] = {}
            for o in cp.ops:
                if o.tag not in fused:
                    for b in (o.bots or (o.bot,)):
                        rd_all.setdefault(b, []).append(o)
            for a in cp.ops:
                rd = rd_all.get(a.top, [])
                if a.tag in fused or a.tag in self.fused_pools or a.tag in pl_second or a.in_place or len(rd) != 1 or a.top in self.slices or a.tag in post_of:
                    continue
                b = rd[0]
                if b.in_place or b.tag in self.fused_pools or b.tag in post_of or {a.type, b.type} != {"Pooling", "LRN"}:
                    continue
                pool, lrn = (a, b) if a.type == "Pooling" else (b, a); lds_pair = False
                if a.type == "LRN" and self.fuse_pool_lrn == "pool_first":     # LRN first: through LDS (the LRN evaluated once per input chunk) where that kernel applies; the thread-per-
                    # output kernel evaluates it per window position -- compute-bound, level with or behind the two kernels -- and is only taken on request (fuse_pool_lrn=True)
                    # (a workgroup kernel; since round 5 the multi-device backend shards it by img like the per-element functions: it declares its group index)
                    if not (os.environ.get("BODAHIP_NO_LRN_POOL_LDS") is None and _nhwc.lrn_pool_lds_rows(_nhwc.nhwc_dims(cp.nodes[lrn.bot]), _nhwc.nhwc_dims(cp.nodes[pool.top]), pool.kern_sz, pool.stride)):
                        continue
                    lds_pair = True
                if pool.bot != cp.in_node and _nhwc.pool_lrn_fusable(_nhwc.nhwc_dims(cp.nodes[pool.bot]), _nhwc.nhwc_dims(cp.nodes[pool.top]), pool.kern_sz, pool.stride, pool.in_pad, int(pool.avg_pool), lrn.lrn[0], lrn.lrn[1], lrn.lrn[3]):
                    self.fused_pool_lrn[a.tag] = (b.tag, a.type == "LRN"); pl_second[b.tag] = a.tag
                    if lds_pair:
                        self.lds_pool_lrn.add(a.tag)
        # sibling convolutions (channels-last nets): same bottom node, same kernel / stride / padding / fused ReLU, plain hip_conv_nhwc members
        group_of: Dict[str, List[PipeOp]] = {}     # tag of a member -> its group (list of ops, definition order)
        if self.nhwc and self.fuse_siblings:
            by_key: Dict[tuple, List[PipeOp]] = {}
            for o in cp.ops:
                if o.type == "Convolution" and not annos[o.tag].has("nhwc_s2d") and not annos[o.tag].get_dims("filts").has("in_grp"):
                    by_key.setdefault((o.bot, tuple(o.kern_sz), tuple(o.stride), tuple(o.in_pad), has_relu[o.tag]), []).append(o)
            for members0 in by_key.values():
                for members in sibling_runs(cp.ops, members0, fused):
                    for k in range(0, len(members), 4):
                        grp = members[k:k + 4]
                        if len(grp) >= 2:
                            for o in grp:
                                group_of[o.tag] = grp
                            self.groups.append(tuple(o.tag for o in grp))
        grp_done = set()
        def vd(node: str) -> Dims:   # dims a node's var is created with
            if not self.nhwc:
                return cp.nodes[node]
            if node == cp.in_node and in_anno is not None:
                return in_anno.get_dims("in")
            return _nhwc.nhwc_dims(cp.nodes[node])
        self._vd = vd
        rtc.create_var_with_dims(cp.in_node, vd(cp.in_node)); self._vars.append(cp.in_node)
        self.in_var = cp.in_node
        if self.nhwc:   # the caller's input arrives in the reference layout; the first call of every pass transposes it
            self.in_var = cp.in_node + "_ref"
            rtc.create_var_with_dims(self.in_var, cp.nodes[cp.in_node]); self._vars.append(self.in_var)
            self.fwd_calls.append(FwdCall("xpose_" + cp.in_node, _nhwc.xpose_call("in", self.in_var, cp.in_node, cp.nodes[cp.in_node], vd(cp.in_node), in_anno, rtc if self.spec_fwd else None), "nhwc_xpose_in"))
        pdims = lambda pn: annos[pn[:-len("_filts")]].get_dims("filts") if (self.nhwc and pn.endswith("_filts")) else cp.params[pn]   # (as stored)
        for pn, pd in cp.params.items():
            rtc.create_var_with_dims(pn, pdims(pn)); self._vars.append(pn); self.op_param_names.append(pn)
        made = set()
        for cat in sorted({t for (t, _, _) in self.slices.values()}):   # Concat outputs that convs write into exist before those convs
            rtc.create_var_with_dims(cat, vd(cat)); self._vars.append(cat); made.add(cat)
        for op in cp.ops:
            if op.tag in fused:
                continue
            if op.type == "Dropout":           # no-op at inference (src/caffepb.cc:233-238): the top IS the bottom
                if not op.in_place:
                    alias[op.top] = vn(op.bot)
                continue
            if not op.in_place and op.top not in made and op.top not in self.slices:
                rtc.create_var_with_dims(op.top, vd(op.top)); self._vars.append(op.top); made.add(op.top)
            if op.type == "Convolution" and op.tag in group_of:
                grp = group_of[op.tag]
                if grp[0].tag in grp_done:
                    continue                 # (emitted with the group's first member)
                grp_done.add(grp[0].tag)
                for o in grp:                # every member's output var exists before the fused call
                    annos[o.tag].nda_vals["conv_has_relu"].v = (has_relu[o.tag],)
                    if o.top not in made and o.top not in self.slices and o.top not in self._vars:
                        rtc.create_var_with_dims(o.top, vd(o.top)); self._vars.append(o.top); made.add(o.top)
                ganno = _nhwc.annotate_group([annos[o.tag] for o in grp])
                gname = "+".join(o.tag for o in grp); gen_fn = f"{_nhwc.GRP_FUNC}__{cp.name}_{grp[0].tag}"
                annos[gname] = ganno          # (a sibling group may in turn join a level set)
                rtc.compile([RtcFuncInfo(gen_fn, "", _nhwc.group_arg_names(len(grp)), ganno)]); self._funcs.append(gen_fn)
                fv, bv = f"{gen_fn}_filts", f"{gen_fn}_biases"
                rtc.create_var_with_dims(fv, ganno.get_dims("filts")); rtc.create_var_with_dims(bv, ganno.get_dims("biases")); self._vars += [fv, bv]
                self._grp_params = getattr(self, "_grp_params", []) + [(fv, bv, ganno.get_dims("grp"), [o.tag for o in grp])]
                am = {"filts": RtcArg.var(fv), "biases": RtcArg.var(bv), "in": RtcArg.var(vn(op.bot)), "stride": RtcArg.ref(ganno.get_dims("stride")),
                      "in_pad": RtcArg.ref(ganno.get_dims("in_pad")), "grp": RtcArg.ref(ganno.get_dims("grp"))}
                for m, o in enumerate(grp):
                    if o.top in self.slices:
                        cat, c_off, _ = self.slices[o.top]
                        am[f"out_{m}"] = RtcArg.var(cat); am[f"out_chan_off_{m}"] = _u32(c_off)
                    else:
                        am[f"out_{m}"] = RtcArg.var(o.top)
                        if vd(o.top).dsz("chan") != o.out_chans:
                            am[f"out_chan_off_{m}"] = _u32(0)
                self.fwd_calls.append(FwdCall(gname, RtcFuncCall(gen_fn, am), _nhwc.GRP_FUNC, sum(cp.conv_op(o).flops() for o in grp)))
                continue
            if op.type == "Convolution":
                cop = cp.conv_op(op)
                anno = annos[op.tag]
                anno.nda_vals["conv_has_relu"].v = (has_relu[op.tag],)
                fn = anno.get_func_name(); gen_fn = f"{fn}__{cp.name}_{op.tag}"
                rtc.compile([RtcFuncInfo(gen_fn, "", [a for a, _ in NATIVE_ARGS[fn]], anno)]); self._funcs.append(gen_fn)
                am = {"filts": RtcArg.var(op.tag + "_filts"), "biases": RtcArg.var(op.tag + "_biases"), "in": RtcArg.var(vn(conv_in.get(op.tag, op.bot))),
                      "stride": RtcArg.ref(anno.get_dims("stride")), "in_pad": RtcArg.ref(anno.get_dims("in_pad")), "out": RtcArg.var(op.top)}
                if op.top in self.slices:
                    cat, c_off, _ = self.slices[op.top]
                    am["out"] = RtcArg.var(cat); am["out_chan_off"] = _u32(c_off)
                elif self.nhwc and vd(op.top).dsz("chan") != op.out_chans:
                    am["out_chan_off"] = _u32(0)     # (the var carries zero pad channels: the conv writes the first out_chans of each row)
                if op.tag in chain_lazy:         # first conv of a 1x1 chain: no call of the pass writes its node; this call materialises it when somebody asks
                    self._lazy[op.top] = FwdCall(op.tag, RtcFuncCall(gen_fn, am), fn, cop.flops())
                    continue
                if op.tag in self.fused_post:    # conv + pooling [+ LRN] as one call: it writes the LAST op's node; the convolution's own node is materialised (by the plain function) on demand
                    ptag, ltag = self.fused_post[op.tag]; last = next(o for o in cp.ops if o.tag == (ltag or ptag))
                    if last.top not in made and last.top not in self._vars:
                        rtc.create_var_with_dims(last.top, vd(last.top)); self._vars.append(last.top); made.add(last.top)
                    pam = dict(am); am["out"] = RtcArg.var(last.top)
                    if vd(last.top).dsz("chan") != op.out_chans:
                        am["out_chan_off"] = _u32(0)
                    plain = self._post_plain[op.tag]; plain.nda_vals["conv_has_relu"].v = (has_relu[op.tag],)
                    pfn = gen_fn + "_plain"
                    rtc.compile([RtcFuncInfo(pfn, "", [a for a, _ in NATIVE_ARGS[fn]], plain)]); self._funcs.append(pfn)
                    self._lazy[op.top] = FwdCall(op.tag, RtcFuncCall(pfn, pam), fn, cop.flops())
                    ftag = "+".join(t for t in (op.tag, ptag, ltag) if t); annos[ftag] = anno
                    self.fwd_calls.append(FwdCall(ftag, RtcFuncCall(gen_fn, am), fn, cop.flops()))
                    continue
                if op.tag in chain_first:        # second conv of the chain: ONE call from the first conv's input to this conv's output
                    first = chain_first[op.tag]; fa = annos[first.tag]
                    canno = annotate_k1_chain(fa, anno, has_relu[first.tag], has_relu[op.tag]); annos[first.tag + "+" + op.tag] = canno
                    cfn = f"{K1_CHAIN_FUNC}__{cp.name}_{first.tag}"
                    rtc.compile([RtcFuncInfo(cfn, "", [a for a, _ in NATIVE_ARGS[K1_CHAIN_FUNC]], canno)]); self._funcs.append(cfn)
                    cam = {"filts": RtcArg.var(first.tag + "_filts"), "biases": RtcArg.var(first.tag + "_biases"), "filts2": am["filts"], "biases2": am["biases"],
                           "in": RtcArg.var(vn(first.bot)), "stride": RtcArg.ref(fa.get_dims("stride")), "in_pad": RtcArg.ref(fa.get_dims("in_pad")), "out": am["out"]}
                    if "out_chan_off" in am:
                        cam["out_chan_off"] = am["out_chan_off"]
                    self.fwd_calls.append(FwdCall(first.tag + "+" + op.tag, RtcFuncCall(cfn, cam), K1_CHAIN_FUNC, cp.conv_op(first).flops() + cop.flops()))
                    continue
                if fn == "hip_conv" and self.filts_kmajor_once and self._plan_reads_filts_kmajor(anno):
                    # the plan reads its filters k-major (the staging-wave kernel, csrc/kernels/conv_big_f32.hip): the net holds that copy -- made ONCE, below in
                    # refresh_group_params(), as the reference transposes its filters at set-up (xpose_filts, src/rtc_fwd.cc:229-243) -- and the call skips its own pass
                    fd = anno.get_dims("filts"); kmv = op.tag + "_filts_km"
                    kmd = Dims(("k", "out_chan"), (fd.dsz("in_chan") * fd.dsz("y") * fd.dsz("x") + 128, (fd.dsz("out_chan") + 3) // 4 * 4), "float")
                    rtc.create_var_with_dims(kmv, kmd); self._vars.append(kmv)
                    xfn = f"{FILTS_KMAJOR_FUNC}__{cp.name}_{op.tag}"
                    rtc.compile([RtcFuncInfo(xfn, "", [a for a, _ in NATIVE_ARGS[FILTS_KMAJOR_FUNC]], Op({"type": "Convolution", "func_name": FILTS_KMAJOR_FUNC}, {}))]); self._funcs.append(xfn)
                    self._km_params.append(RtcFuncCall(xfn, {"filts": am["filts"], "filts_km": RtcArg.var(kmv)}))
                    am["filts_km"] = RtcArg.var(kmv)
                self.fwd_calls.append(FwdCall(op.tag, RtcFuncCall(gen_fn, am), fn, cop.flops()))
            elif self.nhwc and op.tag in post_of:                   # taken into its convolution's launch (fuse_post): the pooling's node on demand (from the convolution's, on demand too); the LRN's is what that launch writes
                if op.type == "Pooling":
                    self._lazy[op.top] = FwdCall(op.tag, _nhwc.pool_call(vn(op.bot), op.top, vd(op.bot), vd(op.top), op.kern_sz, op.stride, op.in_pad, int(op.avg_pool), rtc if self.spec_fwd else None), "nhwc_pool")
                    self._lazy_pre[op.top] = [op.bot]
            elif self.nhwc and op.tag in self.fused_pool_lrn:       # first op of a pooling <-> LRN pair: its node is only materialised when somebody asks for it
                if op.type == "Pooling":
                    self._lazy[op.top] = FwdCall(op.tag, _nhwc.pool_call(vn(op.bot), op.top, vd(op.bot), vd(op.top), op.kern_sz, op.stride, op.in_pad, int(op.avg_pool), rtc), "nhwc_pool")
                else:
                    self._lazy[op.top] = FwdCall(op.tag, _nhwc.lrn_call(vn(op.bot), op.top, vd(op.bot), *op.lrn, rtc=rtc), "nhwc_lrn")
            elif self.nhwc and op.tag in pl_second:                 # second op of the pair: ONE call from the first op's input to this op's output
                first = next(o for o in cp.ops if o.tag == pl_second[op.tag]); pool, lrn = (first, op) if first.type == "Pooling" else (op, first)
                if first.tag in self.lds_pool_lrn:
                    self.fwd_calls.append(FwdCall(first.tag + "+" + op.tag, _nhwc.lrn_pool_lds_call(vn(first.bot), op.top, vd(first.bot), vd(op.top), pool.kern_sz, pool.stride, pool.in_pad,
                                                                                                   *lrn.lrn, rtc=rtc), "nhwc_lrn_pool_lds"))
                    continue
                self.fwd_calls.append(FwdCall(first.tag + "+" + op.tag, _nhwc.pool_lrn_call(vn(first.bot), op.top, vd(first.bot), vd(op.top), pool.kern_sz, pool.stride, pool.in_pad,
                                                                                       *lrn.lrn, lrn_first=(first.type == "LRN"), rtc=rtc), "nhwc_pool_lrn"))
            elif self.nhwc and op.type == "Pooling" and op.tag in self.fused_pools:     # taken into its convolution: only materialised when somebody asks for the node
                self._lazy[op.top] = FwdCall(op.tag, _nhwc.pool_call(vn(op.bot), op.top, vd(op.bot), vd(op.top), op.kern_sz, op.stride, op.in_pad, int(op.avg_pool),
                                                                      rtc if self.spec_fwd else None), "nhwc_pool")
            elif self.nhwc and op.type == "Pooling":
                self.fwd_calls.append(FwdCall(op.tag, _nhwc.pool_call(vn(op.bot), op.top, vd(op.bot), vd(op.top), op.kern_sz, op.stride, op.in_pad, int(op.avg_pool),
                                                                      rtc if self.spec_fwd else None), "nhwc_pool"))
            elif self.nhwc and op.type == "LRN":
                if op.in_place:   # the channels-last kernel reloads halo chunks of its INPUT at wave edges: in == out would read what neighbouring waves already stored
                    raise UnsupErr(f"channels-last LRN {op.tag} in place is not supported (the kernel reads neighbouring chunks of its input)")
                self.fwd_calls.append(FwdCall(op.tag, _nhwc.lrn_call(vn(op.bot), op.top, vd(op.bot), *op.lrn, rtc=rtc if self.spec_fwd else None), "nhwc_lrn"))
            elif self.nhwc and op.type == "ReLU":
                self.fwd_calls.append(FwdCall(op.tag, _nhwc.relu_call(vn(op.bot), vd(op.top)), "nhwc_relu"))
            elif self.nhwc and op.type == "Concat":
                c_done = 0
                for bi, b in enumerate(op.bots):
                    if b not in self.slices:        # (else: already written in place by its conv)
                        self.fwd_calls.append(FwdCall(f"{op.tag}.{bi}", _nhwc.copy_call(vn(b), op.top, vd(b), vd(op.top), c_done), "nhwc_copy"))
                    c_done += cp.nodes[b].dsz("chan")
            elif op.type == "Pooling":
                i, o = cp.nodes[op.bot], cp.nodes[op.top]
                none = lambda yx: Nda(Dims(("y", "x"), tuple(yx), "none"), "none")
                pop = Op({"type": "Pooling", "func_name": "pool"}, {"in": Nda(i), "out": Nda(o), "kern_sz": none(op.kern_sz), "stride": none(op.stride),
                                                                    "in_pad": none(op.in_pad), "avg_pool": Nda(None, "uint32_t", (int(op.avg_pool),))})
                H, W, OH, OW = i.dsz("y"), i.dsz("x"), o.dsz("y"), o.dsz("x")
                nonempty = all(min(H, oy * op.stride[0] - op.in_pad[0] + op.kern_sz[0]) > max(0, oy * op.stride[0] - op.in_pad[0]) for oy in (0, OH - 1)) and \
                    all(min(W, ox * op.stride[1] - op.in_pad[1] + op.kern_sz[1]) > max(0, ox * op.stride[1] - op.in_pad[1]) for ox in (0, OW - 1))
                uncond = int(self.spec_fwd and nonempty and op.kern_sz[0] * op.kern_sz[1] <= 64)     # taps as independent, unconditional loads (see the template)
                sig = pop.to_str() + f"|uncond={uncond}"
                cache = rtc.__dict__.setdefault("_pool_funcs", {})      # one generated function per distinct signature (rtc_func_sigs_map_t)
                if sig not in cache:
                    inst = instantiate(_POOL_T, pop, f"fwd_pool__{len(cache)}")
                    inst.src = f"#define BODAHIP_POOL_UNCOND {uncond}\n" + inst.src
                    rtc.compile([RtcFuncInfo(inst.func_name, inst.src, inst.arg_names, pop)])
                    cache[sig] = inst
                inst = cache[sig]
                am = {"in": RtcArg.var(vn(op.bot)), "out": RtcArg.var(op.top), "avg_pool": _u32(op.avg_pool), "kern_sz": RtcArg.ref(pop.get_dims("kern_sz")),
                      "stride": RtcArg.ref(pop.get_dims("stride")), "in_pad": RtcArg.ref(pop.get_dims("in_pad"))}
                pcall = FwdCall(op.tag, RtcFuncCall(inst.func_name, am, tpb=inst.tpb, blks=inst.blks), "fwd_pool")
                if op.tag in self.fused_pools:     # taken into its convolution (fp32 nets, round 5): only materialised when somebody asks for the node
                    self._lazy[op.top] = pcall
                else:
                    self.fwd_calls.append(pcall)
            elif op.type == "Concat":
                o = cp.nodes[op.top]; hw = o.dsz("y") * o.dsz("x"); chw_out = o.dsz("chan") * hw
                c_done = 0
                for bi, b in enumerate(op.bots):
                    d = cp.nodes[b]; n = d.dims_prod(); chw_in = d.dsz("chan") * hw
                    if b in self.slices:        # already written in place by its conv
                        c_done += d.dsz("chan"); continue
                    am = {"in": RtcArg.var(vn(b)), "out": RtcArg.var(op.top), "n_in": _u32(n), "chw_in": _u32(chw_in), "chw_out": _u32(chw_out), "off_out": _u32(c_done * hw)}
                    self.fwd_calls.append(FwdCall(f"{op.tag}.{bi}", RtcFuncCall("fwd_copy", am, tpb=_TPB, blks=(n + _TPB - 1) // _TPB), "fwd_copy"))
                    c_done += d.dsz("chan")
            elif op.type == "ReLU":
                n = cp.nodes[op.top].dims_prod()
                self.fwd_calls.append(FwdCall(op.tag, RtcFuncCall("fwd_relu", {"inout": RtcArg.var(vn(op.bot)), "n": _u32(n)}, tpb=_TPB, blks=(n + _TPB - 1) // _TPB), "fwd_relu"))
            elif op.type == "LRN":
                d = cp.nodes[op.bot]
                ls, alpha, beta, k = op.lrn
                f32 = lambda v: Nda(None, "float", (float(v),))
                cblk_sz = 32
                work = Dims(("img", "cblk", "cblk_sz", "y", "x"), (d.dsz("img"), -(-d.dsz("chan") // cblk_sz), cblk_sz, d.dsz("y"), d.dsz("x")), "none")
                lop = Op({"type": "LRN", "func_name": "lrn"}, {"in": Nda(d), "out": Nda(d), "alpha": f32(alpha), "beta": f32(beta), "k": f32(k),
                                                               "local_size": Nda(None, "uint32_t", (int(ls),)), "work": Nda(work, "none")})
                fastpow = int(self.spec_fwd and k > 0.0 and alpha >= 0.0)      # x^-beta as exp2(-beta log2 x): x = k + alpha/n * sum of squares > 0
                sig = lop.to_str() + f"|fastpow={fastpow}"
                cache = rtc.__dict__.setdefault("_pool_funcs", {})
                if sig not in cache:
                    inst = instantiate(_LRN_T, lop, f"fwd_lrn__{len(cache)}")
                    inst.src = f"#define BODAHIP_LRN_FASTPOW {fastpow}\n" + inst.src
                    rtc.compile([RtcFuncInfo(inst.func_name, inst.src, inst.arg_names, lop)])
                    cache[sig] = inst
                inst = cache[sig]
                am = {"in": RtcArg.var(vn(op.bot)), "out": RtcArg.var(op.top), "alpha": _f32(alpha), "beta": _f32(beta), "k": _f32(k), "local_size": _u32(ls),
                      "work": RtcArg.ref(work)}
                self.fwd_calls.append(FwdCall(op.tag, RtcFuncCall(inst.func_name, am, tpb=inst.tpb, blks=inst.blks), "fwd_lrn"))
        self._alias = alias
        self._annos = annos
        if self.nhwc and self.fuse_levels:
            self._fuse_level_sets(cp)
        # params: given arrays (copy_ndas_to_vars, src/rtc_fwd.cc:524) or the deterministic on-device pattern
        for pn in self.op_param_names:
            dst = pn
            if self.nhwc and pn.endswith("_filts"):   # filters: uploaded / generated in the reference layout, transposed once (src/rtc_fwd.cc:229-243 does the same at init)
                dst = pn + "_ref"; rtc.create_var_with_dims(dst, cp.params[pn])
            if op_params is not None and pn in op_params:
                rtc.copy_nda_to_var(dst, op_params[pn])
            else:
                arg = "filts" if pn.endswith("_filts") else "biases"
                rtc.run(gd.gen_call("Convolution", arg, dst, cp.params[pn], gen_mode, 0.0))
            if dst != pn:
                rtc.run(_nhwc.xpose_call("filts", dst, pn, cp.params[pn], pdims(pn), annos[pn[:-len("_filts")]])); rtc.finish_and_sync(); rtc.release_var(dst)
        rtc.finish_and_sync()
        self.refresh_group_params()
        rtc.release_per_call_id_data()

    def _fuse_level_sets(self, cp: "ConvPipe") -> None:
        """Channels-last nets: the calls of a forward pass are levelled by their true dependencies (level = 1 + the deepest call it must run after: read-after-write,
        write-after-read and write-after-write hazards between the calls' vars, `_call_deps`), and the plain convolutions of one level -- an inception module's 3x3,
        5x5 and pool-projection convs; an auxiliary classifier's conv beside the trunk's -- become ONE hip_conv_nhwc_set launch, every member on its own specialised
        kernel code.  As separate launches they are 100-200 tiles each on 256 CUs and a dependency-wired hipGraph does not overlap them (a cross-branch edge costs
        about what such a kernel takes; measured: GoogLeNet at 64 images, chain 69.9 k img/s, dependency-wired 67.6 k, with level sets 77.4 k).  The call list is
        re-ordered level by level (a valid schedule: every hazard is a dependency), calls keeping their relative order inside a level."""
        from . import nhwc as _nhwc
        rtc = self.rtc
        deps = self._call_deps()
        level = [0] * len(self.fwd_calls)
        for i, ds in enumerate(deps):
            level[i] = 1 + max((level[d] for d in ds), default=-1)
        by_level: Dict[int, List[int]] = {}
        for i, l in enumerate(level):
            by_level.setdefault(l, []).append(i)
        new_calls: List[FwdCall] = []
        flops_of = {o.tag: cp.conv_op(o).flops() for o in cp.ops if o.type == "Convolution"}
        for l in sorted(by_level):
            idxs = by_level[l]
            def joins(i: int) -> bool:   # a plain convolution (round 4: one whose own plan does not slice K -- unsliced in a set GoogLeNet's 4x4 auxiliary-head conv,
                c = self.fwd_calls[i]    # K = 2048 on 8 tiles, set the set's pace: 44 us against 12 sliced on its own)
                if c.func == _nhwc.GRP_FUNC:
                    return self.sets_take_groups   # (a sibling group as a member: the GROUPS form of the implicit-GEMM kernel inside the wrapper)
                if c.func != _nhwc.FUNC or not _nhwc.set_eligible(self._annos[c.tag]):
                    return False
                if os.environ.get("BODAHIP_NHWC_SPLITK2") is None:
                    return True     # (round 5: K slices are reduced INSIDE the launch -- a sliced member is one kernel like any other and brings its slices along)
                from .rtc import explain_plan   # the two-kernel form of the slices (slabs in the shared scratch + a reduce pass): such a member stays a call of its own
                return "_s" not in explain_plan(self._annos[c.tag], getattr(self, "_num_cus", 256)).split()[1]
            elig = [i for i in idxs if joins(i)]
            out_tn = {(self._annos[self.fwd_calls[i].tag].get_dims("out_0") if self.fwd_calls[i].func == _nhwc.GRP_FUNC else self._annos[self.fwd_calls[i].tag].get_dims("out")).tn for i in elig}
            if len(elig) < 2 or len(out_tn) != 1:
                new_calls += [self.fwd_calls[i] for i in idxs]; continue
            new_calls += [self.fwd_calls[i] for i in idxs if i not in elig]
            for k in range(0, len(elig), 16):
                part = elig[k:k + 16]
                if len(part) < 2:
                    new_calls += [self.fwd_calls[i] for i in part]; continue
                members = [self.fwd_calls[i] for i in part]
                sanno = _nhwc.annotate_set([self._annos[c.tag] for c in members])
                gen_fn = f"{_nhwc.SET_FUNC}__{cp.name}_{members[0].tag}"
                rtc.compile([RtcFuncInfo(gen_fn, "", _nhwc.multi_arg_names(len(members)), sanno)]); self._funcs.append(gen_fn)
                am = {"multi": RtcArg.ref(sanno.get_dims("multi"))}
                for m, c in enumerate(members):
                    for an, v in c.rfc.arg_map.items():
                        am[f"{an}_{m}"] = v
                new_calls.append(FwdCall("+".join(c.tag for c in members), RtcFuncCall(gen_fn, am), _nhwc.SET_FUNC, sum(flops_of[t] for c in members for t in c.tag.split("+"))))
                self.level_sets.append(tuple(c.tag for c in members))
        self.fwd_calls = new_calls

    def refresh_group_params(self) -> None:
        """Stacked filters / biases of the fused sibling convolutions, from the members' own (already transposed) params: init time only (and again after a
        caller overwrote params, e.g. a weight broadcast)."""
        rtc = self.rtc
        for call in getattr(self, "_km_params", []):     # k-major filter copies of the staging-wave convolutions (filts_kmajor_once)
            rtc.run(call)
        for fv, bv, grp, tags in getattr(self, "_grp_params", []):
            from . import nhwc as _nhwc
            offs = _nhwc.group_row_offsets(grp)
            fd, bd = rtc.get_var_dims(fv), rtc.get_var_dims(bv)
            F = np.zeros(fd.sizes, dtype=np.uint16); Bv = np.zeros(bd.sizes, dtype=np.float32)
            for off, tag in zip(offs, tags):
                f = rtc.copy_var_to_nda(tag + "_filts"); b = rtc.copy_var_to_nda(tag + "_biases")
                F[off:off + f.shape[0]] = f; Bv[off:off + b.shape[0]] = b
            rtc.copy_nda_to_var(fv, F); rtc.copy_nda_to_var(bv, Bv)
        rtc.finish_and_sync()

    def _plan_reads_filts_kmajor(self, anno: Op) -> bool:
        from .rtc import explain_plan
        try:
            plan = explain_plan(anno, getattr(self, "_num_cus", 256)).split()
        except (UnsupErr, RtErr):
            return False
        return plan[0] == "bodahip_conv_big_f32" and "-DI_VW=0" in plan

    def var_of(self, node: str) -> str:
        return self._alias.get(node, node)

    # -- run_fwd: set inputs -> run all calls -> get outputs (src/rtc_fwd.cc:529-577)
    def run_fwd(self, to_set_vns: Sequence[str], fwd: Dict[str, np.ndarray], to_get_vns: Sequence[str]) -> None:
        rtc = self.rtc
        if self.enable_double_run:
            for c in self.fwd_calls:
                rtc.run(c.rfc)
        rtc.finish_and_sync()
        for v in to_set_vns:
            rtc.copy_nda_to_var(self.in_var if (self.nhwc and v == self.cp.in_node) else self.var_of(v), fwd[v])
        rtc.finish_and_sync()
        for c in self.fwd_calls:
            c.call_id = rtc.run(c.rfc)
        rtc.finish_and_sync()
        for v in to_get_vns:
            if v in self._lazy:      # (a fused pooling's output: no call of the pass writes it)
                self._materialise(v)
            if v in self.slices:     # a conv output that only exists as a channel range of its Concat output
                cat, c_off, ch = self.slices[v]
                fwd[v] = np.ascontiguousarray(self._fetch(cat)[:, c_off:c_off + ch])
            else:
                fwd[v] = self._fetch(v)
        self.compute_dur_ms = rtc.get_dur(self.fwd_calls[0].call_id, self.fwd_calls[-1].call_id) if self.fwd_calls else 0.0
        self.per_call_ms = [(c.tag, c.func, rtc.get_dur(c.call_id, c.call_id), c.flops) for c in self.fwd_calls]
        if self.per_call_fn:
            with open(self.per_call_fn, "w") as f:
                f.write(f"net.args.runtime={self.compute_dur_ms / 1000.0}\n")
                for tag, func, ms, _ in self.per_call_ms:
                    f.write(f"per_layer_time['{tag}']=per_layer_time.get('{tag}',0.0) + {ms / 1000.0} # {func} \n")
        rtc.release_per_call_id_data()

    def _materialise(self, v: str) -> None:
        """Run the call that writes a node no call of the pass writes any more (after the lazy nodes it reads)."""
        for d in self._lazy_pre.get(v, ()):
            if d in self._lazy:
                self._materialise(d)
        self.rtc.run(self._lazy[v].rfc); self.rtc.finish_and_sync()

    def _fetch(self, node: str) -> np.ndarray:
        """A node's value in the reference layout (img:chan:y:x float)."""
        rtc = self.rtc
        if not self.nhwc:
            return rtc.copy_var_to_nda(self.var_of(node))
        from . import nhwc as _nhwc
        ref = self.cp.nodes[node]; tmp = "__fetch_tmp"
        rtc.create_var_with_dims(tmp, ref)
        try:
            d = self._vd(node)
            if d.dsz("chan") != ref.dsz("chan"):   # padded channels: read through a channel count of the stored row
                raise UnsupErr(f"fetch of node {node!r} with padded channels is not supported")
            rtc.run(_nhwc.xpose_call("out", tmp, self.var_of(node), ref, d)); rtc.finish_and_sync()
            return rtc.copy_var_to_nda(tmp)
        finally:
            rtc.release_var(tmp)

    def run_fwd_device_only(self) -> float:
        """Run all calls once with inputs already resident (benchmarks); -> ms first-call-start to last-call-end."""
        rtc = self.rtc
        ids = [rtc.run(c.rfc) for c in self.fwd_calls]
        rtc.finish_and_sync()
        ms = rtc.get_dur(ids[0], ids[-1])
        self.per_call_ms = [(c.tag, c.func, rtc.get_dur(i, i), c.flops) for c, i in zip(self.fwd_calls, ids)]
        rtc.release_per_call_id_data()
        return ms

    # -- hipGraph form of run_fwd_device_only: the call list captured once, replayed with one host call per forward pass
    def capture_graph(self, parallel: bool = False) -> int:
        """Capture the forward call list into a hipGraph (after at least one ordinary run, so that every lazily built kernel and
        table exists); -> number of captured calls.  With parallel=True the graph gets the calls' true dependencies instead of
        the launch order (read-after-write, write-after-write and write-after-read hazards between the vars they read and
        write), so independent branches of the net -- inception modules -- may overlap on the GPU."""
        rtc = self.rtc
        if getattr(self, "_graph", None) is not None:
            rtc.graph_destroy(self._graph)
        rtc.finish_and_sync()
        rtc.graph_begin()
        for c in self.fwd_calls:
            rtc.run(c.rfc)
        if parallel:
            try:
                self.call_deps = self._call_deps()
            except Exception:
                rtc.graph_destroy(rtc.graph_end()[0])      # (leave no capture open behind a host-side error)
                raise
            self._graph = rtc.graph_end_deps(self.call_deps); n = len(self.fwd_calls)
        else:
            self._graph, n = rtc.graph_end()
        return n

    def _call_deps(self) -> List[List[int]]:
        """deps[i] = the earlier calls that call i must run after.  A call reads its `in` / `inout` vars and writes its `out` /
        `inout` vars (convs also read filts / biases, which nothing writes during a forward pass); Concat copies fill disjoint
        channel ranges of one var and are not ordered among themselves."""
        writers: Dict[str, List[int]] = {}   # var -> the call(s) that produced its current contents
        readers: Dict[str, List[int]] = {}   # var -> calls that read it since
        deps: List[List[int]] = []
        for i, c in enumerate(self.fwd_calls):
            am = c.rfc.arg_map
            rd = [am[a].n for a in am if (a in ("in", "inout") or (a.startswith("in_") and a[3:].isdigit())) and am[a].is_var()]   # (in_<m>: the members of a set)
            outs = [a for a in am if (a in ("out", "inout") or (a.startswith("out_") and not a.startswith("out_chan_off"))) and am[a].is_var()]
            wr = [am[a].n for a in outs]
            if c.func == "nhwc_xpose_in":     # (the layout pass of the net's input: reads <in>_ref, writes <in>)
                rd, wr = [am["in_ref"].n], [am["in"].n]
            slice_outs = {t for t, _, _ in getattr(self, "slices", {}).values()}   # (Concat outputs that convs write channel ranges of)
            part_of = {am[a].n: (c.func in ("fwd_copy", "nhwc_copy") or am[a].n in slice_outs) for a in outs}   # writers of disjoint channel ranges of one var: unordered among themselves
            d = set()
            for v in rd:
                d.update(writers.get(v, []))
            for v in wr:
                if not (part_of.get(v, False) and not readers.get(v)):
                    d.update(writers.get(v, []))
                d.update(readers.get(v, []))
            d.discard(i)
            for v in rd:
                readers.setdefault(v, []).append(i)
            for v in wr:
                if part_of.get(v, False) and not readers.get(v):
                    writers.setdefault(v, []).append(i)
                else:
                    writers[v] = [i]; readers[v] = []
            deps.append(sorted(d))
        return deps

    def run_graph(self) -> float:
        """One forward pass as one graph launch; -> ms of the whole replay."""
        rtc = self.rtc
        cid = rtc.graph_launch(self._graph)
        rtc.finish_and_sync()
        ms = rtc.get_dur(cid, cid)
        rtc.release_per_call_id_data()
        return ms

    def get_info_log(self) -> str:
        return "\n".join(f"{c.tag}: {c.func}" for c in self.fwd_calls)

    def release(self) -> None:
        rtc = self.rtc
        rtc.finish_and_sync()
        if getattr(self, "_graph", None) is not None:
            rtc.graph_destroy(self._graph); self._graph = None
        for f in self._funcs:
            rtc.release_func(f)
        for v in self._vars:
            rtc.release_var(v)
        self._funcs, self._vars, self.fwd_calls, self._grp_params, self.groups = [], [], [], [], []
        self._km_params = []
        self.fused_post, self._lazy_pre = {}, {}
        self.k1_chains, self._lazy, self.fused_pools, self.fused_pool_lrn, self.level_sets, self.lds_pool_lrn = [], {}, {}, {}, [], set()   # (a second init() starts from a clean slate)

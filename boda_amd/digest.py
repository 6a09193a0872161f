"""nda_digest_t and wisdom files (host-side harness logic, numpy only).

Restates the behaviour of
  * nda_digest_T<T>::{get_samp_strides,get_sis,set_from_nda,get_samp}   src/boda_base.cc:214-272
  * mrd_comp (tolerance scaled by sqrt(n/1000) for checksums of n>1000)  src/boda_base.cc:278-311
  * bwrite/bread of a digest                                             src/boda_base.cc:329-363
  * min_sig_mag_rel_diff                                                 src/boda_base.cc:140-154
  * wisdom text records                                                  src/op-tuner.cc:42-126
The sample offsets come from boost::random::mt19937 + uniform_int_distribution<uint64_t>
(third-party, not vendored in the reference tree): standard MT19937 and boost's bucket-rejection
integer draw, pinned by reproducing every stored sample vector of the reference's goldens.
"""
from __future__ import annotations
import math
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .op import Dims, Op, RtErr, parse_op

NDD_VER1 = 0xDADA0101
_PRIMES = (1, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29)
# std::hash<std::string>(var name) of libstdc++ as stored in the reference's goldens (src/rtc_prof.cc:306,335)
KNOWN_SEEDS = {"c": 10959529184379665549, "out": 470894893395316877}


class MT19937:
    """Standard 32-bit Mersenne Twister with init_genrand seeding (== boost::random::mt19937(seed))."""

    def __init__(self, seed: int):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.idx = mt, 624

    def next(self) -> int:
        if self.idx >= 624:
            mt = self.mt
            for i in range(624):
                y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if (y & 1) else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def _boost_uniform_u64(gen: MT19937, rng: int) -> int:
    """boost::random::uniform_int_distribution<uint64_t>(0, rng) on a 32-bit engine."""
    if rng == 0:
        return 0
    brange = 0xFFFFFFFF
    if rng == brange:
        return gen.next()
    if rng > brange:
        raise RtErr("digest: tensors with >= 2^32 elements are not supported")
    bucket = brange // (rng + 1)
    if brange % (rng + 1) == rng:
        bucket += 1
    while True:
        r = gen.next() // bucket
        if r <= rng:
            return r


def sample_plan(dims: Dims, seed: int) -> List[Tuple[int, int, int]]:
    """[(stride, offset, num_subsamps)] in the reference's order."""
    n = dims.dims_prod()
    strides = {p for p in _PRIMES if p <= n} | set(dims.strides) | {n}
    gen = MT19937(seed & 0xFFFFFFFF)
    out = []
    for stride in sorted(strides):
        if stride == 0 or stride > n:
            raise RtErr("digest: bad stride")
        num_offsets = (stride + 1).bit_length() - 1
        seen = set()
        for _ in range(num_offsets):
            off = _boost_uniform_u64(gen, stride - 1)
            if off in seen:
                continue
            seen.add(off)
            out.append((stride, off, (n - off) // stride))
    return out


@dataclass
class Digest:
    dims: Dims
    seed: int
    min_v: float
    max_v: float
    samps: np.ndarray  # float32
    self_cmp_mrd: float = 0.0

    # ---- construction from data (product-side; sequential fp32 sums via cumsum)
    @staticmethod
    def from_array(v: np.ndarray, dims: Dims, seed: int) -> "Digest":
        flat = np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
        if flat.size != dims.dims_prod():
            raise RtErr("digest: data size != dims")
        plan = sample_plan(dims, seed)
        samps = np.empty(len(plan), dtype=np.float32)
        for i, (stride, off, _) in enumerate(plan):
            sl = flat[off::stride]
            # strictly sequential fp32 accumulation, as the reference's scalar loop does
            samps[i] = np.cumsum(sl, dtype=np.float32)[-1] if sl.size else np.float32(0)
        return Digest(dims, seed, float(flat.min()), float(flat.max()), samps)

    # ---- binary form
    def to_bytes(self) -> bytes:
        def bstr(s: str) -> bytes:
            b = s.encode()
            return struct.pack("<I", len(b)) + b
        out = b"\x01" + bstr(self.dims.tn) + struct.pack("<Id", NDD_VER1, self.self_cmp_mrd)
        out += struct.pack("<I", len(self.dims.sizes))
        for sz, st, nm in zip(self.dims.sizes, self.dims.strides, self.dims.names):
            out += struct.pack("<II", sz, st) + bstr(nm)
        out += bstr(self.dims.tn) + struct.pack("<QB", self.dims.dims_prod(), 1)
        out += struct.pack("<Qff", self.seed, self.min_v, self.max_v)
        out += struct.pack("<I", len(self.samps)) + np.asarray(self.samps, dtype="<f4").tobytes()
        return out

    def to_hex(self) -> str:
        return self.to_bytes().hex().upper()

    @staticmethod
    def from_hex(h: str) -> "Digest":
        b = bytes.fromhex(h.strip())
        pos = 0

        def take(fmt):
            nonlocal pos
            vals = struct.unpack_from(fmt, b, pos)
            pos += struct.calcsize(fmt)
            return vals

        def tstr():
            nonlocal pos
            (n,) = take("<I")
            s = b[pos:pos + n].decode()
            pos += n
            return s

        (nonnull,) = take("<B")
        if nonnull != 1:
            raise RtErr("digest: null digest")
        tn = tstr()
        if tn != "float":
            raise RtErr(f"digest: only float digests are on this path (got {tn})")
        magic, mrd = take("<Id")
        if magic != NDD_VER1:
            raise RtErr("digest: bad magic")
        (nd,) = take("<I")
        names, sizes, strides = [], [], []
        for _ in range(nd):
            sz, st = take("<II")
            names.append(tstr()); sizes.append(sz); strides.append(st)
        tn2 = tstr()
        strides_sz, valid = take("<QB")
        dims = Dims(tuple(names), tuple(sizes), tn2)
        if tuple(strides) != dims.strides or strides_sz != dims.dims_prod() or not valid:
            raise RtErr("digest: padded/invalid strides are not supported")
        seed, mn, mx = take("<Qff")
        (ns,) = take("<I")
        samps = np.frombuffer(b, dtype="<f4", count=ns, offset=pos).copy()
        pos += 4 * ns
        if pos != len(b):
            raise RtErr("digest: trailing bytes")
        return Digest(dims, seed, mn, mx, samps, mrd)

    # ---- comparison
    def mrd_comp(self, o: "Digest", mrd: float) -> str:
        """'' if equal within tolerance, else a description (reference: non-empty string == failure)."""
        if self.dims != o.dims:
            return f"nda_digest dims mismatch: v1.dims={self.dims.pretty()} v2.dims={o.dims.pretty()}"
        if self.seed != o.seed:
            return f"nda_digest seed mismatch: v1.seed={self.seed} v2.seed={o.seed}"
        plan = sample_plan(self.dims, self.seed)
        if not (len(plan) == len(self.samps) == len(o.samps)):
            return f"nda_digest sample count mismatch: plan={len(plan)} v1={len(self.samps)} v2={len(o.samps)}"
        ret = []

        def chk(tag, v1, v2, tol):
            rd = min_sig_mag_rel_diff(1.0, float(v1), float(v2))
            if not (rd <= tol):  # NaN fails
                ret.append(f" [{tag}]: v1={v1} v2={v2}")
        chk("min_v", self.min_v, o.min_v, mrd)
        chk("max_v", self.max_v, o.max_v, mrd)
        for (stride, off, nsub), s1, s2 in zip(plan, self.samps, o.samps):
            adj = mrd * (math.sqrt(nsub / 1000.0) if nsub > 1000 else 1.0)
            chk(f"stride={stride},offset={off}", s1, s2, adj)
        return "\n".join(ret)

    def worst_scaled_rd(self, o: "Digest") -> float:
        """max over (min,max,samples) of rel-diff / tolerance-scale: must be < mrd to pass."""
        plan = sample_plan(self.dims, self.seed)
        w = max(min_sig_mag_rel_diff(1.0, self.min_v, o.min_v), min_sig_mag_rel_diff(1.0, self.max_v, o.max_v))
        for (stride, off, nsub), s1, s2 in zip(plan, self.samps, o.samps):
            sc = math.sqrt(nsub / 1000.0) if nsub > 1000 else 1.0
            w = max(w, min_sig_mag_rel_diff(1.0, float(s1), float(s2)) / sc)
        return w


def min_sig_mag_rel_diff(min_sig_mag: float, v1: float, v2: float) -> float:
    amax = max(min_sig_mag, abs(v1), abs(v2))
    return abs(v2 - v1) / amax


@dataclass
class SsdsDiff:
    """ssds_diff_t (src/boda_base.cc:156-206): o2 - o1 statistics."""
    num_diff: int
    ssds: float
    sds: float
    mad: float
    mrd: float
    avg1: float
    avg2: float
    sz: int

    @staticmethod
    def of(o1: np.ndarray, o2: np.ndarray) -> "SsdsDiff":
        a = np.asarray(o1, dtype=np.float64).reshape(-1)
        b = np.asarray(o2, dtype=np.float64).reshape(-1)
        if a.size != b.size:
            raise RtErr("ssds_diff: size mismatch")
        d = b - a
        amax = np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))
        rd = np.abs(d) / amax
        with np.errstate(invalid="ignore"):
            mrd = float(np.max(rd)) if a.size else 0.0
            if np.isnan(rd).any():
                mrd = float("nan")
        return SsdsDiff(int(np.count_nonzero(a != b)), float(np.dot(d, d)), float(d.sum()),
                        float(np.max(np.abs(d))) if a.size else 0.0, mrd,
                        float(a.mean()) if a.size else 0.0, float(b.mean()) if a.size else 0.0, a.size)

    def has_nan(self) -> bool:
        return math.isnan(self.ssds) or math.isnan(self.sds) or math.isnan(self.mad) or math.isnan(self.mrd)

    def basic_str(self) -> str:
        aad = math.sqrt(self.ssds / self.sz) if self.sz else 0.0
        ad = self.sds / self.sz if self.sz else 0.0
        return (f"cnt={self.num_diff} sum_squared_diffs={self.ssds:g} avg_abs_diff={aad:g} max_abs_diff={self.mad:g} "
                f"sum_diffs={self.sds:g} avg_diff={ad:g} max_rel_diff={self.mrd:g} avg1={self.avg1:g} avg2={self.avg2:g}")


# ------------------------------------------------------------------------------------------------
# wisdom files
# ------------------------------------------------------------------------------------------------
@dataclass
class OpRun:
    be_plat_tag: str
    rt_secs: float
    err: str = ""
    op: Optional[Op] = None  # annotated op actually run (only when err is empty)


@dataclass
class OpTuneWisdom:
    op_tune: str  # NESI dump text; tunes are matched by this exact string (src/rtc_prof.cc:257-261)
    runs: Dict[str, OpRun] = field(default_factory=dict)


@dataclass
class OpWisdom:
    op: Op
    kgs: List[Tuple[str, Digest]] = field(default_factory=list)
    wisdoms: List[OpTuneWisdom] = field(default_factory=list)


def read_wisdoms(path: str) -> List[OpWisdom]:
    with open(path) as f:
        lines = [l.rstrip("\n") for l in f]
    i, out = 0, []

    def nxt() -> str:
        nonlocal i
        if i >= len(lines):
            raise RtErr("wisdom: unexpected end of file")
        l = lines[i]
        i += 1
        return l
    while i < len(lines):
        l = nxt()
        if not l:
            continue
        if l != "op_wisdom_t":
            raise RtErr(f"wisdom: expected op_wisdom_t at line {i}, got {l!r}")
        ow = OpWisdom(parse_op(nxt()))
        while True:
            l = nxt()
            if l == "/op_wisdom_t":
                break
            if l == "kg":
                vn = nxt()
                ow.kgs.append((vn, Digest.from_hex(nxt())))
            elif l == "op_tune_wisdom_t":
                otw = OpTuneWisdom(nxt())
                while True:
                    l = nxt()
                    if l == "/op_tune_wisdom_t":
                        break
                    if l != "op_run_t":
                        raise RtErr(f"wisdom: expected op_run_t at line {i}, got {l!r}")
                    tag = nxt(); secs = float(nxt()); err = nxt()
                    run = OpRun(tag, secs, err)
                    if not err:
                        run.op = parse_op(nxt())
                    otw.runs[tag] = run
                ow.wisdoms.append(otw)
            else:
                raise RtErr(f"wisdom: unexpected line {l!r}")
        out.append(ow)
    return out


def write_wisdoms(path: str, ows: List[OpWisdom]) -> None:
    with open(path, "w") as f:
        for ow in ows:
            f.write("op_wisdom_t\n" + ow.op.to_str() + "\n")
            for vn, dg in ow.kgs:
                f.write("kg\n" + vn + "\n" + dg.to_hex() + "\n")
            for otw in ow.wisdoms:
                f.write("op_tune_wisdom_t\n" + otw.op_tune + "\n")
                for tag in sorted(otw.runs):
                    r = otw.runs[tag]
                    f.write("op_run_t\n" + r.be_plat_tag + "\n" + repr(r.rt_secs) + "\n" + r.err + "\n")
                    if not r.err:
                        f.write((r.op.to_str() if r.op else "()") + "\n")
                f.write("/op_tune_wisdom_t\n")
            f.write("/op_wisdom_t\n")

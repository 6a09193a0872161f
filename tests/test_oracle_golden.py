"""Pins the CPU oracle (oracle/boda_oracle.c) and the host-side digest code (boda_amd/digest.py)
against every known-answer digest the reference's own tests hold for this path
(test/good_tr/*/wisdom.wis, committed under tests/golden/wisdom/)."""
import os
import numpy as np
import pytest

from boda_amd.digest import Digest, read_wisdoms, sample_plan, write_wisdoms, KNOWN_SEEDS
from oracle import boda_oracle as bo

MRD = 2e-4  # src/rtc_prof.cc:161


def _check_file(golden_dir, name, mode, max_batch=None):
    ws = read_wisdoms(os.path.join(golden_dir, "wisdom", name + ".wis"))
    assert ws
    n = 0
    for ow in ws:
        if max_batch is not None and ow.op.get_type() == "Convolution" and ow.op.conv_geom()["B"] > max_batch:
            continue
        r = bo.run_op(ow.op, mode=mode)
        assert len(ow.kgs) == 1
        vn, kg = ow.kgs[0]
        assert kg.seed == KNOWN_SEEDS[vn]
        mn, mx, samps, st, of = bo.digest(r[vn], kg.dims.sizes, kg.seed)
        mine = Digest(kg.dims, kg.seed, mn, mx, samps)
        assert kg.mrd_comp(mine, MRD) == "", ow.op.to_str()
        # the oracle follows the reference's fp32 fma order exactly: digests are in fact bit-identical
        assert np.array_equal(kg.samps, samps) and kg.min_v == mn and kg.max_v == mx, ow.op.to_str()
        n += 1
    return n


def test_sgemm_gen600(golden_dir):
    assert _check_file(golden_dir, "sgemm-gen600", 600) == 1


def test_sgemm_gen600_exact_answer():
    # mode 600: a[k,m] = 1000*m + k, b = identity  ->  c[m,n] = 1000*m + n   (gen_data_sgemm_{a,b}.cucl)
    a = bo.gen_sgemm_a(256, 128, 600); b = bo.gen_sgemm_b(256, 256, 600)
    c = bo.sgemm(a, b)
    m, n = np.meshgrid(np.arange(128), np.arange(256), indexing="ij")
    assert np.array_equal(c, (1000 * m + n).astype(np.float32))


def test_sgemm_gen5(golden_dir):
    assert _check_file(golden_dir, "sgemm-gen5", 5) == 1


def test_conv_gen5_and_debug(golden_dir):
    assert _check_file(golden_dir, "conv-gen5", 5) == 1
    assert _check_file(golden_dir, "conv-debug", 5) == 2


def test_conv_full_gen5_all_204(golden_dir):
    assert _check_file(golden_dir, "conv-full-gen5", 5) == 204


def test_conv_3x3_all_42(golden_dir):
    assert _check_file(golden_dir, "ops-prof-conv-3x3-cudnn-boda", 5) == 42


def test_product_digest_matches_oracle_digest(golden_dir):
    """boda_amd.digest (numpy, product-side harness) == oracle C digest == stored golden."""
    ws = read_wisdoms(os.path.join(golden_dir, "wisdom", "conv-debug.wis"))
    for ow in ws:
        r = bo.run_op(ow.op)
        vn, kg = ow.kgs[0]
        prod = Digest.from_array(r[vn], kg.dims, kg.seed)
        assert prod.to_hex() == kg.to_hex()
        plan = sample_plan(kg.dims, kg.seed)
        _, _, _, st, of = bo.digest(r[vn], kg.dims.sizes, kg.seed)
        assert [p[0] for p in plan] == list(st) and [p[1] for p in plan] == list(of)


def test_digest_hex_roundtrip_and_detects_corruption(golden_dir):
    ws = read_wisdoms(os.path.join(golden_dir, "wisdom", "sgemm-gen5.wis"))
    vn, kg = ws[0].kgs[0]
    assert Digest.from_hex(kg.to_hex()).to_hex() == kg.to_hex()
    bad = Digest(kg.dims, kg.seed, kg.min_v, kg.max_v, kg.samps.copy())
    bad.samps[7] *= np.float32(1.01)
    assert "stride=" in kg.mrd_comp(bad, MRD)
    nan = Digest(kg.dims, kg.seed, float("nan"), kg.max_v, kg.samps.copy())
    assert "min_v" in kg.mrd_comp(nan, MRD)


def test_wisdom_roundtrip(golden_dir, tmp_path):
    src = os.path.join(golden_dir, "wisdom", "ops-prof-conv-3x3-cudnn-boda.wis")
    ws = read_wisdoms(src)
    assert len(ws) == 42
    p = tmp_path / "w.wis"
    write_wisdoms(str(p), ws)
    ws2 = read_wisdoms(str(p))
    assert [w.op.to_str() for w in ws] == [w.op.to_str() for w in ws2]
    assert [k[1].to_hex() for w in ws for k in w.kgs] == [k[1].to_hex() for w in ws2 for k in w.kgs]
    # run records (op_tune_wisdom_t / op_run_t): excerpt of the reference's test/wisdom-merged.wis
    wm = read_wisdoms(os.path.join(golden_dir, "wisdom", "wisdom-merged-head3.wis"))
    assert len(wm) == 3 and all(len(w.wisdoms) >= 1 for w in wm)
    runs = [r for w in wm for t in w.wisdoms for r in t.runs.values()]
    assert any(r.be_plat_tag.startswith("ocl:") for r in runs) and any(r.err == "" and r.rt_secs > 0 for r in runs)
    write_wisdoms(str(p), wm)
    wm2 = read_wisdoms(str(p))
    assert [(t.op_tune, sorted(t.runs)) for w in wm for t in w.wisdoms] == [(t.op_tune, sorted(t.runs)) for w in wm2 for t in w.wisdoms]
    assert [r.op.to_str() for w in wm for t in w.wisdoms for r in t.runs.values() if not r.err] == \
           [r.op.to_str() for w in wm2 for t in w.wisdoms for r in t.runs.values() if not r.err]


def test_det_hash_rand_range_and_determinism():
    v = np.array([bo.det_hash_rand(i) for i in range(0, 200000, 37)], np.float32)
    assert v.min() >= -5.0 and v.max() <= 5.0 and abs(float(v.mean())) < 0.1
    assert bo.det_hash_rand(12738732) == bo.det_hash_rand(12738732)


def test_conv_oracle_vs_torch_independent():
    """Independent cross-check of the oracle's conv semantics (cross-correlation, padding, stride)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    for (B, C, H, W, OC, K, S, P) in [(2, 3, 23, 19, 8, 5, 2, 2), (1, 7, 9, 9, 5, 3, 1, 1), (3, 4, 8, 8, 6, 1, 1, 0), (2, 3, 31, 31, 4, 11, 4, 0)]:
        i = rng.standard_normal((B, C, H, W), dtype=np.float32)
        f = rng.standard_normal((OC, C, K, K), dtype=np.float32)
        b = rng.standard_normal((OC,), dtype=np.float32)
        ref = torch.relu(torch.nn.functional.conv2d(torch.from_numpy(i).double(), torch.from_numpy(f).double(),
                                                    torch.from_numpy(b).double(), stride=S, padding=P)).float().numpy()
        out = bo.conv_fwd(i, f, b, (S, S), (P, P), True)
        assert out.shape == ref.shape and bo.mrd(ref, out) < 1e-5
        out_nr = bo.conv_fwd(i, f, b, (S, S), (P, P), False)
        assert out_nr.min() < 0


def test_pool_lrn_relu_oracle_vs_independent_definitions():
    """The non-conv oracle ops against their textbook definitions: torch's pooling (Caffe-style: ceil sizes, padding pixels
    excluded from max and from the average's divisor) and a direct float64 window sum for LRN."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    rng = np.random.default_rng(3)
    for (B, C, H, W, K, S, P) in [(2, 5, 13, 13, 3, 2, 0), (1, 4, 14, 14, 3, 2, 1), (2, 3, 7, 9, 3, 1, 1), (1, 2, 6, 6, 6, 1, 0), (2, 3, 12, 12, 2, 2, 0)]:
        x = rng.standard_normal((B, C, H, W), dtype=np.float32)
        t = torch.from_numpy(x)
        mx = bo.pool_fwd(x, (K, K), (S, S), (P, P), False)
        ref = F.max_pool2d(t, K, S, P, ceil_mode=True).numpy()
        assert mx.shape == ref.shape and np.array_equal(mx, ref)
        av = bo.pool_fwd(x, (K, K), (S, S), (P, P), True)
        ones = torch.ones_like(t).double()
        ssum = F.avg_pool2d(t.double(), K, S, P, ceil_mode=True, count_include_pad=True, divisor_override=1)
        cnt = F.avg_pool2d(ones, K, S, P, ceil_mode=True, count_include_pad=True, divisor_override=1)
        assert av.shape == ssum.shape and bo.mrd((ssum / cnt).float().numpy(), av) < 1e-5
    for (B, C, H, W, ls, alpha, beta, k) in [(2, 11, 3, 4, 5, 1e-2, 0.75, 2.0), (1, 3, 2, 2, 5, 1e-4, 0.75, 1.0), (1, 9, 2, 3, 3, 5e-2, 0.5, 1.5)]:
        x = rng.standard_normal((B, C, H, W), dtype=np.float32) * 3
        got = bo.lrn_fwd(x, ls, alpha, beta, k)
        xd = x.astype(np.float64); want = np.empty_like(xd)
        for c in range(C):
            lo, hi = max(0, c - ls // 2), min(C, c + ls // 2 + 1)
            want[:, c] = xd[:, c] * (k + (xd[:, lo:hi] ** 2).sum(1) * alpha / ls) ** -beta
        assert bo.mrd(want.astype(np.float32), got) < 1e-5
    x = np.array([-1.5, -0.0, 0.0, 2.0, np.nan, -np.inf, np.inf], np.float32)
    y = bo.relu(x)
    assert np.array_equal(y[[0, 1, 2, 3, 5, 6]], np.array([0, 0, 0, 2, 0, np.inf], np.float32)) and np.isnan(y[4]) and not np.signbit(y[1])

"""op-list parsing (both text forms) + flop/byte accounting of the reference."""
import os
import pytest
from boda_amd.op import parse_op, read_ops, parse_lexp, RtErr, Dims


def test_all_fixture_op_lists_parse(golden_dir):
    counts = {"sgemm-ops-full.txt": 17, "sgemm-ops-tiny.txt": 3, "sgemm-ops-debug.txt": 1,
              "conv-ops-1-5-20-nin-alex-gn.txt": 204, "conv-ops-kern-3x3-batch-1-5-20-nin-alex-gn.txt": 42,
              "conv-ops-small.txt": 9, "conv-ops-debug-tmp.txt": 2}
    for fn, n in counts.items():
        ops = read_ops(os.path.join(golden_dir, "ops", fn))
        assert len(ops) == n
        for o in ops:
            assert parse_op(o.to_str()) == o  # canonical print/parse round trip


def test_legacy_and_current_forms_agree():
    legacy = "(type=sgemm,dims_vals=(a=(K=64,M=32),b=(K=64,N=16),c=(M=32,N=16)))"
    cur = "(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=64,M=32)),b=(dims=(K=64,N=16)),c=(dims=(M=32,N=16))))"
    assert parse_op(legacy) == parse_op(cur)
    assert parse_op(cur).flops() == 2 * 32 * 16 * 64 and parse_op(cur).algo_bytes() == 4 * (64 * 32 + 64 * 16 + 32 * 16)


def test_conv_accounting_matches_baseline_md(golden_dir):
    # AlexNet conv1 at B=20: M=20*55*55, N=96, K=3*11*11  (src/latex-util.H:126-133)
    op = read_ops(os.path.join(golden_dir, "ops", "conv-ops-debug.txt"))[0]
    assert op.flops() == 2 * (20 * 55 * 55) * 96 * 363
    assert op.algo_bytes() == 4 * (20 * 3 * 227 * 227 + 20 * 96 * 55 * 55 + 96 * 363 + 96)


def test_errors():
    with pytest.raises(RtErr):
        parse_op("(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=64,M=32)),b=(dims=(K=32,N=16)),c=(dims=(M=32,N=16))))")
    with pytest.raises(RtErr):
        parse_op("(str_vals=(type=Convolution),nda_vals=(in=(dims=(img=1,chan=3,y=8,x=8))))")
    with pytest.raises(RtErr):
        parse_lexp("(a=b")
    assert parse_lexp("(a=(b=c\\,d),e=f)") == [("a", [("b", "c,d")]), ("e", "f")]
    assert Dims.make(K=4, M=8).strides == (8, 1)


def test_package_shape_data_is_the_fixture(golden_dir):
    """boda_amd/data/ops/sgemm-ops-full.txt (what bench.py reads: the product never reads tests/) == the reference-identical fixture."""
    import os
    from boda_amd.op import data_path
    assert open(data_path("ops", "sgemm-ops-full.txt"), "rb").read() == open(os.path.join(golden_dir, "ops", "sgemm-ops-full.txt"), "rb").read()

"""wis-merge / wis-ana (src/op-tuner.cc:126-392) over the run-record wisdom fixture (the first three ops of the reference's test/wisdom-merged.wis: 14 tunes,
21 runs each on three platforms), and the per-op best-tile table that feeds recorded wisdom back to the native planner."""
import io
import os
import pytest

from boda_amd.cnn_op import OpTune, add_codegen_annotations, set_tile_wisdom
from boda_amd.digest import OpRun, OpTuneWisdom, OpWisdom, read_wisdoms, write_wisdoms
from boda_amd.op import RtErr, parse_op
from boda_amd.wis_ana import TileWisdom, fmt_g, get_op_flops, wis_ana
from boda_amd.wis_merge import main as merge_main, merge_wisdoms, op_ref_key


@pytest.fixture(scope="module")
def head3(golden_dir):
    return read_wisdoms(os.path.join(golden_dir, "wisdom", "wisdom-merged-head3.wis"))


def test_op_order_is_the_reference_s(head3):
    """op_base_t::operator<: the nda's dims compare by SIZE first (then stride, then name), so img=5 sorts before img=20 -- not text order."""
    assert [ow.op.get_dims("in").dsz("img") for ow in head3] == [1, 5, 20]                # (the reference wrote the file from its by-op set)
    assert sorted(head3, key=lambda ow: op_ref_key(ow.op)) == head3
    assert sorted(head3, key=lambda ow: ow.op.to_str()) != head3                           # text order would put img=20 before img=5
    a = parse_op("(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=4,M=4)),b=(dims=(K=4,N=4)),c=(dims=(M=4,N=4))))")
    assert op_ref_key(a) > op_ref_key(head3[0].op)                                         # str_vals first: type=sgemm > type=Convolution


def test_merge_splits_and_reunites(head3, tmp_path):
    """Split the fixture by platform into three files, merge them in a scrambled order: the same ops in op order, every tune's runs reunited; tunes come out
    sorted by their text (a set keyed on str(op_tune)); the reader takes the writer's output."""
    parts = {}
    for ow in head3:
        for t in ow.wisdoms:
            for tag, r in t.runs.items():
                parts.setdefault(tag, {}).setdefault(ow.op.to_str(), OpWisdom(ow.op, [], [])).wisdoms.append(OpTuneWisdom(t.op_tune, {tag: r}))
    fns = []
    for i, (tag, byop) in enumerate(sorted(parts.items())):
        fn = str(tmp_path / f"p{i}.wis"); write_wisdoms(fn, list(reversed(list(byop.values())))); fns.append(fn)
    out_fn = str(tmp_path / "merged.wis")
    assert merge_main([fns[2], fns[0], fns[1], "--wisdom-out-fn", out_fn]) == 0
    merged = read_wisdoms(out_fn)
    assert [m.op.to_str() for m in merged] == [ow.op.to_str() for ow in head3]
    for m, ow in zip(merged, head3):
        assert [t.op_tune for t in m.wisdoms] == sorted(t.op_tune for t in ow.wisdoms)
        want = {t.op_tune: {tag: r.rt_secs for tag, r in t.runs.items()} for t in ow.wisdoms}
        assert {t.op_tune: {tag: r.rt_secs for tag, r in t.runs.items()} for t in m.wisdoms} == want
        assert all(r.op is not None and r.op.to_str() == ow_r.op.to_str() for t in m.wisdoms for tag, r in t.runs.items()
                   for ow_t in ow.wisdoms if ow_t.op_tune == t.op_tune for ow_r in [ow_t.runs[tag]])
    # the same platform twice for one (op, tune): an error, never an overwrite
    with pytest.raises(RtErr, match="never overwritten"):
        merge_wisdoms([head3, head3])
    # kgs: dropped by default, the first file's kept on request
    a = OpWisdom(head3[0].op, [("out", None)], []); b = OpWisdom(head3[0].op, [("other", None)], [])
    assert merge_wisdoms([[a], [b]])[0].kgs == [] and merge_wisdoms([[a], [b]], keep_kgs=True)[0].kgs == [("out", None)]


def test_wis_ana_columns(head3):
    """The CSV: header, op text, 2MNK, AOM / POM / REF seconds in the reference's `ostream << double` form; platform filter and error runs dropped first."""
    csv, out = io.StringIO(), io.StringIO()
    n, rows, aom = wis_ana(head3, s_plat="TITAN", csv_out=csv, out=out)
    lines = csv.getvalue().splitlines()
    assert lines[0] == "OP FLOPS boda-manual-tune boda-autotuned REF" and len(lines) == 4
    assert out.getvalue() == "tot_runs=42\n" and n == 42                                   # 3 ops x 14 runs on the two TITAN platform tags
    for line, ow in zip(lines[1:], head3):
        op_s, flops, a, p, r = line.rsplit(" ", 4)
        assert op_s == ow.op.to_str() and int(flops) == get_op_flops(ow.op) == ow.op.flops()
        titan = [(run.rt_secs, t.op_tune) for t in ow.wisdoms for tag, run in t.runs.items() if "TITAN" in tag and not run.err]
        assert p == fmt_g(min(titan)[0]) and r == "nan"
        assert a == fmt_g(next(s for s, tune in titan if tune == aom))
    assert aom == "(use_be=nvrtc,use_culibs=1,MNt=8 8,MNb=8 16,tconv_max_ksz=11 11)"       # ran all three ops, least total time
    # a reference tune: its runs leave the selection and fill the REF column
    csv2 = io.StringIO()
    n2, rows2, _ = wis_ana(head3, s_plat="nvrtc.*TITAN", ref_tune=aom, show_aom=False, pom_tag="boda-autotuned-TITAN", ref_tag="NVIDIA-cuDNNv5-library", csv_out=csv2, out=io.StringIO())
    l2 = csv2.getvalue().splitlines()
    assert l2[0] == "OP FLOPS boda-autotuned-TITAN NVIDIA-cuDNNv5-library"
    for line, (ow, poa) in zip(l2[1:], rows2):
        assert line.rsplit(" ", 1)[1] == fmt_g(poa.ref_r.rt_secs) and poa.min_tune != aom and poa.ref_r.be_plat_tag.startswith("nvrtc:")
    assert n2 == sum(1 for ow in head3 for t in ow.wisdoms for tag in t.runs if tag.startswith("nvrtc:") and "TITAN" in tag and t.op_tune != aom)
    # --s-img / --min-flops are permanent filters; number form: 6 significant digits, nan lower-case
    assert wis_ana(head3, s_img=5, out=io.StringIO())[1][0][0].op.get_dims("in").dsz("img") == 5
    assert len(wis_ana(head3, min_flops=5e7, out=io.StringIO())[1]) == 1
    assert fmt_g(0.000117024) == "0.000117024" and fmt_g(9.7248e-05) == "9.7248e-05" and fmt_g(float("nan")) == "nan" and fmt_g(1234567.0) == "1.23457e+06"
    # the ops table of `--ops-out-fn` (conv_op_info_to_latex_t with print_format 2, brief): KSZ & S & OC & B & in dims & %.3g flops
    ops = io.StringIO(); wis_ana(head3, show_aom=False, show_pom=False, show_ref=False, ops_out=ops, ops_out_brief=True, out=io.StringIO())
    assert ops.getvalue().splitlines()[0] == "1 & 1 & 16 & 1 & $  28 \\dx 28 \\dx 192 $ & 4.82e+06\\\\ "


def test_recorded_wisdom_overrides_the_planner(tmp_path):
    """The tuning loop the reference's way: ops-prof style runs under several op_tunes (here: tile=... keys) -> wisdom file -> wis-ana's per-op minimum ->
    a per-op best-tile table -> the annotation gives that tile to the op's function -> the native planner uses it instead of its cost model."""
    from boda_amd import rtc
    op = parse_op("(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan=256)),filts=(dims=(out_chan=256,in_chan=96,y=5,x=5)),in=(dims=(img=256,chan=96,y=27,x=27)),"
                  "in_pad=(tn=none,dims=(y=2,x=2)),kern_sz=(tn=none,dims=(y=5,x=5)),out=(dims=(img=256,chan=256,y=27,x=27)),out_chans=(tn=uint32_t,v=256),stride=(tn=none,dims=(y=1,x=1))))")
    other = parse_op(op.to_str().replace("img=256", "img=128"))
    auto = rtc.explain_plan(add_codegen_annotations(op, OpTune()))
    assert not auto.startswith("bodahip_conv_f32 128x128x")
    runs = lambda secs: {"hip:gfx950": OpRun("hip:gfx950", secs, "", add_codegen_annotations(op, OpTune()))}
    ow = OpWisdom(op, [], [OpTuneWisdom(OpTune().to_str(), runs(1.7e-3)), OpTuneWisdom(OpTune(hip_tile="128x128x16x2x2x2").to_str(), runs(1.5e-3)),
                           OpTuneWisdom(OpTune(hip_tile="64x64x16x2x2x2").to_str(), runs(2.5e-3))])
    ow2 = OpWisdom(other, [], [OpTuneWisdom(OpTune().to_str(), runs(0.8e-3)), OpTuneWisdom(OpTune(hip_tile="64x64x16x2x2x2").to_str(), runs(0.9e-3))])   # the planner's own choice is best
    wfn = str(tmp_path / "w.wis"); write_wisdoms(wfn, [ow, ow2])
    tw = TileWisdom.from_wisdoms(read_wisdoms(wfn), s_plat="hip:")
    assert tw.table == {op.to_str(): ("128x128x16x2x2x2", 1.5e-3)}
    tfn = str(tmp_path / "tiles.txt"); tw.save(tfn)
    assert TileWisdom.load(tfn).table == tw.table
    a = add_codegen_annotations(op, OpTune(), tile_wisdom=tw)
    assert a.str_vals["hip_tile"] == "128x128x16x2x2x2"
    tuned = rtc.explain_plan(a)                                                          # (the planner reads the function's own tile, as run() does)
    assert tuned.startswith("bodahip_conv_f32 128x128x") and tuned.split()[1].endswith("_w2x2") and tuned.split()[1] != auto.split()[1]
    assert "hip_tile" not in add_codegen_annotations(other, OpTune(), tile_wisdom=tw).str_vals
    assert add_codegen_annotations(op, OpTune(hip_tile="64x64x16x2x2x2"), tile_wisdom=tw).str_vals["hip_tile"] == "64x64x16x2x2x2"   # an explicit tune wins
    assert "hip_tile" not in add_codegen_annotations(op, OpTune(hip_dtype="bf16"), tile_wisdom=tw).str_vals                           # recorded for the fp32 function only
    try:                                                                                                                               # process-wide installation (what BODAHIP_TILE_WISDOM does at import)
        set_tile_wisdom(tfn)
        assert add_codegen_annotations(op, OpTune()).str_vals["hip_tile"] == "128x128x16x2x2x2"
    finally:
        set_tile_wisdom(None)
    assert "hip_tile" not in add_codegen_annotations(op, OpTune()).str_vals

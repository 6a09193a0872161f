"""cnn_op_info's row text (boda_amd/cnn_op_info.py <- src/latex-util.H:21-140, src/str_util.cc:230-263).  The number format is pinned on rows the
reference's author recorded from its own runs (doc/sgemm-notes.txt:9-16,19-26: `$128 \\dx 128 \\dx 128$ & \\verb|cublas_sgemm| & 64.2us & 65.3GF/s & 990m`
..., peak 6600e9): runtime, rate and percent-of-peak strings are reproduced from the runtime alone."""
import io
import math

from boda_amd.cnn_op_info import OpInfoToLatex, pp_bytes, pp_flops, pp_fps, pp_secs, pp_val
from boda_amd.op import parse_op

RECORDED = [  # (n, runtime string, rate string, % of 6600e9 string) -- doc/sgemm-notes.txt:9-16 (cublas) and :19-26 (sgemm)
    (128, "64.2us", "65.3GF/s", "990m"), (256, "62.5us", "537GF/s", "8.14"), (384, "76.5us", "1.48TF/s", "22.4"), (512, "121us", "2.22TF/s", "33.7"),
    (768, "218us", "4.16TF/s", "63.1"), (1024, "625us", "3.44TF/s", "52.1"), (1536, "1.37ms", "5.29TF/s", "80.1"), (2048, "3.56ms", "4.83TF/s", "73.1"),
    (128, "78.5us", "53.5GF/s", "810m"), (256, "142us", "237GF/s", "3.59"), (1024, "1.48ms", "1.45TF/s", "22.0"), (2048, "7.56ms", "2.27TF/s", "34.4")]


def _num(s, unit):
    """'64.2us' / '1.48TF/s' / '990m' -> value (suffix letters of pp_val: munp / KMGTP)."""
    body = s[:-len(unit)] if unit else s
    mult = {"m": 1e-3, "u": 1e-6, "n": 1e-9, "p": 1e-12, "K": 1e3, "M": 1e6, "G": 1e9, "T": 1e12, "P": 1e15}
    return float(body[:-1]) * mult[body[-1]] if body[-1] in mult else float(body)


def test_number_format_reproduces_the_reference_recorded_rows():
    for n, rt, rate, pct in RECORDED:
        secs, fps, p = _num(rt, "s"), _num(rate, "F/s"), _num(pct, "")
        assert pp_secs(secs) == rt and pp_fps(fps) == rate and pp_val(p) == pct           # the strings are fixed points of the formatter
        assert abs(2.0 * n ** 3 / secs / fps - 1.0) < 5e-3 and abs(fps / 6600e9 * 100.0 / p - 1.0) < 5e-3   # and consistent: rate = 2n^3 / runtime, % of 6600e9
    assert pp_val(0.81) == "810m" and pp_val(0.99) == "990m" and pp_val(8.14) == "8.14" and pp_val(80.1) == "80.1" and pp_val(537.0) == "537"
    assert pp_val(1234.0) == "1.23K" and pp_val(999.6) == "1000" and pp_val(5e-7) == "500n" and pp_val(float("nan")) == "NAN"
    assert pp_flops(2.0 * 2048 ** 3) == "17.2GF" and pp_bytes(3 * 4 * 2048 ** 2) == "50.3MB"


def test_rows_for_a_convolution_and_an_sgemm():
    cv = parse_op("(str_vals=(type=Convolution),nda_vals=(biases=(dims=(out_chan=256)),filts=(dims=(out_chan=256,in_chan=256,y=1,x=1)),in=(dims=(img=20,chan=256,y=27,x=27)),"
                  "in_pad=(tn=none,dims=(y=0,x=0)),kern_sz=(tn=none,dims=(y=1,x=1)),out=(dims=(img=20,chan=256,y=27,x=27)),out_chans=(tn=uint32_t,v=256),stride=(tn=none,dims=(y=1,x=1))))")
    tl = OpInfoToLatex(cv)
    assert (tl.M, tl.K, tl.N) == (20 * 27 * 27, 256, 256) and tl.flops == 2 * 20 * 27 * 27 * 256 * 256 and tl.bytes == 4 * (2 * 20 * 256 * 27 * 27 + 256 * 256 + 256)
    info = tl.info_row()
    assert info.startswith("1 & 1 & 256 & 20 & $  27 \\dx 27 \\dx 256 $ & $  27 \\dx 27 \\dx 256 $ &  $ 14580 \\dx 256 \\dx 256 $ & ") and info.endswith("\\\\ \n")
    # the row the reference's author recorded for this very op on a phone GPU (doc/sgemm-notes.txt:235: 36.5ms -> 52.4GF/s, 20.5 % of 256e9)
    eff = tl.eff_row("k1conv_simd", 36.5e-3, 256e9)
    assert eff == "1 & 1 & 256 & $ 20 \\dx 27 \\dx 27 \\dx 256 $ & \\verb|k1conv_simd| &  36.5ms & 52.4GF/s & 20.5 \\\\ \n"
    sg = parse_op("(str_vals=(type=sgemm),nda_vals=(a=(dims=(K=2048,M=2048)),b=(dims=(K=2048,N=2048)),c=(dims=(M=2048,N=2048))))")
    row = OpInfoToLatex(sg).eff_row("sgemm", 5.07e-3, 6600e9, 3.56e-3)
    assert row == " $ 2048 $ & 50.3MB & 17.2GF & 341  & 3.56ms & 4.83TF/s  & 5.07ms & 3.39TF/s  & 0.70x \\\\ \n"
    raw = OpInfoToLatex(sg, print_format=1).eff_row("sgemm", 5.07e-3, 6600e9, 3.56e-3)
    assert "17179869184.0" in raw and "0.00507" in raw

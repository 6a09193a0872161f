import numpy as np, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import _conv_op, _run
from boda_amd.rtc import make_rtc
from oracle import boda_oracle as bo
from boda_amd.ops_prof import OpsBackend
rtc = make_rtc("(be=hip)", 0); rtc.init(); be = OpsBackend(rtc)
shapes = [(4, 20, 8, 8, 100, 3, 3, 1, 1), (2, 6, 10, 10, 12, 3, 3, 1, 1), (2, 6, 10, 10, 12, 3, 3, 1, 0), (1, 3, 12, 12, 16, 3, 3, 1, 1), (3,3,227,227,96,11,11,4,0), (2, 3, 35, 35, 96, 11, 11, 4, 0)]
for s in shapes:
    op = _conv_op(*s)
    outs, _ = _run(be, op, 5, include_ins=True)
    g = op.conv_geom()
    want = bo.conv_fwd(outs["in"], outs["filts"], outs["biases"], (g["SY"], g["SX"]), (g["PY"], g["PX"]), True)
    got = outs["out"]
    ll = rtc.last_launch()
    bad = np.argwhere(want != got)
    print(s, ll, "nbad", len(bad), "of", want.size)
    if len(bad):
        print(" first bad", bad[:6].tolist(), "last", bad[-3:].tolist())
        for ax, nm in enumerate(["img", "oc", "y", "x"]):
            u, c = np.unique(bad[:, ax], return_counts=True)
            print("  ", nm, dict(zip(u.tolist()[:20], c.tolist()[:20])))
        b = bad[0]; print("  want/got", want[tuple(b)], got[tuple(b)])

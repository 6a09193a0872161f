"""-m gpu, sorts LAST: timing-attribution and other non-parity properties of the backend.  Nothing in here compares arithmetic with the oracle, and
nothing in here bounds wall-clock time: a host stall (GC, scheduler) between two launches must not be able to fail -- or, under `pytest -x`, shadow --
a parity test.  tests/conftest.py orders the GPU files so that every parity file runs before this one and before tests/test_gpu_zz_bench.py."""
import gc
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from boda_amd.cnn_op import NATIVE_ARGS, OpTune, add_codegen_annotations
from boda_amd.op import RtErr
from test_gpu_parity import _conv_op, be  # noqa: F401  (module-scoped backend fixture)


def _timing_invariants(mode, each, whole):
    """Structure only (no wall-clock bound).  stream: the per-call spans tile first-begin .. last-end exactly.  call / kernel: disjoint spans inside
    the whole.  kernel: events are bound to the call's own dispatches, so identical launches take comparable time (within 3x of the median) whatever
    the host did between them."""
    if not (all(d > 0 for d in each) and whole >= max(each)):
        return f"non-positive span or whole < max: {mode} {each} {whole}"
    if mode == "stream":
        if abs(sum(each) - whole) > 1e-3 * whole + 1e-4:
            return f"stream spans do not add up: {each} {whole}"
    elif sum(each) > whole * 1.001 + 1e-3:
        return f"spans exceed the whole: {mode} {each} {whole}"
    if mode == "kernel":
        med = float(np.median(each))
        if not all(d < 3 * med + 1e-3 for d in each):
            return f"kernel-bound spans differ: {each}"
    return None


def test_timing_modes(be):
    """get_dur under the three attributions of stream time (tune key `timing`): call = a marker pair around every call (default, the reference's semantics);
    stream = end markers only -- per-call durations then add up EXACTLY to first-begin .. last-end; kernel = events bound to the call's own dispatches."""
    from boda_amd.rtc import RtcArg, RtcFuncCall, RtcFuncInfo
    rtc = be.rtc
    op = _conv_op(8, 64, 28, 28, 64, 3, 3, 1, 1); anno = add_codegen_annotations(op, OpTune()); fn = anno.get_func_name()
    rtc.compile([RtcFuncInfo("tm_f", "", [a for a, _ in NATIVE_ARGS[fn]], anno)])
    am, made = {}, []
    for an, io in NATIVE_ARGS[fn]:
        if io == "REF":
            am[an] = RtcArg.ref(anno.get_dims(an)); continue
        rtc.create_var_with_dims("tm_" + an, anno.get_dims(an)); made.append("tm_" + an); am[an] = RtcArg.var("tm_" + an)
    try:
        rtc.finish_and_sync(); rtc.release_per_call_id_data()
        for mode in ("stream", "kernel", "call", ""):
            rtc.set_tune("timing", mode)
            call = RtcFuncCall("tm_f", am)
            for _ in range(3):
                rtc.run(call)                                   # warm-up: code object resident, clocks up
            rtc.finish_and_sync(); rtc.release_per_call_id_data()
            err = None
            for attempt in range(2):                            # one retry: the invariants are structural, but a driver hiccup is not a parity failure
                gc.collect(); gc.disable()
                try:
                    ids = [rtc.run(call) for _ in range(6)]
                    rtc.finish_and_sync()
                finally:
                    gc.enable()
                each = [rtc.get_dur(i, i) for i in ids]; whole = rtc.get_dur(ids[0], ids[-1])
                rtc.release_per_call_id_data()
                err = _timing_invariants(mode, each, whole)
                if err is None:
                    break
            assert err is None, err
        with pytest.raises(RtErr):
            rtc.set_tune("timing", "sometimes")
        # switching the attribution while call ids are outstanding is an error (the caller still holds them), not a silent release
        rtc.set_tune("timing", "call"); cid = rtc.run(RtcFuncCall("tm_f", am)); rtc.finish_and_sync()
        rtc.set_tune("timing", "call")                         # (the mode it already has: nothing happens, the id stays valid)
        assert rtc.get_dur(cid, cid) > 0
        with pytest.raises(RtErr, match="outstanding"):
            rtc.set_tune("timing", "stream")
        assert rtc.get_dur(cid, cid) > 0
    finally:
        rtc.finish_and_sync(); rtc.release_per_call_id_data()
        rtc.set_tune("timing", "")
        for vn in made:
            rtc.release_var(vn)
        rtc.release_func("tm_f"); rtc.release_per_call_id_data()

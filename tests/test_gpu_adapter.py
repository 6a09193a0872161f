"""-m gpu: Boda's rtc_test flow (src/rtc_compute.cc:135-194) through the C++ virtuals of the compiled be=hip adapter (adapter/hip_util.cc ->
C ABI -> libbodahip.so): my_dot compiled by hiprtc, c == a + b to 1e-6, and an unsupported request surfacing as unsup_err."""
import subprocess
import pytest

pytestmark = pytest.mark.gpu


def test_rtc_test_through_the_adapter_virtuals():
    from boda_amd.build import build_adapter
    exe = build_adapter()
    r = subprocess.run([exe, "run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "All is Well." in r.stdout and "plat_tag=hip:" in r.stdout, (r.stdout, r.stderr)


def test_rtc_test_through_the_adapter_on_a_multi_device_backend():
    """The adapter's NESI field `devices` (here 0:0) makes init() build ONE backend over several shards (bodahip_create_multi): the same rtc_test
    flow -- replicated vars, a generated function run on every device -- through the same virtuals."""
    from boda_amd.build import build_adapter
    exe = build_adapter()
    r = subprocess.run([exe, "run-multi"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "All is Well." in r.stdout and "*2" in r.stdout, (r.stdout, r.stderr)

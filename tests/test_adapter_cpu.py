"""the Boda-side adapter (adapter/hip_util.cc = the text of INTEGRATION.md section 1: `hip_compute_t : rtc_compute_t`,
NESI type_id "hip", forwarding every virtual to the C ABI) is COMPILED against a restatement of the reference interface
(adapter/shim/rtc_compute.H <- src/rtc_compute.H:35-123, with the reference's member types) and LINKED with libbodahip.so; on a GPU
Boda's own rtc_test flow (src/rtc_compute.cc:135-194: compile my_dot, three vars from vect_float, run, read back, compare) is driven
through the C++ virtuals (tests/test_gpu_adapter.py)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from boda_amd.build import build_adapter


def test_adapter_compiles_links_and_constructs_the_backend():
    exe = build_adapter()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "adapter linked: be=hip" in r.stdout, (r.stdout, r.stderr)


def test_adapter_text_is_the_one_in_integration_md():
    """The compiled file IS the documented adapter, not a variant of it."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = doc.index("```cpp\n// hip_util.cc") + len("```cpp\n"); b = doc.index("```", a)
    src = open(os.path.join(ROOT, "adapter", "hip_util.cc")).read()
    assert doc[a:b] in src
    for member in ("init", "get_plat_tag", "create_var_with_dims", "create_var_with_dims_as_reshaped_view_of_var", "release_var", "get_var_dims",
                   "set_var_to_zero", "compile", "release_func", "release_all_funcs", "run", "finish_and_sync", "release_per_call_id_data",
                   "get_dur", "profile_start", "profile_stop", "copy_nda_to_var", "copy_var_to_nda", "get_var_raw_native_pointer"):
        assert f" {member}( " in src, member   # all 19 virtuals of src/rtc_compute.H:45-80 are overridden


def test_adapter_maps_a_fatal_backend_error_to_rt_err():
    """Without a GPU, init() fails inside the backend; the adapter must surface it as an exception carrying the backend's message (rc 2)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run itself is covered by tests/test_gpu_adapter.py")
    exe = build_adapter()
    r = subprocess.run([exe, "run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device available" in r.stdout and "no CPU fallback" in r.stdout, (r.stdout, r.stderr)

"""The C-ABI library loads and exports every symbol include/mdisp.h declares (no GPU needed)."""
import ctypes
import os
import re

from cameracalibration_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdisp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdisp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libmdisp.so does not export {name}"
    assert set(declared) == set(_lib.EXPORTS)


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.mdisp_version() >= 100
    assert isinstance(lib.mdisp_last_error(), bytes)


def test_no_device_fails_loudly_without_fallback():
    lib = _lib.load()
    if lib.mdisp_device_count() > 0:
        return
    h = ctypes.c_void_p()
    rc = lib.mdisp_create(ctypes.byref(h), -1, None)
    assert rc == _lib.E_NODEVICE
    assert b"no CPU fallback" in lib.mdisp_last_error()


def test_multi_device_entry_points_fail_loudly_without_a_device():
    import numpy as np
    from cameracalibration_amd import api
    from cameracalibration_amd._lib import MdispError
    lib = _lib.load()
    if lib.mdisp_device_count() > 0:
        return
    img = np.zeros((32, 48), np.uint8)
    par = api.sgbm_params(0, 16, 8, 32, 10, 1, 8, 1)
    for call in (lambda: api.sgbm_compute_batch_multi([0], [img], [img], par), lambda: api.sgbm_compute_rows_multi([0], img, img, par)):
        try:
            call()
            raise AssertionError("expected MDISP_E_NODEVICE")
        except MdispError as e:
            assert e.code == _lib.E_NODEVICE and "no CPU fallback" in str(e)


def test_rectify_maps_and_synth_are_host_only():
    import numpy as np
    from cameracalibration_amd import api
    K = np.array([[500.0, 0, 31.5], [0, 500.0, 23.5], [0, 0, 1]])
    m1, m2 = api.initUndistortRectifyMap(K, np.zeros(5), np.eye(3), K, (64, 48))
    xs, ys = np.meshgrid(np.arange(64), np.arange(48))
    assert np.array_equal(m1[..., 0], xs) and np.array_equal(m1[..., 1], ys)
    assert not m2.any()
    left, right, gt = api.synth_pair(0x5EED0001, 48, 64, 16)
    assert left.shape == right.shape == gt.shape == (48, 64)
    assert right.max() == 255 and right.min() == 0 and gt.max() <= 15


def test_no_wide_buffer_store_is_followed_by_an_overwrite_of_its_data():
    """A 16-byte buffer store with its scalar offset in a register needs a wait state before a vector instruction that
    overwrites its data registers, and the compiler does not leave one for that form (scripts/check_store_hazard.py):
    the device code of the built library must not contain such a pair."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "cameracalibration_amd", "lib", "libmdisp.so")
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(root, "scripts", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(os.path.join(mod.LLVM, "llvm-objdump")):
        import pytest
        pytest.skip("no llvm-objdump")
    # the pattern itself, on listings of the faulty form and of the guarded one
    assert mod.regs("v[14:17]") == {14, 15, 16, 17} and mod.regs("v14") == {14}
    faulty = "0000000000001000 <kernel>:\n\tbuffer_store_dwordx4 v[14:17], v55, s[52:55], s57 offen nt sc1\n\tv_max_i32_e32 v14, 0, v45\n"
    guarded = faulty.replace("\tv_max", "\ts_nop 0\n\tv_max")
    other = faulty.replace("v_max_i32_e32 v14", "v_max_i32_e32 v18")
    immediate = faulty.replace(", s57 offen", ", 0 offen")         # the compiler covers this form itself
    assert len(mod.scan_text(faulty)[0]) == 1 and mod.scan_text(faulty)[1] == 1
    assert mod.scan_text(guarded) == ([], 1) and mod.scan_text(other) == ([], 1) and mod.scan_text(immediate) == ([], 0)
    import sys
    argv = sys.argv
    try:
        sys.argv = ["check_store_hazard.py", lib]
        assert mod.main() == 0
    finally:
        sys.argv = argv

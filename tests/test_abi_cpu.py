"""CPU-side checks of the C ABI: the library loads and exports every symbol the
header declares (no compute calls without a GPU)."""
import ctypes

import yvhip


def test_library_exports_every_header_symbol():
    syms = yvhip.header_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(yvhip.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert missing == [], missing
    assert yvhip.MISSING == []
    assert sorted(yvhip._SIGS) == syms          # every declared symbol has a ctypes signature


def test_version_and_error_strings():
    assert yvhip.lib.yv_version() >= 100
    assert yvhip.lib.yv_error_string(0) == b"ok"
    assert b"argument" in yvhip.lib.yv_error_string(-1)


def test_argument_validation_without_gpu():
    # argument errors are detected on the host before any HIP call
    assert yvhip.lib.yv_custom_nms(None, None, None, 1, 20000, 0.45, None, None, None, 0, None) == -1
    assert yvhip.lib.yv_custom_nms_ws_bytes(4, 100) == 0
    assert yvhip.lib.yv_custom_nms_ws_bytes(2, 8400) == 2 * 8400 * 16
    assert yvhip.lib.yv_efficient_nms(None, None, 1, 8400, 5, 0.25, 0.65, 100, 4096, None, None, None, None, None) == -1

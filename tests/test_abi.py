"""The C-ABI library loads and exports every symbol include/qpn_hip.h declares (no compute calls:
this runs without a GPU).  Also: there is NO CPU fallback -- engine creation must fail loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "qpn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qpn_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    import qpn_amd  # noqa: F401
    from qpn_amd import _lib
    names = _declared()
    assert "qpn_solve_avi_batch" in names and "qpn_verify_nodes" in names and len(names) >= 14
    assert set(names) == set(_lib.ABI_SYMBOLS)
    lib = _lib.load_library()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.qpn_abi_version() == 1
    assert lib.qpn_strerror(-3).decode() == "no gfx950 device visible"


def test_default_opts_are_the_reference_tolerances():
    import qpn_amd  # noqa: F401
    from qpn_amd import _lib
    lib = _lib.load_library()
    o = _lib.AviOpts()
    lib.qpn_avi_default_opts(ctypes.byref(o))
    assert o.check_tol == 1e-6 and o.comp_tol == 1e-2          # src/avi.jl:148, src/avi_solutions.jl:511


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run rather than route to any CPU path."""
    import torch
    import qpn_amd
    from qpn_amd.engine import QpnError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(QpnError):
        qpn_amd.Engine(0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower().replace("# oracle", "").replace("cpu oracle", "").replace("the oracle", ""), \
                    f"{f} mentions the oracle"

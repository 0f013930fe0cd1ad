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


def test_library_does_not_read_the_environment():
    """Routes are chosen by arguments and qpn_ctx_set_option alone: the product build of the library does not import getenv
    (the developer switches exist in -DQPN_DEV_SWITCHES builds only; QPN_HIP_LIB is the loaders' variable)."""
    import shutil
    import subprocess
    import qpn_amd  # noqa: F401
    from qpn_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    path = _lib.LIB_PATH
    if os.environ.get("QPN_HIP_LIB"):
        pytest.skip("a developer library is selected")
    out = subprocess.run([nm, "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    undefined = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
    assert "getenv" not in undefined and "secure_getenv" not in undefined
    src_dir = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "csrc")
    for f in os.listdir(src_dir):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(src_dir, f)).read()
            assert len(re.findall(r"\bgetenv\s*\(", txt)) == (1 if f == "qpn_internal.h" else 0), f

"""MI355X-native engine for the node-AVI hot path of QuadraticProgramNetworks.jl.

The directory is named after the reference (``quadraticprogramnetworks.jl_amd``); because of the
dot it is imported through the root-level shim ``qpn_amd`` (``import qpn_amd``).

Layout:
  csrc/            hand-written HIP kernels (gfx950) + the C-ABI (include/qpn_hip.h)
  _lib.py          ctypes binding of libqpn_hip.so -- fails loudly if the library is missing
  engine.py        thin host wrapper: batched solve / check / comp_indices / assemble / verify
  avi.py           AVI, GAVI, solve_avi, solve_gavi, convert, check_avi_solution, solve_qep, ...
                   (host-side mirror of src/avi.jl for the hot path)
  qp_processing.py verify_solution, solve_qp, process_qp      (mirror of src/qp_processing.jl)
  avi_solutions.py comp_indices                                (mirror of src/avi_solutions.jl:511-612)
  programs.py      QPNet data model: the input contract of the hot path (src/programs.jl)
  algorithm.py     solve / solve_base! inner loop               (src/algorithm.jl, src/requests.jl)
  examples.py      setup(:simple_bilevel | :four_player_matrix_game | :robust_avoid_simple | synthetic)
  sharding.py      node-range sharding + RCCL all-gather of the primal iterate
"""
from ._lib import LibraryMissing, load_library  # noqa: F401
from .engine import Engine, Nodes, default_engine  # noqa: F401

SUCCESS, RAY_TERM, MAX_ITERS, FAILURE = 1, 2, 3, 4
ROW_STD, ROW_GAVI = 0, 1

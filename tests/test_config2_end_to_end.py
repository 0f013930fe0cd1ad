"""BASELINE config 2, `setup(:robust_avoid_simple)` (examples/robust_avoid_simple.jl:1-93: three levels, 18 variables), through
`solve()` itself: outer loop -> the lower levels -> process_qp on every node with the device-made local pieces as solution
graphs (src/qp_processing.jl:193-198, :231) -> solve_qep per level.  What is checked without the reference at hand (no Julia, no
PATH: parity with the reference's own iterates is unpinned):
* the run ends `solved`: every node of every level passed verify_solution under the selected sub-pieces (src/algorithm.jl:44-101);
* the point is a fixed point: solve() started from it returns it unchanged;
* the HIP-backed run and the oracle-backed run end at the same point (1e-8);
* every constraint of the net holds at the point (1e-6).
The local pieces are the first generation of the reference's solution graphs (no `combine` exploration): the
point is an equilibrium with respect to those pieces."""
import numpy as np
import pytest

import qpn_amd  # noqa: F401
from qpn_amd import algorithm, examples


def _run(engine, seed):
    net = examples.setup("robust_avoid_simple", seed=seed)
    ret = algorithm.solve(net, engine=engine)
    assert ret["solved"]
    x = ret["x_opt"]
    again = algorithm.solve(examples.setup("robust_avoid_simple", seed=seed), x, engine=engine)
    assert again["solved"] and np.max(np.abs(again["x_opt"] - x)) <= 1e-9
    for con in net.constraints.values() if isinstance(net.constraints, dict) else net.constraints:
        A, l, u = con.poly.vectorize()
        ax = A @ x
        assert np.all(ax >= l - 1e-6) and np.all(ax <= u + 1e-6)
    return x


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_robust_avoid_simple_through_solve_on_the_oracle_engine(seed):
    from oracle_engine import OracleEngine
    _run(OracleEngine(), seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_robust_avoid_simple_through_solve_on_the_hip_engine(engine, seed):
    from oracle_engine import OracleEngine
    xh = _run(engine, seed)
    xo = _run(OracleEngine(), seed)
    assert np.max(np.abs(xh - xo)) <= 1e-8

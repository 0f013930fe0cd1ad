"""Property tests (hypothesis) of the host pieces around the hot path: the node partition of the multi-GPU sharding,
the row normalisation of Poly (src/sets.jl:68-92), the interchange round trip, and the reduced-block assembly
against the plain numpy statement."""
import numpy as np
from hypothesis import given, settings, strategies as st

import qpn_amd  # noqa: F401
from qpn_amd import interchange, sharding
from qpn_amd.programs import Poly, QPNet

import problems as P


@given(total=st.integers(0, 100_000), world=st.integers(1, 8))
def test_node_ranges_partition_the_nodes(total, world):
    r = [sharding.node_range(total, world, k) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    sizes = [hi - lo for lo, hi in r]
    assert max(sizes) - min(sizes) <= 1 and all(s >= 0 for s in sizes)


finite = st.floats(-50, 50, allow_nan=False, allow_infinity=False)


@given(rows=st.lists(st.tuples(st.lists(finite, min_size=3, max_size=3), finite, st.floats(0, 10)), min_size=1, max_size=6))
@settings(max_examples=60, deadline=None)
def test_poly_normalisation_keeps_the_set_and_is_idempotent(rows):
    A = np.array([r[0] for r in rows]); l = np.array([r[1] for r in rows]); u = l + np.array([r[2] for r in rows])
    p = Poly(A, l, u)
    q = Poly(*p.vectorize())
    # What the reference's Slice guarantees (src/sets.jl:76-88): coefficients below 1e-8 are dropped BEFORE the row is
    # scaled by its leading coefficient, so a coefficient in [1e-8, 1e-8 * lead) survives the first pass and falls under
    # the threshold afterwards: normalising twice may zero it.  Poly mirrors that order on purpose.  Idempotence holds
    # exactly when no scaled coefficient lies in (0, 1e-8); otherwise the second pass only zeroes such entries.
    Pa, Pl, Pu = p.vectorize(); Qa, Ql, Qu = q.vectorize()
    tiny = (np.abs(Pa) > 0) & (np.abs(Pa) < 1e-8)
    assert np.array_equal(np.where(tiny, 0.0, Pa), Qa) or np.allclose(Pa, Qa, rtol=0, atol=1e-8)
    if not tiny.any():
        assert np.array_equal(Pa, Qa) and np.array_equal(Pl, Ql) and np.array_equal(Pu, Qu)      # idempotent
    rng = np.random.default_rng(0)
    for x in rng.standard_normal((20, 3)) * 5:
        raw = bool(np.all(A @ x >= l - 1e-9) and np.all(A @ x <= u + 1e-9))
        if abs(np.min(np.abs(np.concatenate([A @ x - l, A @ x - u])))) > 1e-6:      # away from a boundary
            assert p.contains(x, tol=1e-9) == raw
    Pn, ln, un = p.vectorize()
    for i in range(Pn.shape[0]):
        nz = np.nonzero(Pn[i])[0]
        assert nz.size == 0 or Pn[i, nz[0]] == 1.0                                   # leading coefficient +1


@given(nv=st.integers(1, 5), seed=st.integers(0, 10_000))
@settings(max_examples=25, deadline=None)
def test_interchange_round_trip_of_random_nets(tmp_path_factory, nv, seed):
    rng = np.random.default_rng(seed)
    net = QPNet(nv)
    ncon = int(rng.integers(1, 4)); nqp = int(rng.integers(1, 4))
    for _ in range(ncon):
        r = int(rng.integers(1, 4))
        l = rng.standard_normal(r)
        net.add_constraint(rng.standard_normal((r, nv)), np.where(rng.random(r) < 0.3, -np.inf, l), l + rng.random(r))
    for k in range(nqp):
        G = rng.standard_normal((nv, nv))
        net.add_qp(G @ G.T, rng.standard_normal(nv), sorted(set(int(c) for c in rng.integers(1, ncon + 1, size=2))),
                   [int(v) for v in rng.choice(nv, size=int(rng.integers(1, nv + 1)), replace=False)], k=float(rng.standard_normal()))
    net.add_edges([(i, i + 1) for i in range(1, nqp)])
    net.assign_constraint_groups()
    d = str(tmp_path_factory.mktemp("net"))
    interchange.save_qpnet(d, net)
    back = interchange.load_qpnet(d)
    assert back.num_vars == net.num_vars and back.network_depth_map == net.network_depth_map
    for pid in net.qps:
        assert np.array_equal(net.qps[pid].f.Q, back.qps[pid].f.Q) and net.qps[pid].var_indices == back.qps[pid].var_indices
        assert net.qps[pid].f.k == back.qps[pid].f.k and net.qps[pid].constraint_indices == back.qps[pid].constraint_indices
    for cid in net.constraints:
        for a, b in zip(net.constraints[cid].poly.vectorize(), back.constraints[cid].poly.vectorize()):
            assert np.array_equal(a, b)
        assert net.constraints[cid].group_mapping == back.constraints[cid].group_mapping


@given(n=st.integers(1, 6), m=st.integers(0, 6), p=st.integers(0, 3), seed=st.integers(0, 1000))
@settings(max_examples=40, deadline=None)
def test_oracle_assembly_equals_the_numpy_statement(n, m, p, seed):
    from oracle import binding as ob
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((n, n)); R = rng.standard_normal((n, p)); qd = rng.standard_normal(n)
    A = rng.standard_normal((m, n)); B = rng.standard_normal((m, p)); l = -rng.random(m); u = rng.random(m); w = rng.standard_normal(p)
    M, q, lo, hi, kind = ob.assemble_node(Q, R, qd, A, B, l, u, w)
    Mn, qn, ln, un, kn = P.reduced_blocks(Q[None], R[None], qd[None], A[None], B[None], l[None], u[None], w)
    assert np.array_equal(M, Mn[0]) and np.allclose(q, qn[0], atol=1e-13)
    assert np.array_equal(lo, ln[0]) and np.array_equal(hi, un[0]) and np.array_equal(kind, kn[0])

"""QPNet interchange: write -> read round trips of the example nets, the same solve
result from a loaded net (CPU oracle engine as the arithmetic), the byte layout the Julia writer produces
(fortran_order NPY files + a compact meta.json, built by hand here), and refusal of malformed directories."""
import json
import os
import struct

import numpy as np
import pytest

import qpn_amd  # noqa: F401
from qpn_amd import algorithm, examples, interchange


def _same_net(a, b):
    assert a.num_vars == b.num_vars and sorted(a.qps) == sorted(b.qps) and sorted(a.constraints) == sorted(b.constraints)
    for pid in a.qps:
        qa, qb = a.qps[pid], b.qps[pid]
        assert np.array_equal(qa.f.Q, qb.f.Q) and np.array_equal(qa.f.q, qb.f.q) and qa.f.k == qb.f.k
        assert qa.constraint_indices == qb.constraint_indices and qa.var_indices == qb.var_indices
    for cid in a.constraints:
        for x, y in zip(a.constraints[cid].poly.vectorize(), b.constraints[cid].poly.vectorize()):
            assert np.array_equal(x, y)
        assert a.constraints[cid].group_mapping == b.constraints[cid].group_mapping
    assert a.network_edges == b.network_edges and a.reachable_nodes == b.reachable_nodes
    assert a.network_depth_map == b.network_depth_map
    assert a.options.__dict__ == b.options.__dict__
    assert np.array_equal(a.default_initialization, b.default_initialization)


@pytest.mark.parametrize("name", ["simple_bilevel", "four_player_matrix_game", "robust_avoid_simple", "synthetic_pairs"])
def test_round_trip_of_the_example_nets(tmp_path, name):
    net = examples.setup(name)
    net.set_options(max_iters=77, levels_to_remove_subsets={1, 2})
    d = str(tmp_path / name)
    interchange.save_qpnet(d, net)
    back = interchange.load_qpnet(d)
    _same_net(net, back)
    # bounds at infinity survive, every array is plain float64 NPY, nothing needs pickle
    for f in os.listdir(d):
        if f.endswith(".npy"):
            assert np.load(os.path.join(d, f), allow_pickle=False).dtype == np.float64
    interchange.save_qpnet(str(tmp_path / "again"), back)
    assert json.load(open(os.path.join(d, "meta.json"))) == json.load(open(tmp_path / "again" / "meta.json"))


def test_loaded_net_solves_like_the_original(tmp_path):
    from oracle_engine import OracleEngine
    net = examples.setup("simple_bilevel", gen_solution_map=True)
    d = str(tmp_path / "sb")
    interchange.save_qpnet(d, net)
    back = interchange.load_qpnet(d)
    eng = OracleEngine()
    for w in ([-2.0, -3.0], [1.0, 0.0], [0.0, 0.0]):
        x0 = np.array(w + [0.0, 0.0])
        ra = algorithm.solve(net, x0.copy(), engine=eng)
        rb = algorithm.solve(back, x0.copy(), engine=eng)
        assert np.array_equal(ra["x_opt"], rb["x_opt"])


def _julia_npy(path, arr):
    """The bytes julia/QPNExport.jl:write_npy emits: v1.0 header, fortran_order True, column-major data."""
    arr = np.asarray(arr, dtype="<f8")
    shape = f"({arr.shape[0]},)" if arr.ndim == 1 else "(" + ", ".join(str(s) for s in arr.shape) + ")"
    d = "{'descr': '<f8', 'fortran_order': True, 'shape': " + shape + ", }"
    pad = (64 - (10 + len(d) + 1) % 64) % 64
    header = (d + " " * pad + "\n").encode()
    with open(path, "wb") as fh:
        fh.write(b"\x93NUMPY\x01\x00" + struct.pack("<H", len(header)) + header + np.asfortranarray(arr).tobytes(order="F"))


def test_directory_as_the_julia_writer_lays_it_out(tmp_path):
    """simple_bilevel as QPNExport.jl would write it (examples/simple_bilevel.jl:6-35): vars [w1 w2 x y], follower
    min (y - x)^2 s.t. y >= 0 (node 1, owns y), leader min (x - w1)^2 + (y - w2)^2 (node 2, owns x), edge 2 -> 1."""
    d = tmp_path / "jl"; d.mkdir()
    Q1 = np.zeros((4, 4)); Q1[2, 2] = Q1[3, 3] = 2.0; Q1[2, 3] = Q1[3, 2] = -2.0
    Q2 = 2.0 * np.eye(4); Q2[0, 2] = Q2[2, 0] = -2.0; Q2[1, 3] = Q2[3, 1] = -2.0
    A = np.array([[0.0, 0.0, 0.0, 1.0]])
    for name, arr in [("qp1_Q", Q1), ("qp1_q", np.zeros(4)), ("qp2_Q", Q2), ("qp2_q", np.zeros(4)), ("con1_A", A),
                      ("con1_l", np.array([0.0])), ("con1_u", np.array([np.inf])), ("default_initialization", np.zeros(4))]:
        _julia_npy(d / (name + ".npy"), arr)
    meta = ('{"arrays": {"con1_A": [1, 4], "con1_l": [1], "con1_u": [1], "default_initialization": [4], "qp1_Q": [4, 4], '
            '"qp1_q": [4], "qp2_Q": [4, 4], "qp2_q": [4]}, "constraints": [{"group_mapping": {"1": 1}, "id": 1}], '
            '"format": "qpnet-interchange/1", "index_base": 1, "network_edges": {"1": [], "2": [1]}, "num_vars": 4, '
            '"options": {"max_iters": 150, "shared_variable_mode": "SHARED_DUAL", "levels_to_remove_subsets": null}, '
            '"qps": [{"constraint_indices": [1], "id": 1, "k": 0.0, "var_indices": [4]}, '
            '{"constraint_indices": [], "id": 2, "k": 0.0, "var_indices": [3]}]}')
    (d / "meta.json").write_text(meta)
    net = interchange.load_qpnet(str(d))
    assert net.num_vars == 4 and net.qps[1].var_indices == [3] and net.qps[2].var_indices == [2]      # 0-based here
    assert net.network_edges == {1: set(), 2: {1}} and net.num_levels() == 2
    assert np.array_equal(net.qps[2].f.Q, Q2) and np.array_equal(net.constraints[1].poly.A, A)
    assert net.constraints[1].poly.u[0] == np.inf and net.constraints[1].group_mapping == {1: 1}
    ref = examples.setup("simple_bilevel")
    for pid in (1, 2):
        assert np.array_equal(ref.qps[pid].f.Q, net.qps[pid].f.Q) and ref.qps[pid].var_indices == net.qps[pid].var_indices


def test_malformed_directories_are_refused(tmp_path):
    net = examples.setup("simple_bilevel")
    d = str(tmp_path / "n"); interchange.save_qpnet(d, net)
    meta = json.load(open(os.path.join(d, "meta.json")))
    bad = dict(meta, format="something-else")
    json.dump(bad, open(os.path.join(d, "meta.json"), "w"))
    with pytest.raises(ValueError):
        interchange.load_qpnet(d)
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"))
    np.save(os.path.join(d, "qp1_Q.npy"), np.zeros((3, 3)))
    with pytest.raises(ValueError):
        interchange.load_qpnet(d)
    np.save(os.path.join(d, "qp1_Q.npy"), np.zeros((4, 4), dtype=np.float32))
    with pytest.raises(ValueError):
        interchange.load_qpnet(d)

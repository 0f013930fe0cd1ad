"""Row A6 on the device: qpn_assemble_pools (combine_gavis, src/avi.jl:305-377, and convert, :113-128) against the host
mirror avi.combine_gavis / combine_gavis_reduced / convert -- matrices, bounds and kinds bit-equal (it is data movement),
q within 1e-14 (the device adds the parameter terms with fma in ascending order).  Pools of the reference's examples
(four-player Nash game; the three levels of robust_avoid_simple with fixture pieces) and random pools, shared and
per-item inputs, and BASELINE config 3 end to end: 1 000 payoff draws assembled by ONE call and solved."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = np.inf


def _host_forms(avi, n_total, dec, par, lab, w):
    g_ref = avi.combine_gavis(n_total, dec, par, lab)
    a = avi.convert(g_ref)
    ref = dict(M=a.M, q=a.N @ w + a.o, l=a.l, u=a.u, kind=np.zeros(len(a.l), np.uint8))
    red = None
    if sum(len(lab[i]["dvars"]) for i in lab) == len(dec):
        g = avi.combine_gavis_reduced(n_total, dec, par, lab)
        red = dict(M=np.vstack([g.M, g.A]), q=np.concatenate([g.N @ w + g.o, g.B @ w]), l=np.concatenate([g.l1, g.l2]),
                   u=np.concatenate([g.u1, g.u2]), kind=np.concatenate([np.zeros(len(g.l1), np.uint8), np.ones(len(g.l2), np.uint8)]))
    return ref, red


def _check(dev_out, host, what):
    Mc, q, lo, hi, kind = (np.asarray(a) for a in dev_out)
    M = Mc.T if Mc.ndim == 2 else np.swapaxes(Mc, 1, 2)[0]
    assert M.shape == host["M"].shape, what
    assert np.array_equal(M, host["M"]), what
    assert np.array_equal(lo[0], host["l"]) and np.array_equal(hi[0], host["u"]) and np.array_equal(kind[0], host["kind"]), what
    assert np.max(np.abs(q[0] - host["q"]), initial=0.0) <= 1e-14 * max(1.0, np.max(np.abs(host["q"]), initial=0.0)), what


def _pool_of(net, level, S):
    from qpn_amd import avi
    pool = sorted(net.network_depth_map[level])
    dec = sorted(set().union(*[set(net.decision_inds(i)) for i in pool]))
    par = [i for i in range(net.num_vars) if i not in set(dec)]
    lab = {i: avi.create_labeled_gavi_from_qp(net, i, S) for i in pool}
    return pool, dec, par, lab


def test_four_player_pool_both_forms(engine):
    from qpn_amd import avi, examples
    net = examples.setup("four_player_matrix_game", seed=3)
    pool, dec, par, lab = _pool_of(net, 1, {})
    w = np.zeros(0)
    ref, red = _host_forms(avi, 8, dec, par, lab, w)
    b = avi.pool_blocks(8, dec, par, lab)
    _check(avi.assemble_pool_batch(b, w, engine=engine, form="reference"), ref, "four-player reference form")
    _check(avi.assemble_pool_batch(b, w, engine=engine, form="reduced"), red, "four-player reduced form")
    assert ref["M"].shape[0] == 32 and red["M"].shape[0] == 16            # SURVEY section 8 size table


def test_robust_avoid_pools_all_levels_with_fixture_pieces(engine, oracle):
    """BASELINE config 2: the AVIs solve_qep forms at the three levels of robust_avoid_simple (N_ref 52 at level 3,
    growing with the child pieces above it; SURVEY section 8 size table), children's pieces taken from committed fixtures
    (tests/golden/robust_avoid_pieces.json: polyhedral pieces of the kind local_piece produces, rows over all 18
    variables).  Device assembly == host mirror; the assembled AVIs solve identically on the HIP path and the oracle
    (status, masks, primals), and pass A3.  Degenerate regime (Q = 0): pinned by A3 + residual, DESIGN.md section 2."""
    import json, os
    from qpn_amd import avi, examples
    from qpn_amd.engine import colmajor
    from qpn_amd.programs import Poly
    net = examples.setup("robust_avoid_simple")
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "robust_avoid_pieces.json")))
    x = np.asarray(fx["x"])
    sizes = {}
    for level in (3, 2, 1):
        S = {int(k): Poly(np.asarray(v["A"]), np.asarray(v["l"], dtype=float), np.asarray(v["u"], dtype=float), normalise=False)
             for k, v in fx["pieces"].items() if int(k) in set().union(*[net.network_edges[i] for i in net.network_depth_map[level]])}
        pool, dec, par, lab = _pool_of(net, level, S)
        w = x[par]
        ref, red = _host_forms(avi, net.num_vars, dec, par, lab, w)
        b = avi.pool_blocks(net.num_vars, dec, par, lab)
        out_ref = avi.assemble_pool_batch(b, w, engine=engine, form="reference")
        _check(out_ref, ref, f"robust_avoid level {level} reference form")
        sizes[level] = ref["M"].shape[0]
        forms = [("reference", out_ref, ref)]
        if red is not None:
            out_red = avi.assemble_pool_batch(b, w, engine=engine, form="reduced")
            _check(out_red, red, f"robust_avoid level {level} reduced form")
            forms.append(("reduced", out_red, red))
        for name, out, host in forms:
            Mc, q, lo, hi, kind = out
            z0 = np.zeros_like(q)
            z0[0, :len(dec)] = x[dec]                                                   # z0 = [x[dec]; 0], src/avi.jl:404
            if name == "reference":
                g = avi.combine_gavis(net.num_vars, dec, par, lab)
                z0[0, len(g.l1) + len(g.l2):] = g.A @ z0[0, :g.M.shape[1]] + g.B @ w      # slack start, src/avi.jl:108
            rg = engine.solve_avi_batch(Mc, q, lo, hi, z0=z0, kind=kind)
            rc = oracle.solve_avi_batch(host["M"][None], q, lo, hi, z0=z0, kind=kind)
            assert rg["status"][0] == rc["status"][0] == 1, (level, name, rg["status"], rc["status"])
            assert np.array_equal(rg["active"], rc["active"]) and np.max(np.abs(rg["z"] - rc["z"])) <= 1e-9, (level, name)
            assert rg["resid"][0] <= 1e-8
            deg, _ = engine.check_avi_batch(Mc if Mc.ndim == 3 else Mc, q, lo, hi, rg["z"], kind=kind)
            assert deg[0] == 0
    assert sizes[3] == 52 and 60 <= sizes[2] <= 120 and 60 <= sizes[1] <= 120, sizes       # SURVEY section 8 size table


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_pools_with_shared_variables_and_batches(engine, seed):
    """Random pool shapes incl. overlapping decision sets (reference form only) and players without constraints;
    per-item and shared inputs mixed; device == numpy statement of src/avi.jl:305-377 + :113-128."""
    from qpn_amd import avi
    rng = np.random.default_rng(seed)
    for trial in range(6):
        players = int(rng.integers(1, 5)); nd = int(rng.integers(2, 9)); p = int(rng.integers(0, 4))
        disjoint = trial % 2 == 0
        if disjoint:
            perm = rng.permutation(nd); cuts = np.sort(rng.choice(np.arange(1, nd), size=min(players - 1, nd - 1), replace=False))
            dv = [sorted(int(v) for v in part) for part in np.split(perm, cuts)]
        else:
            dv = [sorted(int(v) for v in rng.choice(nd, size=int(rng.integers(1, nd + 1)), replace=False)) for _ in range(players)]
            missing = set(range(nd)) - set().union(*map(set, dv))
            dv[0] = sorted(set(dv[0]) | missing)
        players = len(dv)
        disjoint = sum(len(d) for d in dv) == nd              # (a random draw may be disjoint by chance)
        n_total = nd + p
        dec = list(range(nd)); par = list(range(nd, n_total))
        lab = {}
        for i, d in enumerate(dv):
            mi = int(rng.integers(0, 5))
            A = rng.standard_normal((mi, n_total))
            M1 = np.hstack([rng.standard_normal((len(d), n_total)), np.zeros((len(d), len(d))), -A[:, d].T])
            lab[i + 1] = dict(dvars=d, M1=M1, q1=rng.standard_normal(len(d)), M2=A, l2=-rng.random(mi), u2=rng.random(mi))
        w = rng.standard_normal(p)
        ref, red = _host_forms(avi, n_total, dec, par, lab, w)
        b = avi.pool_blocks(n_total, dec, par, lab)
        _check(avi.assemble_pool_batch(b, w, engine=engine, form="reference"), ref, f"random pool {seed}/{trial} reference")
        if disjoint:
            _check(avi.assemble_pool_batch(b, w, engine=engine, form="reduced"), red, f"random pool {seed}/{trial} reduced")
            # a batch: per-item linear terms and parameters, shared matrices -> one shared M
            cnt = 5
            qds = rng.standard_normal((cnt, len(b["qd"]))); ws = rng.standard_normal((cnt, p))
            Mc, q, lo, hi, kind = avi.assemble_pool_batch(b, ws, engine=engine, form="reduced", qd=qds)
            assert Mc.ndim == 2 and q.shape[0] == cnt
            for k in range(cnt):
                lab_k = {i: dict(lab[i]) for i in lab}
                o = 0
                for i in sorted(lab_k):
                    ni = len(lab_k[i]["dvars"]); lab_k[i]["q1"] = qds[k, o:o + ni]; o += ni
                _, red_k = _host_forms(avi, n_total, dec, par, lab_k, ws[k])
                assert np.array_equal(Mc.T, red_k["M"]) and np.allclose(q[k], red_k["q"], rtol=0, atol=1e-13)
                assert np.array_equal(lo[k], red_k["l"]) and np.array_equal(hi[k], red_k["u"])
        else:
            with pytest.raises(Exception):
                avi.assemble_pool_batch(b, w, engine=engine, form="reduced")     # overlapping decision sets: reference form only


def test_four_player_1000_draws_one_assembly_call(engine, oracle):
    """BASELINE config 3: 1 000 random payoff draws of the four-player game; the whole batch of pool AVIs is built by ONE
    qpn_assemble_pools call (shared M, strideM = 0: only q differs between draws) and solved by one qpn_solve_avi_batch;
    a sample of draws is compared with the per-draw host assembly, the whole batch with the oracle, every draw's
    equilibrium certified by check_avi_solution."""
    from qpn_amd import avi, examples
    draws = 1000
    rng = np.random.Generator(np.random.Philox(key=[20240422, 3]))
    cs = rng.standard_normal((draws, 4, 4, 2))
    net0 = examples.setup("four_player_matrix_game", constellations=cs[0])
    pool, dec, par, lab = _pool_of(net0, 1, {})
    b = avi.pool_blocks(8, dec, par, lab)
    # the linear terms of all draws at once: player i's q over its own two variables (examples/four_player_matrix_game.jl:149-157),
    # q_i = -2 c_ii + 2 sum_{j != i} c_ij
    qd = np.stack([-2.0 * cs[:, i, i, :] + 2.0 * (cs[:, i, :, :].sum(axis=1) - cs[:, i, i, :]) for i in range(4)], axis=1).reshape(draws, 8)
    Mc, q, lo, hi, kind = avi.assemble_pool_batch(b, np.zeros(0), engine=engine, form="reduced", qd=qd)
    assert Mc.shape == (16, 16) and q.shape == (draws, 16)
    for d in (0, 1, 17, 500, 999):                       # per-draw host assembly (setup + create_labeled_gavi + combine)
        net = examples.setup("four_player_matrix_game", constellations=cs[d])
        _, _, _, lab_d = _pool_of(net, 1, {})
        g = avi.combine_gavis_reduced(8, dec, par, lab_d)
        assert np.array_equal(Mc.T, np.vstack([g.M, g.A]))
        assert np.max(np.abs(q[d] - np.concatenate([g.o, np.zeros(8)]))) <= 1e-13
        assert np.array_equal(lo[d], np.concatenate([g.l1, g.l2])) and np.array_equal(hi[d], np.concatenate([g.u1, g.u2]))
    rg = engine.solve_avi_batch(Mc, q, lo, hi, kind=kind)
    rc = oracle.solve_avi_batch(Mc.T.copy(), q, lo, hi, kind=kind)
    assert np.all(rg["status"] == 1) and np.array_equal(rg["active"], rc["active"])
    assert np.max(np.abs(rg["z"] - rc["z"])) <= 1e-9 and np.max(rg["resid"]) <= 1e-8
    deg, _ = engine.check_avi_batch(Mc, q, lo, hi, rg["z"], kind=kind)
    assert np.all(deg == 0)

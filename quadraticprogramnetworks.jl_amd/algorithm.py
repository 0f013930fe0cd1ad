"""Outer equilibrium loop: solve (src/requests.jl:1-22) and solve_base! (src/algorithm.jl:1-127).
Control skeleton on the host; the per-node map of process_qp (:44-52) and the per-level solve_qep
(:95) are the batch boundaries served by the GPU engine."""
from __future__ import annotations

import numpy as np

from .avi import AVISolveError, solve_qep
from .level_batch import process_level
from .polyhedra import remove_subsets_many


class CyclingError(RuntimeError):
    pass


def solve(qpn, x_init=None, engine=None, rng=None):
    """solve(qpn) / solve(qpn, x_init) -> dict(solved, x_opt, Sol, x_fail)."""
    if x_init is None:
        x_init = qpn.default_initialization
    rng = rng if rng is not None else np.random.default_rng(1)
    qpn.iterate_cache = {}
    qpn.__dict__["_process_memo"] = {}          # level_batch.process_level / the subset reduction below: results of the last sweep
    qpn.__dict__["_subset_memo"] = {}
    return solve_base(qpn, np.asarray(x_init, dtype=np.float64), level=1, proj_vectors=[], rng=rng, engine=engine)


def solve_base(qpn, x_init, level=1, proj_vectors=None, rng=None, engine=None):
    x = np.array(x_init, dtype=np.float64, copy=True)
    opts = qpn.options
    try:
        if level == 1 and not proj_vectors:
            for _ in range(opts.num_projections):
                proj_vectors.append(rng.standard_normal(len(x)))                     # :10-12
        for _iters in range(opts.max_iters):                                         # :13
            proj_vals = np.array([x @ v for v in proj_vectors])
            if opts.check_for_cycling:                                               # :16-30
                if opts.num_projections == 0:
                    raise CyclingError("Cycling check requested, but num_projections == 0.")
                cache = qpn.iterate_cache.setdefault(level, None)
                if cache is None:
                    qpn.iterate_cache[level] = [proj_vals]
                else:
                    if any(np.allclose(proj_vals, pv, rtol=np.sqrt(np.finfo(float).eps), atol=0) for pv in cache):
                        raise CyclingError("Cycling detected (noticed solution iterate returned to a previous value).")
                    cache.append(proj_vals)
            if level < qpn.num_levels():                                             # :32-42
                low = solve_base(qpn, x, level=level + 1, proj_vectors=proj_vectors, rng=rng, engine=engine)
                if not low["solved"]:
                    return dict(solved=False, x_fail=x, x_opt=None)
                S = low["Sol"]; x = low["x_opt"]
            else:
                S = {}
            players = sorted(qpn.network_depth_map[level])                           # :44
            child_level = sorted(set().union(*[qpn.network_edges[i] for i in players]))
            results = process_level(qpn, players, x, S, engine=engine,
                                    exploration_vertices=opts.exploration_vertices)                    # :47-49, one batch
            equilibrium = True
            sub_assign = {i: S[i][0] for i in child_level}                           # :54
            sub_ids = {i: 0 for i in child_level}
            if any(r["failed"] for r in results):                                    # :57-66
                return dict(solved=False, x_fail=x, x_opt=None)
            graphs = {}
            for pid, r in zip(players, results):                                     # :68-90
                if not r["solution"]:
                    equilibrium = False
                    if level < qpn.num_levels():
                        for child, sp in r["subpiece_assignments"].items():
                            sub_assign[child] = S[child][sp]; sub_ids[child] = sp
                else:
                    graphs[pid] = r["S"]
            lv = opts.levels_to_remove_subsets                                       # None = NaturalNumbers(): every level
            if graphs and (lv is None or level in lv):                               # :84, all nodes of the level in one batch
                ids = [pid for pid in graphs if graphs[pid] is not None and len(graphs[pid]) > 1]
                smemo = qpn.__dict__.setdefault("_subset_memo", {})
                for pid in ids:                                                      # the graph process_level handed back unchanged:
                    ent = smemo.get(pid)                                             # its reduction is known
                    if ent is not None and ent[0] is graphs[pid]:
                        graphs[pid] = ent[1]
                ids = [pid for pid in ids if not (pid in smemo and smemo[pid][1] is graphs[pid])]
                if ids:
                    from .avi import _eng
                    for pid, kept in zip(ids, remove_subsets_many([graphs[pid] for pid in ids], _eng(engine))):
                        smemo[pid] = (graphs[pid], kept)
                        graphs[pid] = kept
            S.update(graphs)
            if not equilibrium:                                                      # :91-109
                try:
                    xnew = solve_qep(qpn, players, x, sub_assign, engine=engine,
                                     settled={pid for pid, r in zip(players, results) if r["solution"]})
                except AVISolveError:
                    return dict(solved=False, x_fail=x, x_opt=None)
                if np.linalg.norm(xnew - x) < 1e-4:                                  # :96-99
                    raise RuntimeError("Detected disagreement in solution status between qp solution "
                                       "processer and equilibrium solver.")
                x = xnew
                continue
            if level == 1:
                qpn.iterate_cache = {}
            return dict(solved=True, x_opt=x, Sol=S)                                 # :116
        raise RuntimeError("Can't find solution")                                    # :119
    except (RuntimeError, CyclingError) as err:                                      # :120-126
        qpn.iterate_cache = {}
        return dict(solved=False, x_fail=x, x_opt=None, error=str(err))

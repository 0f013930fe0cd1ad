"""QPNet interchange: a numeric QPNet as a directory of plain files, so that nets
built by the reference's Julia `setup(...)` can travel to a GPU box that has no Julia, and back.

    <dir>/meta.json     format tag, sizes, ids, index lists, edges, options
    <dir>/*.npy         one NPY file (version 1.0, '<f8', either memory order) per array

The fields are the reference's own (src/programs.jl:16-92): per QP `f.Q, f.q, f.k, constraint_indices,
var_indices`; per constraint `vectorize(poly)` = (A, l, u) (src/sets.jl:213-221) and `group_mapping`;
`network_edges`; `options`; `default_initialization`.  meta.json keeps the reference's 1-based variable
indices and node / constraint ids; the Python model is 0-based for variables and converts on the way.
The Julia writer is `julia/QPNExport.jl` (plain Julia: an NPY writer and a JSON emitter of a few lines,
no package beyond SparseArrays).  Nothing here is read with a loader that can execute code
(`numpy.load(allow_pickle=False)`, `json`).
"""
from __future__ import annotations

import json
import os

import numpy as np

from .programs import Constraint, Poly, QPNet, QPNetOptions

FORMAT = "qpnet-interchange/1"


def _opt_to_json(o: QPNetOptions) -> dict:
    d = dict(o.__dict__)
    lv = d.get("levels_to_remove_subsets")
    d["levels_to_remove_subsets"] = None if lv is None else sorted(int(v) for v in lv)    # None = NaturalNumbers()
    return d


def save_qpnet(path: str, net: QPNet) -> None:
    os.makedirs(path, exist_ok=True)
    files = {}

    def put(name, arr):
        np.save(os.path.join(path, name + ".npy"), np.ascontiguousarray(arr, dtype=np.float64), allow_pickle=False)
        files[name] = list(np.shape(arr))

    meta = {"format": FORMAT, "num_vars": net.num_vars, "index_base": 1, "qps": [], "constraints": [],
            "network_edges": {str(k): sorted(int(v) for v in vs) for k, vs in sorted(net.network_edges.items())},
            "options": _opt_to_json(net.options)}
    for pid, qp in sorted(net.qps.items()):
        put(f"qp{pid}_Q", qp.f.Q); put(f"qp{pid}_q", qp.f.q)
        meta["qps"].append({"id": int(pid), "k": float(qp.f.k),
                            "constraint_indices": [int(c) for c in qp.constraint_indices],
                            "var_indices": [int(v) + 1 for v in qp.var_indices]})
    for cid, con in sorted(net.constraints.items()):
        A, l, u = con.poly.vectorize()
        put(f"con{cid}_A", A); put(f"con{cid}_l", l); put(f"con{cid}_u", u)
        meta["constraints"].append({"id": int(cid), "group_mapping": {str(k): int(v) for k, v in sorted(con.group_mapping.items())}})
    put("default_initialization", net.default_initialization)
    meta["arrays"] = files
    with open(os.path.join(path, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


def load_qpnet(path: str) -> QPNet:
    with open(os.path.join(path, "meta.json")) as fh:
        meta = json.load(fh)
    if meta.get("format") != FORMAT:
        raise ValueError(f"{path}: not a {FORMAT} directory (format = {meta.get('format')!r})")
    base = int(meta.get("index_base", 1))
    nv = int(meta["num_vars"])

    def get(name, shape):
        a = np.load(os.path.join(path, name + ".npy"), allow_pickle=False)
        if a.dtype != np.float64:
            raise ValueError(f"{name}.npy: dtype {a.dtype}, expected float64")
        a = np.ascontiguousarray(a)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}.npy: shape {a.shape}, expected {tuple(shape)}")
        return a

    net = QPNet(nv)
    for c in sorted(meta["constraints"], key=lambda c: c["id"]):
        cid = int(c["id"])
        rows = int(meta["arrays"][f"con{cid}_l"][0])
        A = get(f"con{cid}_A", (rows, nv)); l = get(f"con{cid}_l", (rows,)); u = get(f"con{cid}_u", (rows,))
        # rows are taken as they are: the writer's Poly already normalised them (src/sets.jl:68-92)
        net.constraints[cid] = Constraint(Poly(A, l, u, normalise=False), {int(k): int(v) for k, v in c["group_mapping"].items()})
    for q in sorted(meta["qps"], key=lambda q: q["id"]):
        pid = int(q["id"])
        Q = get(f"qp{pid}_Q", (nv, nv)); qv = get(f"qp{pid}_q", (nv,))
        got = net.add_qp(Q, qv, [int(c) for c in q["constraint_indices"]], [int(v) - base for v in q["var_indices"]], k=float(q["k"]))
        if got != pid:
            raise ValueError(f"{path}: QP ids must be 1..N in order (got {pid} at position {got})")
        missing = [c for c in q["constraint_indices"] if int(c) not in net.constraints]
        if missing:
            raise ValueError(f"{path}: QP {pid} names unknown constraints {missing}")
    edges = [(int(i), int(j)) for i, js in meta["network_edges"].items() for j in js]
    net.add_edges(edges)
    opts = dict(meta.get("options", {}))
    lv = opts.pop("levels_to_remove_subsets", None)
    net.set_options(**opts)
    net.options.levels_to_remove_subsets = None if lv is None else set(int(v) for v in lv)
    net.default_initialization = get("default_initialization", (nv,))
    return net

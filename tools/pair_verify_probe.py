"""One pair of a synthetic_pairs net through solve() on the HIP engine, every verify of the outer loop repeated on the CPU oracle
(developer aid: where do the two engines' verify decisions part?).  usage: python tools/pair_verify_probe.py <pair> [n] [m]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import qpn_amd
from qpn_amd import algorithm, examples, level_batch
from oracle import binding as ob

warnings.simplefilter("ignore")
k = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 16; m = int(sys.argv[3]) if len(sys.argv) > 3 else n
eng = qpn_amd.default_engine(0)
net = examples.setup("synthetic_pairs", pairs=1, n=n, m=m, first=k)
orig = level_batch.verify_items


def vi(qpn, items, x, engine, tol=1e-4):
    recs, batches, out = orig(qpn, items, x, engine, tol=tol)
    for b in batches:
        xd, w = b.gather(x)
        for kk, i in enumerate(b.where):
            s, lam, pth = ob.verify_solution(b.Qc[kk].T, b.Rc[kk].T, b.qd[kk], b.Ac[kk].T, b.Bc[kk].T, b.l[kk], b.u[kk], xd[kk], w[kk])
            flag = "" if (bool(s) == out[i]["solution"] and pth == out[i]["path"]) else "   <<< DIFFERS"
            qt = b.Qc[kk].T @ xd[kk] + b.Rc[kk].T @ w[kk] + b.qd[kk]
            lh = np.zeros(b.m); 
            if out[i]["lam"] is not None: lh[:len(out[i]["lam"])] = out[i]["lam"]
            rh = np.linalg.norm(b.Ac[kk] @ lh - qt); ro = np.linalg.norm(b.Ac[kk] @ lam - qt)
            if flag and not os.path.exists(os.path.join(ROOT, "gpurun_out", "verify_case.npz")):
                np.savez(os.path.join(ROOT, "gpurun_out", "verify_case.npz"), Qc=b.Qc[kk], Rc=b.Rc[kk], qd=b.qd[kk], Ac=b.Ac[kk], Bc=b.Bc[kk],
                         l=b.l[kk], u=b.u[kk], xd=xd[kk], w=w[kk], lam_hip=lh, lam_oracle=lam)
            print(f"  verify pid {recs[i]['pid']} n={b.n} m={b.m}: HIP sol {out[i]['solution']} path {out[i]['path']} |res| {rh:.3e} min lam*sg ... ; oracle sol {bool(s)} path {pth} |res| {ro:.3e}{flag}")
    return recs, batches, out


level_batch.verify_items = vi
ret = algorithm.solve(net, engine=eng)
print("solved", ret["solved"], ret.get("error"))

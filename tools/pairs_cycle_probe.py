"""Which pairs of a synthetic_pairs net keep moving when solve() gives up with "Cycling detected"?  (developer aid)
usage: python tools/pairs_cycle_probe.py [pairs] [n] [m]"""
import sys, os, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import warnings
import numpy as np
import qpn_amd
from qpn_amd import algorithm, examples
import qpn_amd.algorithm as alg

warnings.simplefilter("ignore")
pairs, n, m = (int(v) for v in (sys.argv[1:4] + ["1000", "16", "16"][len(sys.argv) - 1:]))
eng = qpn_amd.default_engine(0) if os.environ.get("QPN_PROBE_ENGINE", "hip") == "hip" else __import__("oracle_engine").OracleEngine()
hist = []
orig = alg.solve_qep


def sq(qpn, players, x, S, engine=None, **kw):
    xn = orig(qpn, players, x, S, engine=engine, **kw)
    uns = [i for i in players if i not in (kw.get('settled') or set())]
    print('solve_qep: players', len(players), 'unsettled', len(uns), uns[:8], '|dx| %.3e' % np.linalg.norm(xn - x), flush=True)
    hist.append((len(players), xn.copy()))
    return xn


alg.solve_qep = sq
net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
t = time.time()
ret = algorithm.solve(net, engine=eng)
print("solved", ret["solved"], ret.get("error"), "%.1fs" % (time.time() - t), "solve_qep calls", len(hist))
if not ret["solved"] and len(hist) >= 3:
    xs = [h[1] for h in hist]
    last = xs[-1]
    same = [i for i in range(len(xs) - 1) if np.allclose(xs[i], last, rtol=1e-7, atol=0)]
    print("iterates equal to the last one:", same, "of", len(xs))
    d = np.abs(xs[-1] - xs[-2]).reshape(pairs, 2 * n).max(axis=1)
    d3 = np.abs(xs[-1] - xs[-3]).reshape(pairs, 2 * n).max(axis=1)
    mov = np.nonzero(d > 1e-9)[0]
    print("pairs moving in the last step:", mov.tolist()[:40], "their |dx|", d[mov][:10])
    print("pairs differing from two steps back:", np.nonzero(d3 > 1e-9)[0].tolist()[:40])

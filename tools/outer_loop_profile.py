"""cProfile of algorithm.solve on a synthetic_pairs net on the HIP engine (developer aid): where the host time of the outer
loop goes.  usage: python tools/outer_loop_profile.py [pairs] [n] [m]"""
import cProfile, os, pstats, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import qpn_amd
from qpn_amd import algorithm, examples

warnings.simplefilter("ignore")
pairs, n, m = (int(v) for v in (sys.argv[1:4] + ["250", "32", "32"][len(sys.argv) - 1:]))
eng = qpn_amd.default_engine(0)
net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
# per call of the node solve: shape, seconds, how many records the fused kernels declined (status -1 is internal: the count
# shows up as the general kernel's share -- here only shape and time)
import collections, numpy as np
shapes = collections.defaultdict(lambda: [0, 0.0, 0])
_orig = eng.solve_nodes
def _wrapped(Qc, Rc, qd, Ac, Bc, l, u, w, **kw):
    t0 = time.time()
    out = _orig(Qc, Rc, qd, Ac, Bc, l, u, w, **kw)
    key = (np.shape(qd)[1], np.shape(l)[1])
    e = shapes[key]; e[0] += 1; e[1] += time.time() - t0; e[2] += np.shape(qd)[0]
    return out
eng.solve_nodes = _wrapped
pr = cProfile.Profile()
t = time.time()
pr.enable()
ret = algorithm.solve(net, engine=eng)
pr.disable()
print("solved", ret["solved"], ret.get("error"), "%.1fs" % (time.time() - t))
print({k: round(v, 3) for k, v in eng.seconds.items()})
print(dict(eng.calls))
for key, (cnt, sec, recs) in sorted(shapes.items()):
    print(f"solve_nodes (n, m) = {key}: {cnt} calls, {recs} records, {sec:.3f} s")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(40)

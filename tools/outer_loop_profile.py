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
pr = cProfile.Profile()
t = time.time()
pr.enable()
ret = algorithm.solve(net, engine=eng)
pr.disable()
print("solved", ret["solved"], ret.get("error"), "%.1fs" % (time.time() - t))
print({k: round(v, 3) for k, v in eng.seconds.items()})
print(dict(eng.calls))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(40)

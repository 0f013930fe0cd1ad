"""Developer probe: does the memory-side cache (256 MB) hold part of the records from one sweep to the next, and does the
ORDER in which a sweep takes the nodes decide what the next sweep's first round finds there?  10 000 nodes x 32 vars
(348 MB of records: more than the cache, so a cyclic pass in the same order is the LRU worst case).  Two static orders on
alternate sweeps, both with the shortest nodes last:  X = [A | B | C],  Y = [B | A | C]  (A, B: 4 096 nodes each).
Usage: python tools/mall_probe.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cnt, n, m, p = 10000, 32, 32, 8
eng = qpn_amd.Engine(0)
dev = "cuda:0"
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m, p)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
rec = (t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u))
w0 = synthetic.shared_params(p)
ring = t(w0[None, :] + 0.25 * np.random.Generator(np.random.Philox(key=[synthetic.SEED, 2 ** 41])).standard_normal((64, p)))
h = eng.upload_nodes(*rec)
x = torch.zeros((cnt, n), dtype=torch.float64, device=dev)
out = None
h.set_schedule(0)
pv = []
for k in range(64):
    out = h.solve(ring[k], out=out, x_out=x)
    pv.append(out["pivots"].cpu().numpy().copy())
mean_piv = np.stack(pv).mean(0)
lpt = np.argsort(-mean_piv, kind="stable").astype(np.int32)


def run(label, orders, fixed=False):
    dord = [torch.tensor(o, dtype=torch.int32, device=dev) for o in orders]
    for k in range(100):
        eng.set_node_order(dord[k % len(dord)])
        h.solve(ring[0 if fixed else k % 64], out=out, x_out=x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(steps):
        eng.set_node_order(dord[k % len(dord)])
        h.solve(ring[0 if fixed else k % 64], out=out, x_out=x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label:72s} {ms*1e3:8.2f} us  {cnt/ms/1e3:7.2f} M/s", flush=True)


Aa, Bb, Cc = lpt[:4096], lpt[4096:8192], lpt[8192:]
X = lpt
Y = np.concatenate([Bb, Aa, Cc]).astype(np.int32)
rev = lpt[::-1].copy()
for fixed in (False, True):
    tag = "fixed w" if fixed else "ring"
    run(f"X, X (longest-first every sweep; + the 40 KB order copy), {tag}", [X, X], fixed)
    run(f"X, Y ([A|B|C] then [B|A|C]), {tag}", [X, Y], fixed)
    run(f"Y, Y, {tag}", [Y, Y], fixed)
    # what the last sweep read last comes first: [A|B|C] then [C'|B|A]-like would put the short nodes first; instead keep
    # the tail short and only rotate the two full rounds by halves
    Z = np.concatenate([Bb[2048:], Aa, Bb[:2048], Cc]).astype(np.int32)
    run(f"X, Z ([A|B|C] then [B2|A|B1|C]), {tag}", [X, Z], fixed)
    run(f"X, reverse (upper bound on cache reuse, bad tail), {tag}", [X, rev], fixed)

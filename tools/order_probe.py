"""Developer probe: how the order in which the fused kernel's wavefronts take the nodes changes the sweep time
(10 000 nodes, ring of 64 parameter vectors, resident records).  Usage: python tools/order_probe.py [steps]"""
import os, sys, time
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
for k in range(900):
    out = h.solve(ring[k % 64], out=out, x_out=x)
torch.cuda.synchronize()
print("info", h.info())
# pivot statistics over the ring
pv = []
h.set_schedule(0)
for k in range(64):
    out = h.solve(ring[k], out=out, x_out=x)
    pv.append(out["pivots"].cpu().numpy().copy())
pv = np.stack(pv)
print("pivots mean %.2f; per-node std over the ring %.2f; std of per-node mean %.2f" % (pv.mean(), pv.std(0).mean(), pv.mean(0).std()))
mean_piv = pv.mean(0)


def timed(label, fixed=False):
    for k in range(50):
        h.solve(ring[0 if fixed else k % 64], out=out, x_out=x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(steps):
        h.solve(ring[0 if fixed else k % 64], out=out, x_out=x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    print(f"{label:58s} {ms*1e3:8.2f} us  {cnt/ms/1e3:7.2f} M/s  frac {34848*cnt/(ms*1e-3)/8e12:.4f}", flush=True)
    return ms


h.set_schedule(0); eng.set_node_order(None)
timed("natural order")
timed("natural order, fixed w", fixed=True)
h.set_schedule(16)
timed("handle schedule: longest-first, refresh/16 (ring)")
timed("handle schedule: longest-first, refresh/16 (fixed w)", fixed=True)
h.set_schedule(1)
timed("handle schedule: refresh every sweep (ring)")
h.set_schedule(0)
lpt = np.argsort(-mean_piv, kind="stable").astype(np.int32)
eng.set_node_order(lpt)
timed("static order: longest-first by MEAN pivots over the ring")
# first round mixed: slot s of the resident set (blocks 1024 s .. 1024 s + 1023) gets long, short, long, short
def mixed(order, pattern):
    first = order[:4096]; rest = order[4096:]
    longs = first[:2048]; shorts_pool = order  # take the shortest 2048 of ALL nodes into the first round
    a = order[:2048]; b = order[-2048:]; mid = order[2048:-2048]
    slots = {"LSLS": [a[:1024], b[:1024], a[1024:], b[1024:]], "LLSS": [a[:1024], a[1024:], b[:1024], b[1024:]],
             "SLSL": [b[:1024], a[:1024], b[1024:], a[1024:]]}[pattern]
    return np.concatenate(slots + [mid]).astype(np.int32)
for pat in ("LSLS", "LLSS", "SLSL"):
    eng.set_node_order(mixed(lpt, pat))
    timed(f"first round {pat} (2048 longest + 2048 shortest), rest longest-first")
rng = np.random.default_rng(0)
eng.set_node_order(rng.permutation(cnt).astype(np.int32))
timed("random order")
# interleave: every 4th block long ... (each SIMD's 4 slots get a spread of lengths in every round)
q = lpt.reshape(4, -1)          # quartiles by length
eng.set_node_order(np.stack([q[0], q[3], q[1], q[2]], 1).reshape(-1).astype(np.int32))
timed("interleaved quartiles (L,S,ML,MS) block by block")
eng.set_node_order(np.concatenate([q[0], q[2], q[1], q[3]]).astype(np.int32))
timed("quartile order Q1,Q3,Q2,Q4")

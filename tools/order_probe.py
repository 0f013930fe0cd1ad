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


def planned_order(dur, P=4096, mode="closed"):
    """Static plan (slot -> list of jobs, longest first inside a slot) from predicted durations, returned as the
    dispatch order = jobs by planned start time.  closed: slots with q jobs take the largest items (snake pairs),
    slots with q + 1 the rest in balanced groups; ffd: first-fit-decreasing into P bins of the smallest capacity
    that works."""
    B = len(dur)
    desc = np.argsort(-dur, kind="stable")
    a = dur[desc]
    slots = [[] for _ in range(P)]
    if mode == "closed":
        q, r = divmod(B, P)
        assert q == 2
        k3, k2 = r, P - r
        for j in range(k2):
            slots[j] = [desc[j], desc[2 * k2 - 1 - j]]
        c = desc[2 * k2:]
        h2 = k3 // 2
        for j in range(k3):
            if j < h2: s_, t_ = k3 - 1 - 2 * j, j + h2
            else: jp = j - h2; s_, t_ = k3 - 2 - 2 * jp, jp
            s_ = min(max(s_, 0), k3 - 1); t_ = min(t_, k3 - 1)
            slots[k2 + j] = [c[j], c[k3 + s_], c[2 * k3 + t_]]
        used = np.zeros(B, bool)
        for sl in slots:
            for v in sl: used[v] = True
        missing = list(np.nonzero(~used)[0])          # odd k3: the formulas may leave a few out / double: repair
        seen = set(); fixed = []
        for sl in slots:
            out_ = []
            for v in sl:
                if v in seen: out_.append(missing.pop())
                else: out_.append(v)
                seen.add(out_[-1])
            fixed.append(out_)
        slots = fixed
    else:
        import bisect
        lo_, hi_ = a.sum() / P, a.sum() / P * 1.5
        def pack(T):
            loads = []; bins = []
            # first fit decreasing with a sorted list of (remaining capacity) is best-fit; use plain first-fit over open bins
            rem = np.full(P, T); cnt_ = 0; res = [[] for _ in range(P)]
            ptr = 0
            for idx, d in zip(desc, a):
                # first bin with rem >= d
                cand = np.nonzero(rem[:max(cnt_, 1) + 1 if cnt_ < P else P] >= d)[0]
                if len(cand) == 0: return None
                b_ = cand[0]
                if b_ >= P: return None
                rem[b_] -= d; res[b_].append(idx); cnt_ = max(cnt_, b_ + 1)
            return res
        for _ in range(12):
            mid_ = 0.5 * (lo_ + hi_)
            r_ = pack(mid_)
            if r_ is None: lo_ = mid_
            else: hi_ = mid_; slots = r_
    start = np.zeros(B); slot_of = np.zeros(B, int)
    sums = []
    for si, sl in enumerate(slots):
        sl = sorted(sl, key=lambda v: -dur[v]); t_ = 0.0
        for v in sl:
            start[v] = t_; slot_of[v] = si; t_ += dur[v]
        sums.append(t_)
    sums = np.array(sums)
    order = np.lexsort((slot_of, start)).astype(np.int32)
    return order, sums


if len(sys.argv) > 2 and sys.argv[2] == "plan":
    h.set_schedule(16)
    timed("handle schedule: longest-first, refresh/16 (ring)")
    timed("handle schedule: longest-first, refresh/16 (fixed w)", fixed=True)
    h.set_schedule(0)
    out = h.solve(ring[0], out=out, x_out=x); torch.cuda.synchronize()
    piv0 = out["pivots"].cpu().numpy().astype(np.float64)
    for c0 in (12.0, 20.0, 26.0):
        for mode in ("closed", "ffd"):
            order, sums = planned_order(piv0 - c0, mode=mode)
            assert sorted(order.tolist()) == list(range(cnt))
            eng.set_node_order(order)
            timed(f"planned starts ({mode}, dur = piv - {c0:.0f}), exact counts, fixed w  [slot sums {sums.min():.0f}..{sums.max():.0f}]", fixed=True)
            order, sums = planned_order(mean_piv - c0, mode=mode)
            eng.set_node_order(order)
            timed(f"planned starts ({mode}, dur = piv - {c0:.0f}), ring-mean counts, ring")
    eng.set_node_order(np.argsort(-piv0, kind="stable").astype(np.int32))
    timed("static longest-first, exact counts, fixed w", fixed=True)
    sys.exit(0)

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

#!/usr/bin/env python3
"""Developer aid: very large batches (index arithmetic beyond 2^31 bytes / elements): 300 000 nodes of n = m = 32 (10 GB of records)
and 40 000 of n = m = 64 through the resident-records route; all solved, residuals <= 1e-8, a random subset against the oracle,
and the rows of the last nodes equal to the same nodes solved on their own.  Usage: python tools/big_batch_check.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from oracle import binding
eng = qpn_amd.Engine(0)
dev = "cuda:0"
SIZES = ((32, 32, int(os.environ.get("NODES32", "300000")), 30), (64, 64, 40_000, 10))
for n, m, cnt, reps in SIZES:
    # a block of 1 000 distinct nodes repeated (generating 300 000 distinct ones on the host would take minutes), the last block distinct
    base = P.synth_nodes(70_000 + n, 1000, n, m)
    last = P.synth_nodes(71_000 + n, 1000, n, m)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    recs = []
    for k, (b, l_) in enumerate(zip(base, last)):
        b = colmajor(b) if k in (0, 1, 3, 4) else b
        l_ = colmajor(l_) if k in (0, 1, 3, 4) else l_
        tb, tl = t(b), t(l_)
        recs.append(torch.cat([tb.repeat((cnt // 1000 - 1,) + (1,) * (tb.dim() - 1)), tl], dim=0).contiguous())
    w = t(P.shared_params())
    print(f"n=m={n}: {cnt} nodes, records {sum(r.numel() for r in recs) * 8 / 2**30:.1f} GiB", flush=True)
    h = eng.upload_nodes(*recs)
    out = h.solve(w)
    torch.cuda.synchronize()
    st = out["status"].cpu().numpy(); res = out["resid"].cpu().numpy()
    assert (st == 1).all() and res.max() <= 1e-8, (int((st != 1).sum()), float(res.max()))
    z = out["z"]
    # the last 1 000 nodes on their own
    h2 = eng.upload_nodes(*[r[-1000:].contiguous() for r in recs]); o2 = h2.solve(w); torch.cuda.synchronize()
    assert torch.equal(o2["z"], z[-1000:]) and torch.equal(o2["active"], out["active"][-1000:])
    # every repetition of the first block gives the same rows
    assert torch.equal(z[:1000], z[1000 * (cnt // 1000 - 2):1000 * (cnt // 1000 - 1)])
    idx = np.sort(np.random.default_rng(1).choice(1000, 40, replace=False))
    M, q, lo, hi, kd = P.reduced_blocks(*[a[idx] for a in last], P.shared_params())
    rc = binding.solve_avi_batch(M, q, lo, hi, kind=kd)
    zz = z[-1000:].cpu().numpy()[idx]
    assert np.array_equal(out["active"][-1000:].cpu().numpy()[idx], rc["active"]) and np.max(np.abs(zz - rc["z"])) <= 1e-9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out = h.solve(w, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"   all solved, max residual {res.max():.1e}, subset == oracle; {ms:.2f} ms per sweep = {cnt / ms / 1e3:.1f} M solves/s", flush=True)
    h.close(); h2.close(); del recs, out, z
    torch.cuda.empty_cache()

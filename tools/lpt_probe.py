#!/usr/bin/env python3
"""Developer probe: launch time of the fused node kernel under different wavefront -> node schedules
(qpn_set_node_order).  Results are identical under every schedule; only the launch shape changes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
cnt, n, m = int(os.environ.get("CNT", "10000")), 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w)]
x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
def run(order, tag):
    eng.set_node_order(None if order is None else order.astype(np.int32))
    out = None
    for _ in range(5): out = eng.solve_nodes(*args, out=out, x_out=x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(60): out = eng.solve_nodes(*args, out=out, x_out=x)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:44s} {e0.elapsed_time(e1) / 60 * 1000:7.1f} us/step")
    return out["pivots"].cpu().numpy()
piv = run(None, "natural order")
lpt = np.argsort(-piv, kind="stable")
run(lpt, "longest first (LPT)")
S = 4096
# first round: 3/4 longest + 1/4 shortest, interleaved; rest LPT
k = S // 4
short = lpt[-k:][::-1]; long_ = lpt[:S - k]; rest = lpt[S - k:-k]
first = np.empty(S, dtype=np.int64); first[0::4] = short; idx = np.ones(S, bool); idx[0::4] = False; first[idx] = long_
run(np.concatenate([first, rest]), "first round: every 4th slot a SHORT node")
k = S // 2
short = lpt[-k:][::-1]; long_ = lpt[:S - k]; rest = lpt[S - k:-k]
first = np.empty(S, dtype=np.int64); first[0::2] = short; first[1::2] = long_
run(np.concatenate([first, rest]), "first round: every 2nd slot a SHORT node")
# LPT within rounds but rounds reversed (short round first)
run(np.concatenate([lpt[2 * S:], lpt[S:2 * S], lpt[:S]]) if cnt > 2 * S else lpt, "rounds reversed (short round first)")
# block-cyclic: position i gets lpt[(i % S) * ceil(cnt/S) + i // S] (each slot gets a long, a medium and a short one)
rounds = -(-cnt // S)
grid = np.full((S, rounds), -1, dtype=np.int64); grid.flat[:cnt] = lpt        # row-major: slot s holds lpt[s*rounds + j]
bc = grid.T.reshape(-1); bc = bc[bc >= 0]
run(bc, "block-cyclic (each slot long+medium+short)")
run(np.concatenate([lpt[-S:], lpt[:-S]]), "shortest 4096 first, then the rest longest-first")
run(np.concatenate([lpt[-S:][::-1], lpt[:-S]]), "shortest 4096 first (ascending), then LPT")
mid = lpt[S:2 * S]
run(np.concatenate([mid, lpt[:S], lpt[2 * S:]]), "middle 4096 first, then longest, then shortest")
run(np.concatenate([lpt[S // 2: S // 2 + S], lpt[:S // 2], lpt[S // 2 + S:]]), "offset window first, then LPT")
eng.set_node_order(None)

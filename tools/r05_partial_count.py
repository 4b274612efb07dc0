import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import juliagrid.jl_amd as jg
s = jg.powerSystem(load_case("case_ACTIVSg10k")); jg.acModel_(s)
Y = s.model.ac.nodalMatrix
policy = 1 | 4 | (1 << 49) | 8 | (1 << 50) | (47 << 16) | (127 << 24) | (12 << 4)
plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=policy)
e_row, e_col, e_src, t_ptr = (plan.get(k) for k in ("e_row", "e_col", "e_src", "t_ptr"))
t_a, t_d, t_b = plan.get("t_a"), plan.get("t_d"), plan.get("t_b")
hdr, data, launches, task_of, misc = plan.top_tables()
print("tasks", hdr.shape[0], "pivots in top", int((task_of >= 0).sum()), "launches", launches[:, :4].tolist(), "misc", misc.tolist())
owner = np.minimum(e_row, e_col)
top_ent = task_of[owner] >= 0
ent = np.repeat(np.arange(e_row.size), np.diff(t_ptr))
piv = e_col[t_a]                       # pivot k of a term
sel = top_ent[ent] & (task_of[piv] < 0)
print("terms of bottom pivots into top-owned entries", int(sel.sum()), "entries touched", np.unique(ent[sel]).size, "top-owned entries", int(top_ent.sum()))
ops = np.unique(np.concatenate([t_a[sel], t_d[sel], t_b[sel]]))
own = np.unique(ent[sel])
own_read = int((e_src[own] >= 0).sum())
blk = 2048 * 8
print("distinct operand blocks", ops.size, "own blocks", own.size, "(read", own_read, ")")
print("compulsory GB: operands %.3f + own %.3f = %.3f" % (ops.size * blk / 1e9, (own.size + own_read) * blk / 1e9, (ops.size + own.size + own_read) * blk / 1e9))
print("every term its own operand block (no reuse): %.3f GB; three blocks per term %.3f GB" % (sel.sum() * blk / 1e9, 3 * sel.sum() * blk / 1e9))
# emission instead: one store + one load per term
print("emission (store + extend-add load per term): %.3f GB" % (2 * sel.sum() * blk / 1e9))
# per contributing subtree: subtree root = bottom pivot whose parent is a top pivot
u_ptr, u_col = plan.get("u_ptr"), plan.get("u_col")
n = Y.n
parent = np.array([u_col[u_ptr[k]] if u_ptr[k + 1] > u_ptr[k] else -1 for k in range(n)])
root = np.full(n, -1)
for k in range(n - 1, -1, -1):
    if task_of[k] >= 0: continue
    p = parent[k]
    root[k] = k if (p < 0 or task_of[p] >= 0) else root[p]
pairs = np.unique(np.stack([root[piv[sel]], ent[sel]], axis=1), axis=0)
print("(subtree, entry) pairs", pairs.shape[0], "-> per-subtree update matrices: store + load %.3f GB" % (2 * pairs.shape[0] * blk / 1e9), "subtrees", np.unique(pairs[:, 0]).size)

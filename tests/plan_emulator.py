"""numpy replay of the static block-LU / triangular-solve schedules (TEST INFRASTRUCTURE).

Mirrors what the HIP kernels k_lu / k_fwd / k_bwd do for ONE scenario, walking the schedule exactly
as the device does (launch -> task -> step -> item) and asserting that every value an item reads
was produced in an earlier launch or an earlier step of the same task: a race detector for the
schedule.  Not used by the product.
"""
import numpy as np


def _walk(sch):
    """yield (launch_index, task, step, item)"""
    for li, (t0, t1, _w, _wpi, _chunk, _ib, _ie, _fused) in enumerate(sch["launches"]):
        for t in range(t0, t1):
            for s in range(sch["task_ptr"][t], sch["task_ptr"][t + 1]):
                for idx in range(sch["step_ptr"][s], sch["step_ptr"][s + 1]):
                    yield li, t, s, int(sch["items"][idx])


def dfactor(D):
    """2x2 LU with in-block partial pivoting, stored like the device: [1/u11, u12, l (+4 if swapped), 1/u22]."""
    a, b, c, d = D[0, 0], D[0, 1], D[1, 0], D[1, 1]
    sw = abs(c) > abs(a)
    if sw:
        a, b, c, d = c, d, a, b
    l = c / a
    u22 = d - l * b
    return np.array([[1.0 / a, b], [l + 4.0 if sw else l, 1.0 / u22]])


def dsolve(F, R):
    """D^-1 R for a stored factor block F; R is a 2-vector or a 2x2 block (column by column)."""
    sw = F[1, 0] > 2.0
    l = F[1, 0] - 4.0 if sw else F[1, 0]
    R = np.asarray(R, dtype=float)
    a, b = (R[1], R[0]) if sw else (R[0], R[1])
    y2 = (b - l * a) * F[1, 1]
    y1 = (a - F[0, 1] * y2) * F[0, 0]
    return np.array([y1, y2])


class Replay:
    def __init__(self, plan):
        self.p = plan
        g = plan.get
        self.perm, self.e_row, self.e_col, self.e_src = g("perm"), g("e_row"), g("e_col"), g("e_src")
        self.t_ptr, self.t_a, self.t_d, self.t_b, self.diag = g("t_ptr"), g("t_a"), g("t_d"), g("t_b"), g("diag")
        self.l_ptr, self.l_ent, self.l_col = g("l_ptr"), g("l_ent"), g("l_col")
        self.u_ptr, self.u_ent, self.u_col = g("u_ptr"), g("u_ent"), g("u_col")
        self.nE = self.e_row.size
        self.n = plan.n

    @staticmethod
    def _visible(stamp, src, li, t, s):
        """src produced strictly before (launch li, task t, step s)?"""
        if stamp[src] is None:
            return False
        l2, t2, s2 = stamp[src]
        return l2 < li or (l2 == li and t2 == t and s2 < s)

    def factor(self, A, rhs):
        """A: [nnz_blocks,2,2] caller CSR order; rhs [n,2] original order.
        Returns X [nE,2,2] (U, unscaled Lh, factored diagonal blocks) and y [n,2] (pivot order)."""
        nE = self.nE
        X = np.zeros((nE, 2, 2))
        Y = np.zeros((self.n, 2))
        stamp = [None] * (nE + self.n)          # entries, then rhs rows
        for li, t, s, it in _walk(self.p.schedule("fact")):
            assert stamp[it] is None, f"item {it} scheduled twice"
            vis = lambda src: self._visible(stamp, src, li, t, s)
            if it < nE:
                e = it
                acc = A[self.e_src[e]].copy() if self.e_src[e] >= 0 else np.zeros((2, 2))
                for k in range(self.t_ptr[e], self.t_ptr[e + 1]):
                    a, d, b = self.t_a[k], self.t_d[k], self.t_b[k]
                    assert vis(a) and vis(d) and vis(b), "LU schedule race"
                    acc -= X[a] @ dsolve(X[d], X[b])
                X[e] = dfactor(acc) if self.e_row[e] == self.e_col[e] else acc
            else:
                k = it - nE
                y = rhs[self.perm[k]].copy()
                for p in range(self.l_ptr[k], self.l_ptr[k + 1]):
                    c = self.l_col[p]
                    assert vis(self.l_ent[p]) and vis(self.diag[c]) and vis(nE + c), "forward schedule race"
                    y -= X[self.l_ent[p]] @ dsolve(X[self.diag[c]], Y[c])
                Y[k] = y
            stamp[it] = (li, t, s)
        assert all(st is not None for st in stamp), "items missing from the factorisation schedule"
        return X, Y

    def backsolve(self, X, Y):
        W = Y.copy()
        stamp = [None] * self.n
        out = np.zeros((self.n, 2))
        for li, t, s, k in _walk(self.p.schedule("bwd")):
            y = W[k].copy()
            for p in range(self.u_ptr[k], self.u_ptr[k + 1]):
                assert self._visible(stamp, self.u_col[p], li, t, s), "bwd schedule race"
                y -= X[self.u_ent[p]] @ W[self.u_col[p]]
            W[k] = dsolve(X[self.diag[k]], y)
            out[self.perm[k]] = W[k]
            stamp[k] = (li, t, s)
        assert all(st is not None for st in stamp)
        return out


def block_jacobian_from_csc(n, ycolptr, yrowval, typ, pq, pvpq, jcolptr, jrowval, jnz):
    """Scatter a reference-ordered Jacobian (CSC, rows/cols pvpq then pq) into n x n 2x2 blocks with the
    Ybus pattern in ROW-CSR order (= transposed CSC pointer order), identity padding for PV / slack."""
    import scipy.sparse as sp
    dim = jcolptr.size - 1
    J = sp.csc_matrix((jnz, jrowval - 1, jcolptr - 1), shape=(dim, dim)).tocsr()
    rowptr = ycolptr - 1          # symmetric pattern: CSR of Y == CSC pointers of the transpose
    col = yrowval - 1
    A = np.zeros((col.size, 2, 2))
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            j = col[p]
            ri = [pvpq[i] - 1 if pvpq[i] else -1, pq[i] - 1 if pq[i] else -1]
            cj = [pvpq[j] - 1 if pvpq[j] else -1, pq[j] - 1 if pq[j] else -1]
            for a in range(2):
                for b in range(2):
                    if ri[a] >= 0 and cj[b] >= 0:
                        A[p, a, b] = J[ri[a], cj[b]]
                    elif i == j and a == b:
                        A[p, a, b] = 1.0
    return rowptr.astype(np.int32), col.astype(np.int32), A

"""numpy replay of the static block-LU / triangular-solve replay tables (TEST INFRASTRUCTURE).

Mirrors what the HIP kernels (k_fact_level / k_fact_top / k_bwd_level) do for ONE scenario:
it walks the very tables the device reads (segment -> chunk -> wave -> 64-byte record, jg_symbolic.hpp)
and asserts that every value a record reads was produced in an EARLIER dependency level (all items of a
level run concurrently on the device): a race detector for the schedule.  Not used by the product.
"""
import numpy as np


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


def dsolve_right(F, r):
    """r D^-1 for a row vector r (what a task stages for its row items: Lh(p,k) D(k)^-1 row by row)."""
    sw = F[1, 0] > 2.0
    l = F[1, 0] - 4.0 if sw else F[1, 0]
    z1 = r[0] * F[0, 0]
    z2 = (r[1] - z1 * F[0, 1]) * F[1, 1]
    y1 = z1 - z2 * l
    return np.array([z2, y1]) if sw else np.array([y1, z2])


class Replay:
    """inplace=True mirrors policy bit 0: the caller's blocks already sit in the factor storage and
    off-diagonal entries without update terms are not scheduled at all."""

    def __init__(self, plan, inplace=False, symmetric=False, prefactor=False, producer=True, jordan=False, single=False):
        """prefactor mirrors policy bit 2: the pivots nobody updates (pre_pivot) are level-0 items -- finished by the producer
        (producer=True: their diagonal blocks arrive factorised, their rhs rows are in place) or by the plan's PRE tables.
        jordan mirrors Engine::jordan on a plan with policy bit 49: the top tasks eliminate above the diagonal too and leave Jordan rows
        behind the factor entries; the backward sweep walks the "bwdj" tables."""
        self.p = plan
        self.jordan = jordan
        self.single = single            # policy bit 60: the items below the top come from the thread-per-item tables of a single instance (k_fact1_bottom / k_fact1_partial)
        self.prefactor, self.producer = prefactor, producer
        g = plan.get
        self.perm, self.e_row, self.e_col, self.e_src = g("perm"), g("e_row"), g("e_col"), g("e_src")
        self.t_ptr, self.diag = g("t_ptr"), g("diag")
        self.l_ptr, self.u_ptr = g("l_ptr"), g("u_ptr")
        self.nE = self.e_row.size
        self.n = plan.n
        self.inplace = inplace
        self.symmetric = symmetric      # policy bit 1: only the upper entries are scheduled (LDL' through transposed reads)
        self.fseg, self.frec = plan.replay_tables("fact")
        self.bseg, self.brec = plan.replay_tables("bwd")
        self.thdr, self.tdata, self.tlaunch, self.task_of, self.tinfo = plan.top_tables()
        self.n_jordan = int(self.tinfo[7]) if self.tinfo.size > 7 else 0
        self.fact_tasks = bool(self.tinfo.size > 8 and int(self.tinfo[8]))
        if jordan:
            assert self.tinfo.size > 7 and int(self.tinfo[6]) == 1, "not a Jordan plan"
            self.bseg, self.brec = plan.replay_tables("bwdj")

    @staticmethod
    def _waves(seg, rec):
        """yield (level, wave-group key, sub, [records of the wave]) in device order"""
        for si, (base, nchunks, wpi, rpw, level, _last, _items, _pad) in enumerate(seg):
            for c in range(nchunks):
                for w in range(16):
                    r0 = base + (c * 16 + w) * rpw
                    yield level, (si, c, w // wpi), w % wpi, rec[r0:r0 + rpw]

    def factor(self, A, rhs):
        """A: [nnz_blocks,2,2] caller CSR order; rhs [n,2] original order.
        Returns X [nE,2,2] (U, unscaled Lh, factored diagonal blocks) and y [n,2] (pivot order)."""
        nE = self.nE
        X = np.zeros((nE + self.n_jordan, 2, 2))                # Jordan rows sit behind the factor entries
        X[nE:] = np.nan
        Y = np.zeros((self.n, 2))
        level_of = np.full(nE + self.n, -1)      # level at which an entry / rhs row becomes final (-1: never yet)
        partial = np.zeros(nE + self.n, dtype=bool)          # task-owned items that hold their bottom terms only
        part_level = np.zeros(nE + self.n, dtype=int)
        if self.inplace:
            has = self.e_src >= 0
            X[:nE][has] = A[self.e_src[has]]
            work = np.diff(self.t_ptr)
            untouched = has & (work == 0) & (self.e_row != self.e_col)
            if self.task_of.size:                           # task-owned entries become final inside their task
                untouched &= self.task_of[np.minimum(self.e_row, self.e_col)] < 0
            level_of[:nE][untouched] = 0         # already final before the first level
        pre = self.p.get("pre_pivot").astype(bool) if self.prefactor else np.zeros(self.n, dtype=bool)
        assert self.prefactor or not pre.any()
        if self.prefactor:
            work = np.diff(self.t_ptr)
            lcount = np.diff(self.l_ptr)
            assert np.array_equal(pre, (work[self.diag] == 0) & (lcount == 0) & ((self.task_of < 0) if self.task_of.size else True)), \
                "level 0 = the pivots whose diagonal block and rhs row receive no term"
            if self.producer:                    # what the Jacobian assembly does for these pivots
                for k in np.flatnonzero(pre):
                    X[self.diag[k]] = dfactor(X[self.diag[k]])
                    Y[k] = rhs[self.perm[k]]
                    level_of[self.diag[k]] = 0
                    level_of[nE + k] = 0
        acc, meta = {}, {}

        def flush():                              # the level is complete: its items become final together
            for key, (lev, kind, ident) in meta.items():
                v = acc[key]
                if kind == 3:
                    assert level_of[nE + ident] < 0 and not partial[nE + ident], f"rhs row {ident} scheduled twice"
                    Y[ident] = v
                    if self.task_of.size and self.task_of[ident] >= 0:
                        partial[nE + ident] = True
                        part_level[nE + ident] = lev
                    else:
                        level_of[nE + ident] = lev
                else:
                    assert level_of[ident] < 0 and not partial[ident], f"entry {ident} scheduled twice"
                    X[ident] = dfactor(v) if kind == 2 else v
                    if self._top_owned(ident):                   # bottom terms only, stored raw: final inside its task
                        assert kind != 2, "a task-owned diagonal block must not be factorised by a level item"
                        partial[ident] = True
                        part_level[ident] = lev
                    else:
                        level_of[ident] = lev
            acc.clear()
            meta.clear()

        current = None
        tables = [] if self.fact_tasks else [(lv, k, sb, rc) for lv, k, sb, rc in self._waves(self.fseg, self.frec)]
        if self.single:
            tables = self._single_items()
        if self.prefactor and not self.producer:             # plain blocks: the PRE tables run ahead of level 1, as level 0
            pseg, prec = self.p.replay_tables("pre")
            tables = [(0, ("pre",) + k, sb, rc) for lv, k, sb, rc in self._waves(pseg, prec)] + tables
        for level, key, sub, recs in tables:
            if level != current:
                assert current is None or level > current, "segments out of level order"
                flush()
                current = level
            kind, ident, src = int(recs[0][0]), int(recs[0][1]), int(recs[0][2])
            if kind < 0:
                continue
            if sub == 0:
                assert key not in meta, "two leaders in one wave group"
                meta[key] = (level, kind, ident)
                if kind == 3:
                    acc[key] = np.array(rhs[src], dtype=float)
                elif src >= 0:
                    acc[key] = (X[src] if self.inplace else A[src]).copy()
                    assert not self.inplace or src == ident
                else:
                    acc[key] = np.zeros((2, 2))
            part = np.zeros(2) if kind == 3 else np.zeros((2, 2))
            for r in recs:
                assert (int(r[0]), int(r[1])) == (kind, ident), "continuation record of another item"
                for t in range(int(r[3])):
                    a, d, b = (int(v) for v in r[4 + 3 * t: 7 + 3 * t])
                    tr, a = a >> 30, a & ((1 << 30) - 1)          # symmetric plans read Lh(i,k) as U(k,i)'
                    assert 0 <= level_of[a] < level and 0 <= level_of[d] < level, "LU schedule race"
                    La = X[a].T if tr else X[a]
                    if kind == 3:
                        assert 0 <= level_of[nE + b] < level, "forward schedule race"
                        part -= La @ dsolve(X[d], Y[b])
                    else:
                        assert 0 <= level_of[b] < level, "LU schedule race"
                        part -= La @ dsolve(X[d], X[b])
            acc[key] = acc[key] + part            # the leader (sub 0) comes first and seeds the accumulator
        flush()
        if self.fact_tasks:                                   # policy bit 50: the same items as TASKS (jg_symbolic.hpp, k_fact_task)
            for si, (base, ntasks, spw, rpw, level, _last, _items, _pad) in enumerate(self.fseg):
                if level != current:
                    assert current is None or level > current, "segments out of level order"
                    flush()
                    current = level
                for c in range(ntasks):
                    self._task(si, c, self.frec[base + c * 8 * rpw: base + (c + 1) * 8 * rpw].reshape(8, rpw, 16), spw, level, A, rhs, X, Y, level_of, acc, meta)
            flush()
            terms_seen = int((self.frec[self.frec[:, 0] & 7 != 7, 3] & 0xff).sum())
        elif self.single:
            terms_seen = sum(int(r[3]) for _lv, _k, _sb, rc in tables for r in rc)
        else:
            terms_seen = 0
            for r in self.frec:
                if r[0] >= 0:
                    terms_seen += int(r[3])
        terms_seen += self._top_tasks(X, Y, level_of, partial, part_level, (current or 0))
        work = np.diff(self.t_ptr)
        if self.symmetric:
            lower = self.e_row > self.e_col
            assert (level_of[:nE][lower] <= 0).all(), "a symmetric plan must not schedule the lower entries"
            # inside a task the elimination runs on the full (mirrored) front: both triangles of its terms are executed
            in_task = self.task_of[self.e_col[self.p.get("t_a")]] >= 0 if self.task_of.size else np.zeros(int(self.t_ptr[-1]), dtype=bool)
            ent = np.repeat(np.arange(nE), work)
            assert terms_seen == int((~lower[ent] | in_task).sum()) + int(self.l_ptr[-1])
            assert (level_of[:nE][~lower] >= 0).all() and (level_of[nE:] >= 0).all()
        else:
            assert terms_seen == int(self.t_ptr[-1]) + int(self.l_ptr[-1]), "update terms lost or duplicated in the records"
            assert (level_of >= 0).all(), "items missing from the factorisation schedule"
        return X, Y

    def _task(self, si, c, recs, spw, level, A, rhs, X, Y, level_of, acc, meta):
        """One factorisation TASK (jg_symbolic.hpp): recs [8 waves][rpw][16].  Staging records first (shared operands premultiplied by their
        pivot block into slots), then rounds of item records; shares of a split item meet at the end of a TK_BAR round."""
        nE = self.nE
        FIRST, LAST, BAR, SIDE, DIRECT = 8, 16, 32, 64, 128
        rpw = recs.shape[1]
        slots = {}
        assert 1 <= spw <= rpw
        for w in range(8):
            for j in range(rpw):
                r = recs[w, j]
                nst = int(r[3]) >> 8
                assert 0 <= nst <= 2 and (nst == 0 or j < spw), "staging entries sit in the first spw records of a wave"
                for u in range(nst):
                    aw, d, sl = int(r[10 + 3 * u]), int(r[11 + 3 * u]), int(r[12 + 3 * u])
                    a, tr, right = aw & ((1 << 24) - 1), (aw >> 30) & 1, (aw >> 29) & 1
                    assert sl not in slots and 0 <= sl < 30 and sl % 8 == w, "slot staged twice / by the wrong wave"
                    assert 0 <= level_of[a] < level and 0 <= level_of[d] < level, "LU schedule race (staged operand)"
                    Aa = X[a].T if tr else X[a]
                    if right:                                 # Lh(p,k) D(k)^-1: the solve runs over the ROWS of the operand
                        slots[sl] = (np.stack([dsolve_right(X[d], Aa[0]), dsolve_right(X[d], Aa[1])], axis=0), a, d, 0)
                    else:                                     # D(k)^-1 U(k,p)
                        slots[sl] = (np.stack([dsolve(X[d], Aa[:, 0]), dsolve(X[d], Aa[:, 1])], axis=1), a, d, 1)
        assert sorted(slots) == list(range(len(slots))), "slots are numbered without holes"
        state = [None] * 8                                    # per wave: (kind, ident, sub, wpi, value, first round)
        for rd in range(rpw):
            bar = [int(recs[w, rd, 0]) & BAR for w in range(8)]
            assert len(set(bar)) == 1, "TK_BAR must mark a round for every wave"
            done = {}
            for w in range(8):
                r = recs[w, rd]
                h = int(r[0])
                kind = h & 7
                if kind == 7:
                    assert state[w] is None and int(r[3]) & 0xff == 0, "idle record inside a share"
                    continue
                assert kind in (0, 2, 3)
                ident, src, nt, sub, wpi, side = int(r[1]), int(r[2]), int(r[3]) & 0xff, (h >> 8) & 7, (h >> 12) & 15, 1 if h & SIDE else 0
                assert wpi in (1, 2, 4, 8) and sub < wpi and (w - sub) % wpi == 0, "shares of an item sit in neighbouring, aligned waves"
                if h & FIRST:
                    assert state[w] is None
                    if sub == 0:
                        if kind == 3:
                            v = np.array(rhs[src], dtype=float)
                        elif src >= 0:
                            v = (X[src] if self.inplace else A[src]).copy()
                            assert not self.inplace or src == ident
                        else:
                            v = np.zeros((2, 2))
                    else:
                        v = np.zeros(2) if kind == 3 else np.zeros((2, 2))
                    state[w] = [kind, ident, sub, wpi, v]
                st = state[w]
                assert st is not None and st[:4] == [kind, ident, sub, wpi], "continuation record of another item"
                assert not (kind == 3 and side), "a rhs row is a row item"
                for t in range(nt):
                    if h & DIRECT:
                        assert t < 2
                        a, d, b = (int(v) for v in r[4 + 3 * t: 7 + 3 * t])
                        tr, a = a >> 30, a & ((1 << 30) - 1)
                        assert 0 <= level_of[a] < level and 0 <= level_of[d] < level, "LU schedule race"
                        La = X[a].T if tr else X[a]
                        if kind == 3:
                            assert 0 <= level_of[nE + b] < level, "forward schedule race"
                            st[4] = st[4] - La @ dsolve(X[d], Y[b])
                        else:
                            assert 0 <= level_of[b] < level, "LU schedule race"
                            st[4] = st[4] - La @ dsolve(X[d], X[b])
                    else:
                        assert t < 6
                        word = int(r[4 + t])
                        mem, sl = word & ((1 << 24) - 1), word >> 24
                        S, sa, sd, sside = slots[sl]
                        assert sside == side, "a row item reads a column slot"
                        if kind == 3:
                            assert 0 <= level_of[nE + mem] < level, "forward schedule race"
                            st[4] = st[4] - S @ Y[mem]
                        else:
                            assert 0 <= level_of[mem] < level, "LU schedule race"
                            st[4] = st[4] - (X[mem] @ S if side else S @ X[mem])
                if h & LAST:
                    assert wpi == 1 or bar[w], "the last round of a split item is a TK_BAR round"
                    done[w] = st
                    state[w] = None
            for w, st in sorted(done.items()):                # owners collect their shares in wave order (fixed order on the device)
                kind, ident, sub, wpi, v = st
                if sub != 0:
                    assert (w - sub) in done and done[w - sub][:2] == [kind, ident], "share without its owner in the same round"
                    continue
                for x in range(1, wpi):
                    assert (w + x) in done and done[w + x][:4] == [kind, ident, x, wpi], "shares of an item must end in the same round"
                    v = v + done[w + x][4]
                key = (si, c, w, rd)
                assert key not in meta
                meta[key] = (level, kind, ident)
                acc[key] = v
        assert all(s is None for s in state), "a share without its last record"

    def _single_items(self):
        """The thread-per-item tables as the stream factor() walks: f1 = (workgroup, level, item) with the levels replayed across all workgroups at once -- which is
        only the device's order if every operand of an item is produced by ITS workgroup (asserted here) -- then f2 (the partial sums) as one more level."""
        g = self.p.get
        info = g(85)
        assert info[0] == 1, "the plan carries no single-instance tables"
        nlev = int(info[2])
        rec, f1_first, f2_first = g(81).reshape(-1, 16), g(82), g(84)
        f1_wg = g(83).reshape(-1, nlev + 1)
        nE = self.nE

        def item(first):                                      # record `first` = the item's first record: word 3 = terms | first continuation record << 10
            nt, cont = int(rec[first][3]) & 1023, int(rec[first][3]) >> 10
            nrec = max(1, (nt + 3) // 4)
            out = np.concatenate([rec[first: first + 1], rec[cont: cont + nrec - 1]]).copy() if nrec > 1 else rec[first: first + 1].copy()
            for i, r in enumerate(out):
                r[3] = min(4, nt - 4 * i) if nt > 4 * i else 0
                assert i == 0 or (int(r[0]), int(r[1])) == (int(out[0][0]), int(out[0][1]))
            return out
        wg_of = {}                                            # item (entry, or nE + rhs row) -> workgroup
        for w in range(f1_wg.shape[0]):
            for j in range(f1_wg[w, 0], f1_wg[w, nlev]):
                r = rec[f1_first[j]]
                wg_of[int(r[1]) + (nE if r[0] == 3 else 0)] = w
        out = []
        for l in range(nlev):
            for w in range(f1_wg.shape[0]):
                for j in range(f1_wg[w, l], f1_wg[w, l + 1]):
                    recs = item(int(f1_first[j]))
                    kind = int(recs[0][0])
                    for r in recs:
                        for t in range(int(r[3])):
                            a, d, b = (int(v) for v in r[4 + 3 * t: 7 + 3 * t])
                            for op in (a, d, b + nE if kind == 3 else b):
                                assert wg_of.get(op, w) == w, "an item below the top reads an item of another workgroup"
                    out.append((l + 1, ("f1", w, j), 0, recs))
        for j, first in enumerate(f2_first):
            out.append((nlev + 1, ("f2", j), 0, item(int(first))))
        return out

    def _top_owned(self, e):
        return self.task_of.size > 0 and self.task_of[min(self.e_row[e], self.e_col[e])] >= 0

    def _top_tasks(self, X, Y, level_of, partial, part_level, bottom_levels):
        """Replays the multifrontal top (jg_symbolic.hpp, k_fact_top): per launch, per task: the entry map fills the dense
        front (+ rhs column), the children's update blocks are pulled from the stack through their inverse maps, the chain's
        pivots are eliminated densely (blocks outside the pattern are zero and stay zero), owned entries go back, the update
        block goes to the stack.  Asserts that everything a task reads exists before its launch.  Returns the update terms."""
        hdr, data, launches = self.thdr, self.tdata, self.tlaunch
        nE = self.nE
        if hdr.shape[0] == 0:
            assert not partial.any()
            return 0
        stacks = [np.full(int(self.tinfo[3 + c]), np.nan) for c in range(3)]     # one stack per interleave class (1 / 4 / 16 scenarios)
        assert int(self.tinfo[1]) == sum(int(self.tinfo[3 + c]) for c in range(3))
        stack_level = {}
        seen_tasks = 0
        terms = 0
        done_pivot = np.zeros(self.n, dtype=bool)
        prev = 0
        u_col = self.p.get("u_col")
        wgmap = self.p.get(78)
        geom_of_stack = {}                                    # stack offset of a task's update block -> log2 of its scenario interleave
        for li, (tb, ntk, cls, tlevel, grouped, wgb, nwg, _pad) in enumerate(launches):
            assert tlevel >= prev and tb == seen_tasks and cls in (2, 3, 4), "launches out of order"   # (policy bit 3: a level may take one launch per class)
            prev = tlevel
            lev = bottom_levels + tlevel
            results = []
            if grouped:                                      # every (task, scenario block) of the launch exactly once in its workgroup map
                want = sorted((ti << 8) | b for ti in range(tb, tb + ntk) for b in range(64 >> int(hdr[ti][13])))
                assert sorted(int(v) for v in wgmap[wgb: wgb + nwg]) == want, "workgroup map of a grouped launch"
            else:
                assert nwg == 0
            for ti in range(tb, tb + ntk):
                m, e, root, base, soff, nchild, piv_off, child_off, dent_off, tcls, tl, fprime, lgo, lg = (int(v) for v in hdr[ti][:14])
                jbase = int(hdr[ti][14])
                f = m + e
                if self.tinfo.size > 7 and int(self.tinfo[6]):
                    assert (jbase >= nE and jbase + m * e <= nE + self.n_jordan and lg == 0) if e > 0 else jbase == -1
                else:
                    assert jbase == -1
                assert (lg > 0) == bool(grouped) and lg in (0, 2, 4) and lgo in (0, 2, 4)
                if soff >= 0:
                    assert (lgo, soff) not in geom_of_stack
                    geom_of_stack[(lgo, soff)] = lgo
                tp = data[base + piv_off: base + piv_off + m].astype(np.int64)          # the task's pivots, ascending, the root last
                assert np.all(np.diff(tp) > 0) and tp[-1] == root
                par = [int(u_col[self.u_ptr[k]]) if self.u_ptr[k + 1] > self.u_ptr[k] else -1 for k in tp]
                assert all(pk in set(tp.tolist()) for pk in par[:-1]), "a task is a connected piece of the elimination tree"
                tgrid = 16 >> (lg // 2)                       # threads per front dimension: 16 / 8 / 4 at 1 / 4 / 16 scenarios per workgroup
                assert tl == tlevel and tcls <= cls and fprime == f + 1 <= tgrid * tcls and m >= 1
                assert (tcls == 4 and (lg == 2 or fprime <= 16)) if lg else (tcls == 2 or fprime > 16 * (tcls - 1))
                assert np.all(self.task_of[tp] == ti) and np.sum(self.task_of == ti) == m
                ext = self._front_ext(root)
                assert ext.size == e
                piv = np.concatenate([tp, ext])
                F = np.zeros((f, fprime, 2, 2))
                emap = data[base: base + f * fprime].reshape(f, fprime)
                owned, seen = [], set()
                for r in range(f):
                    for c in range(fprime):
                        cd = int(emap[r, c])
                        if cd == -1:
                            assert (r >= m and (c >= m)) or (c < f and min(r, c) < m), "only the update block and pattern holes are unmapped"
                            continue
                        if cd <= -2:
                            assert c == f and r < m and -(cd + 2) == tp[r] and partial[nE + tp[r]]
                            F[r, c, :, 0] = Y[tp[r]]
                            continue
                        ent, fl = cd & 0x0fffffff, cd >> 28
                        assert min(r, c) < m and c < f
                        rr, cc = (c, r) if fl & 2 else (r, c)
                        assert (self.e_row[ent], self.e_col[ent]) == (piv[rr], piv[cc])
                        assert bool(fl & 2) == (self.symmetric and r > c) and (not (fl & 2) or fl & 4)
                        assert bool(fl & 4) == (r == c or bool(fl & 2))
                        assert level_of[ent] < 0, "a task-owned entry was finished by somebody else"
                        if fl & 1:
                            assert not partial[ent] and self.e_src[ent] < 0
                            v = np.zeros((2, 2))
                        else:
                            assert partial[ent] or (self.inplace and self.e_src[ent] >= 0), "task loads an entry nobody wrote"
                            v = X[ent].copy()
                        F[r, c] = v.T if fl & 2 else v
                        if not (fl & 2):
                            assert ent not in seen
                            seen.add(ent)
                            if not (fl & 4):
                                owned.append((ent, r, c))
                # every pattern entry of the chain's rows / columns is mapped
                for q in range(m):
                    k = int(tp[q])
                    assert self.diag[k] in seen
                    want = set(int(x) for x in self.p.get("u_ent")[self.u_ptr[k]: self.u_ptr[k + 1]])
                    assert want <= seen
                dent = data[base + dent_off: base + dent_off + m]
                assert np.array_equal(dent, self.diag[tp])
                cd = base + child_off
                for _ in range(nchild):
                    coff, ce = int(data[cd]), int(data[cd + 1])
                    inv = data[cd + 2: cd + 2 + fprime]
                    assert stack_level[(lg, coff)] < tlevel, "child task in the same or a later launch level"     # KeyError: the child left its
                    C = stacks[lg >> 1][coff: coff + ce * (ce + 1) * 4].reshape(ce, ce + 1, 2, 2)                # block in another interleave
                    assert not np.isnan(C).any(), "update blocks of one interleave class overlap"
                    assert inv[f] == ce and sorted(inv[:f][inv[:f] >= 0].tolist()) == list(range(ce))
                    for r in np.flatnonzero(inv[:f] >= 0):
                        for c in np.flatnonzero(inv >= 0):
                            F[r, c] += C[inv[r], inv[c]]
                    cd += 2 + fprime
                D = [None] * m
                D[0] = dfactor(F[0, 0])
                for q in range(m):
                    k = int(tp[q])
                    s = int(self.u_ptr[k + 1] - self.u_ptr[k])
                    loc = [self._loc(tp, root, int(c)) for c in u_col[self.u_ptr[k]: self.u_ptr[k + 1]]]
                    assert q + 1 == m or (s > 0 and min(loc) > q)
                    Z = np.zeros((fprime, 2, 2))
                    for c in range(q + 1, fprime):
                        Z[c] = np.stack([dsolve(D[q], F[q, c][:, 0]), dsolve(D[q], F[q, c][:, 1])], axis=1)
                    # blocks outside struct(q) are exactly zero: the dense update touches only struct(q) x (struct(q) + rhs)
                    nzr = [i for i in range(q + 1, f) if np.any(F[i, q] != 0)]
                    nzc = [c for c in range(q + 1, f) if np.any(F[q, c] != 0)]
                    assert set(nzr) <= set(loc) and set(nzc) <= set(loc)
                    for i in range(q + 1, f):
                        for c in range(q + 1, fprime):
                            F[i, c] = F[i, c] - F[i, q] @ Z[c]
                    if self.jordan:                          # ... and above the diagonal: rows of finished pivots lose column q
                        for i in range(q):
                            for c in range(q + 1, fprime):
                                F[i, c] = F[i, c] - F[i, q] @ Z[c]
                    terms += s * (s + 1)
                    if q + 1 < m:
                        D[q + 1] = dfactor(F[q + 1, q + 1])
                assert not np.isnan(F).any()
                results.append((ti, owned, F, D, tp, m, e, soff, dent, lgo, jbase))
            for ti, owned, F, D, tp, m, e, soff, dent, lgo, jbase in results:      # tasks of one launch are independent of each other
                f = m + e
                for ent, r, c in owned:
                    X[ent] = F[r, c]
                    if self.jordan and r < m and c >= m:     # U(pivot, external): superseded by the Jordan row, not stored by the task
                        X[ent] = np.nan
                    level_of[ent] = lev
                    partial[ent] = False
                if self.jordan and e > 0:
                    assert np.isnan(X[jbase: jbase + m * e]).all(), "Jordan rows of two tasks overlap"
                    X[jbase: jbase + m * e] = F[:m, m:f].reshape(m * e, 2, 2)
                for q in range(m):
                    X[dent[q]] = D[q]
                    level_of[dent[q]] = lev
                    partial[dent[q]] = False
                    Y[tp[q]] = F[q, f, :, 0]
                    level_of[nE + tp[q]] = lev
                    partial[nE + tp[q]] = False
                    done_pivot[tp[q]] = True
                if e > 0:
                    assert soff >= 0
                    assert np.isnan(stacks[lgo >> 1][soff: soff + e * (e + 1) * 4]).all(), "update blocks of one interleave class overlap"
                    stacks[lgo >> 1][soff: soff + e * (e + 1) * 4] = F[m:, m:].reshape(-1)
                    stack_level[(lgo, soff)] = tlevel
                else:
                    assert soff == -1
            seen_tasks += ntk
        assert seen_tasks == hdr.shape[0] and not partial.any()
        assert np.array_equal(done_pivot, self.task_of >= 0)
        return terms

    def _front_ext(self, root):
        return self.p.get("u_col")[self.u_ptr[root]: self.u_ptr[root + 1]]

    def _loc(self, tp, root, piv):
        hit = np.flatnonzero(tp == piv)
        if hit.size:
            return int(hit[0])
        ext = self._front_ext(root).tolist()
        return len(tp) + ext.index(piv)

    def backsolve(self, X, Y):
        """Replays the backward tables: wave-record rows and CHAIN tasks (segments with wpi == 0: consecutive pivots of a
        supernode solved by one workgroup: external part first, then the in-chain triangle from the last pivot up)."""
        W = Y.copy()
        level_of = np.full(self.n, -1)
        out = np.zeros((self.n, 2))
        chain = self.p.get("bwd_chain")
        terms = 0
        for si, (base, nchunks, wpi, rpw, level, _last, _items, _pad) in enumerate(self.bseg):
            if wpi <= 0:                                     # chain tasks: general (wpi 0, 16 waves) or small (wpi -1, 8 waves, at most 8 rows)
                for t in range(nchunks):
                    nb, nE, off, wpr = (int(v) for v in self.brec[base + t][:4])
                    if wpi == 0:
                        assert 2 <= nb <= 32 and nE <= 72 and wpr in (1, 2, 4, 8) and (wpr == 1 or wpr * nb <= 16)
                    else:
                        assert wpi == -1 and 2 <= nb <= 8 and nE <= 72 and wpr in (1, 2, 4) and wpr * nb <= 8 and (wpr == 4 or 2 * wpr * nb > 8)
                    rows = chain[off:off + 3 * nb].reshape(nb, 3)
                    ecol = chain[off + 3 * nb: off + 3 * nb + nE]
                    uext = chain[off + 3 * nb + nE: off + 3 * nb + nE + nb * nE].reshape(nb, nE)
                    uin = chain[off + 3 * nb + nE + nb * nE: off + 3 * nb + nE + nb * nE + nb * nb].reshape(nb, nb)
                    assert np.all(np.diff(rows[:, 0]) == 1), "a chain is a run of consecutive pivots"
                    acc = np.zeros((nb, 2))
                    for p in range(nb):
                        k, bus, dg = (int(v) for v in rows[p])
                        assert dg == self.diag[k] and bus == self.perm[k] and level_of[k] < 0
                        acc[p] = Y[k]
                        for q in range(nE):
                            assert 0 <= level_of[ecol[q]] < level, "bwd chain race (external column)"
                            assert self.e_row[uext[p, q]] == k and self.e_col[uext[p, q]] == ecol[q]
                            acc[p] -= X[uext[p, q]] @ W[ecol[q]]
                            terms += 1
                    for c in range(nb - 1, -1, -1):
                        k = int(rows[c, 0])
                        W[k] = dsolve(X[int(rows[c, 2])], acc[c])
                        out[int(rows[c, 1])] = W[k]
                        for p in range(c):
                            e = int(uin[p, c])
                            assert self.e_row[e] == rows[p, 0] and self.e_col[e] == k
                            acc[p] -= X[e] @ W[k]
                            terms += 1
                    for p in range(nb):
                        level_of[int(rows[p, 0])] = level
                continue
            acc, meta = {}, {}
            for c in range(nchunks):
                for w in range(16):
                    r0 = base + (c * 16 + w) * rpw
                    recs, key, sub = self.brec[r0:r0 + rpw], (c, w // wpi), w % wpi
                    k = int(recs[0][0])
                    if k < 0:
                        continue
                    part = np.zeros(2)
                    for r in recs:
                        assert int(r[0]) == k
                        for t in range(int(r[3])):
                            ent, col = int(r[4 + 2 * t]), int(r[5 + 2 * t])
                            assert 0 <= level_of[col] < level, "bwd schedule race"
                            if ent >= self.nE:                   # Jordan row: block (local pivot, external column) of the pivot's task
                                h = self.thdr[self.task_of[k]]
                                tpv = self.tdata[int(h[3]) + int(h[6]): int(h[3]) + int(h[6]) + int(h[0])]
                                q, il = (ent - int(h[14])) % int(h[1]), (ent - int(h[14])) // int(h[1])
                                assert self.jordan and tpv[il] == k and self._front_ext(int(h[2]))[q] == col
                            else:
                                assert self.e_row[ent] == k and self.e_col[ent] == col and not (self.jordan and self.task_of[k] >= 0)
                            part -= X[ent] @ W[col]
                            terms += 1
                    if sub == 0:
                        meta[key] = (k, int(recs[0][1]), int(recs[0][2]))
                        acc[key] = Y[k] + part
                    else:
                        acc[key] = acc[key] + part
            for key, (kk, bus, dg) in meta.items():           # rows of one segment are independent of each other
                assert dg == self.diag[kk] and bus == self.perm[kk] and level_of[kk] < 0
                W[kk] = dsolve(X[dg], acc[key])
                out[bus] = W[kk]
            for key, (kk, bus, dg) in meta.items():
                level_of[kk] = level
        assert (level_of >= 0).all(), "rows missing from the backward schedule"
        if self.jordan:                                      # a task pivot has one term per external column of its task instead of its U row
            want = 0
            for k in range(self.n):
                kk = int(self.thdr[self.task_of[k]][2]) if self.task_of[k] >= 0 else k
                want += int(self.u_ptr[kk + 1] - self.u_ptr[kk])
            assert terms == want, "terms lost or duplicated in the Jordan records"
        else:
            assert terms == int(self.u_ptr[-1]), "U terms lost or duplicated in the records"
        assert not np.isnan(out).any()
        return out

    def selected_inverse(self, X):
        """Replays the selected-inverse tables (symmetric matrix): Z on the upper factor pattern + diagonal, [nE,2,2];
        asserts that every Z block a record reads was finished in an earlier level."""
        seg, rec = self.p.replay_tables("sel")
        Z = np.zeros((self.nE, 2, 2))
        level_of = np.full(self.nE, -1)
        acc, meta = {}, {}

        def flush():
            for key, (lev, target, dg, kind) in meta.items():
                assert level_of[target] < 0, "Z entry scheduled twice"
                T = acc[key]
                R = np.eye(2) - T if kind == 1 else -T
                Z[target] = np.stack([dsolve(X[dg], R[:, 0]), dsolve(X[dg], R[:, 1])], axis=1)
                level_of[target] = lev
            acc.clear()
            meta.clear()

        current, terms = None, 0
        for level, key, sub, recs in self._waves(seg, rec):
            if level != current:
                assert current is None or level > current
                flush()
                current = level
            target = int(recs[0][0])
            if target < 0:
                continue
            part = np.zeros((2, 2))
            for r in recs:
                assert int(r[0]) == target
                for t in range(int(r[3])):
                    u, z = int(r[4 + 2 * t]), int(r[5 + 2 * t])
                    tr, z = z >> 30, z & ((1 << 30) - 1)
                    assert 0 <= level_of[z] < level, "selected-inverse schedule race"
                    assert self.e_row[u] == self.e_row[target] or self.e_col[target] == self.e_row[target]
                    part += X[u] @ (Z[z].T if tr else Z[z])
                    terms += 1
            if sub == 0:
                meta[key] = (level, target, int(recs[0][1]), int(recs[0][2]))
                acc[key] = part
            else:
                acc[key] = acc[key] + part
        flush()
        upper = self.e_col >= self.e_row
        assert (level_of[upper] >= 0).all() and (level_of[~upper] < 0).all()
        return Z

    def _wpi_of(self, key):
        return int(self.bseg[key[0]][2])


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

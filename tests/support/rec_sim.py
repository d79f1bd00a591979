"""numpy interpreter of the RECORD STREAMS of the register-front kernel (test infrastructure).

Executes, for ONE problem and on the CPU, what cannoles.jl_amd/csrc/kernels2.hip executes from the streams written
by csrc/analysis.cpp::write_forward_records (layout in csrc/plan.h): plain assembly entries, raw values and products
of the on-the-fly condensation, the children's extend-add tables, elimination from the highest local index, the
packed L panels, and the backward records.  It validates the host side of that path without a GPU.
Never used by the product.
"""
import numpy as np

(R_NPIV, R_NUPD, R_RECLEN, R_NASM, R_NCHILD, R_UOFF, R_FLAGS, R_FSOFF, R_LPTR_LO, R_LPTR_HI, R_NASMV, R_ASM_OFF,
 R_CHILD_OFF, R_NPROD, R_NRAW, R_NRD) = range(16)
R_HDR = 16
FAST_IMG_TRI, FAST_IMG_DOUBLES = 136, 152   # plan.h
RF_U_GLOBAL, RF_FS_GLOBAL, RF_ROWS = 1, 2, 4
ROWS_KM = 5                                   # plan.h: row form of the condensation products
ROWS_NPAIR = ROWS_KM * (ROWS_KM + 1) // 2 + ROWS_KM
ROWS_PW = (ROWS_NPAIR + 3) // 4
ROWS_WORDS = 16 * (1 + ROWS_KM + 1 + ROWS_PW)
B_NPIV, B_NUPD, B_RECLEN, B_XOFF, B_PXOFF, B_LPTR_LO, B_LPTR_HI, B_CLS = range(8)
B_HDR = 8
B_ROWS_FLAG, BROWS_WORDS = 256, 16 * (ROWS_KM + 3)   # plan.h: residual rows recovered by the backward records


BAND_HW = 4   # csrc/kernels2.hip: CNL_BAND_HW


def band_rows(band_word, i):
    """rows a < i that pivot i updates under the band form `band_word` of a fast front's record (R_FSOFF: nfix | hw << 8, 0 = all):
    the nfix lowest rows and the hw rows right below the pivot (csrc/kernels2.hip, eliminate16_dpp<LATE, BNF>)"""
    if band_word == 0:
        return set(range(i))
    nfix, hw = band_word & 255, band_word >> 8
    assert hw == BAND_HW and nfix in (2, 3), band_word
    return {a for a in range(i) if a < nfix or a >= i - hw}


def tri(i):
    return i * (i + 1) // 2


class RecSim:
    def __init__(self, plan, nnz_src, n_rhs):
        """nnz_src / n_rhs: length of the value array / right-hand side the records address (the caller's arrays when
        the records condense on the fly, the condensed buffer otherwise)."""
        self.rec = plan.array("rec").astype(np.int64)
        self.brec = plan.array("brec").astype(np.int64)
        assert self.rec.size, "the plan is not served by the register-front kernel"
        self.ns = plan.info["nsuper"]
        self.lsize = plan.info["lsize"]
        self.nnz, self.n_rhs = nnz_src, n_rhs

    def forward(self, vals, rhs, eig_tol):
        """Returns (L storage, npos, nzero) of the fronts (the condensed residual pivots are NOT included)."""
        rec = self.rec
        src_val = lambda s: vals[s] if s < self.nnz else rhs[s - self.nnz]
        L = np.zeros(self.lsize + 64)
        U = {}
        npos = nzer = 0
        off = 0
        self.stats = dict(plain=0, products=0, raw=0, strided=0, packed=0)
        self.owned = self.own_pos = self.own_zer = 0  # condensed residual pivots counted by the fronts that own them
        for _ in range(self.ns):
            H = rec[off:off + R_HDR]
            npiv, nupd, nasm, nasmv = int(H[R_NPIV]), int(H[R_NUPD]), int(H[R_NASM]), int(H[R_NASMV])
            flags, cls = int(H[R_FLAGS]) & 0xff, int(H[R_FLAGS]) >> 8
            aoff, coff = int(H[R_ASM_OFF]), int(H[R_CHILD_OFF])
            nprod, nrd_own, nraw = int(H[R_NPROD]) & 0xffff, int(H[R_NPROD]) >> 16, int(H[R_NRAW])
            nrd, nrawv = int(H[R_NRD]) & 0xffff, int(H[R_NRD]) >> 16
            f = 1 + nupd + npiv
            strided = cls == 16 and not (flags & RF_FS_GLOBAL)
            self.stats["strided" if strided else "packed"] += 1
            img = np.zeros(FAST_IMG_DOUBLES if strided else tri(f) + 16)   # fast fronts: packed triangle + padding slots (plan.h)
            pos_of = lambda a, b: tri(a) + b
            r = rec[off:off + int(H[R_RECLEN])]
            # plain entries: matrix values first, right-hand side last; every group padded to rounds of 16
            assert nasmv % 16 == 0 and nasm % 16 == 0 and nasmv <= nasm
            for e in range(nasm):
                s, p = int(r[aoff + e]), int(r[aoff + nasm + e])
                assert (s >= self.nnz) == (e >= nasmv), "entries are not grouped by array"
                img[p] += src_val(s)
            self.stats["plain"] += nasm
            # raw values and products
            raw_off = aoff + 2 * nasm
            if flags & RF_ROWS:
                self._row_products(r, raw_off, nraw, nprod, nrd, nrd_own, img, src_val, eig_tol)
                nraw = nprod = nrd = nrd_own = nrawv = 0
            assert nrawv % 16 == 0 and nraw % 16 == 0
            jraw = np.array([src_val(int(r[raw_off + t])) for t in range(nraw)])
            for t in range(nraw):
                assert (int(r[raw_off + t]) >= self.nnz) == (t >= nrawv)
            # the first nrd_own pivots are owned by this front: it counts them in the inertia
            assert nrd_own <= nrd
            self.owned += nrd_own
            self.own_pos += int((jraw[:nrd_own] > eig_tol).sum())
            self.own_zer += int((np.abs(jraw[:nrd_own]) <= eig_tol).sum())
            jr = jraw.copy()
            jr[:nrd] = -1.0 / jraw[:nrd]
            poff = raw_off + nraw
            for e in range(nprod):
                if strided:
                    w = int(r[poff + e])
                    p, ia, ib, idd = w & 255, (w >> 8) & 127, (w >> 15) & 127, (w >> 22) & 127
                    if nraw == 0:
                        continue
                else:
                    p, w = int(r[poff + 2 * e]), int(r[poff + 2 * e + 1])
                    ia, ib, idd = w & 1023, (w >> 10) & 1023, (w >> 20) & 1023
                img[p] += jr[ia] * jr[ib] * jr[idd]
            self.stats["products"] += nprod
            self.stats["raw"] += nraw
            # extend-add
            co = coff
            for _c in range(int(H[R_NCHILD])):
                cu, tuc, cfl = int(r[co]), int(r[co + 1]), int(r[co + 2])
                Uc = U.pop((cfl, cu))
                assert len(Uc) == tuc
                for t in range(tuc):
                    img[int(r[co + 4 + t])] += Uc[t]
                co += 4 + ((tuc + 3) & ~3)
            # eliminate from the highest local index; L panel: row i at lptr + tri(i) - tri(nupd + 1)
            F = np.zeros((f, f))
            for a in range(f):
                for b in range(a + 1):
                    F[a, b] = img[pos_of(a, b)]
            lptr = int(H[R_LPTR_LO]) | (int(H[R_LPTR_HI]) << 31)
            tu = tri(1 + nupd)
            band = int(H[R_FSOFF]) if strided else 0
            self.stats["band"] = self.stats.get("band", 0) + (1 if band else 0)
            for i in range(f - 1, nupd, -1):
                d = F[i, i]
                npos += d > eig_tol
                nzer += abs(d) <= eig_tol
                w = F[i, :i].copy()
                lv = w / d
                L[lptr + tri(i) - tu: lptr + tri(i) - tu + i] = lv
                L[lptr + tri(i) - tu + i] = d
                rows_i = band_rows(band, i)
                for a in range(i):
                    if a not in rows_i:   # the kernel does not contain this update: the operand must be an EXACT zero
                        assert w[a] == 0.0, f"band form: pivot {i} of a front (npiv {npiv}, nupd {nupd}) has a non-zero {w[a]} in column {a}"
                        self.stats["skipped"] = self.stats.get("skipped", 0) + 1
                        continue
                    F[a, :a + 1] -= w[a] * lv[:a + 1]
            U[(1 if flags & RF_U_GLOBAL else 0, int(H[R_UOFF]))] = np.array([F[a, b] for a in range(nupd + 1) for b in range(a + 1)])
            off += int(H[R_RECLEN])
        assert all(len(u) == 1 for u in U.values()), "update matrices left on the stack"
        return L, int(npos), int(nzer)

    def _row_products(self, r, raw_off, nraw, nprod, nrows, nown, img, src_val, eig_tol):
        """RF_ROWS (plan.h): lane l = residual row l; (J_p w) J_q with w = -1/d_r goes to the position byte of pair (p, q)"""
        assert nraw == ROWS_WORDS and nprod == 0 and nown <= nrows <= 16
        self.stats["rowform"] = self.stats.get("rowform", 0) + 1
        for l in range(16):
            dsrc = int(r[raw_off + l])
            assert dsrc < self.nnz
            dv = src_val(dsrc)
            if l < nown:
                self.owned += 1
                self.own_pos += int(dv > eig_tol)
                self.own_zer += int(abs(dv) <= eig_tol)
            w = -1.0 / dv
            J = [src_val(int(r[raw_off + 16 * (1 + q) + l])) for q in range(ROWS_KM)]
            for q in range(ROWS_KM):
                assert int(r[raw_off + 16 * (1 + q) + l]) < self.nnz
            rs = int(r[raw_off + 16 * (1 + ROWS_KM) + l])
            assert rs >= self.nnz
            rh = src_val(rs)
            words = [int(r[raw_off + 16 * (2 + ROWS_KM + g) + l]) & 0xffffffff for g in range(ROWS_PW)]
            pos = lambda k: (words[k >> 2] >> (8 * (k & 3))) & 255
            k = 0
            for p in range(ROWS_KM):
                for q in range(p + 1):
                    if l >= nrows:
                        assert pos(k) >= FAST_IMG_TRI
                    img[pos(k)] += (J[p] * w) * J[q]
                    self.stats["products"] += pos(k) < FAST_IMG_TRI
                    k += 1
            for q in range(ROWS_KM):
                img[pos(k)] += (rh * w) * J[q]
                self.stats["products"] += pos(k) < FAST_IMG_TRI
                k += 1
        self.stats["raw"] += nraw

    def _back_rows(self, br, off, H, f, x, d, vals, rhs):
        """B_ROWS_FLAG (plan.h): lane l recovers the residual component of row l the front owns from the front's own x"""
        clsw = int(H[B_CLS])
        if not clsw & B_ROWS_FLAG:
            return 0
        assert vals is not None, "the backward records recover residual components: vals / rhs needed"
        roff = (B_HDR + f + 3) & ~3
        assert int(H[B_RECLEN]) == roff + BROWS_WORDS
        sec = br[off + roff: off + roff + BROWS_WORDS]
        nrows = clsw >> 16
        for l in range(16):
            iw = int(sec[16 * (ROWS_KM + 2) + l])
            assert bool(iw & (1 << 23)) == (l < nrows)
            if l >= nrows:
                continue
            nm = (iw >> 20) & 7
            dsrc, rsrc = int(sec[l]), int(sec[16 * (ROWS_KM + 1) + l])
            assert dsrc < self.nnz <= rsrc
            acc = -rhs[rsrc - self.nnz]
            for p in range(nm):
                li = (iw >> (4 * p)) & 15
                assert 1 <= li < f
                acc += vals[int(sec[16 * (1 + p) + l])] * x[li]
            assert np.isnan(d[rsrc - self.nnz]), "residual component written twice"
            d[rsrc - self.nnz] = acc / vals[dsrc]
        return nrows

    def backward(self, L, n_out, vals=None, rhs=None):
        """d = -x from the factor storage (z = D^-1 L^-1 b sits in column 0 of the panels)."""
        br = self.brec
        d = np.full(n_out, np.nan)
        xs = {}
        off = 0
        for _ in range(self.ns):
            H = br[off:off + B_HDR]
            npiv, nupd, xoff, pxoff = int(H[B_NPIV]), int(H[B_NUPD]), int(H[B_XOFF]), int(H[B_PXOFF])
            lptr = int(H[B_LPTR_LO]) | (int(H[B_LPTR_HI]) << 31)
            f = 1 + nupd + npiv
            idx = br[off + B_HDR: off + B_HDR + f]
            x = np.zeros(f)
            for l in range(1, nupd + 1):
                x[l] = xs[pxoff + int(idx[l])]
            tu = tri(1 + nupd)
            for i in range(nupd + 1, f):
                row = L[lptr + tri(i) - tu: lptr + tri(i) - tu + i]
                x[i] = row[0] - np.dot(row[1:i], x[1:i])
                d[int(idx[i])] = -x[i]
            for l in range(1, f):
                xs[xoff + l] = x[l]
            self._back_rows(br, off, H, f, x, d, vals, rhs)
            off += int(H[B_RECLEN])
        return d


class StagedSim(RecSim):
    """The same streams executed the way the STAGED kernel runs a latency plan (csrc/plan.h, struct Task): stage by stage,
    every task with an empty LDS stack of its own, task roots handing their update matrix over through the global scratch
    and reading the parent's solution from d.  Update matrices are stored like the kernel stores them — rows of 16 lanes,
    ascending, so that up to 15 doubles past the matrix are clobbered (NaN here) — which is what makes overlapping slots
    visible.  `reverse` runs the tasks of a stage in the other order (they must be independent); `dataflow` runs the tasks in a
    depth-first order of the task tree given by the parent fields (children before their parent, whatever their stage),
    which is what the single-launch execution with dependency counters relies on."""

    def __init__(self, plan, nnz_src, n_rhs):
        super().__init__(plan, nnz_src, n_rhs)
        self.tasks = plan.array("tasks").astype(np.int64).reshape(-1, 8)   # struct Task (csrc/plan.h)
        self.stage_ptr = plan.array("stage_ptr").astype(np.int64)
        assert len(self.tasks) > 1, "not a staged plan"

    def _store_u(self, mem, uoff, F, nupd):
        for a in range(nupd + 1):
            row = np.full(16, np.nan)
            row[:a + 1] = F[a, :a + 1]
            mem[uoff + tri(a): uoff + tri(a) + 16] = row

    def dataflow_order(self):
        """tasks in depth-first post-order of the parent links; checks the child counts"""
        nt = len(self.tasks)
        kids = [[] for _ in range(nt)]
        for t in range(nt):
            par = int(self.tasks[t][6])
            if par >= 0:
                assert int(self.tasks[par][0]) > int(self.tasks[t][0]), "a parent task must belong to a later stage"
                kids[par].append(t)
        for t in range(nt):
            assert len(kids[t]) == int(self.tasks[t][7])
        order, stack = [], [(t, False) for t in range(nt - 1, -1, -1) if int(self.tasks[t][6]) < 0]
        while stack:
            t, done = stack.pop()
            if done:
                order.append(t)
            else:
                stack.append((t, True))
                stack.extend((c, False) for c in kids[t])
        assert sorted(order) == list(range(nt))
        return order

    def run(self, vals, rhs, eig_tol, n_out, reverse=False, gs_doubles=None, lds_doubles=4096, dataflow=False):
        rec, br = self.rec, self.brec
        src_val = lambda s: vals[s] if s < self.nnz else rhs[s - self.nnz]
        L = np.zeros(self.lsize + 64)
        gs = np.full((gs_doubles or 1 << 22) + 64, np.nan)
        npos = nzer = 0
        nst = len(self.stage_ptr) - 1
        covered = np.zeros(self.ns, bool)
        # ---- forward, children first
        if dataflow:
            fwd_lists = [self.dataflow_order()]
        else:
            fwd_lists = []
            for st in range(nst):
                tl = list(range(int(self.stage_ptr[st]), int(self.stage_ptr[st + 1])))
                assert all(int(self.tasks[t][0]) == st for t in tl)
                fwd_lists.append(list(reversed(tl)) if reverse else tl)
        for tl in fwd_lists:
            for t in tl:
                stage, f0, f1, off, boff, is_root = (int(v) for v in self.tasks[t][:6])
                lds = np.full(lds_doubles, np.nan)
                for s in range(f0, f1):
                    assert not covered[s]
                    covered[s] = True
                    H = rec[off:off + R_HDR]
                    npiv, nupd, nasm, nasmv = int(H[R_NPIV]), int(H[R_NUPD]), int(H[R_NASM]), int(H[R_NASMV])
                    flags, cls = int(H[R_FLAGS]) & 0xff, int(H[R_FLAGS]) >> 8
                    aoff, coff = int(H[R_ASM_OFF]), int(H[R_CHILD_OFF])
                    nprod, nrd_own, nraw = int(H[R_NPROD]) & 0xffff, int(H[R_NPROD]) >> 16, int(H[R_NRAW])
                    nrd = int(H[R_NRD]) & 0xffff
                    f = 1 + nupd + npiv
                    strided = cls == 16 and not (flags & RF_FS_GLOBAL)
                    img = np.zeros(FAST_IMG_DOUBLES if strided else tri(f) + 16)   # fast fronts: packed triangle + padding slots (plan.h)
                    pos_of = lambda a, b: tri(a) + b
                    r = rec[off:off + int(H[R_RECLEN])]
                    for e in range(nasm):
                        img[int(r[aoff + nasm + e])] += src_val(int(r[aoff + e]))
                    raw_off = aoff + 2 * nasm
                    if flags & RF_ROWS:
                        self.stats = getattr(self, "stats", None) or dict(plain=0, products=0, raw=0, strided=0, packed=0)
                        self.owned = self.own_pos = self.own_zer = 0
                        self._row_products(r, raw_off, nraw, nprod, nrd, nrd_own, img, src_val, eig_tol)
                        npos += self.own_pos
                        nzer += self.own_zer
                        nraw = nprod = nrd = nrd_own = 0
                    jraw = np.array([src_val(int(r[raw_off + q])) for q in range(nraw)])
                    npos += int((jraw[:nrd_own] > eig_tol).sum())
                    nzer += int((np.abs(jraw[:nrd_own]) <= eig_tol).sum())
                    jr = jraw.copy()
                    jr[:nrd] = -1.0 / jraw[:nrd]
                    poff = raw_off + nraw
                    for e in range(nprod):
                        if strided:
                            w = int(r[poff + e])
                            p, ia, ib, idd = w & 255, (w >> 8) & 127, (w >> 15) & 127, (w >> 22) & 127
                            if nraw == 0:
                                continue
                        else:
                            p, w = int(r[poff + 2 * e]), int(r[poff + 2 * e + 1])
                            ia, ib, idd = w & 1023, (w >> 10) & 1023, (w >> 20) & 1023
                        img[p] += jr[ia] * jr[ib] * jr[idd]
                    co = coff
                    for _c in range(int(H[R_NCHILD])):
                        cu, tuc, cfl = int(r[co]), int(r[co + 1]), int(r[co + 2])
                        Uc = (gs if cfl else lds)[cu:cu + tuc]
                        assert not np.isnan(Uc).any(), f"front {s}: child update matrix at {'global' if cfl else 'LDS'} offset {cu} is not intact"
                        for q in range(tuc):
                            img[int(r[co + 4 + q])] += Uc[q]
                        co += 4 + ((tuc + 3) & ~3)
                    F = np.zeros((f, f))
                    for a in range(f):
                        for b in range(a + 1):
                            F[a, b] = img[pos_of(a, b)]
                    lptr = int(H[R_LPTR_LO]) | (int(H[R_LPTR_HI]) << 31)
                    tu = tri(1 + nupd)
                    band = int(H[R_FSOFF]) if strided else 0
                    for i in range(f - 1, nupd, -1):
                        dd = F[i, i]
                        npos += dd > eig_tol
                        nzer += abs(dd) <= eig_tol
                        w = F[i, :i].copy()
                        lv = w / dd
                        L[lptr + tri(i) - tu: lptr + tri(i) - tu + i] = lv
                        L[lptr + tri(i) - tu + i] = dd
                        rows_i = band_rows(band, i)
                        for a in range(i):
                            if a not in rows_i:   # not compiled into the kernel's band-form elimination: must be an exact zero
                                assert w[a] == 0.0, f"band form: pivot {i} of front {s} has a non-zero {w[a]} in column {a}"
                                continue
                            F[a, :a + 1] -= w[a] * lv[:a + 1]
                    uglob = bool(flags & RF_U_GLOBAL)
                    if s == f1 - 1:
                        assert uglob, "a task root must leave its update matrix in the global scratch"
                    self._store_u(gs if uglob else lds, int(H[R_UOFF]), F, nupd)
                    off += int(H[R_RECLEN])
        assert covered.all(), "fronts outside every task"
        # ---- backward, parents first
        d = np.full(n_out, np.nan)
        for tl in reversed(fwd_lists):
            for t in reversed(tl):
                stage, f0, f1, off, boff, is_root = (int(v) for v in self.tasks[t][:6])
                xs = {}
                for _ in range(f1 - f0):
                    H = br[boff:boff + B_HDR]
                    npiv, nupd, xoff, pxoff = int(H[B_NPIV]), int(H[B_NUPD]), int(H[B_XOFF]), int(H[B_PXOFF])
                    lptr = int(H[B_LPTR_LO]) | (int(H[B_LPTR_HI]) << 31)
                    f = 1 + nupd + npiv
                    idx = br[boff + B_HDR: boff + B_HDR + f]
                    x = np.zeros(f)
                    for l in range(1, nupd + 1):
                        if pxoff == -2:
                            x[l] = -d[int(idx[l])]
                        else:
                            x[l] = xs[pxoff + int(idx[l])]
                    assert not np.isnan(x).any(), "solution of the parent not available"
                    tu = tri(1 + nupd)
                    for i in range(nupd + 1, f):
                        row = L[lptr + tri(i) - tu: lptr + tri(i) - tu + i]
                        x[i] = row[0] - np.dot(row[1:i], x[1:i])
                        d[int(idx[i])] = -x[i]
                    for l in range(1, f):
                        xs[xoff + l] = x[l]
                    self._back_rows(br, boff, H, f, x, d, vals, rhs)
                    boff += int(H[B_RECLEN])
        return d, int(npos), int(nzer)

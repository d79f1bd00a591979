"""CPU interpreter of the band program (csrc/band.h, written by csrc/band.cpp): executes the step / row / epoch blocks exactly as
the kernels of csrc/band.hip do — operands through per-epoch 64-byte pieces into a lane block, window of 5 band slots + 1 border
slot, factor records through the out ring, junction of the two parts — vectorised over a batch of problems.  Test
infrastructure: it pins the GENERATOR against the oracle without a GPU."""
import numpy as np

HW, NB, EPOCH, NPIECE, LREC = 4, 5, 8, 15, 6
NS = EPOCH   # window slots: variable number t of a part lives in slot t % NS; the border row is index NS
IN_OFF, LOUT_OFF, LOUT_MAX = 0, NPIECE * 8, 32
DX_OFF = LOUT_OFF     # the out ring: factor records of half an epoch (forward), solution components of an epoch (backward)
DX_MAX = 8
DR_OFF = DX_OFF + DX_MAX
DR_MAX = 24
ZERO_OFF = LOUT_OFF + LOUT_MAX
LANE = ZERO_OFF + 1
SW, RW, EW, BW = 20, 8, 44, 4
BS_FLAGS, BS_DG0, BS_RHO, BS_OD = 0, 1, 4, 5
BS_BC0 = BS_OD + 2 * HW
BS_BC1, BS_RX, BS_LB, BS_LX, BS_DX, BS_BORDER = BS_BC0 + 1, BS_BC0 + 2, BS_BC0 + 3, BS_BC0 + 4, BS_BC0 + 5, BS_BC0 + 6
BF_ENTER_B, BF_PIVOT_B, BF_PIVOT_X = 1, 2, 4
BR_DI, BR_J0, BR_RR, BR_DR = 0, 1, 1 + NB, 2 + NB
BE_FP, BE_BP = 0, NPIECE
BE_LBASE, BE_LCNT, BE_LBASE2, BE_LCNT2, BE_DXLO, BE_DXCNT, BE_DRLO, BE_DRCNT, BE_NSTEP, BE_FOFF, BE_BOFF, BE_OPLEN = (2 * NPIECE + i for i in range(12))


class BandSim:
    def __init__(self, plan):
        info = plan.array("band_info")
        self.ok = bool(info[0])
        if not self.ok:
            return
        self.nparts, self.m0, self.n, self.N, self.nnz, self.lsize = (int(v) for v in info[1:7])
        self.parts = []
        for q in range(self.nparts):
            pi = plan.array(f"band_part{q}")
            self.parts.append(dict(nsteps=int(pi[0]), nepochs=int(pi[1]), npiv=int(pi[2]), nevents=int(pi[3]), loff=int(pi[4]),
                                   fops=plan.array(f"band_fops{q}"), bops=plan.array(f"band_bops{q}"),
                                   epochs=plan.array(f"band_epochs{q}").reshape(-1, EW), borders=plan.array(f"band_borders{q}").reshape(-1, BW)))

    # ---- one factorisation attempt (+ forward substitution) of every problem -------------------------------------------
    def _load_pieces(self, blk, pieces, arrays):
        for k, pc in enumerate(pieces):
            if pc < 0:              # -1: unused
                continue
            arr, base = arrays[pc >> 28], pc & ((1 << 28) - 1)
            w = 8                   # a piece: eight consecutive elements of one array (64 bytes)
            assert base + w <= arr.shape[1], (base, arr.shape)
            blk[:, IN_OFF + 8 * k: IN_OFF + 8 * k + w] = arr[:, base: base + w]

    def forward(self, vals, rhs, rho, ovr, tol, Lst):
        B = vals.shape[0]
        npos = np.zeros(B, np.int64)
        nzer = np.zeros(B, np.int64)
        wins = []
        for q, P in enumerate(self.parts):
            S = np.zeros((NS + 1, NS + 1, B))
            c = np.zeros((NS + 1, B))
            blk = np.zeros((B, LANE))
            ops, o = P["fops"], 0
            Lq = Lst[:, P["loff"]:]
            starts = np.concatenate([[0], np.cumsum(P["epochs"][:, BE_NSTEP])])
            assert starts[-1] == P["nsteps"]
            ep_of = np.repeat(np.arange(P["nepochs"]), P["epochs"][:, BE_NSTEP])
            for u in range(P["nsteps"]):
                if u == starts[ep_of[u]]:
                    E = P["epochs"][ep_of[u]]
                    if u:   # factor records of the previous epoch's second half
                        Ep = P["epochs"][ep_of[u] - 1]
                        Lq[:, Ep[BE_LBASE2]: Ep[BE_LBASE2] + Ep[BE_LCNT2]] = blk[:, LOUT_OFF: LOUT_OFF + Ep[BE_LCNT2]]
                    blk[:, :ZERO_OFF] = np.nan   # stale operands must not be read
                    self._load_pieces(blk, E[BE_FP: BE_FP + NPIECE], (vals, rhs))
                    assert o == E[BE_FOFF] and E[BE_OPLEN] <= 256
                st = ops[o: o + SW]
                fl = int(st[BS_FLAGS])
                nrows = (fl >> 8) & 255
                assert (st[1:BS_LB + 2] % 8 == 0).all()
                v = lambda off: blk[:, off // 8]
                es = u % NS
                live = [(u - HW + k) % NS for k in range(NB)]   # live slots: [0] = this step's pivot .. [HW] = the entering variable
                ps = live[0]
                # enter
                diag = (v(st[BS_DG0]) + v(st[BS_DG0 + 1])) + v(st[BS_DG0 + 2])
                rv = v(st[BS_RHO])
                if not (fl >> 16) & 1:
                    rv = np.where(ovr, rho, rv)
                S[es, es] = diag + rv
                for k in range(1, HW + 1):
                    s = (es - k) % NS
                    S[es, s] = S[s, es] = v(st[BS_OD + 2 * (k - 1)]) + v(st[BS_OD + 2 * (k - 1) + 1])
                S[NS, es] = S[es, NS] = v(st[BS_BC0]) + v(st[BS_BC1])
                c[es] = v(st[BS_RX])
                # rows
                for i in range(nrows):
                    rb = ops[o + SW + RW * i: o + SW + RW * (i + 1)]
                    dr = v(rb[BR_DI])
                    npos += dr > tol
                    nzer += np.abs(dr) <= tol
                    w = -1.0 / dr
                    J = [v(rb[BR_J0 + k]) for k in range(NB)]
                    tr = v(rb[BR_RR]) * w
                    for ka in range(NB):
                        a = live[ka]
                        ta = J[ka] * w
                        for kb in range(ka + 1):
                            b = live[kb]
                            S[a, b] = S[a, b] + ta * J[kb]
                            S[b, a] = S[a, b]
                        c[a] = c[a] + tr * J[ka]
                # border pivot
                if fl & BF_PIVOT_B:
                    bt = P["borders"][st[BS_BORDER]]
                    S[NS, NS] = S[NS, NS] + vals[:, bt[0]]
                    c[NS] = c[NS] + rhs[:, bt[1]]
                    d = S[NS, NS].copy()
                    npos += d > tol
                    nzer += np.abs(d) <= tol
                    w = np.stack([S[NS, live[k]] for k in range(NB)])
                    l = w / d
                    z = c[NS] / d
                    for ka in range(NB):
                        a = live[ka]
                        for kb in range(ka + 1):
                            b = live[kb]
                            S[a, b] = S[a, b] - w[ka] * l[kb]
                            S[b, a] = S[a, b]
                        c[a] = c[a] - w[ka] * z
                    off = st[BS_LB] // 8
                    blk[:, off: off + NB] = l.T
                    blk[:, off + NB] = z
                    S[NS, :] = 0.0
                    S[:, NS] = 0.0
                    c[NS] = 0.0
                # band pivot
                if fl & BF_PIVOT_X:
                    d = S[ps, ps].copy()
                    npos += d > tol
                    nzer += np.abs(d) <= tol
                    w = S[:, ps].copy()
                    l = w / d
                    z = c[ps] / d
                    oth = live[1:] + [NS]
                    for ia, a in enumerate(oth):
                        for b in oth[: ia + 1]:
                            S[a, b] = S[a, b] - w[a] * l[b]
                            S[b, a] = S[a, b]
                        c[a] = c[a] - w[a] * z
                    off = st[BS_LX] // 8
                    for k in range(1, NB):
                        blk[:, off + k - 1] = l[live[k]]
                    blk[:, off + 4] = l[NS]
                    blk[:, off + 5] = z
                    S[ps, :] = np.nan   # a pivoted slot holds nothing until the next variable enters it
                    S[:, ps] = np.nan
                    c[ps] = np.nan
                o += SW + RW * nrows
                if u - starts[ep_of[u]] == EPOCH // 2 - 1 or (u + 1 == starts[ep_of[u] + 1] and u - starts[ep_of[u]] < EPOCH // 2 - 1):
                    Ec = P["epochs"][ep_of[u]]   # factor records of the epoch's first half
                    Lq[:, Ec[BE_LBASE]: Ec[BE_LBASE] + Ec[BE_LCNT]] = blk[:, LOUT_OFF: LOUT_OFF + Ec[BE_LCNT]]
                    blk[:, LOUT_OFF: ZERO_OFF] = np.nan
            Ep = P["epochs"][P["nepochs"] - 1]
            Lq[:, Ep[BE_LBASE2]: Ep[BE_LBASE2] + Ep[BE_LCNT2]] = blk[:, LOUT_OFF: LOUT_OFF + Ep[BE_LCNT2]]
            wins.append((S, c))
        junction = None
        if self.nparts == 2:
            n, m0 = self.n, self.m0
            sL = [(m0 + i) % NS for i in range(HW)]
            sR = [(n - 1 - m0 - i) % NS for i in range(HW)]
            (SL, cL), (SR, cR) = wins
            SJ = np.zeros((HW, HW, B))
            cJ = np.zeros((HW, B))
            for i in range(HW):
                for j in range(HW):
                    SJ[i, j] = SL[sL[i], sL[j]] + SR[sR[i], sR[j]]
                cJ[i] = cL[sL[i]] + cR[sR[i]]
            lj = np.zeros((HW, HW, B))
            zj = np.zeros((HW, B))
            for i in range(HW):
                d = SJ[i, i].copy()
                npos += d > tol
                nzer += np.abs(d) <= tol
                w = SJ[:, i].copy()
                zj[i] = cJ[i] / d
                for a in range(i + 1, HW):
                    lj[a, i] = w[a] / d
                for a in range(i + 1, HW):
                    for b in range(i + 1, a + 1):
                        SJ[a, b] = SJ[a, b] - w[a] * lj[b, i]
                        SJ[b, a] = SJ[a, b]
                    cJ[a] = cJ[a] - w[a] * zj[i]
            junction = (lj, zj)
        return npos, nzer, junction

    def backward(self, vals, rhs, Lst, junction, d):
        B = vals.shape[0]
        n, m0 = self.n, self.m0
        xj = None
        if self.nparts == 2:
            lj, zj = junction
            xj = np.zeros((HW, B))
            for i in range(HW - 1, -1, -1):
                xj[i] = zj[i] - sum(lj[a, i] * xj[a] for a in range(i + 1, HW))
                d[:, m0 + i] = -xj[i]
        for q, P in enumerate(self.parts):
            xs = np.zeros((NS + 1, B))
            if xj is not None:
                for i in range(HW):
                    xs[((m0 + i) if q == 0 else (n - 1 - m0 - i)) % NS] = xj[i]
            blk = np.zeros((B, LANE))
            ops, o = P["bops"], 0
            Lq = Lst[:, P["loff"]:]
            starts = np.concatenate([[0], np.cumsum(P["epochs"][:, BE_NSTEP])])
            ep_of = np.repeat(np.arange(P["nepochs"]), P["epochs"][:, BE_NSTEP])
            for u in range(P["nsteps"] - 1, -1, -1):
                if u == starts[ep_of[u] + 1] - 1:
                    E = P["epochs"][ep_of[u]]
                    blk[:, :ZERO_OFF] = np.nan
                    self._load_pieces(blk, E[BE_BP: BE_BP + NPIECE], (vals, rhs, Lq))
                    assert o == E[BE_BOFF]
                st = ops[o: o + SW]
                fl = int(st[BS_FLAGS])
                nrows = (fl >> 8) & 255
                v = lambda off: blk[:, off // 8]
                live = [(u - HW + k) % NS for k in range(NB)]
                ps = live[0]
                if fl & BF_PIVOT_X:
                    off = st[BS_LX] // 8
                    x = blk[:, off + 5].copy()
                    for k in range(1, NB):
                        x = x - blk[:, off + k - 1] * xs[live[k]]
                    x = x - blk[:, off + 4] * xs[NS]
                    xs[ps] = x
                    blk[:, st[BS_DX] // 8] = -x
                if fl & BF_PIVOT_B:
                    off = st[BS_LB] // 8
                    x = blk[:, off + NB].copy()
                    for k in range(NB):
                        x = x - blk[:, off + k] * xs[live[k]]
                    xs[NS] = x
                    d[:, P["borders"][st[BS_BORDER]][2]] = -x
                for i in range(nrows):
                    rb = ops[o + SW + RW * i: o + SW + RW * (i + 1)]
                    acc = -v(rb[BR_RR])
                    for k in range(NB):
                        acc = acc + v(rb[BR_J0 + k]) * xs[live[k]]
                    blk[:, rb[BR_DR] // 8] = acc / v(rb[BR_DI])
                if fl & BF_ENTER_B:
                    xs[NS] = 0.0
                o += SW + RW * nrows
                if u == starts[ep_of[u]]:
                    E = P["epochs"][ep_of[u]]
                    d[:, E[BE_DXLO]: E[BE_DXLO] + E[BE_DXCNT]] = blk[:, DX_OFF: DX_OFF + E[BE_DXCNT]]
                    d[:, E[BE_DRLO]: E[BE_DRLO] + E[BE_DRCNT]] = blk[:, DR_OFF: DR_OFF + E[BE_DRCNT]]

    # ---- newton_system! (src/CaNNOLeS.jl:1008-1052), batched ----------------------------------------------------------
    def newton_system(self, vals, rhs, nvar, rho_old, params):
        vals = np.atleast_2d(vals)
        rhs = np.atleast_2d(rhs)
        B = vals.shape[0]
        tol, kdec, kinc, klarge, rho0, rhomax, rhomin = params[0], params[2], params[3], params[4], params[5], params[6], params[7]
        rho = np.zeros(B)
        ro = np.broadcast_to(np.asarray(rho_old, float), (B,)).copy()
        nf = np.zeros(B, np.int64)
        done = np.zeros(B, bool)
        succ = np.zeros(B, bool)
        ovr = np.zeros(B, bool)
        wrote = np.zeros(B)
        Lst = np.zeros((B, self.lsize))
        while True:
            with np.errstate(all="ignore"):
                npos, nzer, junction = self.forward(vals, rhs, rho, ovr, tol, Lst)
            ok = (npos == nvar) & (nzer == 0)
            for b in range(B):
                if done[b]:
                    continue
                nf[b] += 1
                if ok[b]:
                    done[b] = succ[b] = True
                elif nf[b] == 1:
                    rho[b] = rho0 if ro[b] == 0.0 else max(rhomin, kdec * ro[b])
                    ovr[b] = True
                    wrote[b] = rho[b]
                elif rho[b] <= rhomax:
                    rho[b] = klarge * rho[b] if ro[b] == 0.0 else kinc * rho[b]
                    if rho[b] <= rhomax:
                        wrote[b] = rho[b]
                    else:
                        done[b] = True
                else:
                    done[b] = True
            if done.all():
                break
        for b in range(B):
            if nf[b] > 1:
                if rho[b] <= rhomax:
                    ro[b] = rho[b]
                vals[b, -nvar:] = wrote[b]
        d = np.zeros((B, self.N))
        with np.errstate(all="ignore"):
            self.backward(vals, rhs, Lst, junction, d)
        return d, succ, rho, ro, nf

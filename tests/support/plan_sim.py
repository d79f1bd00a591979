"""numpy interpreter of the multifrontal plan (test infrastructure).

Executes, for ONE problem and on the CPU, exactly the schedule the HIP kernel
executes (cannoles.jl_amd/csrc/kernels.hip): assembly rounds, extend-add,
reversed-order pivot elimination with the independent-pivot shortcut, panel
store, stack moves, and the backward pass.  It validates the host analysis
(index maps, static stack offsets) without a GPU.  It is never used by the
product.
"""
import numpy as np

HDR = ["npiv", "nupd", "foff", "ubase", "seg_begin", "seg_end", "child_begin", "child_end", "rel_begin",
       "first_piv", "xoff", "parent", "lptr_lo", "lptr_hi", "indep", "pad"]


def tri(i):
    return i * (i + 1) // 2


class PlanSim:
    """Executes the plan of `plan` (a hipldl.Plan).  When the plan condenses the residual block
    (csrc/condense.h) the pre/post passes are emulated with numpy as well."""

    def __init__(self, plan):
        self.info = plan.info
        self.ncond = plan.info.get("ncond", 0)
        self.Nout, self.nnzout = plan.info["N"], plan.info["nnz"]
        if self.ncond:
            self.c_ptr = plan.array("c_ptr").astype(np.int64)
            self.c_a = plan.array("c_a").astype(np.int64)
            self.c_b = plan.array("c_b").astype(np.int64)
            self.c_d = plan.array("c_d").astype(np.int64)
            self.orig_of = plan.array("orig_of").astype(np.int64)
            self.r_orig = plan.array("r_orig").astype(np.int64)
            self.r_dsrc = plan.array("r_dsrc").astype(np.int64)
            self.r_ptr = plan.array("r_ptr").astype(np.int64)
            self.r_jsrc = plan.array("r_jsrc").astype(np.int64)
            self.r_jx = plan.array("r_jx").astype(np.int64)
        fr = plan.array("fronts").reshape(-1, 16)
        self.fr = {k: fr[:, i].astype(np.int64) for i, k in enumerate(HDR)}
        self.ns = fr.shape[0]
        self.seg_ptr = plan.array("seg_ptr").astype(np.int64)
        self.asm_pos = plan.array("asm_pos").astype(np.int64)
        self.asm_src = plan.array("asm_src").astype(np.int64)
        self.child_idx = plan.array("child_idx").astype(np.int64)
        self.rel_idx = plan.array("rel_idx").astype(np.int64)
        self.perm = plan.array("inner_perm").astype(np.int64)
        self.N = len(self.perm)                       # order of the (condensed) system the fronts work on
        # inner "nnz": rhs entries of the assembly lists are encoded as nnz_inner + index
        self.nnz = self.info["nnz"] if not self.ncond else int(self.c_ptr.size - 1 - self.N)
        fmax = self.info["fmax"]
        self.ti, self.tj = np.tril_indices(fmax)
        self.lptr = self.fr["lptr_lo"] | (self.fr["lptr_hi"] << 31)

    def check_layout(self):
        """static stack offsets never overlap live data (forward pass)"""
        f = self.fr
        for s in range(self.ns):
            fs = 1 + f["nupd"][s] + f["npiv"][s]
            kids = self.child_idx[f["child_begin"][s]:f["child_end"][s]]
            top = f["foff"][s]
            for c in kids:
                assert f["ubase"][c] + tri(1 + f["nupd"][c]) <= top, "child update overlaps the front"
            assert f["ubase"][s] <= f["foff"][s]
            assert f["foff"][s] + tri(fs) <= self.info["fwd_peak"]
            assert f["xoff"][s] + fs <= self.info["bwd_peak"]

    def factor(self, vals, rhs, nvar, eig_tol, rho=None):
        f = self.fr
        W = np.full(self.info["fwd_peak"] + 8, np.nan)
        L = np.full(self.info["lsize"], np.nan)
        src_all = np.concatenate([np.asarray(vals, float), np.zeros(self.N) if rhs is None else np.asarray(rhs, float)])
        if rho is not None:
            src_all[self.nnz - nvar:self.nnz] = rho
        npos = nzer = 0
        for s in range(self.ns):
            nupd, npiv = f["nupd"][s], f["npiv"][s]
            fs = 1 + nupd + npiv
            tf, tu = tri(fs), tri(1 + nupd)
            F = W[f["foff"][s]:f["foff"][s] + tf]
            F[:] = 0.0
            for r in range(f["seg_begin"][s], f["seg_end"][s]):
                e0, e1 = self.seg_ptr[r], self.seg_ptr[r + 1]
                pos = self.asm_pos[e0:e1]
                assert len(np.unique(pos)) == len(pos), "duplicate position inside an assembly round"
                F[pos] += src_all[self.asm_src[e0:e1]]
            for c in self.child_idx[f["child_begin"][s]:f["child_end"][s]]:
                nu = f["nupd"][c]
                tuc = tri(1 + nu)
                U = W[f["ubase"][c]:f["ubase"][c] + tuc]
                rel = self.rel_idx[f["rel_begin"][c]:f["rel_begin"][c] + 1 + nu]
                dest = tri(rel[self.ti[:tuc]]) + rel[self.tj[:tuc]]
                assert len(np.unique(dest)) == len(dest)
                assert np.all(rel[self.ti[:tuc]] >= rel[self.tj[:tuc]])
                F[dest] += U
            idep = fs - f["indep"][s]
            for i in range(fs - 1, nupd, -1):
                ulim = idep if i >= idep else i
                row = F[tri(i):tri(i) + i + 1]
                dp = row[i]
                npos += dp > eig_tol
                nzer += abs(dp) <= eig_tol
                w = row[:ulim].copy()
                with np.errstate(all="ignore"):
                    l = w / dp
                row[:ulim] = l
                tul = tri(ulim)
                with np.errstate(all="ignore"):
                    F[:tul] -= l[self.ti[:tul]] * w[self.tj[:tul]]
            L[self.lptr[s]:self.lptr[s] + tf - tu] = F[tu:tf]
            if f["ubase"][s] != f["foff"][s]:
                W[f["ubase"][s]:f["ubase"][s] + tu] = F[:tu].copy()
        return L, int(npos), int(nzer)

    def backward(self, L):
        f = self.fr
        X = np.full(self.info["bwd_peak"] + 8, np.nan)
        d = np.zeros(self.N)
        for s in range(self.ns - 1, -1, -1):
            nupd, npiv = f["nupd"][s], f["npiv"][s]
            fs = 1 + nupd + npiv
            tu = tri(1 + nupd)
            panel = L[self.lptr[s]:self.lptr[s] + tri(fs) - tu]
            xo = f["xoff"][s]
            if f["parent"][s] >= 0:
                rel = self.rel_idx[f["rel_begin"][s]:f["rel_begin"][s] + 1 + nupd]
                xp = f["xoff"][f["parent"][s]]
                X[xo + 1:xo + 1 + nupd] = X[xp + rel[1:]].copy()
            for i in range(nupd + 1, fs):
                row = panel[tri(i) - tu:tri(i) - tu + i + 1]
                xi = row[0] - np.dot(row[1:i], X[xo + 1:xo + i])
                X[xo + i] = xi
                d[self.perm[f["first_piv"][s] + (fs - 1 - i)]] = -xi
        return d

    def condense(self, vals, rhs):
        """numpy twin of condense_kernel: [K2 slots | rho tail | condensed rhs]"""
        x = np.concatenate([np.asarray(vals, float), np.asarray(rhs, float)])
        out = np.zeros(len(self.c_ptr) - 1)
        plain = self.c_b < 0
        with np.errstate(all="ignore"):
            contrib = np.where(plain, x[self.c_a], -(x[self.c_a] * x[np.maximum(self.c_b, 0)]) / x[np.maximum(self.c_d, 0)])
        slot = np.repeat(np.arange(len(out)), np.diff(self.c_ptr))
        np.add.at(out, slot, contrib)
        return out

    def expand(self, vals, rhs, d2):
        d = np.zeros(self.Nout)
        d[self.orig_of] = d2
        for q in range(len(self.r_orig)):
            k = slice(self.r_ptr[q], self.r_ptr[q + 1])
            s_ = rhs[self.r_orig[q]] + np.dot(vals[self.r_jsrc[k]], d2[self.r_jx[k]])
            d[self.r_orig[q]] = -s_ / vals[self.r_dsrc[q]]
        return d

    def newton_system(self, vals, rhs, nvar, nequ, ncon, rho_old, params):
        if self.ncond:
            cb = self.condense(vals, rhs)
            nmat = len(cb) - self.N
            dr = np.asarray(vals)[self.r_dsrc]
            xpos, xzer = int((dr > params[0]).sum()), int((np.abs(dr) <= params[0]).sum())
            d2, ok, rho, rho_old, nfact = self._newton_inner(cb[:nmat].copy(), cb[nmat:], nvar, rho_old, params, xpos, xzer)
            return (self.expand(np.asarray(vals), np.asarray(rhs), d2) if ok else np.zeros(self.Nout)), ok, rho, rho_old, nfact
        return self._newton_inner(vals, rhs, nvar, rho_old, params, 0, 0)

    def _newton_inner(self, vals, rhs, nvar, rho_old, params, xpos, xzer):
        """same ladder as the kernel / src/CaNNOLeS.jl:1019-1051"""
        nvar_ = nvar
        def inert(npos, nzer):
            return npos + xpos == nvar_ and nzer + xzer == 0
        eig_tol, kdec, kinc, klarge, rho0, rhomax, rhomin = params[0], params[2], params[3], params[4], params[5], params[6], params[7]
        rho, nfact = 0.0, 0
        L, npos, nzer = self.factor(vals, rhs, nvar, eig_tol)
        nfact += 1
        ok = inert(npos, nzer)
        if not ok:
            rho = rho0 if rho_old == 0 else max(rhomin, kdec * rho_old)
            L, npos, nzer = self.factor(vals, rhs, nvar, eig_tol, rho)
            nfact += 1
            ok = inert(npos, nzer)
            while not ok and rho <= rhomax:
                rho = klarge * rho if rho_old == 0 else kinc * rho
                if rho <= rhomax:
                    L, npos, nzer = self.factor(vals, rhs, nvar, eig_tol, rho)
                    nfact += 1
                    ok = inert(npos, nzer)
            if rho <= rhomax:
                rho_old = rho
        d = self.backward(L) if ok else np.zeros(self.N)
        return d, ok, rho, rho_old, nfact

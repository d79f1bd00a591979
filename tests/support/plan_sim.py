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
    def __init__(self, plan):
        self.info = plan.info
        fr = plan.array("fronts").reshape(-1, 16)
        self.fr = {k: fr[:, i].astype(np.int64) for i, k in enumerate(HDR)}
        self.ns = fr.shape[0]
        self.seg_ptr = plan.array("seg_ptr").astype(np.int64)
        self.asm_pos = plan.array("asm_pos").astype(np.int64)
        self.asm_src = plan.array("asm_src").astype(np.int64)
        self.child_idx = plan.array("child_idx").astype(np.int64)
        self.rel_idx = plan.array("rel_idx").astype(np.int64)
        self.perm = plan.array("perm").astype(np.int64)
        self.N, self.nnz = self.info["N"], self.info["nnz"]
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

    def newton_system(self, vals, rhs, nvar, nequ, ncon, rho_old, params):
        """same ladder as the kernel / src/CaNNOLeS.jl:1019-1051"""
        eig_tol, kdec, kinc, klarge, rho0, rhomax, rhomin = params[0], params[2], params[3], params[4], params[5], params[6], params[7]
        rho, nfact = 0.0, 0
        L, npos, nzer = self.factor(vals, rhs, nvar, eig_tol)
        nfact += 1
        ok = npos == nvar and nzer == 0
        if not ok:
            rho = rho0 if rho_old == 0 else max(rhomin, kdec * rho_old)
            L, npos, nzer = self.factor(vals, rhs, nvar, eig_tol, rho)
            nfact += 1
            ok = npos == nvar and nzer == 0
            while not ok and rho <= rhomax:
                rho = klarge * rho if rho_old == 0 else kinc * rho
                if rho <= rhomax:
                    L, npos, nzer = self.factor(vals, rhs, nvar, eig_tol, rho)
                    nfact += 1
                    ok = npos == nvar and nzer == 0
            if rho <= rhomax:
                rho_old = rho
        d = self.backward(L) if ok else np.zeros(self.N)
        return d, ok, rho, rho_old, nfact

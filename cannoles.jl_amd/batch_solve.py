"""Batched outer loop over the batched C ABI.

B problems that share one KKT sparsity pattern are solved concurrently: every problem runs the restated
`solve!` (outer_loop.solve, /root/reference/src/CaNNOLeS.jl:418-864) in its own host thread, and all Newton
systems of one "round" go to the device together through ONE batched `cnl_newton_system` call on a handle
created with batch = B (`NewtonBroker`).  Problems that have finished keep their last system in their slot
(solved again, result ignored), so the batch shape never changes; problems that take fewer inner iterations
simply post less often.  The per-problem logic is exactly the single-problem loop, so a batch of one
reproduces it.

The batched Newton step is the HIP path (`hipldl.newton_system_`, i.e. `cnl_newton_system`); there is no CPU fallback:
without the extension or a device the first round raises.  The batched executor is a parameter of the broker
(`executor(vals, rhs, rho_old, params, dims) -> (d, ok, rho, rho_old, nfact)`, like the per-rank executor of
sharding.run_shard); its default is the HIP handle, the CPU test-suite passes one built on the oracle to exercise the
rendezvous logic where no GPU exists.
"""
import threading

import numpy as np


class NewtonBroker:
    """Rendezvous of B outer loops on one batched linear-solver handle."""

    def __init__(self, B, device=0, executor=None):
        self.B, self.device = int(B), device
        self._batched = executor
        self.cv = threading.Condition()
        self.handle = None
        self.active = self.B
        self.posted = set()
        self.gen = 0
        self.ncalls = 0
        self.failed = None

    # --- the two callables outer_loop.solve expects ------------------------------------------------------
    def make_solver(self, slot):
        def mk(N, rows, cols, vals, nvar, nequ, ncon):
            with self.cv:
                if self.handle is None:
                    self.rows, self.cols = np.array(rows, np.int64), np.array(cols, np.int64)
                    self.dims = (int(N), int(nvar), int(nequ), int(ncon))
                    if self._batched is None:
                        from . import hipldl
                        self.handle = hipldl.HIPLDLStruct(N, self.rows, self.cols, None, nvar, nequ, ncon, batch=self.B,
                                                          device=self.device)
                    else:
                        self.handle = object()
                    nnz = len(self.rows)
                    self.vals = np.tile(np.asarray(vals, float), (self.B, 1))
                    self.rhs = np.zeros((self.B, N))
                    self.rho_old = np.zeros(self.B)
                    self.out = None
                    assert self.vals.shape == (self.B, nnz)
                elif (int(N), int(nvar), int(nequ), int(ncon)) != self.dims or not (
                        np.array_equal(rows, self.rows) and np.array_equal(cols, self.cols)):
                    raise ValueError("all problems of a batch must share one KKT pattern")
            return slot
        return mk

    def newton_system(self, slot, nvar, nequ, ncon, rhs, vals, rho_old, params):
        with self.cv:
            self.vals[slot] = vals
            self.rhs[slot] = rhs
            self.rho_old[slot] = rho_old
            self.posted.add(slot)
            gen = self.gen
            if len(self.posted) >= self.active:
                self._run(params)
            else:
                while self.gen == gen and self.failed is None:
                    self.cv.wait()
            if self.failed is not None:
                raise RuntimeError("batched Newton step failed") from self.failed
            d, ok, rho, ro, nf = self.out
            vals[-nvar:] = self.vals[slot, -nvar:]  # rho slots as the reference leaves them (src/CaNNOLeS.jl:1027,1041)
            return d[slot].copy(), bool(ok[slot]), float(rho[slot]), float(ro[slot]), int(nf[slot])

    def finish(self, slot, params):
        """The problem in `slot` is done: the others no longer wait for it."""
        with self.cv:
            self.active -= 1
            self.posted.discard(slot)
            if self.active > 0 and len(self.posted) >= self.active:
                self._run(params)

    # --- one batched device call ---------------------------------------------------------------------------
    def _run(self, params):
        N, nvar, nequ, ncon = self.dims
        try:
            if self._batched is None:
                from . import hipldl
                d = np.zeros((self.B, N))
                self.out = hipldl.newton_system_(d, nvar, nequ, ncon, self.rhs, self.vals, self.handle, self.rho_old, params)
                if self.B == 1:
                    d1, ok, rho, ro, nf = self.out
                    self.out = (d1.reshape(1, N), np.array([ok]), np.array([rho]), np.array([ro]), np.array([nf]))
            else:
                self.out = self._batched(self.rows, self.cols, self.dims, self.rhs, self.vals, self.rho_old, params)
        except Exception as e:  # wake everybody up, they re-raise
            self.failed = e
        self.ncalls += 1
        self.posted.clear()
        self.gen += 1
        self.cv.notify_all()

    def close(self):
        if self._batched is None and self.handle is not None:
            self.handle.close()
        self.handle = None


def solve_batch(models, params=None, device=0, executor=None, **kw):
    """Runs outer_loop.solve for every model of `models` (same pattern), Newton systems batched on the device.
    Returns (list of result dicts, number of batched device calls)."""
    from . import outer_loop
    if params is None:
        from . import hipldl
        params = hipldl.default_params()
    B = len(models)
    broker = NewtonBroker(B, device, executor)
    results = [None] * B
    errors = [None] * B

    def run(k):
        try:
            results[k] = outer_loop.solve(models[k], broker.make_solver(k),
                                          lambda LDLT, n, m, p, rhs, vals, ro, prm: broker.newton_system(k, n, m, p, rhs, vals, ro, prm),
                                          params, **kw)
        except Exception as e:
            errors[k] = e
        finally:
            broker.finish(k, params)

    threads = [threading.Thread(target=run, args=(k,)) for k in range(B)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    ncalls = broker.ncalls
    broker.close()
    for e in errors:
        if e is not None:
            raise e
    return results, ncalls

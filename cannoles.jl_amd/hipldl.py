"""Host-side mirror of the reference's linear-solver plugin surface, over the C ABI.

The reference's extension point is `abstract type LinearSolverStruct`
(/root/reference/src/solver_types.jl:1) with
  * a constructor `Backend(N, rows, cols, vals)`        (src/CaNNOLeS.jl:323,327)
  * `get_vals(LDLT)`                                      (src/solver_types.jl:25,67)
  * `try_to_factorize(LDLT, vals, nvar, nequ, ncon, eig_tol)::Bool` (solver_types.jl:79-98)
  * `solve_ldl!(rhs, LDLT.factor, d)::Bool`               (solver_types.jl:69-77)
and the driver `newton_system!` (src/CaNNOLeS.jl:1008-1052).  The same names
and argument meanings are kept here (`!` spelled `_`), so the parity tests read
like the reference's own code.  All arithmetic happens in libcannoles_hip.so
(HIP kernels); this module only marshals numpy / torch buffers through ctypes.
Julia would bind the same symbols with `ccall`.

There is no CPU fallback: if the shared library is missing, or no HIP device
is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CANNOLES_HIP_LIB") or os.path.join(_HERE, "libcannoles_hip.so")  # same override as the Julia glue

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_lib = None

# symbols declared in include/cannoles_hip.h (checked by the CPU test-suite)
ABI_SYMBOLS = [
    "cnl_last_error", "cnl_version", "cnl_default_params",
    "cnl_options_init", "cnl_plan_create_ex", "cnl_create_ex", "cnl_dataflow_timeouts",
    "cnl_plan_create", "cnl_plan_create_for_batch", "cnl_plan_destroy", "cnl_plan_info", "cnl_plan_get", "cnl_plan_order_name",
    "cnl_create", "cnl_destroy", "cnl_get_plan",
    "cnl_factorize", "cnl_solve", "cnl_newton_system",
    "cnl_factorize_dev", "cnl_solve_dev", "cnl_newton_system_dev",
    "cnl_set_timing", "cnl_last_kernel_ms", "cnl_get_config", "cnl_launch_counts",
    "cnl_residual_vectors_dev", "cnl_trial_point_dev", "cnl_prepare_newton_system_dev",
    "cnl_cgls_multipliers_dev",
    "cnl_multi_create_ex", "cnl_multi_factorize_dev", "cnl_multi_solve_dev", "cnl_multi_newton_system_dev", "cnl_multi_synchronize",
    "cnl_multi_create", "cnl_multi_destroy", "cnl_multi_shards", "cnl_multi_factorize", "cnl_multi_solve", "cnl_multi_newton_system",
    "cnl_outer_begin_dev", "cnl_outer_newton_done_dev", "cnl_outer_extrapolated_dev", "cnl_outer_trial_done_dev", "cnl_outer_end_dev",
    "cnl_outer_ls_begin_dev", "cnl_outer_ls_test_dev", "cnl_outer_ls_step_dev", "cnl_outer_ls_take_dev",
    "cnl_layout_len", "cnl_interleave_dev", "cnl_deinterleave_dev", "cnl_residual_vectors_jac_dev", "cnl_cgls_multipliers_jac_dev",
]


PLAN_AUTO, PLAN_THROUGHPUT, PLAN_LATENCY = 0, 1, 2
LAYOUT_PROBLEM_MAJOR, LAYOUT_INTERLEAVED = 0, 1   # cnl_options.batch_layout (include/cannoles_hip.h)
IL_GROUP = 32


def il_blocks(length):
    """blocks of eight doubles per problem of an interleaved array (one spare block; csrc/band.h: band_il_blocks)"""
    return (int(length) + 7) // 8 + 1


def il_len(batch, length):
    return (int(batch) + IL_GROUP - 1) // IL_GROUP * il_blocks(length) * IL_GROUP * 8


def il_index(p, e, length):
    """position of element e of problem p in an array interleaved over groups of 32 problems in blocks of eight doubles
    (CNL_LAYOUT_INTERLEAVED, csrc/band.h: band_il_index); p, e may be numpy arrays"""
    return ((p // IL_GROUP * il_blocks(length) + e // 8) * IL_GROUP + p % IL_GROUP) * 8 + e % 8


class cnl_options(C.Structure):
    """struct cnl_options of include/cannoles_hip.h (field order and types must match)"""
    _fields_ = [("struct_size", C.c_int32), ("plan_kind", C.c_int32), ("staged_max_batch", C.c_int64)] + [
        (k, C.c_int32) for k in ("verbose", "band_kernel", "dense_backend", "staged", "dataflow", "device_ladder", "host_ladder", "split_tail",
                                 "multi_share_plan", "batch_layout")] + [("force_order", C.c_char * 32), ("tuning", C.c_char * 192)]


_PUBLIC_OPTIONS = {f[0] for f in cnl_options._fields_} - {"struct_size", "tuning"}


class cnl_outer_state(C.Structure):
    """struct cnl_outer_state of include/cannoles_hip.h (row f3: bookkeeping of the batched outer loop); field order must match"""
    _fields_ = ([(k, C.c_int64) for k in ("B", "n", "m", "p", "P", "N", "nnzjF", "nnzjc", "max_inner")] +
                [(k, C.c_double) for k in ("dmin", "rhomax", "delta_dec", "smax")] +
                [(k, C.c_void_p) for k in (
                    "status", "it", "flags", "nf_new", "ok_new",
                    "inner", "nfact", "nlin",
                    "phase0", "act", "need", "brk", "ext", "lsm", "rej", "chk", "done_in", "tired", "small_res",
                    "normdual", "normprimal", "combined", "combined_hat", "delta", "ndh", "nph", "fx", "epsk", "epstol", "epsF", "epsc", "rho_old",
                    "d", "d_new", "ro_tmp", "rho_new",
                    "x", "r", "Fx", "cx", "Jv", "Jcv", "lam", "rhs_cur",
                    "xt", "rt", "Ft", "ct", "Jt", "Jct", "lamt", "rhs_t", "nrm_t",
                    "xt_e", "rt_e", "lamt_e")] +
                [(k, C.c_double) for k in ("gammaA", "eps2")] +
                [(k, C.c_void_p) for k in ("ls_g", "xl", "Fl", "cl", "lam_ls", "alpha", "Dphi", "phix", "eta", "nbk", "bt")])


def Options(**kw):
    """cnl_options with the library's defaults (cnl_options_init), then the given switches: explicit arguments instead of
    environment variables — e.g. Options(plan_kind=PLAN_THROUGHPUT), Options(dataflow=0), Options(force_order="nd32+early").
    Fields of the public structure are set directly; any other switch of csrc/options.h (lean_kernel, band_form,
    band_problems_per_group, ...) goes into its `tuning` string as key=value — the library rejects an unknown key."""
    o = cnl_options()
    lib().cnl_options_init(C.byref(o))
    assert o.struct_size == C.sizeof(cnl_options), "cnl_options layout differs from the library's"
    tun = []
    for k, v in kw.items():
        if k == "force_order":
            v = v.encode() if isinstance(v, str) else v
        if k == "tuning":
            tun.append(v.decode() if isinstance(v, bytes) else str(v))
        elif k in _PUBLIC_OPTIONS:
            setattr(o, k, v)
        else:
            tun.append(f"{k}={int(v)}")
    text = ",".join(t for t in tun if t)
    if len(text) >= 192:
        raise ValueError("cnl_options.tuning holds 191 characters")
    o.tuning = text.encode()
    return o


def _optref(options):
    if options is None:
        return None
    if isinstance(options, dict):
        options = Options(**options)
    return C.byref(options)


class CnlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cannoles_hip error {code}: {msg}")
        self.code = code


def loaded_hip_runtimes():
    """paths of the HIP runtimes (libamdhip64) mapped into this process"""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def _one_hip_runtime():
    """ONE HIP runtime per process, whatever the import order.  libcannoles_hip.so needs `libamdhip64.so.7`.  A PyTorch wheel
    bundles its own copy (same SONAME, found through torch's rpath under the name `libamdhip64.so`):
      * torch imported first: the loader satisfies our dependency with torch's copy (SONAME match) — one runtime;
      * our library first, torch later: ours would pull in the system ROCm's runtime, torch then its bundled copy — two runtimes,
        and the one that initialises second finds no device.
    So when this process has no HIP runtime yet and a torch wheel with a bundled runtime is INSTALLED (not imported here), that
    copy is loaded first: our library binds to it and a later `import torch` finds it already loaded.  Without torch (the Julia
    drop-in, a C caller) the system runtime is used as linked."""
    if loaded_hip_runtimes():
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # fall back to the runtime the library is linked against


def lib():
    """Load libcannoles_hip.so (built by __graft_entry__.build / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        _one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int32
        L.cnl_last_error.restype = C.c_char_p
        L.cnl_version.restype = i32
        L.cnl_default_params.argtypes = [vp]
        L.cnl_plan_create.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64]
        L.cnl_plan_create_for_batch.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64, i64]
        L.cnl_options_init.argtypes = [vp]
        L.cnl_options_init.restype = None
        L.cnl_plan_create_ex.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64, i64, vp]
        L.cnl_create_ex.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64, i64, C.c_int, vp]
        L.cnl_dataflow_timeouts.argtypes = [vp, C.POINTER(i64)]
        L.cnl_plan_destroy.argtypes = [vp]
        L.cnl_plan_destroy.restype = None
        L.cnl_plan_info.argtypes = [vp, _i64p]
        L.cnl_plan_get.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64)]
        L.cnl_plan_order_name.argtypes = [vp]
        L.cnl_plan_order_name.restype = C.c_char_p
        L.cnl_create.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64, i64, C.c_int]
        L.cnl_destroy.argtypes = [vp]
        L.cnl_get_plan.argtypes = [vp]
        L.cnl_get_plan.restype = vp
        L.cnl_factorize.argtypes = [vp, vp, dbl, vp, vp, vp]
        L.cnl_solve.argtypes = [vp, vp, vp]
        L.cnl_newton_system.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_factorize_dev.argtypes = [vp, vp, dbl, vp, vp]
        L.cnl_solve_dev.argtypes = [vp, vp, vp, vp]
        L.cnl_newton_system_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_residual_vectors_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_trial_point_dev.argtypes = [vp, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp]
        L.cnl_prepare_newton_system_dev.argtypes = [vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_cgls_multipliers_dev.argtypes = [vp, vp, vp, vp, vp, dbl, dbl, i64, C.c_int, vp, vp]
        L.cnl_multi_create.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64, i64, vp, C.c_int]
        L.cnl_multi_create_ex.argtypes = [C.POINTER(vp), i64, i64, _i64p, _i64p, i64, i64, i64, i64, vp, C.c_int, vp]
        L.cnl_multi_factorize_dev.argtypes = [vp, vp, dbl, vp, vp]
        L.cnl_multi_solve_dev.argtypes = [vp, vp, vp, vp]
        L.cnl_multi_newton_system_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_multi_synchronize.argtypes = [vp, vp]
        L.cnl_multi_destroy.argtypes = [vp]
        L.cnl_multi_shards.argtypes = [vp, C.POINTER(i64), vp, vp, vp]
        L.cnl_multi_factorize.argtypes = [vp, vp, dbl, vp, vp, vp]
        L.cnl_multi_solve.argtypes = [vp, vp, vp]
        L.cnl_multi_newton_system.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_set_timing.argtypes = [vp, C.c_int]
        L.cnl_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.cnl_get_config.argtypes = [vp, _i64p]
        L.cnl_launch_counts.argtypes = [_i64p]
        L.cnl_layout_len.argtypes = [vp, C.c_int, C.POINTER(i64)]
        L.cnl_residual_vectors_jac_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cnl_cgls_multipliers_jac_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, dbl, dbl, i64, C.c_int, vp, vp]
        L.cnl_interleave_dev.argtypes = [vp, C.c_int, vp, vp, vp]
        L.cnl_deinterleave_dev.argtypes = [vp, C.c_int, vp, vp, vp]
        for fn in ("cnl_outer_begin_dev", "cnl_outer_extrapolated_dev", "cnl_outer_trial_done_dev", "cnl_outer_end_dev", "cnl_outer_ls_begin_dev",
                   "cnl_outer_ls_step_dev", "cnl_outer_ls_take_dev"):
            getattr(L, fn).argtypes = [vp, vp]
        L.cnl_outer_newton_done_dev.argtypes = [vp, C.c_int, vp]
        L.cnl_outer_ls_test_dev.argtypes = [vp, C.c_int, vp]
        if L.cnl_version() < 0 and not os.environ.get("CANNOLES_HIP_ALLOW_EXPERIMENT"):
            raise RuntimeError(f"{LIB_PATH} is an EXPERIMENT build (cnl_version() = {L.cnl_version()}: timing probes / diagnostic "
                               "stamps compiled in, results may be wrong); set CANNOLES_HIP_ALLOW_EXPERIMENT=1 to load it on purpose")
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise CnlError(rc, lib().cnl_last_error().decode())


def default_params():
    """ParamCaNNOLeS(Float64) (src/CaNNOLeS.jl:48-62) as
    [eig_tol, dmin, kdec, kinc, klargeinc, rho0, rhomax, rhomin, gammaA]."""
    p = np.zeros(9)
    lib().cnl_default_params(p.ctypes.data)
    return p


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64).ravel())


_INFO_KEYS = ["N", "nnz", "nnzK", "nsuper", "nnzL", "nnzL_exact", "lsize", "fmax", "fwd_peak", "bwd_peak",
              "panel_max", "flops", "nasm", "v2_classes", "v2_lds", "ncond"]


def _plan_info(p):
    info = np.zeros(16, np.int64)
    _check(lib().cnl_plan_info(p, info))
    d = {k: int(info[i]) for i, k in enumerate(_INFO_KEYS)}
    d["order"] = lib().cnl_plan_order_name(p).decode()
    c, l = d.pop("v2_classes"), d.pop("v2_lds")
    if c >= 0:
        d["v2"] = {"fronts16": c & 0xfffff, "fronts32": (c >> 20) & 0xfffff, "fronts64": c >> 40,
                   "ustack": l & 0xfffff, "staging": (l >> 20) & 0xfffff, "reclen": l >> 40}
    else:
        d["v2"] = None
    return d


def _plan_array(p, name):
    n = C.c_int64(0)
    _check(lib().cnl_plan_get(p, name.encode(), None, C.byref(n)))
    out = np.zeros(max(n.value, 1), np.int32)
    cnt = C.c_int64(out.size)
    _check(lib().cnl_plan_get(p, name.encode(), out.ctypes.data, C.byref(cnt)))
    return out[:n.value]


class Plan:
    """Host-only symbolic analysis (what `ldl_analyze` is to the reference).
    Needs no GPU; used by the CPU test-suite and for sizing."""

    def __init__(self, N, rows, cols, nvar, nequ, ncon, batch=None, options=None):
        self.rows, self.cols = _i64(rows), _i64(cols)
        p = C.c_void_p()
        if options is not None:   # explicit options (batch None: the large-batch analysis unless the options say otherwise)
            _check(lib().cnl_plan_create_ex(C.byref(p), int(N), len(self.rows), self.rows, self.cols, int(nvar), int(nequ), int(ncon),
                                            int(batch or 0), _optref(options)))
        elif batch is None:  # the large-batch analysis
            _check(lib().cnl_plan_create(C.byref(p), int(N), len(self.rows), self.rows, self.cols, int(nvar), int(nequ), int(ncon)))
        else:              # what cnl_create(batch) runs
            _check(lib().cnl_plan_create_for_batch(C.byref(p), int(N), len(self.rows), self.rows, self.cols, int(nvar), int(nequ),
                                                   int(ncon), int(batch)))
        self._p = p
        self.info = _plan_info(p)

    def array(self, name):
        return _plan_array(self._p, name)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().cnl_plan_destroy(self._p)
            self._p = None


class _Factor:
    """What `LDLT.factor` is to the reference: the object `solve_ldl!` dispatches on."""

    def __init__(self, owner):
        self._owner = owner


class HIPLDLStruct:
    """`struct HIPLDLStruct <: LinearSolverStruct` — drop-in for LDLFactStruct
    (src/solver_types.jl:45-65).  `batch > 1` is the batched twin: one shared
    pattern, problem-major values `vals[b, :]`."""

    def __init__(self, N, rows, cols, vals, nvar=None, nequ=None, ncon=None, batch=1, device=0, options=None):
        self.N = int(N)
        self.rows, self.cols = _i64(rows), _i64(cols)
        self.nnz = len(self.rows)
        self.batch = int(batch)
        if nvar is None:
            raise ValueError("nvar/nequ/ncon are required (the Julia glue passes them from the solver)")
        self.nvar, self.nequ, self.ncon = int(nvar), int(nequ), int(ncon)
        # `vals` is aliased, not copied: the driver mutates the array returned by get_vals
        self.vals = vals if vals is not None else np.ones((self.batch, self.nnz) if self.batch > 1 else self.nnz)
        h = C.c_void_p()
        if options is None:
            _check(lib().cnl_create(C.byref(h), self.N, self.nnz, self.rows, self.cols, self.nvar, self.nequ, self.ncon,
                                    self.batch, int(device)))
        else:
            _check(lib().cnl_create_ex(C.byref(h), self.N, self.nnz, self.rows, self.cols, self.nvar, self.nequ, self.ncon,
                                       self.batch, int(device), _optref(options)))
        self._h = h
        self.factor = _Factor(self)
        self.info = _plan_info(lib().cnl_get_plan(h))
        cfg = np.zeros(8, np.int64)
        _check(lib().cnl_get_config(h, cfg))
        self.config = {"tpp": int(cfg[0]), "ppb": int(cfg[1]), "lds_bytes": int(cfg[2]), "lds_work": int(cfg[3]), "grid": int(cfg[4]),
                       "kernel": {2: "v2", 3: "dense", 4: "v2-staged"}.get(int(cfg[5]) & 15, "v1"), "wpb": int(cfg[6]), "lds2_bytes": int(cfg[7]),
                       "lean": bool(int(cfg[5]) & 16), "tail": bool(int(cfg[5]) & 32), "band": bool(int(cfg[5]) & 64), "f1_tiles": bool(int(cfg[5]) & 128),
                       "band_nl": (int(cfg[5]) >> 8) & 255, "band_parts": (int(cfg[5]) >> 16) & 255, "band_movers": bool((int(cfg[5]) >> 24) & 1),
                       "batch_layout": (int(cfg[5]) >> 25) & 1, "rhs_interleaved": bool((int(cfg[5]) >> 26) & 1)}

    def plan_array(self, name):
        return _plan_array(lib().cnl_get_plan(self._h), name)

    def close(self):
        if getattr(self, "_h", None):
            lib().cnl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timing(self, on=True):
        _check(lib().cnl_set_timing(self._h, 1 if on else 0))

    def uses_lean_kernel(self):
        """True when newton_system / factorize of this handle run the kernels' LEAN instantiation (cnl_get_config, cfg[5] bit 4)"""
        cfg = np.zeros(8, np.int64)
        _check(lib().cnl_get_config(self._h, cfg))
        return bool(int(cfg[5]) & 16)

    def dataflow_timeouts(self):
        n = C.c_int64(0)
        _check(lib().cnl_dataflow_timeouts(self._h, C.byref(n)))
        return int(n.value)

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(lib().cnl_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)


def launch_counts():
    """launches per kernel family since the library was loaded: {"band", "register_front", "general"} (cnl_launch_counts)"""
    c = np.zeros(3, np.int64)
    _check(lib().cnl_launch_counts(c))
    return {"band": int(c[0]), "register_front": int(c[1]), "general": int(c[2])}


def get_vals(LDLT):
    """get_vals(LDLT::LinearSolverStruct) = LDLT.vals (src/solver_types.jl:67)."""
    return LDLT.vals


def _f64c(a, shape=None):
    a = np.asarray(a)
    if a.dtype != np.float64 or not a.flags.c_contiguous:
        raise TypeError("expected a C-contiguous float64 array (Float64 only; other element types stay on LDLFactStruct)")
    return a


def try_to_factorize(LDLT, vals, nvar, nequ, ncon, eig_tol, return_inertia=False):
    """success = try_to_factorize(LDLT, vals, nvar, nequ, ncon, eig_tol) (src/solver_types.jl:79-98).
    Batched handles return arrays."""
    assert (nvar, nequ, ncon) == (LDLT.nvar, LDLT.nequ, LDLT.ncon)
    vals = _f64c(vals)
    assert vals.size == LDLT.batch * LDLT.nnz
    B = LDLT.batch
    succ = np.zeros(B, np.int32)
    npos = np.zeros(B, np.int64)
    nzer = np.zeros(B, np.int64)
    _check(lib().cnl_factorize(LDLT._h, vals.ctypes.data, float(eig_tol), succ.ctypes.data, npos.ctypes.data, nzer.ctypes.data))
    if B == 1:
        return (bool(succ[0]), int(npos[0]), int(nzer[0])) if return_inertia else bool(succ[0])
    return (succ.astype(bool), npos, nzer) if return_inertia else succ.astype(bool)


def solve_ldl_(rhs, factor, d):
    """solve_ldl!(rhs, factor, d): d = -(K^-1 rhs), returns true (src/solver_types.jl:69-77)."""
    LDLT = factor._owner
    rhs = _f64c(rhs)
    d = _f64c(d)
    assert rhs.size == LDLT.batch * LDLT.N and d.size == rhs.size
    _check(lib().cnl_solve(LDLT._h, rhs.ctypes.data, d.ctypes.data))
    return True


def newton_system_(d, nvar, nequ, ncon, rhs, vals, LDLT, rho_old, params):
    """d, solve_success, rho, rho_old, nfact = newton_system!(d, nvar, nequ, ncon, rhs, vals, LDLT, rho_old, params)
    (src/CaNNOLeS.jl:1008-1052), fused on the device.  `vals` is mutated (rho slots).
    Batched handles take/return arrays for rho_old / success / rho / nfact."""
    assert (nvar, nequ, ncon) == (LDLT.nvar, LDLT.nequ, LDLT.ncon)
    B = LDLT.batch
    vals = _f64c(vals)
    rhs = _f64c(rhs)
    d = _f64c(d)
    assert vals.size == B * LDLT.nnz and rhs.size == B * LDLT.N and d.size == B * LDLT.N
    ro = np.ascontiguousarray(np.broadcast_to(np.asarray(rho_old, dtype=np.float64), (B,)))
    params = np.ascontiguousarray(params, dtype=np.float64)
    rho = np.zeros(B)
    ro_out = np.zeros(B)
    nfact = np.zeros(B, np.int32)
    succ = np.zeros(B, np.int32)
    _check(lib().cnl_newton_system(LDLT._h, vals.ctypes.data, rhs.ctypes.data, d.ctypes.data, ro.ctypes.data,
                                   params.ctypes.data, rho.ctypes.data, ro_out.ctypes.data, nfact.ctypes.data,
                                   succ.ctypes.data))
    if B == 1:
        return d, bool(succ[0]), float(rho[0]), float(ro_out[0]), int(nfact[0])
    return d, succ.astype(bool), rho, ro_out, nfact.astype(np.int64)


def newton_system_dev(LDLT, vals_ptr, rhs_ptr, d_ptr, rho_old_ptr, rho_ptr, nfact_ptr, success_ptr, params, stream=0):
    """Device-resident twin (cnl_newton_system_dev): all *_ptr are device addresses
    (e.g. torch.Tensor.data_ptr()); asynchronous on `stream` (a hipStream_t value)."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    _check(lib().cnl_newton_system_dev(LDLT._h, vals_ptr, rhs_ptr, d_ptr, rho_old_ptr, rho_ptr, nfact_ptr, success_ptr,
                                       params.ctypes.data, stream))


def residual_vectors_dev(LDLT, vals_ptr, r_ptr, lambda_ptr, Fx_ptr, cx_ptr, rhs_ptr, norms_ptr, stream=0):
    """rhs = [dual; primal] with dual = Jx' r - Jc' lambda, primal = [F - r; c], and their infinity norms
    (src/CaNNOLeS.jl:507-508,519-524,528-529,631-632), batched and device-resident (cnl_residual_vectors_dev).
    All *_ptr are device addresses; norms_ptr receives [batch][2] doubles."""
    _check(lib().cnl_residual_vectors_dev(LDLT._h, vals_ptr, r_ptr, lambda_ptr, Fx_ptr, cx_ptr, rhs_ptr, norms_ptr, stream))


def residual_vectors_jac_dev(LDLT, nnzjF, nnzjc, Jx_ptr, Jcx_ptr, r_ptr, lambda_ptr, Fx_ptr, cx_ptr, rhs_ptr, norms_ptr, stream=0):
    """residual_vectors_dev with the Jacobian values read from the model's arrays Jx [batch][nnzjF], Jcx [batch][nnzjc] instead of
    from the J segments of vals (cnl_residual_vectors_jac_dev): no prepare pass in front, any batch_layout"""
    _check(lib().cnl_residual_vectors_jac_dev(LDLT._h, int(nnzjF), int(nnzjc), Jx_ptr, Jcx_ptr, r_ptr, lambda_ptr, Fx_ptr, cx_ptr, rhs_ptr,
                                              norms_ptr, stream))


def trial_point_dev(LDLT, x_ptr, r_ptr, lambda_ptr, d_ptr, max_dlambda, xt_ptr, rt_ptr, lambdat_ptr, dlambda_ptr, stream=0):
    """xt = x + dx, rt = r + dr, dlambda = -d[n+m+1:N] capped at max_dlambda in the 2-norm, lambdat = lambda + dlambda
    (src/CaNNOLeS.jl:654,661-668), batched and device-resident (cnl_trial_point_dev)."""
    _check(lib().cnl_trial_point_dev(LDLT._h, x_ptr, r_ptr, lambda_ptr, d_ptr, float(max_dlambda), xt_ptr, rt_ptr, lambdat_ptr,
                                     dlambda_ptr, stream))


def layout_len(LDLT, which):
    """doubles of the interleaved array of the handle's batch; which = 0: vals, 1: an N-vector per problem (cnl_layout_len)"""
    n = C.c_int64(0)
    _check(lib().cnl_layout_len(LDLT._h, int(which), C.byref(n)))
    return int(n.value)


def interleave_dev(LDLT, which, src_ptr, dst_ptr, stream=0):
    """problem-major -> CNL_LAYOUT_INTERLEAVED on the device, out of place (cnl_interleave_dev)"""
    _check(lib().cnl_interleave_dev(LDLT._h, int(which), src_ptr, dst_ptr, stream))


def deinterleave_dev(LDLT, which, src_ptr, dst_ptr, stream=0):
    """CNL_LAYOUT_INTERLEAVED -> problem-major on the device, out of place (cnl_deinterleave_dev)"""
    _check(lib().cnl_deinterleave_dev(LDLT._h, int(which), src_ptr, dst_ptr, stream))


def prepare_newton_system_dev(LDLT, nnzhF, nnzhc, nnzjF, nnzjc, hF_ptr, hc_ptr, Jx_ptr, Jcx_ptr, delta_ptr, vals_ptr, stream=0):
    """prepare_newton_system! (src/CaNNOLeS.jl:947-981) for a batch on the device (cnl_prepare_newton_system_dev):
    hF / -hc / Jx / Jcx / -delta / 0 into the segments of vals; hF_ptr = 0 leaves H_F alone (Gauss-Newton variants)."""
    _check(lib().cnl_prepare_newton_system_dev(LDLT._h, int(nnzhF), int(nnzhc), int(nnzjF), int(nnzjc), hF_ptr, hc_ptr, Jx_ptr,
                                               Jcx_ptr, delta_ptr, vals_ptr, stream))


def cgls_multipliers_dev(LDLT, vals_ptr, r_ptr, lambda_ptr, Jxtr_ptr=0, atol=None, rtol=None, itmax=0, ones_if_zero=True, iters_ptr=0,
                         stream=0):
    """Least-squares multipliers min ||Jc' lambda - Jx' r|| by CGLS (src/CaNNOLeS.jl:507-518), batched and device-resident
    (cnl_cgls_multipliers_dev).  Default tolerances: sqrt(eps), as Krylov.jl's."""
    eps = float(np.finfo(np.float64).eps)
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    _check(lib().cnl_cgls_multipliers_dev(LDLT._h, vals_ptr, r_ptr, lambda_ptr, Jxtr_ptr, float(atol), float(rtol), int(itmax),
                                          1 if ones_if_zero else 0, iters_ptr, stream))


def cgls_multipliers_jac_dev(LDLT, nnzjF, nnzjc, Jx_ptr, Jcx_ptr, r_ptr, lambda_ptr, Jxtr_ptr=0, atol=None, rtol=None, itmax=0,
                             ones_if_zero=True, iters_ptr=0, stream=0):
    """cgls_multipliers_dev with the Jacobian values read from the model's arrays (cnl_cgls_multipliers_jac_dev)"""
    eps = float(np.finfo(np.float64).eps)
    atol = np.sqrt(eps) if atol is None else atol
    rtol = np.sqrt(eps) if rtol is None else rtol
    _check(lib().cnl_cgls_multipliers_jac_dev(LDLT._h, int(nnzjF), int(nnzjc), Jx_ptr, Jcx_ptr, r_ptr, lambda_ptr, Jxtr_ptr, float(atol),
                                              float(rtol), int(itmax), 1 if ones_if_zero else 0, iters_ptr, stream))


class MultiHIPLDLStruct:
    """One solver object over several devices (cnl_multi_*): the batch is cut into contiguous balanced shards, one handle and
    one host thread per device, no collective.  Same call surface as a batched HIPLDLStruct for the host-pointer calls."""

    def __init__(self, N, rows, cols, nvar, nequ, ncon, batch, devices, options=None):
        self.N, self.nvar, self.nequ, self.ncon, self.batch = int(N), int(nvar), int(nequ), int(ncon), int(batch)
        self.rows, self.cols = _i64(rows), _i64(cols)
        self.nnz = len(self.rows)
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        m = C.c_void_p()
        _check(lib().cnl_multi_create_ex(C.byref(m), self.N, self.nnz, self.rows, self.cols, self.nvar, self.nequ, self.ncon, self.batch,
                                         dv.ctypes.data, len(dv), _optref(options)))
        self._m = m
        n = C.c_int64(0)
        _check(lib().cnl_multi_shards(m, C.byref(n), None, None, None))
        st, ct, dd = np.zeros(n.value, np.int64), np.zeros(n.value, np.int64), np.zeros(n.value, np.int32)
        _check(lib().cnl_multi_shards(m, C.byref(n), st.ctypes.data, ct.ctypes.data, dd.ctypes.data))
        self.shards = [(int(a), int(b), int(c)) for a, b, c in zip(st, ct, dd)]  # (first problem, problems, device)

    def close(self):
        if getattr(self, "_m", None):
            lib().cnl_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def try_to_factorize(self, vals, eig_tol):
        vals = _f64c(vals)
        succ = np.zeros(self.batch, np.int32)
        _check(lib().cnl_multi_factorize(self._m, vals.ctypes.data, float(eig_tol), succ.ctypes.data, None, None))
        return succ.astype(bool)

    def solve_ldl_(self, rhs, d):
        _check(lib().cnl_multi_solve(self._m, _f64c(rhs).ctypes.data, _f64c(d).ctypes.data))
        return True

    def newton_system_(self, d, rhs, vals, rho_old, params):
        B = self.batch
        vals, rhs, d = _f64c(vals), _f64c(rhs), _f64c(d)
        ro = np.ascontiguousarray(np.broadcast_to(np.asarray(rho_old, dtype=np.float64), (B,)))
        params = np.ascontiguousarray(params, dtype=np.float64)
        rho, ro_out = np.zeros(B), np.zeros(B)
        nfact, succ = np.zeros(B, np.int32), np.zeros(B, np.int32)
        _check(lib().cnl_multi_newton_system(self._m, vals.ctypes.data, rhs.ctypes.data, d.ctypes.data, ro.ctypes.data, params.ctypes.data,
                                             rho.ctypes.data, ro_out.ctypes.data, nfact.ctypes.data, succ.ctypes.data))
        return d, succ.astype(bool), rho, ro_out, nfact.astype(np.int64)

    # ---- device-resident twins: one list entry per shard (data_ptr of that shard's array on its device) ----
    @staticmethod
    def _ptrs(ptrs):
        return (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])

    def newton_system_dev(self, vals_ptrs, rhs_ptrs, d_ptrs, rho_old_ptrs, rho_ptrs, nfact_ptrs, success_ptrs, params, streams=None):
        params = np.ascontiguousarray(params, dtype=np.float64)
        st = self._ptrs(streams) if streams is not None else None
        _check(lib().cnl_multi_newton_system_dev(self._m, self._ptrs(vals_ptrs), self._ptrs(rhs_ptrs), self._ptrs(d_ptrs),
                                                 self._ptrs(rho_old_ptrs), self._ptrs(rho_ptrs), self._ptrs(nfact_ptrs),
                                                 self._ptrs(success_ptrs), params.ctypes.data, st))

    def factorize_dev(self, vals_ptrs, eig_tol, success_ptrs, streams=None):
        st = self._ptrs(streams) if streams is not None else None
        _check(lib().cnl_multi_factorize_dev(self._m, self._ptrs(vals_ptrs), float(eig_tol), self._ptrs(success_ptrs), st))

    def solve_dev(self, rhs_ptrs, d_ptrs, streams=None):
        st = self._ptrs(streams) if streams is not None else None
        _check(lib().cnl_multi_solve_dev(self._m, self._ptrs(rhs_ptrs), self._ptrs(d_ptrs), st))

    def synchronize(self, streams=None):
        st = self._ptrs(streams) if streams is not None else None
        _check(lib().cnl_multi_synchronize(self._m, st))

"""The reference-side change that makes `linsolve = :hipldl` selectable (row b of SURVEY 8: the drop-in boundary).

`cannoles.jl_amd/julia/cannoles_hipldl.patch` is the diff a maintainer applies to CaNNOLeS.jl.  These tests run in the
build container only (they need /root/reference to patch a temporary copy of; skipped elsewhere) and check that
  * the patch applies cleanly (`patch --dry-run`, then for real on the copy),
  * on an install WITHOUT HSL — the normal case — `linsolve = :hipldl` reaches the new branch of the if-chain: every
    guard in front of the chain (`/root/reference/src/CaNNOLeS.jl:317-320` rewrites any backend other than
    `:ldlfactorizations` when HSL is not functional) is evaluated for that case,
  * element types other than Float64 and sessions without CaNNOLeSHIP fall back to `:ldlfactorizations`,
  * `:ma57` without HSL still falls back as before,
  * the extension file inside the patch is the one shipped under cannoles.jl_amd/julia/ext/.
No Julia is executed (there is none in the image): the guards are simple boolean expressions over `linsolve`, `T`,
`LIBHSL_isfunctional()` and `hipldl_available(...)`, translated to Python here.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "cannoles.jl_amd", "julia", "cannoles_hipldl.patch")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")) or shutil.which("patch") is None,
                                reason="needs /root/reference (build container) and patch(1)")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    dst = tmp_path_factory.mktemp("ref") / "CaNNOLeS.jl"
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git"))
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=dst, capture_output=True, text=True)
    assert dry.returncode == 0, dry.stdout + dry.stderr
    assert "FAILED" not in dry.stdout and "fuzz" not in dry.stdout, dry.stdout
    real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=dst, capture_output=True, text=True)
    assert real.returncode == 0, real.stdout + real.stderr
    return dst


def _ctor_text(dst):
    src = open(os.path.join(dst, "src", "CaNNOLeS.jl")).read()
    start = src.index("function CaNNOLeSSolver(")
    chain = src.index("LDLT = if linsolve == :ma57", start)
    end = src.index("F = typeof(LDLT)", chain)
    return src[start:chain], src[chain:end]


def _julia_cond_to_python(cond):
    c = cond
    c = re.sub(r":([A-Za-z_][A-Za-z_0-9]*)", r"'\1'", c)          # Symbols
    c = c.replace("&&", " and ").replace("||", " or ")
    c = re.sub(r"!(?!=)", " not ", c)                               # `!x`, `!(…)`; `!=` stays
    c = re.sub(r"hipldl_available\([^)]*\)", "HIPLDL", c)
    c = c.replace("LIBHSL_isfunctional()", "HSL")
    return c


def _selected_backend(dst, linsolve, T, hsl, ext_loaded):
    """Walk the guards in front of the if-chain and then the chain itself, as Julia would."""
    pre, chain = _ctor_text(dst)
    guards = re.findall(r"(?m)^  if (.+)\n((?:    .*\n)+?)  end$", pre)
    for cond, body in guards:
        if "linsolve" not in cond:
            continue
        m = re.search(r"\n?\s*linsolve = :([a-z0-9_]+)", body)
        assert m, f"guard on linsolve that does not reassign it: {cond}"
        env = {"linsolve": linsolve, "T": T, "Float64": "Float64", "HSL": hsl, "HIPLDL": ext_loaded}
        if eval(_julia_cond_to_python(cond), {}, env):
            linsolve = m.group(1)
    branches = re.findall(r"(?:if|elseif) linsolve == :([a-z0-9_]+)\n((?:    .*\n)+)", chain)
    for sym, body in branches:
        if sym == linsolve:
            return sym, body
    return None, ""


def test_patch_applies_and_extension_file_is_the_shipped_one(patched):
    shipped = open(os.path.join(ROOT, "cannoles.jl_amd", "julia", "ext", "CaNNOLeSHIPExt.jl")).read()
    assert open(os.path.join(patched, "ext", "CaNNOLeSHIPExt.jl")).read() == shipped
    proj = open(os.path.join(patched, "Project.toml")).read()
    assert "[weakdeps]" in proj and "[extensions]" in proj and 'CaNNOLeSHIPExt = "CaNNOLeSHIP"' in proj
    uuid = re.search(r'uuid = "([0-9a-f-]+)"', open(os.path.join(ROOT, "cannoles.jl_amd", "julia", "CaNNOLeSHIP", "Project.toml")).read()).group(1)
    assert f'CaNNOLeSHIP = "{uuid}"' in proj
    types = open(os.path.join(patched, "src", "solver_types.jl")).read()
    assert "function linear_solver_struct end" in types and "hipldl_available(rows, cols, vals) = hasmethod(" in types
    main = open(os.path.join(patched, "src", "CaNNOLeS.jl")).read()
    assert "`:hipldl`" in main[main.index("# Keyword arguments"):main.index("max_iter::Int")]


def test_hipldl_reaches_its_branch_without_hsl(patched):
    sym, body = _selected_backend(patched, "hipldl", "Float64", hsl=False, ext_loaded=True)
    assert sym == "hipldl"
    assert "linear_solver_struct(Val(:hipldl), nvar + nequ + ncon, rows, cols, vals, nvar, nequ, ncon)" in body
    assert "vals = get_vals(LDLT)" in body          # solver.vals stays the array the backend holds (CaNNOLeS.jl:324,328)
    # and with HSL present as well
    assert _selected_backend(patched, "hipldl", "Float64", hsl=True, ext_loaded=True)[0] == "hipldl"


def test_unpatched_reference_reroutes_hipldl(tmp_path):
    """The finding this patch answers: in the reference as it stands the guard at src/CaNNOLeS.jl:317-320 turns any backend
    other than :ldlfactorizations into :ldlfactorizations when HSL is missing — a branch added to the chain alone is dead."""
    assert _selected_backend(REF, "hipldl", "Float64", hsl=False, ext_loaded=True)[0] == "ldlfactorizations"


@pytest.mark.parametrize("T,ext", [("Float32", True), ("Float16", True), ("BigFloat", True), ("Float64", False)])
def test_fallback_to_the_cpu_backend(patched, T, ext):
    # /root/reference/test/runtests.jl:102-113 runs Float16/32/BigFloat: those must stay on LDLFactStruct
    assert _selected_backend(patched, "hipldl", T, hsl=False, ext_loaded=ext)[0] == "ldlfactorizations"
    assert _selected_backend(patched, "hipldl", T, hsl=True, ext_loaded=ext)[0] == "ldlfactorizations"


def test_other_backends_unchanged(patched):
    assert _selected_backend(patched, "ma57", "Float64", hsl=False, ext_loaded=True)[0] == "ldlfactorizations"
    assert _selected_backend(patched, "ma57", "Float64", hsl=True, ext_loaded=True)[0] == "ma57"
    assert _selected_backend(patched, "ldlfactorizations", "Float64", hsl=False, ext_loaded=False)[0] == "ldlfactorizations"
    assert _selected_backend(patched, "foo", "Float64", hsl=True, ext_loaded=True)[0] is None      # error("Can't handle …")

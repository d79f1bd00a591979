"""The C ABI from a plain C99 program (tests/c_abi/abi_smoke.c): include/cannoles_hip.h must be self-sufficient for an FFI caller — the
call sequence of the reference's plugin surface (src/solver_types.jl:1-15, src/CaNNOLeS.jl:1008-1052) compiled with
`gcc -std=c99 -pedantic`, no Python, torch or C++ on the caller's side.  CPU: it compiles and links; GPU: it runs fixtures F1 and F3."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "cannoles.jl_amd")


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"),
           "-L", LIBDIR, "-lcannoles_hip", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c99_caller_compiles_and_links(built, tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_c99_caller_runs_the_fixtures(built, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ABI_SMOKE_OK" in r.stdout, (r.stdout, r.stderr)

"""The code-generation hazard behind the staged-execution fault of rounds 4 - 5 (profiles/HISTORY.md 4c): register spills placed in front of the
EXEC restore of a join block (reached with EXEC = 0 when the divergent region in front is skipped: the spills store nothing, the
reloads behind the call return what earlier kernels left in scratch memory).  tools/check_spill_exec.py looks for the pattern in
the ISA hipcc emits for gfx950 (cross-compiled: no GPU needed); the kernels that contain calls are checked in every CPU run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cannoles.jl_amd", "csrc")


def _make_vars():
    """HIPCC, ARCH, CXXFLAGS and the .hip sources of the library, read from csrc/Makefile: the test compiles what ships, with the flags it ships with"""
    import re
    v = {}
    for ln in open(os.path.join(CSRC, "Makefile")):
        m = re.match(r"^(\w+)\s*\??=\s*(.*)$", ln.strip())
        if m:
            v[m.group(1)] = m.group(2)
    return v


_MK = _make_vars()
_HIP_SRCS = [f for f in _MK.get("SRCS", "").split() if f.endswith(".hip")]


def test_every_kernel_source_of_the_makefile_is_checked():
    assert {"kernels.hip", "kernels2.hip", "kernels_aux.hip", "dense.hip", "outer_step.hip", "band.hip"} <= set(_HIP_SRCS)


@pytest.mark.parametrize("src", _HIP_SRCS)
def test_no_spill_in_front_of_an_exec_restore(tmp_path, src):
    hipcc = os.environ.get("HIPCC", _MK.get("HIPCC", "/opt/rocm/bin/hipcc"))
    if not (os.path.isfile(hipcc) and os.access(hipcc, os.X_OK)):
        pytest.skip(f"{hipcc} not present")
    out = tmp_path / (src + ".s")
    flags = [f for f in _MK.get("CXXFLAGS", "-O3 -std=c++17").split() if f != "-fPIC"]
    subprocess.run([hipcc] + flags + [f"--offload-arch={_MK.get('ARCH', 'gfx950')}", "-S", "--cuda-device-only", "-o", str(out), src], cwd=CSRC, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spill_exec.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_the_checker_sees_the_pattern(tmp_path):
    """the shape found in newton2_kernel_t<true, false, false, false, true> of round 5 (abridged)"""
    asm = tmp_path / "bad.s"
    asm.write_text("""_Z3foov:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
	v_mov_b32_e32 v0, s4
.LBB0_2:
	scratch_store_dwordx2 off, v[26:27], off offset:280 ; 8-byte Folded Spill
	s_nop 0
	scratch_store_dwordx4 off, v[194:197], off offset:264 ; 16-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	s_swappc_b64 s[30:31], s[0:1]
	s_endpgm
""")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spill_exec.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 1 and "2 spill store(s)" in r.stdout


def test_the_checker_sees_interleaved_and_agpr_spills(tmp_path):
    """round 6: other instructions between the label and the EXEC restore, an AGPR spill copy, an s_or_saveexec restore"""
    asm = tmp_path / "bad2.s"
    asm.write_text("""_Z3barv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
	v_mov_b32_e32 v0, s4
.LBB0_2:
	s_mov_b32 s7, 0
	v_accvgpr_write_b32 a3, v7 ; 4-byte Folded Spill
	v_add_u32_e32 v1, v2, v3
	scratch_store_dword off, v9, off offset:16 ; 4-byte Folded Spill
	v_writelane_b32 v40, s30, 0 ; 4-byte Folded Spill
	s_or_saveexec_b64 s[2:3], s[0:1]
	s_endpgm
_Z3okv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB1_2
	v_mov_b32_e32 v0, s4
.LBB1_2:
	s_or_b64 exec, exec, s[0:1]
	scratch_store_dword off, v9, off offset:16 ; 4-byte Folded Spill
	s_endpgm
""")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spill_exec.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 1 and "2 spill store(s)" in r.stdout and "1 hazard(s)" in r.stdout

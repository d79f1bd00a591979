"""The code-generation hazard behind the staged-execution fault of rounds 4 - 5: register spills placed in front of the
EXEC restore of a join block (reached with EXEC = 0 when the divergent region in front is skipped: the spills store nothing, the
reloads behind the call return what earlier kernels left in scratch memory).  tools/check_spill_exec.py looks for the pattern in
the ISA hipcc emits for gfx950 (cross-compiled: no GPU needed); the kernels that contain calls are checked in every CPU run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cannoles.jl_amd", "csrc")


@pytest.mark.parametrize("src", ["kernels2.hip", "kernels_aux.hip", "outer_step.hip"])
def test_no_spill_in_front_of_an_exec_restore(tmp_path, src):
    out = tmp_path / (src + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), src], cwd=CSRC, check=True,
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

#!/usr/bin/env python3
"""Looks for the code-generation hazard that caused the staged-execution fault of rounds 4 - 5 (profiles/HISTORY.md section 4c): register SPILL
stores placed in a join block in front of the instruction that restores EXEC.  When the divergent region in front is skipped by
`s_cbranch_execz`, the join block is entered with EXEC = 0 and the stores store nothing; the reloads behind the call they belong to
then return whatever is in scratch memory.

What is flagged (round 6: wider than the one shape found in round 5): every store or AGPR copy the compiler marks `Folded Spill`
(scratch / buffer stores, v_accvgpr_write — all of them honour EXEC; v_writelane SGPR spills do not and are ignored) between the
TARGET LABEL of an `s_cbranch_execz` and the first instruction behind it that writes EXEC (s_or_b64 / s_mov_b64 / s_xor_b64 /
s_andn2_b64 exec, any *_saveexec_*), whatever else is interleaved.  The walk ends at a branch, call, return or the next label
without an EXEC write in between (the label may be a loop head entered with a live mask: not decidable from the listing, not
flagged).  Wave32 code does not occur (gfx950 kernels are wave64; an `exec_lo` write is treated like an EXEC write).
usage: check_spill_exec.py file.s [...]   (ISA from hipcc -S --cuda-device-only); exit code 1 when the pattern is found"""
import re
import sys

EXEC_WRITE = re.compile(r"^(s_\w+\s+exec(_lo|_hi)?\s*,|s_\w*saveexec\w*\s)")
STOP = re.compile(r"^(s_branch|s_cbranch_\w+|s_setpc_b64|s_swappc_b64|s_endpgm|s_call_b64)\b")
SPILL = re.compile(r"Folded Spill")


def scan(path):
    lines = open(path).read().split("\n")
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    func, seen, found = None, set(), []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            func = m.group(1)
        m = re.search(r"s_cbranch_execz\s+(\.LBB\d+_\d+)", ln)
        if not m or m.group(1) not in labels or (func, m.group(1)) in seen:
            continue
        seen.add((func, m.group(1)))
        j = labels[m.group(1)] + 1
        spills = 0
        restored = False
        while j < len(lines):
            t = lines[j].strip()
            j += 1
            if not t or t.startswith(";"):
                continue
            if re.match(r"^\.LBB\d+_\d+:", t) or re.match(r"^_Z\w+:", t) or t.startswith(".Lfunc_end"):
                break
            if EXEC_WRITE.match(t):
                restored = True
                break
            if STOP.match(t):
                break
            if SPILL.search(t) and not t.startswith("v_writelane"):
                spills += 1
        if spills and restored:
            found.append(f"{path}: {(func or '?')[:90]}: {spills} spill store(s) in front of the EXEC restore at {m.group(1)} (line {labels[m.group(1)] + 1})")
    return found


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        bad += scan(p)
    for b in bad:
        print(b)
    print(f"{len(bad)} hazard(s)")
    sys.exit(1 if bad else 0)

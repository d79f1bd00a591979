#!/usr/bin/env python3
"""Looks for the code-generation hazard that caused the staged-execution fault of rounds 4 - 5: register SPILL stores
placed at the head of a join block in front of the instruction that restores EXEC (`s_or_b64 exec, exec, s[..]`).  When the
divergent region in front is skipped by `s_cbranch_execz`, the join block is entered with EXEC = 0 and the stores store nothing;
the reloads behind the call they belong to then return whatever is in scratch memory.
usage: check_spill_exec.py file.s [...]   (ISA from hipcc -S --cuda-device-only); exit code 1 when the pattern is found"""
import re, sys
bad = 0
for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    func = None
    seen = set()
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            func = m.group(1)
        m = re.search(r"s_cbranch_execz\s+(\.LBB\d+_\d+)", ln)
        if not m or m.group(1) not in labels or (func, m.group(1)) in seen:
            continue
        j = labels[m.group(1)] + 1
        spills = 0
        while j < len(lines):
            t = lines[j].strip()
            if t.startswith(";") or not t:
                j += 1; continue
            if re.match(r"^(scratch_store|buffer_store).*Folded Spill", t) or t.startswith("s_nop") or t.startswith("s_waitcnt"):
                spills += "Folded Spill" in t
                j += 1; continue
            break
        if spills and j < len(lines) and re.match(r"^s_or_b64\s+exec,\s*exec", lines[j].strip()):
            seen.add((func, m.group(1)))
            bad += 1
            print(f"{path}: {(func or '?')[:90]}: {spills} spill store(s) in front of the EXEC restore at {m.group(1)} (line {labels[m.group(1)] + 1})")
print(f"{bad} hazard(s)")
sys.exit(1 if bad else 0)

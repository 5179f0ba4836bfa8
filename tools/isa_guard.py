#!/usr/bin/env python
"""ISA guard for the asynchronous SCALAR atomics of the 8-phase GEMM (ec_gemm8.hip: the dynamic tile schedule's tickets).

The kernel issues `s_atomic_add sN, ...` in one inline-asm statement and collects the value behind an `s_waitcnt lgkmcnt(0)` in another one,
many instructions later: between the two the compiler knows nothing of the pending write and would be free to copy sN (reading
the operand, not the ticket).  This check compiles the kernel file to gfx950 assembly and fails if, between an `s_atomic_add sN` and the next
`s_waitcnt` that drains lgkmcnt, any instruction mentions sN.  (Straight-line scan in text order; the kernel's code between the two is
straight-line by construction - issue at the top of a phase, collection at a fixed later phase - so a branch label in between is reported too.)

    python tools/isa_guard.py [file.s]      # without an argument: compiles edgecape_amd/csrc/ec_gemm8.hip
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mentions(line, n):
    """does the instruction line name scalar register n (sN, or a range s[a:b] containing it)?"""
    body = line.split(";")[0]
    if re.search(r"\bs%d\b" % n, body):
        return True
    return any(int(a) <= n <= int(b) for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", body))


def check(path):
    lines = open(path).read().splitlines()
    bad, n_atomics = [], 0
    for i, ln in enumerate(lines):
        m = re.match(r"\s*s_atomic_add\s+s(\d+),", ln)
        if not m:
            continue
        n_atomics += 1
        reg = int(m.group(1))
        for j in range(i + 1, min(i + 4000, len(lines))):
            t = lines[j].strip()
            if not t or t.startswith(";") or t.startswith("."):
                if re.match(r"\.LBB\d+_\d+:", t) or t.startswith("s_endpgm"):
                    pass        # (labels are allowed: the phases' conditionals are if-converted or skip forward)
                continue
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                break
            if mentions(t, reg):
                bad.append((i + 1, j + 1, ln.strip(), t))
                break
        else:
            bad.append((i + 1, None, ln.strip(), "no s_waitcnt lgkmcnt(0) within 4000 lines"))
    return n_atomics, bad


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        sys.path.insert(0, ROOT)
        from edgecape_amd import build     # the compiler, target and flags the shipped library is built with
        path = os.path.join(tempfile.mkdtemp(prefix="isa_guard_"), "ec_gemm8.s")
        cmd = [build.hipcc_path(), f"--offload-arch={build.ARCH}"] + build.CXXFLAGS + ["-S", "--cuda-device-only",
               os.path.join(build.CSRC, "ec_gemm8.hip"), "-o", path]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print("compile failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
            return 2
    n, bad = check(path)
    print(f"{n} scalar atomics, {len(bad)} with their register touched before the collecting wait")
    for b in bad:
        print("  line %s -> %s: %s | %s" % b)
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())

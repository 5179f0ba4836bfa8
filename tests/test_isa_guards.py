"""CPU-side guard on the compiled gfx950 code (no GPU needed: hipcc cross-compiles here)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scalar_atomic_tickets_are_not_touched_before_their_wait():
    """the dynamic tile schedule of the 8-phase GEMM issues `s_atomic_add` and collects the ticket behind an
    `s_waitcnt lgkmcnt(0)` in a SEPARATE inline-asm statement; nothing tells the compiler that the SGPR is pending in between.  The
    shipped binary is checked instead of trusted: tools/isa_guard.py compiles ec_gemm8.hip to assembly and fails if the ticket register
    of any of the scalar atomics is named by an instruction before the wait that collects it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_guard.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 with their register touched" in r.stdout and not r.stdout.startswith("0 scalar"), r.stdout

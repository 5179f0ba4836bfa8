"""Build libedgecape_hip.so (gfx950) in-tree with hipcc.  `python -m edgecape_amd.build [--force]`."""
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libedgecape_hip.so")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]   # every source of the library (tools/isa_guard.py compiles with the same)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sorted(sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h")))


def source_hash():
    """sha256 over every source / header the library is built from (content, not mtimes: the .so travels to the GPU box in a
    snapshot whose timestamps mean nothing)."""
    h = hashlib.sha256()
    for d in deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


STAMP = LIB + ".srchash"


def needs_build():
    """True when the library is missing or was built from different sources than the ones in the tree."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build_locked(verbose=True):
    """build() under an exclusive flock on <LIB>.lock: concurrent processes (the ranks of one job after a kernel edit) queue up,
    the first one compiles, the others find the stamp current when they get the lock and return at once."""
    import fcntl
    with open(LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return build(force=False, verbose=verbose)      # build() re-checks needs_build() itself
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def build_lab(verbose=True):
    """Kernel-lab build (tools/g8_lab.py): the same sources with -DEC_G8_LAB (ablated instantiations of the 8-phase GEMM and their
    timing entry point) into libedgecape_hip_lab.so.  Never loaded by the product path."""
    return build(force=True, verbose=verbose, lab=True)


def build(force=False, verbose=True, lab=False):
    """Compile every .hip source for gfx950 and link the C-ABI shared library. Returns its path."""
    if not force and not lab and not needs_build():
        return LIB
    cc = hipcc_path()
    objdir = os.path.join(HERE, "build", "lab") if lab else os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(objdir, f"{os.path.basename(s)[:-4]}.{os.getpid()}.o")
        objs.append(o)
        cmd = [cc, f"--offload-arch={ARCH}"] + CXXFLAGS + ["-c", s, "-o", o]
        if lab:
            cmd.insert(1, "-DEC_G8_LAB")
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out)
    out = LIB.replace(".so", "_lab.so") if lab else LIB
    tmp = f"{out}.{os.getpid()}.tmp"
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for o in objs:
        try:
            os.remove(o)
        except OSError:
            pass
    if r.returncode != 0:
        raise RuntimeError("link failed: " + " ".join(cmd) + "\n" + r.stdout)
    os.replace(tmp, out)
    if lab:
        return out
    with open(STAMP, "w") as f:
        f.write(source_hash() + "\n")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    if "--lab" in sys.argv:
        print(build_lab())
    elif "--force" in sys.argv:
        build(force=True)
    else:
        build_locked()

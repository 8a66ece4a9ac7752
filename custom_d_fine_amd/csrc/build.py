"""Builds custom_d_fine_amd/csrc/libdfine_hip.so for gfx950 with hipcc (cross-compiles without a
GPU).  `python -m custom_d_fine_amd.csrc.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libdfine_hip.so")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-pass-failed"]
# per-file flags: the gather's f32 atomics must lower to global_atomic_add_f32; the assignment
# kernel's float64 arithmetic must not be contracted into FMAs (bit-exact vs SciPy)
SOURCES = {
    "runtime.cpp": [],
    "msda.hip": ["-munsafe-fp-atomics"],
    "matcher.hip": ["-ffp-contract=off"],
    "dwconv.hip": ["-munsafe-fp-atomics"],
    "bnact.hip": ["-munsafe-fp-atomics"],
    "losses.hip": ["-munsafe-fp-atomics"],
    "optim.hip": ["-munsafe-fp-atomics"],
    "conv.hip": ["-munsafe-fp-atomics"],
    "stem.hip": ["-munsafe-fp-atomics"],
    "postproc.hip": ["-ffp-contract=off"],       # box arithmetic bit-identical to the reference's fp32 ops
    "data.hip": ["-ffp-contract=off"],           # the same for the augmentation box transform
    "cdn.hip": ["-ffp-contract=off"],            # the denoising group's box noise: every fp32 operation individually rounded, like ATen's
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    extra = [f for f in sorted(os.listdir(HERE)) if f.endswith((".hip", ".cpp")) and f not in SOURCES]
    return {**SOURCES, **{f: [] for f in extra}}


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".hip", ".cpp", ".h"))]
    deps.append(os.path.join(HERE, "..", "..", "include", "dfine_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    # one builder at a time: the ranks of a multi-GPU launch import the package together, and a stale library must not be
    # rebuilt by all of them into the same file (the others wait here, then find it up to date)
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    cc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src, flags in sources().items():
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        cmd = [cc] + COMMON + flags + (["-x", "hip"] if src.endswith(".cpp") else []) + \
              ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)          # readers never see a half-written library
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

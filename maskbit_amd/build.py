"""Build libmaskbit_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmaskbit_hip.so")
SOURCES = ["engine.hip", "gemm.hip", "gemm_ht.hip", "norm_embed.hip", "attention.hip", "sampling.hip", "decoder.hip"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    inc = os.path.join(HERE, "..", "include")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, "maskbit_hip.h"), os.path.join(inc, "maskbit_hip_diag.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False, defs: str = "", out: str = "") -> str:
    """The product library (no arguments), or -- `defs` = extra -D flags, `out` = another output path -- an A/B build for the tools (tools/gemm_ab.py,
    tools/forward_ab.py).  The timing-only switches (MB_NO_GELU, MB_MINI_NO_*, ...) produce wrong results by design, so a build with extra flags NEVER
    lands on the product path: it needs an `out` of its own (the environment variable MASKBIT_AMD_BUILD_DEFS of earlier rounds is gone)."""
    if defs and not out:
        raise ValueError("an A/B build (extra -D flags) needs an output path of its own: it must not replace the product library")
    lib = os.path.abspath(out) if out else LIB
    if not force and not defs and not needs_build(lib):
        return lib
    objs = []
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *defs.split()]
    procs = []
    bdir = os.path.join(HERE, "build") if not out else lib + ".objs"
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc(), *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

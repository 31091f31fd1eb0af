"""Builds r3d_amd/_build/libr3d_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m r3d_amd.build [--force]

One object per .hip source (rebuilt when the source or a header is newer), then one shared library.
The library stays in-tree (git-ignored) so it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libr3d_hip.so")
ARCH = "gfx950"
SOURCES = ["abi.hip", "gemm_f32.hip", "rowops.hip", "fusion.hip", "attention.hip", "decoder.hip", "losses.hip", "optim.hip", "embed.hip", "tail.hip", "bnfuse.hip",
           "erank.hip", "posenc.hip", "gemm_bf3.hip", "gemm_ln.hip", "fuser_chain.hip", "decoder_chain.hip", "fuser3.hip"]
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function", "-Wno-pass-failed"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "r3d_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    extra = os.environ.get("R3D_EXTRA_DEFS", "").split()      # profiling builds only (tools/*probe*): e.g. -DR3D_FC_PROBE=3
    if extra:
        force = True
    hdr = _deps_mtime()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return s

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for s in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[r3d_amd.build] compiled {os.path.basename(s)}", flush=True)
    objs = [os.path.join(OUT_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or not os.path.exists(LIB) or force:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[r3d_amd.build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

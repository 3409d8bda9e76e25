#!/usr/bin/env python3
"""Build libtxe.so (gfx950) in-tree:  python taxoexpan_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU.  Objects are compiled in parallel and cached by source mtime; the shared
library lands next to the sources so that it travels with the repo snapshot to the GPU box.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["txe_gemm_nt.hip", "txe_gemm_nn.hip", "txe_gemm_tn.hip", "txe_gemm_split.hip", "txe_gat.hip", "txe_gcn.hip", "txe_project.hip", "txe_dxpos.hip", "txe_readout.hip", "txe_match.hip",
           "txe_graph.hip", "txe_rank.hip", "txe_profile.hip", "txe_egonet.hip", "txe_optim.hip", "txe_loss.hip"]
HEADERS = ["txe_common.h", "txe_gemm.h", "txe_gather.h", "txe_colsum.h", "txe_dxpos.h", "txe_gemm_tnlds.h", "txe_skinny.h", "txe_gemm_split.h"]
LIB = os.path.join(HERE, "libtxe.so")
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("TXE_HIPCC_FLAGS", "").split()


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    deps = [os.path.join(HERE, src)] + [os.path.join(HERE, h) for h in HEADERS]
    if not any(_newer(d, obj) for d in deps):
        return obj, ""
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    if force:
        for s in srcs:
            o = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    warn = "".join(w for _, w in results)
    if verbose and warn.strip():
        print(warn, file=sys.stderr)
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

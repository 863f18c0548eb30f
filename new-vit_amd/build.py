#!/usr/bin/env python3
"""Build libmst_hip.so (gfx950) in-tree with hipcc: one object per csrc/*.hip, compiled in parallel.

    python new-vit_amd/build.py [--force] [--jobs N]

The library lands at new-vit_amd/mst/hip/libmst_hip.so (git-ignored, travels with gpurun snapshots).
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "build"
LIB = Path(os.environ.get("MST_BUILD_LIB", HERE / "mst" / "hip" / "libmst_hip.so"))   # ablation builds: other name
# -amdgpu-mfma-vgpr-form: keep MFMA C/D in VGPRs (gfx950's register file is unified).  hipcc's default put the
# attention scores in AGPRs and spent 127 v_accvgpr_read/write per KV tile moving them to the softmax and back.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
PER_FILE_FLAGS = {"k_block16s.hip": ["-fno-slp-vectorize"],   # the SLP pass re-places the hand-placed GELU / LayerNorm arithmetic (and packs it)
                  "k_attn16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "k_attn32.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmst_hip.so)")


def newer(src: Path, dst: Path, deps) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [src, *deps])


def build(force: bool = False, jobs: int = 8, verbose: bool = True) -> Path:
    global OBJ
    cc = hipcc()
    extra = os.environ.get("MST_EXTRA_FLAGS", "").split()
    if extra:                     # ablation / diagnostic builds get their own object directory
        OBJ = HERE / ("build_" + "_".join(f.strip("-").replace("=", "") for f in extra))
        FLAGS.extend(f for f in extra if f not in FLAGS)
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [HERE.parent / "include" / "mst_hip.h", Path(__file__)]
    srcs = sorted(CSRC.glob("*.hip"))
    todo = [s for s in srcs if force or newer(s, OBJ / (s.stem + ".o"), headers)]

    def compile_one(src: Path):
        cmd = [cc, *FLAGS, *PER_FILE_FLAGS.get(src.name, []), "-c", str(src), "-o", str(OBJ / (src.stem + ".o"))]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r

    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        for src, r in ex.map(compile_one, todo):
            if verbose and (r.stderr.strip() or r.returncode):
                sys.stderr.write(r.stderr)
            if r.returncode:
                raise RuntimeError(f"hipcc failed on {src.name}")
            if verbose:
                print(f"[build] {src.name}")
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or todo or not LIB.exists():
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    build(a.force, a.jobs)

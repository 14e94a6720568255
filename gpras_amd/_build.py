"""Build libgprx.so (HIP, gfx950 only) in-tree with hipcc.  No torch, no cmake."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libgprx.so"
SOURCES = ["gprx.hip"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgprx.so cannot be built")
    return exe


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    built = LIB_PATH.stat().st_mtime
    deps = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hip")) + [PKG_DIR.parent / "include" / "gprx.h"]  # every file gprx.hip may include
    return any(p.exists() and p.stat().st_mtime > built for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into gpras_amd/libgprx.so."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [
        _hipcc(),
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-shared",
        "-Wno-unused-value",
        "-o",
        str(LIB_PATH),
    ] + [str(CSRC / s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))

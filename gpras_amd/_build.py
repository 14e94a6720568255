"""Build libgprx.so (HIP, gfx950 only) in-tree with hipcc.  No torch, no cmake.

The library is several translation units (one object each, compiled in parallel, then linked): `gprx.hip` (the C ABI and every
launch sequence) and the fused sparse evaluation, whose pass kernels are compiled once per kernel id (`-DSF_KID=k`).  An object is
rebuilt when its source, any header its last compile read (hipcc `-MD` dependency file) or the flags changed.
"""

from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = PKG_DIR / "csrc" / "_obj"
LIB_PATH = PKG_DIR / "libgprx.so"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# development builds (tools/chain_stamps.sh: -DGPRX_CHAIN_STAMPS ...): extra defines for every unit, part of the staleness check
FLAGS += os.environ.get("GPRX_EXTRA_FLAGS", "").split()
# (object name, source, extra defines)
UNITS = (
    [("gprx", "gprx.hip", []), ("sf_cell", "sf_cell.hip", []), ("sf_adam", "sf_adam.hip", [])]
    + [(f"sf_pass1_k{k}", "sf_pass1.hip", [f"-DSF_KID={k}"]) for k in range(5)]
    + [(f"sf_pass2_k{k}", "sf_pass2.hip", [f"-DSF_KID={k}"]) for k in range(5)]
)
SOURCES = sorted({u[1] for u in UNITS})


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libgprx.so cannot be built")
    return exe


def _deps_of(obj: Path) -> list[Path] | None:
    """Files the object's last compile read (make-style .d file), or None when unknown."""
    dep = obj.with_suffix(".d")
    if not dep.exists():
        return None
    text = dep.read_text().replace("\\\n", " ")
    _, _, rhs = text.partition(":")
    return [Path(tok) for tok in rhs.split() if tok]


def _unit_stale(name: str, src: str, defines: list[str]) -> bool:
    obj = OBJ_DIR / f"{name}.o"
    if not obj.exists():
        return True
    flags_file = OBJ_DIR / f"{name}.flags"
    if not flags_file.exists() or flags_file.read_text() != " ".join(FLAGS + defines):
        return True
    built = obj.stat().st_mtime
    deps = _deps_of(obj)
    if deps is None:
        deps = sorted(CSRC.glob("*.h")) + [CSRC / src, PKG_DIR.parent / "include" / "gprx.h"]
    return any((not p.exists()) or p.stat().st_mtime > built for p in deps if "/opt/rocm" not in str(p) and not str(p).startswith("/usr/"))


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    if not OBJ_DIR.exists():
        # a snapshot without the object directory (the GPU box receives the built library; objects are scratch): judge by the sources
        built = LIB_PATH.stat().st_mtime
        deps = sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.hip")) + [PKG_DIR.parent / "include" / "gprx.h"]
        return any(p.exists() and p.stat().st_mtime > built for p in deps)
    if any(_unit_stale(*u) for u in UNITS):
        return True
    built = LIB_PATH.stat().st_mtime
    return any((OBJ_DIR / f"{u[0]}.o").stat().st_mtime > built for u in UNITS)


def _compile(unit, verbose: bool) -> None:
    name, src, defines = unit
    obj = OBJ_DIR / f"{name}.o"
    cmd = [_hipcc()] + FLAGS + defines + ["-MD", "-MF", str(obj.with_suffix(".d")), "-c", str(CSRC / src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src} {' '.join(defines)}:\n{res.stdout}\n{res.stderr}")
    (OBJ_DIR / f"{name}.flags").write_text(" ".join(FLAGS + defines))


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> Path:
    """Compile every HIP source for gfx950 into gpras_amd/libgprx.so."""
    if not force and not needs_build():
        return LIB_PATH
    if os.environ.get("GPRX_NO_BUILD"):
        # set by the profiling scripts: under rocprofv3 --pmc the preloaded library has initialised the GPU and a fork + exec of hipcc
        # from this process is the hop this pool refuses
        raise RuntimeError("libgprx.so is stale but GPRX_NO_BUILD is set: build before the profiler line (python -m gpras_amd._build)")
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    todo = [u for u in UNITS if force or _unit_stale(*u)]
    jobs = jobs or max(1, min(len(todo), os.cpu_count() or 1, 8))
    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            list(pool.map(lambda u: _compile(u, verbose), todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(OBJ_DIR / f"{u[0]}.o") for u in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    import sys

    print(build(force="--force" in sys.argv or len(sys.argv) == 1, verbose=True))

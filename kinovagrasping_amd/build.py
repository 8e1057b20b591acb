"""Build the gfx950 shared library in-tree (kinovagrasping_amd/libkinova_sim.so)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libkinova_sim.so"
SOURCES = ["ks_api.hip", "ks_rollout.hip", "ks_mlp.hip", "ks_xchg.hip"]
HEADERS = ["ks_math.h", "ks_model.h", "ks_model_host.h", "ks_core.h", "ks_obs.h", "ks_env.h", "../../include/kinova_sim.h", "../../include/kinova_rollout.h"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this package has no CPU build)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> libkinova_sim.so.  Cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", str(LIB)] + [str(CSRC / s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd, cwd=str(CSRC))
    return LIB

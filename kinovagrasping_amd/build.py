"""Build the gfx950 shared libraries in-tree: kinovagrasping_amd/libkinova_sim.so (the fixed nine-geom topology: every single-geom
object, the whole rollout / learner C ABI) and libkinova_sim_mg.so (the simulator C ABI of include/kinova_sim.h compiled with the
multi-geom capacities, -DKS_MULTI_GEOM: the reference's Bottle / TBottle / Bowl / RBowl objects, csrc/ks_model.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libkinova_sim.so"
LIB_MG = PKG / "libkinova_sim_mg.so"
SOURCES = ["ks_api.hip", "ks_rollout.hip", "ks_mlp.hip", "ks_xchg.hip"]
HEADERS = ["ks_math.h", "ks_model.h", "ks_model_host.h", "ks_core.h", "ks_obs.h", "ks_env.h", "../../include/kinova_sim.h", "../../include/kinova_rollout.h"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this package has no CPU build)")


def needs_build(lib: Path = LIB) -> bool:
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, multi_geom: bool | None = None) -> Path:
    """hipcc --offload-arch=gfx950 -> libkinova_sim.so and libkinova_sim_mg.so (multi_geom: None = both, False / True = that one).
    Cross-compiles without a GPU.  Returns the standard library's path (the multi-geom one's with multi_geom=True)."""
    for mg in ([False, True] if multi_geom is None else [bool(multi_geom)]):
        lib = LIB_MG if mg else LIB
        if not force and not needs_build(lib):
            continue
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + (["-DKS_MULTI_GEOM"] if mg else []) + \
              ["-o", str(lib)] + [str(CSRC / s) for s in SOURCES]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd, cwd=str(CSRC))
    return LIB_MG if multi_geom else LIB

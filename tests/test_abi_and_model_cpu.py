"""CPU checks: the C-ABI library loads and exports every symbol include/kinova_sim.h declares (no compute
without a GPU: ks_create must fail loudly), and the model compiler reproduces the known answers measured
from the reference's STL/MJCF assets (SURVEY.md 8c "builder-generated pins")."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from kinovagrasping_amd import build as kb
from kinovagrasping_amd import model_compiler as mc
from kinovagrasping_amd import sim as ks

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    kb.build()
    return ks.load_library()


def test_every_declared_symbol_is_exported(lib):
    header = (ROOT / "include" / "kinova_sim.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ks_[a-z_0-9]+)\s*\(", header))
    assert {"ks_create", "ks_step", "ks_reset", "ks_load_model", "ks_get_state", "ks_destroy"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/kinova_sim.h but not exported"
    assert declared == set(ks.EXPORTS)
    # the rollout / replay kernels (include/kinova_rollout.h)
    header = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "kinova_rollout.h").read_text(), flags=re.S)
    declared = set(re.findall(r"\b(kr_[a-z_0-9]+)\s*\(", header))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/kinova_rollout.h but not exported"
    assert declared == set(ks.ROLLOUT_EXPORTS)


def test_config_struct_layout_and_defaults(lib):
    cfg = ks.KsConfig()
    lib.ks_default_config(C.byref(cfg))
    assert C.sizeof(cfg) == 48
    assert (cfg.frame_skip, cfg.horizon, cfg.precision) == (15, 30, 32)      # ENV:51, main_DDPGfD.py:384


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a machine without a GPU")
def test_no_cpu_path_fails_loudly(lib):
    cfg = ks.KsConfig()
    lib.ks_default_config(C.byref(cfg))
    ctx = C.c_void_p()
    assert lib.ks_create(C.byref(cfg), 0, C.byref(ctx)) == -2                # KS_ERR_NO_DEVICE
    assert b"no CPU path" in lib.ks_last_error(None)
    with pytest.raises(RuntimeError):
        ks.KinovaSim(64, "CubeS")
    cfg.n_envs = 0
    assert lib.ks_create(C.byref(cfg), 0, C.byref(ctx)) == -1                # KS_ERR_INVALID


def test_model_compiler_known_answers(assets_dir):
    M = mc.read_blob(assets_dir / "CubeS.ksm")
    info = M["mesh_info"]                                                    # volume, hull verts, planes, tris, simplices
    # MuJoCo 1.50's LEGACY mesh inertia (pyramids with absolute volumes: the non-convex hand meshes are over-counted; the exact
    # signed volumes measured in SURVEY 8c are 5.5307e-4, 2.4029e-5, 1.2314e-5 - the convex CubeS is 1.07651e-4 either way)
    np.testing.assert_allclose(info[:, 0], [5.71039e-4, 2.52886e-5, 1.30008e-5, 1.07651e-4], rtol=2e-5)
    assert info[:, 3].astype(int).tolist() == [27908, 2710, 1942, 6344]
    assert abs(info[0, 1] - 755) <= 2 and abs(info[1, 1] - 299) <= 8 and abs(info[2, 1] - 356) <= 8 and info[3, 1] == 24    # (hulls of the float32 geom-frame points, as MuJoCo stores them)
    # mesh geom centres = the legacy centres of mass, equal to MuJoCo 1.50's recorded geom_xpos to 1e-10 (tests/test_mujoco_recorded.py;
    # exact centroids: palm [1e-5, -4.03e-3, -5.969e-2], proximal [0.02041, -0.00818, 0], distal [0.01342, -0.00475, 0])
    np.testing.assert_allclose(M["geom_pos"][1], [2.726e-5, -4.1435e-3, -6.09450e-2], atol=2e-7)
    np.testing.assert_allclose(M["geom_pos"][2], [0.0205051, -0.0079019, 0], atol=5e-7)
    np.testing.assert_allclose(M["geom_pos"][3], [0.0128282, -0.0045348, 0], atol=5e-7)
    np.testing.assert_allclose(sorted(M["body_inertia"][9]), [1.870e-5, 8.568e-5, 8.568e-5], rtol=1e-3)
    # hull adjacency (CSR): every vertex has >= 3 neighbours, symmetric
    off, adj = M["mesh3_adj_off"], M["mesh3_adj"]
    assert len(off) == 25 and off[-1] == len(adj) and (np.diff(off) >= 3).all()
    assert all(i in adj[off[j]:off[j + 1]] for i in range(24) for j in adj[off[i]:off[i + 1]])
    assert M["pairs"].shape == (30, 5) and (M["pairs"][0, :2] == [0, 8]).all() and M["pairs"][0, 2] == 0.3
    # explicit <contact><pair>s carry the PAIR default margin 0 (XML:158-166 give none), the 22 dynamic pairs the geoms' 0.001 (XML:40)
    assert (M["pairs"][:8, 4] == 0).all() and (M["pairs"][8:, 4] == 0.001).all() and (M["pairs"][:8, 1] == 8).all()
    assert M["body_mass"].tolist() == [0, 0, 0.727, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.1]
    # palm geom frame: link frame tilted ~5.7 deg about x (legacy inertia; 5.4 deg with the exact one, SURVEY hard part 6)
    R = mc.quat_to_mat(M["geom_quat"][1])
    assert abs(np.degrees(np.arccos(R[1, 1])) - 5.67) < 0.1 and R[0, 0] > 0.9999
    # obs[33:36] = [s0, s1, 2*s2] of the object's AABB half extents (ENV:529, 706-746)
    np.testing.assert_allclose(M["obj_size_obs"], [0.0167781, 0.0167781, 0.095875], rtol=1e-5)


def test_euler_truncation_and_quaternion():
    np.testing.assert_array_equal(mc.truncated_euler([-0.03600013, -1.57, 4.28785117e-05, 1.2345678]), [-0.03, -1.57, 0.0, 1.234])
    # intrinsic xyz: (-1.57, 0, -1.57) maps link y -> world x, link z -> world y, link x -> world z
    R = mc.quat_to_mat(mc.euler_to_quat([-np.pi / 2, 0, -np.pi / 2]))
    np.testing.assert_allclose(R @ [0, 1, 0], [1, 0, 0], atol=1e-12)
    np.testing.assert_allclose(R @ [0, 0, 1], [0, 1, 0], atol=1e-12)
    np.testing.assert_allclose(R @ [1, 0, 0], [0, 0, 1], atol=1e-12)


def test_all_shapes_have_blobs(assets_dir):
    from kinovagrasping_amd import scenarios
    for shape in scenarios.SHAPES + scenarios.MEDIUM_SHAPES:
        M = mc.read_blob(assets_dir / f"{shape}.ksm")
        assert M["mesh3_vert"].shape[1] == 3 and abs(M["body_mass"][9] - 0.1) < 1e-12
        assert scenarios.start_coord_table(shape).shape[0] == 4499
        assert all(scenarios.has_start_table(shape, o) for o in ("normal", "rotated", "top"))
    # every shape key the reference's experiment stages can ask for has a compiled model (main_DDPGfD.py:1270-1281)
    from kinovagrasping_amd import curriculum
    keys = {s + z for s in curriculum.TRAIN_SHAPES for z in curriculum.TRAIN_SIZES} | {s + z for s in curriculum.TEST_SHAPES for z in curriculum.TEST_SIZES + ["S"]}
    known = set(scenarios.SHAPES + scenarios.MEDIUM_SHAPES + scenarios.EXTRA_SHAPES + scenarios.MULTI_GEOM_SHAPES)      # = the 42 keys of KinovaGripper_Env.all_objects (ENV:150-208)
    assert len(known) == 42
    assert keys <= known, keys - known
    assert all((assets_dir / f"{k}.ksm").exists() for k in known)


def test_example_script_resolves_every_global_name():
    """examples/train_ddpgfd.py: every global its functions load exists after import (a NameError used to hide in the
    expert-collection loop); importing it touches no GPU."""
    import builtins
    import dis
    import importlib.util
    from pathlib import Path
    path = Path(__file__).resolve().parents[1] / "examples" / "train_ddpgfd.py"
    spec = importlib.util.spec_from_file_location("example_train_ddpgfd", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def globals_loaded(code):
        for ins in dis.get_instructions(code):
            if ins.opname == "LOAD_GLOBAL":
                yield ins.argval
        for c in code.co_consts:
            if hasattr(c, "co_code"):
                yield from globals_loaded(c)

    missing = sorted({n for f in vars(mod).values() if callable(f) and getattr(f, "__module__", None) == mod.__name__ and hasattr(f, "__code__")
                      for n in globals_loaded(f.__code__) if not hasattr(mod, n) and not hasattr(builtins, n)})
    assert not missing, missing


def test_c_program_compiles_against_the_headers_and_links_every_symbol(lib, tmp_path):
    """A plain C translation unit (gcc -std=c99 -Wall -Werror, no torch, no C++) that includes include/*.h, checks the
    struct layout a caller relies on and references every declared entry point, linked against libkinova_sim.so and
    run: what a maintainer binding the library from C / cgo / JNI would hit first."""
    import subprocess
    names = []
    for h in ("kinova_sim.h", "kinova_rollout.h"):
        text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / h).read_text(), flags=re.S)
        names += sorted(set(re.findall(r"\b(k[sr]_[a-z_0-9]+)\s*\(", text)))
    src = tmp_path / "abi_check.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "kinova_sim.h"\n#include "kinova_rollout.h"\n'
                   "typedef char cfg_is_48_bytes[sizeof(ks_config) == 48 ? 1 : -1];\n"
                   "typedef char cfg_pair_memory_at_36[offsetof(ks_config, pair_memory) == 36 ? 1 : -1];\n"
                   "typedef void (*fn)(void);\nstatic fn table[] = {" + ", ".join(f"(fn){n}" for n in names) + "};\n"
                   "int main(void) {\n  ks_config c; ks_ctx *ctx = NULL; int rc;\n  ks_default_config(&c);\n"
                   "  if (c.frame_skip != 15 || c.horizon != 30 || c.precision != 32 || c.pair_memory != 1) return 2;\n"
                   "  rc = ks_create(&c, 0, &ctx);\n"
                   "  if (rc == KS_OK) ks_destroy(ctx); else if (rc != KS_ERR_NO_DEVICE || !ks_last_error(NULL)[0]) return 3;\n"
                   '  printf("%d symbols, ks_version %d, ks_create rc %d\\n", (int)(sizeof table / sizeof table[0]), ks_version(), rc);\n  return 0;\n}\n')
    exe = tmp_path / "abi_check"
    libdir = Path(ks._build.LIB).parent
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
                           "-L", str(libdir), "-lkinova_sim", "-Wl,-rpath," + str(libdir), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert f"{len(names)} symbols" in out.stdout and len(names) >= 30


def _register_footprints(src: str, tmp_path) -> dict:
    """{mangled name: (vgprs, agprs)} of every function in a HIP source, from the compiler's device assembly"""
    import subprocess
    out = tmp_path / (src + ".s")
    subprocess.check_call([kb.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out), src],
                          cwd=str(kb.CSRC), stderr=subprocess.DEVNULL)
    regs, name, v = {}, None, 0
    for line in out.read_text().splitlines():
        if line.startswith("\t.size\t_Z"):
            name = line.split()[1].rstrip(",")
        elif line.startswith("; NumVgprs:"):
            v = int(line.split()[2])
        elif line.startswith("; NumAgprs:") and name:
            regs[name] = (v, int(line.split()[2]))
    return regs


def test_stepping_kernel_leaves_registers_for_the_learner(tmp_path):
    """The learner's LDS-free kernels (ks_mlp.hip: *_wave, k_mlp3_split, k_mlp3_bwd_split) run on the registers k_env_step leaves
    free: one of their waves beside the stepping kernel's one wave per SIMD, 512 registers per lane in all (allocated in
    blocks of 8).  A stepping kernel that does not leave room silently serialises the two (measured: 1.15 -> 1.73 ms per
    env-step when a change took it to 418), so the footprint is a contract."""
    step = {k: v for k, v in _register_footprints("ks_api.hip", tmp_path).items() if "k_env_step_f32" in k}      # (the fp32 product's kernel: its budget is set in the source, KS_STEP_NUM_VGPR)
    assert len(step) == 1, step
    (v, a), = step.values()
    up8 = lambda x: (x + 7) // 8 * 8
    learner = {k: v for k, v in _register_footprints("ks_mlp.hip", tmp_path).items() if "_wave" in k or "_split" in k}
    assert learner
    widest = max(up8(lv + la) for lv, la in learner.values())
    assert up8(v + a) + widest <= 512, (v, a, widest, learner)


def test_free_running_rollout_kernel_leaves_registers_for_the_learners_one_wave_kernels(tmp_path):
    """k_rollout (ks_rollout) is resident for a whole K-step launch: the learner can only run beside it if one of its waves fits
    on a SIMD next to one of the rollout's.  pipeline.AsyncTrainer captures the learner with the ONE-WAVE forward / backward /
    weight-gradient kernels (KS_MLP_SPLIT=0) for that reason; the 4-wave split variants (160 registers) do not fit.  The
    allocator is fragile here (a `volatile` on two LDS words cost 28 registers): the footprint is a contract."""
    up8 = lambda x: (x + 7) // 8 * 8
    roll = {k: v for k, v in _register_footprints("ks_api.hip", tmp_path).items() if "k_rolloutILi16ELi16" in k}
    assert len(roll) == 1, roll
    (v, a), = roll.values()
    one_wave = {k: r for k, r in _register_footprints("ks_mlp.hip", tmp_path).items()
                if ("k_mlp3_waveILi16ELi16" in k or "k_mlp3_bwd_waveILi16ELi16" in k or "k_wgrad_wave" in k)}
    assert len(one_wave) >= 3, one_wave
    widest = max(up8(lv + la) for lv, la in one_wave.values())
    assert up8(v) + up8(a) + widest <= 512, (v, a, widest)


def test_learner_and_rollout_glue_kernels_use_no_lds():
    """Everything the learner's stream launches must be able to start beside a stepping kernel that holds ALL of a CU's LDS (a mixed-object
    context's larger hull tables leave less than 1 KB): one glue kernel with a 1 KB block reduction stalled the whole update until
    persistent workgroups exited, and the episodes published meanwhile were dropped (round 4).  ks_rollout.hip (rollout / replay /
    learner glue) and ks_xchg.hip (gradient exchange) declare no LDS at all; ks_mlp.hip only in the one kernel that is allowed to
    (k_mlp3, the lock-step actor launch between two stepping launches)."""
    import re
    from pathlib import Path
    csrc = Path(__file__).resolve().parents[1] / "kinovagrasping_amd" / "csrc"
    for f in ("ks_rollout.hip", "ks_xchg.hip"):
        src = re.sub(r"//.*", "", (csrc / f).read_text())
        assert "__shared__" not in src and "hipcub" not in src, f
    mlp = re.sub(r"//.*", "", (csrc / "ks_mlp.hip").read_text())
    assert len(re.findall(r"__shared__", mlp)) == 3          # H1, H2, P of k_mlp3


def test_orientation_noise_is_zero_mean_truncated_and_seeded():
    """reset(with_noise="zero-mean") (SURVEY note N5's extension): class Euler constants + zero-mean N(0, 0.087), then the reference's
    5-character truncation (ENV:870-874); without an rng exactly the class quaternion."""
    import numpy as np
    from kinovagrasping_amd import scenarios
    from kinovagrasping_amd.model_compiler import euler_to_quat
    for o, base in scenarios.ORIENTATION_EULER.items():
        assert np.array_equal(euler_to_quat(scenarios.hand_euler_for(o)), scenarios.hand_quat_for(o))
        rng = np.random.RandomState(11)
        e = np.stack([scenarios.hand_euler_for(o, rng) for _ in range(4000)])
        assert np.abs(e.mean(0) - np.asarray(base)).max() < 0.01            # zero mean (the truncation is toward zero: < 0.005 bias)
        assert np.abs(e.std(0) - 0.087).max() < 0.006
        assert all(len(repr(float(v))) <= 5 or abs(v) < 1e-4 or float(repr(float(v))[:5]) == v for v in e[:50].ravel())
        rng2 = np.random.RandomState(11)
        assert np.array_equal(e[0], scenarios.hand_euler_for(o, rng2))


def test_reference_with_noise_tables_are_shipped_as_they_are():
    """obj_hand_coords/with_noise/train_coords/<class>/<shape>.txt (the reference's DEFAULT start states, kinova_gripper_env.py:1310,
    1019-1021, 1254-1255): object x, y, z + hand Euler triple per row, first line consumed by the delimiter sniffer (:1012).  Known rows of the
    files, and the statistics SURVEY note N5 measured on them: the Euler columns are noise of std 0.087 around a -0.087 BIAS, about the
    identity in `normal/` and about (-1.57, 0, -1.57) in `top/` - swapped relative to the env's own class constants (ENV:1267-1273)."""
    from kinovagrasping_amd import scenarios
    t = scenarios.noisy_start_table("CubeS", "normal")
    assert t.shape == (4499, 6)
    assert np.allclose(t[0], [0.009388, 0.043207, 0.0654, 0.13371021, -0.00540045, -0.21338375], atol=1e-7)      # line 2 of the file
    assert np.allclose(t[1], [-0.002896, 0.019442, 0.0654, -0.10531564, -0.12903301, -0.12173909], atol=1e-7)
    top = scenarios.noisy_start_table("CubeS", "top")
    assert np.allclose(top[0], [0.033995, 0.024185, 0.05, -1.71440852, -0.08557746, -1.65729666], atol=1e-6)
    for tab, centre in ((t, [0.0, 0.0, 0.0]), (top, [-1.57, 0.0, -1.57]), (scenarios.noisy_start_table("CubeS", "rotated"), [-1.2, 0.0, 0.0])):
        e = tab[:, 3:6]
        assert np.allclose(e.mean(0), np.array(centre) - 0.087, atol=0.006) and np.allclose(e.std(0), 0.087, atol=0.003)
    assert len([s for s in scenarios.SHAPES for o in ("normal", "rotated", "top") if scenarios.noisy_start_table(s, o) is not None]) == 42
    assert scenarios.noisy_start_table("BowlS", "normal") is None
    # the truncation the reference applies when it patches a row's Euler triple into the XML (ENV:870-874)
    assert np.allclose(mc.truncated_euler(top[0, 3:6]), [-1.71, -0.08, -1.65])

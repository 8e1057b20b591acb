"""Start states and action streams of the BASELINE.md configurations (host side, numpy)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from .model_compiler import euler_to_quat, load_model_blob, truncated_euler

ASSETS = Path(__file__).resolve().parent / "assets"

# hand orientation classes, kinova_gripper_env.py:1267-1273 (all non-default xml files)
ORIENTATION_EULER = {"normal": (-1.57, 0.0, -1.57), "rotated": (-1.2, 0.0, 0.0), "top": (0.0, 0.0, 0.0)}
SHAPES = [s + z for s in ["Cube", "Cylinder", "Cube45", "Cone1", "Cone2", "Vase1", "Vase2"] for z in "SB"]  # README.md:59
# the multi-geom objects of the experiment stages (main_DDPGfD.py:1270-1281; shape keys of kinova_gripper_env.py:189-208): `object` plus
# welded pieces.  They run on libkinova_sim_mg.so (sim.KinovaSim picks the library from the model blob).
MULTI_GEOM_SHAPES = [s + z for s in ["Bottle", "Bowl", "TBottle", "RBowl", "Hour"] for z in "SMB"]         # (Hour: the hourglass, `object` + top + bottom)
# the medium size of the README shapes: the experiment mode's test size (main_DDPGfD.py:1280-1281; kinova_gripper_env.py:150-180)
MEDIUM_SHAPES = [s + "M" for s in ["Cube", "Cylinder", "Cube45", "Cone1", "Cone2", "Vase1", "Vase2"]]
# the remaining single-geom families of the env's object table (kinova_gripper_env.py:181-200): vase, lemon stand-in.  The Lemon hull
# (2434 vertices) is beyond the standard library's 1024 and runs on libkinova_sim_mg.so (hull tables in global memory; sim.blob_needs_mg_library)
EXTRA_SHAPES = [s + z for s in ["Vase", "Lemon"] for z in "SMB"]



def model_blob(shape: str) -> bytes:
    """KSMB model blob of a shape (object .ksm merged with the shared hand ray-mesh tables)."""
    return load_model_blob(shape, ASSETS)


def _coord_tables(noise: str, mode: str):
    """the re-encoded coordinate files of one (noise kind, split): obj_hand_coords/<noise>/<mode>_coords (tools/compile_assets.py: coordinate_tables)"""
    if mode not in ("train", "test"):
        raise ValueError(f"start coordinates: mode is 'train' or 'test' (kinova_gripper_env.py:1241-1245), not {mode!r}")
    key = (noise, mode)
    if key not in _table_cache:
        _table_cache[key] = np.load(ASSETS / f"start_coords_{noise}_{mode}.npz")
    return _table_cache[key]


_table_cache = {}


def start_coord_table(shape: str, orientation: str = "normal", mode: str = "train") -> np.ndarray:
    """Rows of obj_hand_coords/no_noise/<mode>_coords/<Orient>/<shape>.txt as the reference's sampler
    sees them (first line consumed by the delimiter sniffer, kinova_gripper_env.py:1012)."""
    return _coord_tables("no_noise", mode)[f"{orientation.capitalize()}/{shape}"].astype(np.float64)


def noisy_start_table(shape: str, orientation: str = "normal", mode: str = "train"):
    """Rows of the reference's obj_hand_coords/with_noise/<mode>_coords/<orient>/<shape>.txt (its DEFAULT start states,
    kinova_gripper_env.py:1310, 1019-1021): object x, y, z and the hand's Euler triple of that row (patched into the XML through the
    5-character truncation, :1254-1255, 870-874).  float64 [rows, 6], or None where the reference ships no such file.  SURVEY note N5: the
    Euler columns are biased by -0.087 rad and swapped between the normal / top classes relative to the object columns - shipped as data
    so that `KinovaGripperVecEnv.reset(with_noise=True)` reproduces the reference's default as it is."""
    t = _coord_tables("with_noise", mode)
    key = f"{orientation}/{shape}"
    return t[key].astype(np.float64) if key in t.files else None


def has_start_table(shape: str, orientation: str = "normal", mode: str = "train") -> bool:
    return f"{orientation.capitalize()}/{shape}" in _coord_tables("no_noise", mode).files


def fallback_start(shape: str, orientation: str, rng=np.random) -> np.ndarray:
    """Object start (x, y, z) where the reference has NO coordinate file for (shape, orientation) - e.g. Normal/BowlS, which the
    `shapes` stage asks for: KinovaGripper_Env.randomize_initial_pos_data_collection (kinova_gripper_env.py:821-849), the
    reference's own rule for an EMPTY file (determine_obj_hand_coords, :1243-1249; for a MISSING file its check_obj_file_empty
    returns False and the open() that follows raises - the stage cannot run there at all).  size = _get_obj_size(): 'rotated'
    starts at the origin, every other class ('normal', 'top' - the old 'side' branch is never selected by name) on a disc of
    radius size[0] / 2; z = size[2] / 2."""
    from .model_compiler import read_blob
    so = read_blob(ASSETS / f"{shape}.ksm")["obj_size_obs"]
    size = np.array([so[0], so[1], so[2] / 2.0])                 # the observation stores [s0, s1, 2 s2] (kinova_gripper_env.py:529)
    if orientation == "rotated":
        x, y = 0.0, 0.0
    else:
        theta = rng.uniform(low=0, high=2 * np.pi)
        r = rng.uniform(low=0, high=size[0] / 2)
        x, y = np.sin(theta) * r, np.cos(theta) * r
    return np.array([x, y, size[-1] / 2])


_object_geom_offset = {}


def reset_body_position(shape: str, commanded_xyz) -> np.ndarray:
    """Where KinovaGripper_Env.reset leaves the object's BODY for a commanded start (x, y, z) (kinova_gripper_env.py:1367-1386): it writes
    the commanded coordinates into the free joint, reads back the pose of the geom named `object` (`_get_obj_pose`, :564-566), and when
    that lies more than 5 cm from the commanded point it writes `commanded + (commanded - geom pose)` instead - i.e. it moves the `object`
    geom's CENTRE onto the commanded point, in all three coordinates.  That is what brings the objects whose STL files carry a CAD origin
    (the bottles' main piece sits 0.19 m from its body origin, the hourglass 7.5 cm, the lemon 5.3 cm) into the hand - and, with the start
    tables' z values that were chosen for the body origin (Bottle: -0.01), what buries them in the floor.  README shapes: offset ~0, unchanged."""
    if shape not in _object_geom_offset:
        from .model_compiler import read_blob
        _object_geom_offset[shape] = read_blob(ASSETS / f"{shape}.ksm")["geom_pos"][8].copy()
    g = _object_geom_offset[shape]                       # the reset quaternion is the identity: geom pose = body position + g
    c = np.asarray(commanded_xyz, dtype=np.float64)
    return c - g if np.linalg.norm(g) > 0.05 else c.copy()


def hand_quat_for(orientation: str) -> np.ndarray:
    """Quaternion of j2s7s300_link_7 for an orientation class, including the 5-character string
    truncation the reference applies when patching the XML (kinova_gripper_env.py:870-874)."""
    return euler_to_quat(truncated_euler(ORIENTATION_EULER[orientation]))


ORIENTATION_NOISE_SIGMA = 0.087      # rad (5 deg), the spread rotation_generation.py:20-25 uses


def hand_euler_for(orientation: str, rng=None, sigma: float = ORIENTATION_NOISE_SIGMA) -> np.ndarray:
    """Euler triple of j2s7s300_link_7 as the reference would patch it into the XML: the class constant (ENV:1267-1273), with
    rng given plus ZERO-MEAN N(0, sigma) per axis (SURVEY note N5: the reference's with_noise tables are biased by -0.087 and
    swapped between classes - an extension replaces them), then truncated to 5 characters (ENV:870-874)."""
    e = np.asarray(ORIENTATION_EULER[orientation], dtype=np.float64)
    if rng is not None:
        e = e + rng.normal(0.0, sigma, 3)
    return truncated_euler(e)


_palm_quat = None


def hand_slide_offsets(orientation: str, shape: str, mode: str = "pose") -> np.ndarray:
    """Start values of the three slide joints, KinovaGripper_Env.determine_hand_location (ENV:1286-1307): zero for
    'normal'; for 'rotated' -T [0.051, -0.075, 0.06] and for 'top' -T [-0.005, -0.155, Z + 0.06] (x, y negated, z kept;
    Z = 0.13 / 0.15 for S / B objects) with T = Tfw[:3, :3], the world -> palm rotation.

    mode "pose": T is the palm rotation of the orientation itself - what the authors intended and what a persistent env
    (the evaluation loops) uses once it has seen the pose: the hand hovers above / beside the object.  Z follows the SHAPE's size letter,
    as intended; a reference env whose object came from the object schedule keeps `self.obj_size` at its __init__ value 'm' (ENV:62; only the
    obj_params hook updates it), i.e. computes Z = 0.14 for every object there (a 1 cm difference in the 'top' hover height of S / B objects).
    mode "fresh-env": zeros - what the training driver actually gets, because it builds a new env for every episode
    (main_DDPGfD.py:381) whose Tfw is still the zero matrix of __init__ (ENV:110) when reset() multiplies by it; the
    'rotated' / 'top' hands then start inside the floor and larger objects inside the hand (SURVEY note N5)."""
    global _palm_quat
    if orientation == "normal" or mode == "fresh-env":
        return np.zeros(3)
    if mode != "pose":
        raise ValueError("hand_slide_offsets: mode is 'pose' or 'fresh-env'")
    from .model_compiler import quat_to_mat, read_blob
    if _palm_quat is None:
        _palm_quat = read_blob(model_blob("CubeS"))["geom_quat"][1]
    Rp = quat_to_mat(hand_quat_for(orientation)) @ quat_to_mat(_palm_quat)
    T = (Rp @ np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)).T
    if orientation == "top":
        Z = 0.15 if shape.endswith("B") else (0.14 if shape.endswith("M") else 0.13)
        v = T @ np.array([-0.005, -0.155, Z + 0.06])
    else:
        v = T @ np.array([0.051, -0.075, 0.06])
    return np.array([-v[0], -v[1], v[2]])


def config5_states(n_envs: int, seed: int = 5, hand_offsets: str = "pose", cohort: int = 1):
    """cohort > 1: the object is drawn once per `cohort` consecutive envs instead of per env (every env's shape is still uniform over
    the 14, orientation / start row / mass / friction stay per env).  With cohort = 16 every shape's env count is a multiple of the
    stepping kernel's 16-env groups: N / 16 groups instead of up to N / 16 + 13, so that each of the free-running rollout's 256
    persistent workgroups steps exactly two groups of an 8192-env batch (bench.py --config 5).
    BASELINE config 5: env i holds one of the README's 14 shapes drawn uniformly, an orientation class from
    the reference's thresholds (ENV:1212-1220: t = rand(); < 0.333 normal, > 0.667 top, else rotated) with the env's
    no-noise Euler constants and hand offsets (ENV:1267-1273, 1286-1307), the object at a row of the matching no_noise
    table, mass ~ U[0.05, 0.15] kg and finger-object friction ~ U[0.5, 1.0]; Generator(PCG64(seed)).
    Returns object_id [N] int32, orientation names [N], qpos0 [16, N], hand_quat [4, N], mass_friction [2, N]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    oid = rng.integers(0, len(SHAPES), n_envs).astype(np.int32)
    if cohort > 1:
        oid = np.repeat(oid[:(n_envs + cohort - 1) // cohort], cohort)[:n_envs].copy()      # (same stream: cohort = 1 is the draw of rounds 2-3)
    t = rng.random(n_envs)
    names = ["normal" if x < 0.333 else ("top" if x > 0.667 else "rotated") for x in t]
    q = np.zeros((16, n_envs))
    hq = np.zeros((4, n_envs))
    q[12] = 1.0
    rows = rng.random(n_envs)
    cache = {}
    for i in range(n_envs):
        key = (SHAPES[oid[i]], names[i])
        if key not in cache:
            cache[key] = (start_coord_table(*key), hand_quat_for(names[i]), hand_slide_offsets(names[i], key[0], hand_offsets))
        tab, quat, off = cache[key]
        q[0:3, i] = off
        q[9:12, i] = tab[int(rows[i] * len(tab))]
        hq[:, i] = quat
    mass, mu = rng.uniform(0.05, 0.15, n_envs), rng.uniform(0.5, 1.0, n_envs)
    return oid, names, q, hq, np.stack([mass, mu])


def config1_state(shape: str = "CubeS"):
    """BASELINE config 1: hand joints 0, object at row 2 of Normal/<shape>.txt, identity quaternion."""
    q = np.zeros(16)
    q[9:12] = start_coord_table(shape)[0]
    q[12] = 1.0
    return q, hand_quat_for("normal")


def config2_states(n_envs: int, shape: str = "CubeS"):
    """BASELINE config 2: env i starts at row 2 + (i mod 4498) of the same table.  Returns
    qpos0 [16, N], hand_quat [4, N]."""
    tab = start_coord_table(shape)
    idx = np.arange(n_envs) % 4498
    q = np.zeros((16, n_envs))
    q[9:12] = tab[idx].T
    q[12] = 1.0
    hq = np.repeat(hand_quat_for("normal")[:, None], n_envs, axis=1)
    return q, hq


def latin_square_object_keys(shape_keys, max_elements: int):
    """Object schedule of KinovaGripper_Env.Generate_Latin_Square (kinova_gripper_env.py:895-964): cyclic rotations of
    the key list - rows keys[k:] + keys[:k] for k = n, n-1, ..., 0 (n = len - 1), repeated - cut at max_elements.
    Episodes take their object from the END of the list (get_object pops, kinova_gripper_env.py:986-989)."""
    keys = list(shape_keys)
    n = len(keys) - 1
    out = []
    while len(out) < max_elements:
        for k in range(n, -1, -1):
            row = keys[k:] + keys[:k]
            out.extend(row[:max_elements - len(out)])
            if len(out) >= max_elements:
                break
    return out


def episode_objects(shape_keys, n_episodes: int):
    """shape of episode 0, 1, ... as the reference's reset() sees them: the Latin-square list popped from its end"""
    return latin_square_object_keys(shape_keys, n_episodes)[::-1]


def select_orientation(shape: str, hand_orientation: str, rng=np.random) -> str:
    """KinovaGripper_Env.select_orienation (kinova_gripper_env.py:1180-1222) with the same np.random draws: 'normal'
    unless hand_orientation == 'random'; RBowl shapes never normal, Lemon shapes never rotated; thresholds 0.333 / 0.667."""
    t = 0.330
    if "RBowl" in shape:
        if hand_orientation == "random":
            t = rng.uniform(0.333, 1)
    elif "Lemon" in shape:
        if hand_orientation == "random":
            c1 = rng.uniform(0, 0.333)
            c2 = rng.uniform(0.667, 1)
            t = rng.choice([c1, c2])
    elif hand_orientation == "random":
        t = rng.rand()
    if t < 0.333:
        return "normal"
    if t > 0.667:
        return "top"
    return "rotated"


def config5_env_params(n_envs: int, seed: int = 5):
    """BASELINE config 5 domain randomisation: object mass ~ U[0.05, 0.15] kg and finger-object
    friction ~ U[0.5, 1.0] per env, Generator(PCG64(seed)); returns (mass [N], mu [N]) float64."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(0.05, 0.15, n_envs), rng.uniform(0.5, 1.0, n_envs)


def config_actions(n_envs: int, n_steps: int = 30, base_seed: int = 1000) -> np.ndarray:
    """Per-env action streams Generator(PCG64(base_seed + i)).uniform(-0.8, 0.8, (n_steps, 4)) as
    float32; returns [n_steps, 4, N].  (Config 1 is base_seed=0 with one env.)"""
    out = np.empty((n_steps, 4, n_envs), dtype=np.float32)
    for i in range(n_envs):
        out[:, :, i] = np.random.Generator(np.random.PCG64(base_seed + i)).uniform(-0.8, 0.8, (n_steps, 4)).astype(np.float32)
    return out

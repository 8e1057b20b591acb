"""Staged experiments of the reference's `experiment` mode (gym-kinova-gripper/main_DDPGfD.py):
which policy a stage starts from, which objects / orientations it trains on, where its files go - and a batched
driver that runs one stage on the GPU simulator.

    stage 0  pretrain_policy                (expert policy, small cube)
    stage 1  1 sizes | 2 shapes | 3 orientations                        each starts from stage 0
    stage 2  4 sizes_shapes_orientations <- 1 | 5 shapes_sizes_orientations <- 2 | 6 orientations_sizes_shapes <- 3

`experiment_info` / `experiment_input` / `experiment_dirs` restate get_experiment_info (main_DDPGfD.py:624-671),
get_exp_input (710-738) and get_experiment_file_structure (674-707); tests/golden/curriculum.json holds the
reference's own answers (tools/gen_golden_curriculum.py).
"""
from __future__ import annotations

import datetime
from pathlib import Path

import numpy as np

from . import scenarios

STAGE0 = "pretrain_policy"
STAGE1 = {"1": ["0", "sizes"], "2": ["0", "shapes"], "3": ["0", "orientations"]}
STAGE2 = {"4": ["1", "sizes_shapes_orientations"], "5": ["2", "shapes_sizes_orientations"], "6": ["3", "orientations_sizes_shapes"]}
# shape / size lists of the experiment mode (main_DDPGfD.py:1270-1282)
TRAIN_SHAPES = ["Cube", "Cylinder", "Cube45", "Vase2", "Bottle", "Bowl", "TBottle"]
TRAIN_SIZES = ["S", "B"]
TEST_SHAPES = ["Vase1", "RBowl"]
TEST_SIZES = ["M"]


def experiment_info(exp_num):
    """(prev_exp_stage, prev_exp_num, prev_exp_name, exp_stage, exp_name) of experiment 1..6.
    The reference's table of third-stage experiments is commented out, so every other number - including the
    `kitchen_sink` experiment 16 its last branch was written for - ends in a NameError there (main_DDPGfD.py:657);
    here they raise ValueError."""
    key = str(exp_num)
    if key in STAGE1:
        return "0", "0", STAGE0, "1", STAGE1[key][1]
    if key in STAGE2:
        prev = STAGE2[key][0]
        return "1", prev, STAGE1[prev][1], "2", STAGE2[key][1]
    raise ValueError(f"invalid experiment number {exp_num!r} (1..6)")


def experiment_input(exp_name: str, shapes, sizes):
    """(requested shape keys, requested orientation) of an experiment name: the name's parts say what varies -
    `shapes` (else only Cube), `sizes` (else only S), `orientations` (else only normal); kitchen_sink = all three."""
    types = ["shapes", "sizes", "orientations"] if exp_name == "kitchen_sink" else exp_name.split("_")
    if "shapes" in types and "sizes" in types:
        req = [shape + size for size in sizes for shape in shapes]
    elif "shapes" in types:
        req = [shape + "S" for shape in shapes]
    elif "sizes" in types:
        req = ["Cube" + size for size in sizes]
    else:
        req = ["CubeS"]
    return req, ("random" if "orientations" in types else "normal")


def experiment_dirs(prev_exp_stage, prev_exp_name, exp_stage, exp_name, with_grasp: bool = False, root=".", create: bool = False):
    """Directory layout of a stage (main_DDPGfD.py:674-707): rl_experiments/<no_grasp|with_grasp>/stage<k>/<name>/
    {policy, replay_buffer, output}; the expert data under expert_replay_data/<...>/combined/<shape>/<orientation>/."""
    grasp = "with_grasp" if with_grasp else "no_grasp"
    root = Path(root)
    exp_dir = root / "rl_experiments" / grasp / f"stage{exp_stage}" / exp_name
    prev_dir = root / "rl_experiments" / grasp / f"stage{prev_exp_stage}" / prev_exp_name
    d = {"exp_dir": exp_dir, "policy_dir": exp_dir / "policy", "replay_dir": exp_dir / "replay_buffer", "output_dir": exp_dir / "output",
         "prev_exp_dir": prev_dir, "prev_policy_dir": prev_dir / "policy", "prev_replay_dir": prev_dir / "replay_buffer",
         "expert_replay_dir": root / "expert_replay_data" / grasp / "combined"}
    if create:
        for k in ("exp_dir", "policy_dir", "replay_dir", "output_dir"):
            d[k].mkdir(parents=True, exist_ok=True)
    return d


def experiment_plan(exp_num, exp_mode: str = "train", with_grasp: bool = False, root="."):
    """Everything main_DDPGfD.py's experiment mode derives from the experiment number before it starts training
    (main_DDPGfD.py:1265-1300): stage info, shape keys, orientation classes, directories."""
    shapes, sizes = (TEST_SHAPES, TEST_SIZES) if exp_mode == "test" else (TRAIN_SHAPES, TRAIN_SIZES)
    prev_stage, prev_num, prev_name, stage, name = experiment_info(exp_num)
    req, orientation = experiment_input(name, shapes, sizes)
    return {"exp_num": int(exp_num), "exp_stage": stage, "exp_name": name, "prev_exp_stage": prev_stage, "prev_exp_num": prev_num,
            "prev_exp_name": prev_name, "requested_shapes": req, "requested_orientation": orientation,
            "requested_orientation_list": ["normal", "rotated", "top"] if orientation == "random" else ["normal"],
            "dirs": experiment_dirs(prev_stage, prev_name, stage, name, with_grasp, root)}


def policy_basename(policy_dir) -> str:
    """name of the checkpoint saved in a policy directory (rl_experiment: glob '*_actor_optimizer', main_DDPGfD.py:781-785)"""
    hits = sorted(Path(policy_dir).glob("*_actor_optimizer"))
    if not hits:
        raise FileNotFoundError(f"no '*_actor_optimizer' checkpoint file in {policy_dir}")
    return hits[0].name[: -len("_actor_optimizer")]


def run_stage(plan, policy, n_envs: int = 1024, rounds: int = 10, updates_per_round: int = 100, expert_prob: float = 0.3, seed: int = 2,
              device: int = 0, load_previous: bool = True, save: bool = True, eval_envs: int | None = None):
    """One stage of the curriculum on the GPU simulator (the batched counterpart of rl_experiment + train_policy,
    main_DDPGfD.py:600-621, 776-800): start from the previous stage's policy and agent replay, mix in the expert
    replay of the stage's shapes at `expert_prob` (DDPGfD.py:232-254), train, evaluate, save policy + replay + info.

    The reference runs max_episode episodes one at a time, objects in Latin-square order, and 100 updates at the end
    of each (main_DDPGfD.py:466-486).  Here a ROUND is one episode of every env (the envs are split over the stage's
    shapes, orientation classes drawn per env by the reference's rule) followed by `updates_per_round` updates.
    Every shape key of the reference's stages has a compiled asset since round 4 - the multi-geom Bottle / Bowl / TBottle / RBowl
    objects included (a stage that holds one runs on libkinova_sim_mg.so, sim.KinovaSim picks it) -; a key without one would be
    listed under `skipped_shapes`.  Where the reference has no start-coordinate file for a (shape, orientation) - Normal/BowlS - the
    start is drawn by the reference's empty-file rule (scenarios.fallback_start).  Returns a dict (num_success, num_total, paths, ...)."""
    import torch

    from .evaluate import eval_policy
    from .multi_shape import MultiShapeSim
    from .replay import DeviceEpisodeReplay
    from .rollout import RolloutEngine

    dirs = plan["dirs"]
    known = scenarios.SHAPES + scenarios.MEDIUM_SHAPES + scenarios.EXTRA_SHAPES + scenarios.MULTI_GEOM_SHAPES
    shapes = [s for s in plan["requested_shapes"] if s in known]
    skipped = [s for s in plan["requested_shapes"] if s not in known]
    if not shapes:
        raise ValueError(f"none of the stage's shapes {plan['requested_shapes']} has a compiled asset")
    rng = np.random.RandomState(seed)
    dev = torch.device("cuda", device)
    if load_previous:
        policy.load(str(dirs["prev_policy_dir"] / policy_basename(dirs["prev_policy_dir"])))
    sim = MultiShapeSim(n_envs, shapes, device=device, auto_reset=False, horizon=30)
    replay = DeviceEpisodeReplay(n_envs, capacity=max(4 * n_envs, 1024), horizon=30, device=dev)
    if load_previous and dirs["prev_replay_dir"].is_dir():
        replay.load(dirs["prev_replay_dir"])
    expert = None
    for s in shapes:                      # expert_replay_data/<grasp>/combined/<shape>/<orientation>/replay_buffer (main_DDPGfD.py:1183-1187)
        p = dirs["expert_replay_dir"] / s / plan["requested_orientation"] / "replay_buffer"
        if p.is_dir():
            if expert is None:
                expert = DeviceEpisodeReplay(n_envs, capacity=max(4 * n_envs, 1024), horizon=30, device=dev)
            expert.load(p)

    def reset_all(the_sim, count):
        """per env: orientation class by the reference's rule, start row from that class's table of the env's shape"""
        q, hq, classes = np.zeros((16, count)), np.zeros((4, count)), []
        q[12] = 1.0
        shape_ids = the_sim.shape_of_env.cpu().numpy()
        for e in range(count):
            shape = the_sim.shapes[shape_ids[e]]
            o = scenarios.select_orientation(shape, plan["requested_orientation"], rng) if plan["requested_orientation"] == "random" else "normal"
            if scenarios.has_start_table(shape, o):
                tab = scenarios.start_coord_table(shape, o)
                q[9:12, e] = tab[rng.randint(0, len(tab))]
            else:
                q[9:12, e] = scenarios.fallback_start(shape, o, rng)
            q[9:12, e] = scenarios.reset_body_position(shape, q[9:12, e])          # the reference reset's 5 cm correction (ENV:1379-1386)
            hq[:, e] = scenarios.hand_quat_for(o)
            classes.append(o)
        return the_sim.reset(torch.as_tensor(q), torch.as_tensor(hq)), classes

    eng = RolloutEngine(sim, policy, replay, expl_noise=0.1)
    # the updates run on the product learner (learner_native: fused MFMA forward / backward launches + kr_* glue), each batch =
    # int(batch_size * (1 - expert_prob)) agent + the rest expert episodes sampled by ONE launch (kr_sample_windows_mixed,
    # DDPGfD.train_batch's mix, DDPGfD.py:232-254); without expert data every episode comes from the agent ring
    from .learner_native import NativeDDPGfDUpdate
    native = NativeDDPGfDUpdate(policy)
    mix = expert is not None and expert.count >= 2
    losses = []
    for r in range(rounds):
        obs0, _ = reset_all(sim, n_envs)
        eng.start(obs0)
        for t in range(30):
            eng.step()
        if replay.count >= 2:
            for u in range(updates_per_round):
                batch = replay.sample_mixed(expert, policy.batch_size, expert_prob) if mix else replay.sample_batch_nstep(policy.batch_size)
                st, ac, ns, rw, nd, w = batch[:6]
                losses.append(native.train_on_batch(st, ac, ns, rw, w))
    # final evaluation: one deterministic episode per env (eval_policy, main_DDPGfD.py:130-272)
    obs0, classes = reset_all(sim, n_envs)
    res = eval_policy(sim, policy, obs0, horizon=30, orientation=plan["requested_orientation"])
    out = {"expert_episodes": 0 if expert is None else int(expert.count), "num_success": res["num_success"], "num_total": n_envs, "avg_reward": res["avg_reward"], "skipped_shapes": skipped, "shapes": shapes,
           "updates": len(losses), "orientation_counts": {c: classes.count(c) for c in sorted(set(classes))}}
    if save:
        for k in ("policy_dir", "replay_dir", "output_dir"):
            Path(dirs[k]).mkdir(parents=True, exist_ok=True)
        stamp = datetime.datetime.now().strftime("%m_%d_%y_%H%M")
        name = f"DDPGfD_kinovaGrip_{stamp}"
        policy.save(str(dirs["policy_dir"] / name))
        replay.save(dirs["replay_dir"])
        text = (f"{'WITH' if 'with_grasp' in str(dirs['exp_dir']) else 'NO'} grasp Experiment {plan['exp_num']}: {plan['exp_name']}, Stage {plan['exp_stage']}\n"
                f"Date: {stamp}\nPrevious experiment: {plan['prev_exp_name']}\nExperiment shapes: {plan['requested_shapes']}\n"
                f"Experiment orientation: {plan['requested_orientation']}\nFinal Policy Evaluation:\n# Success: {out['num_success']}\n"
                f"# Failures: {n_envs - out['num_success']}\n# Total: {n_envs}\nOutput directory: {dirs['exp_dir']}")
        (dirs["output_dir"] / "experiment_info.txt").write_text(text)          # rl_experiment's info file (main_DDPGfD.py:802-822)
        out["policy_path"], out["replay_path"] = str(dirs["policy_dir"] / name), str(dirs["replay_dir"])
    sim.close()
    return out

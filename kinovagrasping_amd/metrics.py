"""Experiment outputs in the reference's file formats: heatmap coordinate arrays, reward box-plot data
and the scalars its training loop sends to tensorboard.  The plots themselves (plotting_code/heatmap_plot.py,
boxplot_plot.py) and tensorboard are not reproduced; these are the data files they are drawn from.

    save_heatmap_coords   plotting_code/heatmap_coords.py:34-99 (filter_heatmap_coords -> save_coordinates)
    save_boxplot_rewards  main_DDPGfD.py:521-528
    ScalarLog             main_DDPGfD.py:310-330 (write_tensor_plot): same tags, written as JSON lines
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

ORIENTATIONS = ("normal", "rotated", "top")


def save_heatmap_coords(success_coords, fail_coords, episode_num, saving_dir):
    """Object start coordinates of an evaluation, split by outcome and hand orientation, as the reference writes them:
    <dir>/<orientation>/{success,fail,total}_{x,y}[_<episode>].npy + heatmap_info.txt.  `success_coords` / `fail_coords`:
    dicts with "x", "y", "orientation" lists (evaluate.eval_policy returns them).  As in the reference, the `total`
    arrays of an orientation are written once from the successes and then AGAIN from the failures when it has any
    (heatmap_coords.py:48-56), so they end up holding the failures wherever an orientation has both."""
    saving_dir = Path(saving_dir)
    idx = lambda c, o: [i for i, x in enumerate(c["orientation"]) if x == o]
    ep = "" if episode_num is None else "_" + str(episode_num)

    def write(coords, indexes, name):
        d = saving_dir / str(coords["orientation"][indexes[0]])
        d.mkdir(parents=True, exist_ok=True)
        np.save(d / f"{name}_x{ep}.npy", np.array([coords["x"][i] for i in indexes]))
        np.save(d / f"{name}_y{ep}.npy", np.array([coords["y"][i] for i in indexes]))

    s_idx, f_idx = [idx(success_coords, o) for o in ORIENTATIONS], [idx(fail_coords, o) for o in ORIENTATIONS]
    for il in s_idx:
        if il:
            write(success_coords, il, "success")
            write(success_coords, il, "total")
    for il in f_idx:
        if il:
            write(fail_coords, il, "fail")
            write(fail_coords, il, "total")
    saving_dir.mkdir(parents=True, exist_ok=True)
    text = (f"Heatmap Coords \nSaved at: {saving_dir}\n\nTotal # Success: {len(success_coords['x'])}\nTotal # Fail: {len(fail_coords['x'])}\n")
    for o, s, f in zip(("Normal", "Rotated", "Top"), s_idx, f_idx):
        text += f"\n{o} Orientation\n# Success: {len(s)}\n# Fail: {len(f)}\n"
    (saving_dir / "heatmap_info.txt").write_text(text)
    return text


def save_boxplot_rewards(boxplot_dir, episode_num, finger_reward, grasp_reward, lift_reward, total_reward):
    """finger/grasp/lift/total_reward_<episode>.npy (main_DDPGfD.py:525-528)"""
    d = Path(boxplot_dir)
    d.mkdir(parents=True, exist_ok=True)
    for name, v in (("finger_reward", finger_reward), ("grasp_reward", grasp_reward), ("lift_reward", lift_reward), ("total_reward", total_reward)):
        np.save(d / f"{name}_{episode_num}.npy", np.asarray(v, dtype=object) if _ragged(v) else np.asarray(v))


def _ragged(v):
    try:
        return len({len(x) for x in v}) > 1
    except TypeError:
        return False


class ScalarLog:
    """add_scalar(tag, value, step) like tensorboardX.SummaryWriter, appended to <dir>/scalars.jsonl; write_eval() uses
    the reference's tags (write_tensor_plot, main_DDPGfD.py:310-330)."""

    def __init__(self, log_dir, eval_freq: int = 200):
        self.dir = Path(log_dir)
        self.dir.mkdir(parents=True, exist_ok=True)
        self.path = self.dir / "scalars.jsonl"
        self.eval_freq = eval_freq

    def add_scalar(self, tag: str, value, step: int):
        with open(self.path, "a") as f:
            f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def write_eval(self, episode_num, avg_reward, avg_rewards, actor_loss, critic_loss, critic_L1loss, critic_LNloss):
        k = f", Avg. {self.eval_freq} episodes"
        self.add_scalar("Episode total reward" + k, avg_reward, episode_num)
        self.add_scalar("Episode finger reward" + k, avg_rewards["finger_reward"], episode_num)
        self.add_scalar("Episode grasp reward" + k, avg_rewards["grasp_reward"], episode_num)
        self.add_scalar("Episode lift reward" + k, avg_rewards["lift_reward"], episode_num)
        self.add_scalar("Actor loss", actor_loss, episode_num)
        self.add_scalar("Critic loss", critic_loss, episode_num)
        self.add_scalar("Critic L1loss", critic_L1loss, episode_num)
        self.add_scalar("Critic LNloss", critic_LNloss, episode_num)

    def read(self):
        return [json.loads(l) for l in open(self.path)] if self.path.exists() else []

#!/usr/bin/env python3
"""Dev-only (needs /root/reference): collect every piece of REAL MuJoCo 1.50 output the reference tree holds into one
fixture, tests/golden/mujoco_recorded.npz.  These are data files of the reference (numbers its authors recorded from
mujoco-py), not source text:

  pose_file            gym-kinova-gripper/Old Code/Pose_file.csv  [58, 7]
                       joint-state rows [wrist, f1_prox, f2_prox, f3_prox, f1_dist, f2_dist, f3_dist] written once per
                       env.step() of the frame_skip = 4 env (kinova_gripper_env_s.py:42,683-696; writer
                       Old Code/main_DDPGfD_OG.py:64-70).  Rows 0-26: fingers 1 and 3 close freely at the servo's
                       maximum command (steady 0.02908 rad per row = 0.8 * 2.5 / 2.75 rad/s * 0.04 s), finger 2 is
                       commanded 0 and sags under gravity; later rows involve the old env's object.
  pose_file_2          gym-kinova-gripper/Old Code/Pose_file_2.csv  [63, 48]
                       state[0:48] ("global" representation, kinova_gripper_env_s.py:181-209) once per env.step() of the same
                       frame_skip = 4 env on `j2s7s300_end_effector_v1_mbox (copy).xml` (ONE slide joint_7, range 0 - 0.2, 4 velocity
                       actuators, box 0.02125 x 0.02125 x 0.055), writer Old Code/main_DDPGfD_OG.py:36-70: columns 0-17 geom_xpos
                       of f1_prox f2_prox f3_prox f1_dist f2_dist f3_dist, 18-20 palm geom_xpos, 21-23 object geom_xpos, 24-30
                       jointpos sensors [wrist, f1, f2, f3 proximal, f1, f2, f3 distal], 31-33 object size, 34-46 site-object
                       distances (kinova_gripper_env_s.py:210-224), 47 dot product (:266-281).  Row 0 = reset (object released
                       5 mm inside the floor), rows 1-3 the box recovers before anything touches it, rows 4-33 finger 1 pushes it
                       across the floor while the hand closes, rows 34-62 grasp and lift.  The ONLY contact trajectory of real
                       MuJoCo in the reference tree.
  demo_*               gym-kinova-gripper/expert_plots/{heatmap_train_{success,fail}_new_{x,y}_arr,success_timesteps,
                       fail_timesteps}.npy: ten recorded demonstrations (expert_data.py:690-921): palm-frame start
                       (x, y) of the object, outcome, env-steps until done.
  heat_success/fail    gym-kinova-gripper/expert_plots/heatmap_plots/{success,fail}_heatmap.png
                       ("Grasp Trial Success / Failure Rate per Initial Pose of Object (CubeS) - Naive Controller"):
                       the rasterised rate maps digitised on a 90 x 45 lattice of 2 mm cells over x in [-0.09, 0.09],
                       y in [0, 0.09] (pixel colour -> percent through the image's own colour bar).
"""
from pathlib import Path

import numpy as np
from PIL import Image

REF = Path("/root/reference/gym-kinova-gripper")
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden" / "mujoco_recorded.npz"


def digitize(png: Path, nx: int = 90, ny: int = 45):
    im = np.asarray(Image.open(png).convert("RGB")).astype(float)
    H, W, _ = im.shape
    dark = im.sum(2) < 100
    rows = np.where(dark.sum(1) > 0.5 * W)[0]                # the axes frame's horizontal lines
    cols = np.where(dark.sum(0) > 0.3 * H)[0]                # frame verticals, then the colour bar's
    yt, yb, x0, x1 = rows.min(), rows.max(), cols[0], cols[1]
    cb = np.where(dark[:, cols[2] + 3:cols[3] - 2].sum(1) > 5)[0]
    top, bot = cb.min(), cb.max()
    r = np.arange(top + 1, bot)
    bar = im[r, (cols[2] + cols[3]) // 2, :]
    pct = 100.0 - 200.0 * (r - top - 0.5) / (bot - top)
    grid = np.zeros((ny, nx))
    for j in range(ny):
        for i in range(nx):
            px = int(round(x0 + (i + 0.5) * (x1 - x0) / nx))
            py = int(round(yb - (j + 0.5) * (yb - yt) / ny))
            c = im[py - 1:py + 2, px - 1:px + 2].reshape(-1, 3).mean(0)
            grid[j, i] = pct[np.abs(bar - c).sum(1).argmin()]
    return grid


def main():
    pose = np.loadtxt(REF / "Old Code" / "Pose_file.csv", delimiter=",")
    pose2 = np.loadtxt(REF / "Old Code" / "Pose_file_2.csv", delimiter=",")
    assert pose2.shape == (63, 48)
    E = REF / "expert_plots"
    sx, sy = np.load(E / "heatmap_train_success_new_x_arr.npy"), np.load(E / "heatmap_train_success_new_y_arr.npy")
    fx, fy = np.load(E / "heatmap_train_fail_new_x_arr.npy"), np.load(E / "heatmap_train_fail_new_y_arr.npy")
    st, ft = np.load(E / "success_timesteps.npy"), np.load(E / "fail_timesteps.npy")
    hs = digitize(E / "heatmap_plots" / "success_heatmap.png")
    hf = -digitize(E / "heatmap_plots" / "fail_heatmap.png")
    hs[np.abs(hs) < 3] = 0.0
    hf[np.abs(hf) < 3] = 0.0
    np.savez_compressed(
        OUT, pose_file=pose, pose_file_2=pose2,
        demo_x=np.r_[sx, fx], demo_y=np.r_[sy, fy], demo_success=np.r_[np.ones(len(sx)), np.zeros(len(fx))].astype(np.int32),
        demo_steps=np.r_[st, ft].astype(np.int32), all_timesteps=np.load(E / "all_timesteps.npy"),
        heat_success=hs.astype(np.float32), heat_fail=hf.astype(np.float32),
        heat_x=(-0.09 + 0.002 * (np.arange(90) + 0.5)), heat_y=(0.002 * (np.arange(45) + 0.5)))
    print(f"wrote {OUT}: pose_file {pose.shape}, pose_file_2 {pose2.shape}, {len(sx)} + {len(fx)} demonstrations, "
          f"{(hs > 0).sum()} success cells, {(hf > 0).sum()} failure cells")


if __name__ == "__main__":
    main()

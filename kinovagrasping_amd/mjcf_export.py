"""Self-contained MJCF from a compiled model blob: what a machine that HAS the third-party `mujoco` Python package needs in
order to cross-check this repo's physics against real MuJoCo (tests/test_mujoco_crosscheck.py, bench.py's optional
`cpu_baseline_mujoco` leg).  Nothing of the reference travels: the XML is rebuilt from the blob (.ksm), whose numbers
the model compiler derived from the reference's MJCF + STL assets - bodies, explicit inertials, joints, the convex hulls as
inline mesh vertices (MuJoCo re-derives the same hull from them), sites, contact pairs, tendons, equalities, actuators,
sensors - with the hand's per-episode orientation passed as a quaternion.

Caveats a reader of the cross-check must know: the rangefinders see convex hulls here (the repo's
own rays use the original triangles), `mujoco` >= 2.1.2 is needed for inline mesh vertices, and a modern MuJoCo is not
MuJoCo 1.50 (5-parameter solimp with the same default curve, native convex collision since 3.2, `autolimits` switched off
explicitly below so that the tendons' `range` stays inactive as in 1.50)."""
from __future__ import annotations

import numpy as np

from .model_compiler import read_blob

BODY = ["world", "root", "j2s7s300_link_7", "j2s7s300_link_finger_1", "j2s7s300_link_finger_tip_1", "j2s7s300_link_finger_2",
        "j2s7s300_link_finger_tip_2", "j2s7s300_link_finger_3", "j2s7s300_link_finger_tip_3", "object"]
GEOM = ["ground", "palm", "f1_prox", "f1_dist", "f2_prox", "f2_dist", "f3_prox", "f3_dist", "object"]
SITE = ["palm", "palm_1", "palm_2", "palm_3", "palm_4", "f1_prox", "f1_prox_1", "f1_dist", "f1_dist_1", "f2_prox", "f2_prox_1",
        "f2_dist", "f2_dist_1", "f3_prox", "f3_prox_1", "f3_dist", "f3_dist_1"]
PARENT = [0, 0, 1, 2, 3, 2, 5, 2, 7, 0]
HINGE = ["j2s7s300_joint_finger_1", "j2s7s300_joint_finger_tip_1", "j2s7s300_joint_finger_2", "j2s7s300_joint_finger_tip_2",
         "j2s7s300_joint_finger_3", "j2s7s300_joint_finger_tip_3"]
SLIDE = ["j2s7s300_slide_x", "j2s7s300_slide_y", "j2s7s300_slide_z"]


def _f(x):
    return repr(float(x))


def _v(a):
    return " ".join(repr(float(x)) for x in np.asarray(a).ravel())


def to_mjcf(blob, hand_quat) -> str:
    """MJCF text of the model in `blob` (bytes or path) with link_7 oriented by hand_quat (w x y z)."""
    M = read_blob(blob)
    dt, impratio, gz, margin, sr0, sr1, d0, d1, dw = M["opt"][:9]
    out = ['<mujoco model="j2s7s300_end_effector">',
           '  <compiler angle="radian" autolimits="false" balanceinertia="false"/>',
           f'  <option timestep="{_f(dt)}" impratio="{_f(impratio)}" gravity="0 0 {_f(gz)}" cone="pyramidal" solver="Newton" iterations="100" tolerance="1e-10"/>',
           # every candidate pair of the compiled model (the XML's 8 explicit pairs AND the 22 the contype / conaffinity rule adds) is
           # written as an explicit <pair> below, so the geoms themselves must not generate pairs a second time
           f'  <default><geom margin="{_f(margin)}" solref="{_f(sr0)} {_f(sr1)}" solimp="{_f(d0)} {_f(d1)} {_f(dw)}" contype="0" conaffinity="0"/></default>', "  <asset>"]
    for k in range(4):
        out.append(f'    <mesh name="mesh{k}" vertex="{_v(M[f"mesh{k}_vert"])}"/>')
    out.append("  </asset>\n  <worldbody>")
    geoms_of = {b: [g for g in range(9) if M["geom_body"][g] == b] for b in range(10)}
    sites_of = {b: [s for s in range(17) if M["site_body"][s] == b] for b in range(10)}
    children = {b: [c for c in range(1, 10) if PARENT[c] == b and c != b] for b in range(10)}

    def geom_xml(g, ind):
        if g == 0:
            return f'{ind}<geom name="ground" type="plane" pos="{_v(M["geom_pos"][0])}" size="{_v(M["geom_size"][0][:2])} 0.1"/>'
        return (f'{ind}<geom name="{GEOM[g]}" type="mesh" mesh="mesh{int(M["geom_mesh"][g])}" pos="{_v(M["geom_pos"][g])}" '
                f'quat="{_v(M["geom_quat"][g])}"/>')

    def body_xml(b, ind):
        quat = hand_quat if b == 2 else M["body_quat"][b]
        out.append(f'{ind}<body name="{BODY[b]}" pos="{_v(M["body_pos"][b])}" quat="{_v(quat)}">')
        if M["body_mass"][b] > 0:
            out.append(f'{ind}  <inertial pos="{_v(M["body_ipos"][b])}" quat="{_v(M["body_iquat"][b])}" mass="{_f(M["body_mass"][b])}" '
                       f'diaginertia="{_v(M["body_inertia"][b])}"/>')
        if b == 2:
            for i in range(3):
                out.append(f'{ind}  <joint name="{SLIDE[i]}" type="slide" axis="{_v(M["slide_axis"][i])}" limited="true" range="{_v(M["slide_range"][i])}" '
                           f'damping="{_f(M["dof_damping"][i])}" armature="{_f(M["dof_armature"][i])}"/>')
        elif 3 <= b <= 8:
            h = b - 3
            lim = "true" if M["hinge_limited"][h] else "false"
            out.append(f'{ind}  <joint name="{HINGE[h]}" type="hinge" axis="0 0 1" limited="{lim}" range="{_v(M["hinge_range"][h])}" '
                       f'damping="{_f(M["dof_damping"][3 + h])}" armature="{_f(M["dof_armature"][3 + h])}"/>')
        elif b == 9:
            out.append(f'{ind}  <joint name="object" type="free" damping="{_f(M["dof_damping"][9])}" armature="{_f(M["dof_armature"][9])}"/>')
        for g in geoms_of[b]:
            out.append(geom_xml(g, ind + "  "))
        for s in sites_of[b]:
            out.append(f'{ind}  <site name="{SITE[s]}" pos="{_v(M["site_pos"][s])}" quat="{_v(M["site_quat"][s])}" size="0.002"/>')
        for c in children[b]:
            body_xml(c, ind + "  ")
        out.append(f"{ind}</body>")

    out.append(geom_xml(0, "    "))
    for c in children[0]:
        body_xml(c, "    ")
    out.append("  </worldbody>\n  <contact>")
    for g1, g2, mu1, mu2, mg in M["pairs"]:
        out.append(f'    <pair geom1="{GEOM[int(g1)]}" geom2="{GEOM[int(g2)]}" condim="3" friction="{_f(mu1)} {_f(mu2)} 0.005 0.0001 0.0001" margin="{_f(mg)}"/>')
    out.append("  </contact>\n  <tendon>")
    for t in range(3):
        c0, c1 = M["tendon_coef"][t]
        out.append(f'    <fixed name="finger_{t + 1}"><joint joint="{HINGE[2 * t]}" coef="{_f(c0)}"/><joint joint="{HINGE[2 * t + 1]}" coef="{_f(c1)}"/></fixed>')
    out.append("  </tendon>\n  <equality>")
    for t in range(3):
        out.append(f'    <tendon tendon1="finger_{t + 1}" solref="{_f(sr0)} {_f(sr1)}" solimp="{_f(d0)} {_f(d1)} {_f(dw)}"/>')
    out.append("  </equality>\n  <actuator>")
    kv_s, gear, cr_s, kv_f, cr_f = M["actuator"]
    for i in range(3):
        out.append(f'    <velocity joint="{SLIDE[i]}" kv="{_f(kv_s)}" ctrllimited="true" ctrlrange="{_f(-cr_s)} {_f(cr_s)}"/>')
        out.append(f'    <motor joint="{SLIDE[i]}" gear="{_f(gear)}"/>')
    for f in range(3):
        out.append(f'    <velocity joint="{HINGE[2 * f]}" kv="{_f(kv_f)}" ctrllimited="true" ctrlrange="{_f(-cr_f)} {_f(cr_f)}"/>')
    out.append("  </actuator>\n  <sensor>")
    for j in SLIDE + HINGE[0::2] + HINGE[1::2]:
        out.append(f'    <jointpos joint="{j}"/>')
    for s in SITE:
        out.append(f'    <rangefinder site="{s}"/>')
    out.append("  </sensor>\n</mujoco>")
    return "\n".join(out)

"""The MJCF-subset model compiler (quadruped-gym_amd/model/compiler.py): mesh mass properties against closed
forms, defaults / childclass inheritance, and -- where the reference checkout is present -- that compiling its
scene.xml reproduces the constants shipped in include/qg_model_data.h bit for bit."""
import os
import textwrap

import numpy as np
import pytest

from quadruped_gym_amd.model import compiler as MC

REF_SCENE = "/root/reference/src/models/quadruped/scene.xml"


def box_obj(path, lx, ly, lz, centre=(0, 0, 0)):
    c = np.array(centre)
    v = np.array([[sx * lx / 2, sy * ly / 2, sz * lz / 2] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) + c
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]   # outward winding
    with open(path, "w") as fh:
        for p in v:
            fh.write("v %.9f %.9f %.9f\n" % tuple(p))
        for q in quads:
            fh.write("f %d %d %d %d\n" % tuple(i + 1 for i in q))


def test_box_mass_properties(tmp_path):
    p = str(tmp_path / "box.obj")
    box_obj(p, 0.2, 0.1, 0.4, centre=(0.05, -0.02, 0.3))
    v, f = MC.load_obj(p)
    for mode in ("exact", "convex"):
        vol, com, I = MC.mesh_properties(v, f, mode)
        assert vol == pytest.approx(0.2 * 0.1 * 0.4, rel=1e-12)
        assert np.allclose(com, [0.05, -0.02, 0.3], atol=1e-12)
        m = vol                                             # unit density
        assert np.allclose(np.diag(I), [m * (0.1 ** 2 + 0.4 ** 2) / 12, m * (0.2 ** 2 + 0.4 ** 2) / 12, m * (0.2 ** 2 + 0.1 ** 2) / 12], rtol=1e-10)
        assert np.abs(I - np.diag(np.diag(I))).max() < 1e-15


def test_contact_point_selection_keeps_extremes():
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(300, 3)) * [0.05, 0.02, 0.01]
    corners = np.array([[sx * 0.3, sy * 0.2, sz * 0.1] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    hv, _ = MC.hull_faces(np.vstack([pts, corners]))
    sel = MC.select_contact_points(np.vstack([pts, corners])[hv], 8)
    assert sorted(map(tuple, np.round(sel, 6))) == sorted(map(tuple, np.round(corners, 6)))     # a box hull -> its 8 corners


def _toy_mjcf(tmp_path):
    box_obj(str(tmp_path / "B.obj"), 0.1, 0.1, 0.02)
    box_obj(str(tmp_path / "L.obj"), 0.02, 0.08, 0.02, centre=(0, 0.04, 0))
    legs = ""
    for k, yaw in enumerate((0, 90, 180, -90)):
        legs += f'''
        <body name="fema_{k}" pos="{0.05 * np.cos(np.deg2rad(yaw + 135)):.6f} {0.05 * np.sin(np.deg2rad(yaw + 135)):.6f} 0.01" euler="0 0 {yaw}">
          <joint name="hip_{k}" class="hip"/> <geom class="link"/>
          <body name="shin_{k}" pos="0 0.08 0" euler="0 90 0"> <joint name="knee_{k}" class="knee"/> <geom class="link"/>
            <body name="foot_{k}" pos="0 0.08 0"> <joint name="ankle_{k}" class="ankle"/> <geom class="link" mass="0.05"/> </body>
          </body>
        </body>'''
    acts = "".join(f'<position joint="{j}_{k}" class="{j if j != "knee" else "knee"}"/>' for k in range(4) for j in ("hip", "knee", "ankle"))
    acts = acts.replace('class="hip"', 'class="hip"')
    xml = textwrap.dedent(f'''
    <mujoco>
      <compiler angle="degree" meshdir="."/>
      <option integrator="implicitfast"/>
      <default><default class="robot">
        <geom type="mesh" friction="0.7" margin="0.002"/>
        <joint axis="0 0 1" type="hinge" damping="0.3" armature="0.002"/>
        <position kp="50" kv="2" timeconst="0.02" forcerange="-2 2" ctrlrange="-1 1" gear="0.5"/>
        <default class="hip"><joint range="-30 30" ref="-10"/></default>
        <default class="knee"><joint range="-40 100" ref="20"/><position ctrlrange="-0.8 0.8"/></default>
        <default class="ankle"><joint range="-90 90"/></default>
        <default class="base"><geom mesh="B" mass="0.3"/></default>
        <default class="link"><geom mesh="L" mass="0.02"/></default>
      </default></default>
      <asset><mesh name="B" file="B.obj"/><mesh name="L" file="L.obj"/></asset>
      <worldbody>
        <geom name="floor" type="plane" size="0 0 0.05"/>
        <body name="FRAME" pos="0 0 0.2" childclass="robot">
          <joint name="root" type="free"/> <geom class="base"/>{legs}
        </body>
      </worldbody>
      <actuator>{acts}</actuator>
      <sensor><jointpos joint="hip_0" name="s0"/><accelerometer site="x" name="acc"/></sensor>
    </mujoco>''')
    path = tmp_path / "toy.xml"
    path.write_text(xml)
    return str(path)


def test_defaults_childclass_and_units(tmp_path):
    m = MC.compile_mjcf(_toy_mjcf(tmp_path))
    assert len(m["bodies"]) == 13 and len(m["joints"]) == 12 and len(m["actuators"]) == 12
    assert m["free_damping"] == 0.3 and m["free_armature"] == 0.002         # the free joint inherits through childclass
    j = m["joints"]
    assert j[0]["ref"] == pytest.approx(np.deg2rad(-10)) and j[0]["range"] == pytest.approx(list(np.deg2rad([-30, 30])))
    assert j[1]["ref"] == pytest.approx(np.deg2rad(20)) and j[2]["ref"] == 0.0
    a = m["actuators"]
    assert a[1]["ctrlrange"] == [-0.8, 0.8] and a[0]["ctrlrange"] == [-1.0, 1.0] and a[0]["gear"] == 0.5 and a[0]["kv"] == 2.0
    assert m["bodies"][3]["mass"] == pytest.approx(0.05) and m["bodies"][1]["mass"] == pytest.approx(0.02)   # explicit mass beats the class
    assert m["contact"]["friction"] == 1.0 and m["contact"]["margin"] == 0.002                                 # max(floor 1.0, geom 0.7)
    assert m["nsensordata"] == 4
    q0 = MC.qpos0(m)
    assert q0[:7] == [0, 0, 0.2, 1, 0, 0, 0] and q0[7] == pytest.approx(np.deg2rad(-10))
    # the shin frame is turned by 90 degrees about y: intrinsic xyz euler convention
    assert np.allclose(MC.quat_to_mat(np.asarray(m["bodies"][2]["quat"])) @ [0, 0, 1], [1, 0, 0], atol=1e-12)


def test_rejects_what_it_does_not_model(tmp_path):
    path = _toy_mjcf(tmp_path)
    bad = open(path).read().replace('integrator="implicitfast"', 'integrator="RK4"')
    p2 = tmp_path / "bad.xml"
    p2.write_text(bad)
    with pytest.raises(ValueError, match="implicitfast"):
        MC.compile_mjcf(str(p2))
    with pytest.raises(FileNotFoundError):
        MC.compile_mjcf(str(tmp_path / "nope.xml"))


@pytest.mark.skipif(not os.path.exists(REF_SCENE), reason="reference checkout not present on this host")
def test_reference_scene_compiles_to_the_shipped_constants():
    from quadruped_gym_amd import _abi
    from quadruped_gym_amd.model.loader import load_model
    m, layout = load_model(REF_SCENE)
    assert bytes(m) == bytes(_abi.default_model())          # same bits as include/qg_model_data.h
    assert layout["nsensordata"] == 33 and len(layout["sensors"]) == 19     # quadruped.xml:174-217
    assert abs(sum(m.body_mass) - 1.110) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mapping", ["lane", "quad"])
def test_a_different_robot_end_to_end(tmp_path, oracle, mapping):
    """A toy quadruped with other masses, geometry, gains and joint offsets, compiled from MJCF on the spot: the
    table-driven kernel variants must track the oracle on that model too (nothing is specific to the shipped numbers)."""
    from quadruped_gym_amd import _abi
    from quadruped_gym_amd.model.loader import load_model
    from quadruped_gym_amd.sim import BatchedSim
    model, _ = load_model(_toy_mjcf(tmp_path))
    task = _abi.default_task()
    n = 80
    rng = np.random.default_rng(4)
    sim = BatchedSim(n, model=model, task=task)
    sim.set_mapping({"lane": _abi.MAP_LANE, "quad": _abi.MAP_QUAD}[mapping])
    assert not sim.baked
    otask = oracle.default_task()
    b = oracle.Batch(model, otask, n)
    b.reset()
    worst = 0.0
    for k in range(60):                     # drop, land, flail: restart both from the oracle's state every step
        q, v, a, _, ns = b.get_state()
        q32, v32, a32 = q.astype(np.float32), v.astype(np.float32), a.astype(np.float32)
        b.set_state(q32.astype(np.float64), v32.astype(np.float64), a32.astype(np.float64), None, ns)
        sim.set_state(q32, v32, a32, None, ns)
        act = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        obs_o, rew_o, _, _ = b.step(act.astype(np.float64))
        obs, rew, _, _ = sim.step(act)
        q1, v1 = sim.get_state()[:2]
        qo, vo = b.get_state()[:2]
        assert np.allclose(q1, qo, atol=5e-5, rtol=1e-5), (k, np.abs(q1 - qo).max())
        assert np.allclose(v1, vo, atol=2e-2, rtol=5e-3), (k, np.abs(v1 - vo).max())
        worst = max(worst, np.abs(q1 - qo).max())
    assert qo[:, 2].min() < 0.1            # the toy robot did reach the floor
    sim.close()

"""CPU restatement (NumPy f64) of the reference's partially observable observation pack:
``POWalkingQuadrupedEnv`` (``src/envs/po_walking_quad.py:10-90``).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the orientation filter is ``ahrs.filters.Madgwick`` (third party,
``requirements.txt``, unpinned, not installed offline) and the module itself needs mujoco / gymnasium to import.
The filter below restates the published algorithm (S. Madgwick, "An efficient orientation filter for inertial
and inertial/magnetic sensor arrays", 2010, IMU form: eqs. 12, 13, 25, 26, 33, 34) with the library's default
IMU gain 0.033 and its conventions (no update when the gyro reads exactly zero; the gradient step is skipped
when the accelerometer reads zero), and ``Quaternion.to_angles`` as roll / pitch / yaw of the unit quaternion.
"""
from __future__ import annotations

import numpy as np

FRAME = 26        # gyro 3 + accel 3 + euler 3 + body_vel xy 2 + ctrl 12 + command vx, vy, theta 3  (po_walking_quad.py:21-27)
GAIN_IMU = 0.033


def q_prod(p, q):
    pw, px, py, pz = p
    qw, qx, qy, qz = q
    return np.array([pw * qw - px * qx - py * qy - pz * qz, pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy - px * qz + py * qw + pz * qx, pw * qz + px * qy - py * qx + pz * qw])


def madgwick_update_imu(q, gyr, acc, dt, gain=GAIN_IMU):
    """One IMU step of the filter (``updateIMU``, po_walking_quad.py:39-43)."""
    q = np.asarray(q, float)
    if not np.linalg.norm(gyr) > 0:
        return q.copy()
    qdot = 0.5 * q_prod(q, np.r_[0.0, gyr])                         # eq. 12
    a_norm = np.linalg.norm(acc)
    if a_norm > 0:
        a = np.asarray(acc, float) / a_norm
        qw, qx, qy, qz = q / np.linalg.norm(q)
        f = np.array([2.0 * (qx * qz - qw * qy) - a[0], 2.0 * (qw * qx + qy * qz) - a[1],
                      2.0 * (0.5 - qx ** 2 - qy ** 2) - a[2]])      # eq. 25
        if np.linalg.norm(f) > 0:
            J = np.array([[-2.0 * qy, 2.0 * qz, -2.0 * qw, 2.0 * qx], [2.0 * qx, 2.0 * qw, 2.0 * qz, 2.0 * qy],
                          [0.0, -4.0 * qx, -4.0 * qy, 0.0]])        # eq. 26
            g = J.T @ f                                             # eq. 34
            gn = np.linalg.norm(g)
            if gn > 0:                 # a vanishing gradient would divide 0 by 0 (the library would return NaN): no correction
                qdot = qdot - gain * g / gn                         # eq. 33
    qn = q + qdot * dt                                              # eq. 13
    return qn / np.linalg.norm(qn)


def to_angles(q):
    """roll, pitch, yaw of the (normalised) quaternion."""
    w, x, y, z = np.asarray(q, float) / np.linalg.norm(q)
    return np.array([np.arctan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y)),
                     np.arcsin(np.clip(2.0 * (w * y - z * x), -1.0, 1.0)),
                     np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))])


class POOracle:
    """Frames and stacking for n envs.  ``time`` is data.time after the step, ``sens`` the 33-value
    sensordata, ``ctrl`` data.ctrl, ``qquat`` data.qpos[3:7] (needed while the estimate still aliases it)."""

    def __init__(self, n, dt, settling_time, obs_window):
        self.n, self.dt, self.settling, self.window = n, dt, settling_time, obs_window
        self.orient = np.tile([1.0, 0, 0, 0], (n, 1))               # po_walking_quad.py:19
        self.alias = np.zeros(n, bool)        # computed_orientation IS the live view data.qpos[3:7] (:67) until an update replaces it
        self.stack = np.zeros((n, obs_window, FRAME))

    def frame(self, i, time, sens, ctrl, qquat, vel_xy, heading_xy, update=True):
        gyro, accel = sens[15:18], sens[12:15]
        q = qquat if self.alias[i] else self.orient[i]
        if update and time > self.settling / 2:                      # :37
            q = madgwick_update_imu(q, gyro, accel, self.dt)
            self.orient[i] = q
            self.alias[i] = False
        theta = np.arctan2(heading_xy[1], heading_xy[0])             # control_inputs.py:69-73
        return np.concatenate([gyro, accel, to_angles(q), sens[30:32], ctrl, vel_xy, [theta]])   # :48-56

    def reset_env(self, i, default_ctrl, qquat_prev, vel_xy_old, heading_xy_old):
        """The frame reset() returns (:59-69): zero sensors, the PREVIOUS orientation estimate, the default ctrl and
        the command of the previous episode (commands are re-sampled after the observation is taken); afterwards the
        estimate aliases data.qpos[3:7]."""
        fr = self.frame(i, 0.0, np.zeros(33), default_ctrl, qquat_prev, vel_xy_old, heading_xy_old, update=False)
        self.stack[i] = fr
        self.alias[i] = True
        return self.stack[i].reshape(-1).copy()

    def step_env(self, i, time, sens, ctrl, qquat, vel_xy, heading_xy):
        fr = self.frame(i, time, sens, ctrl, qquat, vel_xy, heading_xy)
        self.stack[i, :-1] = self.stack[i, 1:].copy()                # :80-83 FIFO
        self.stack[i, -1] = fr
        return self.stack[i].reshape(-1).copy()

"""Host-side camera pose math of the tracer boundary.

Mirrors what `Tracer.__create_camera_parameters` + `SensorPose3DModel.get_sensor_pose` do in the
reference (threedgut_tracer/tracer.py:60-151, 373-383): camera-to-world 4x4 -> world-to-sensor
translation + unit quaternion in XYZW order, float32, start pose == end pose, timestamps [0, 1].
Pinned by tests/golden/pose_golden.npz (generated from the reference's own Python).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class SensorPose3D:
    T_world_sensors: list  # two [t(3), q(xyzw)] float32 arrays (start, end)
    timestamps_us: list


def _world_to_view(c2w34):
    # the reference inverts C2W in float64, rebuilds C2W and inverts again (tracer.py:75-85, 376-383)
    c2w = np.concatenate((np.asarray(c2w34, np.float64)[:3, :4], np.zeros((1, 4))))
    c2w[3, 3] = 1.0
    w2c = np.linalg.inv(c2w)
    rt = np.zeros((4, 4))
    rt[:3, :3] = w2c[:3, :3]
    rt[:3, 3] = w2c[:3, 3]
    rt[3, 3] = 1.0
    rt = np.linalg.inv(np.linalg.inv(rt))
    return np.float32(rt)


def so3_matrix_to_quat_xyzw(R):
    """Largest-diagonal/trace branch selection as tracer.py:88-136, float32 arithmetic."""
    R = np.asarray(R, np.float32)
    dec = np.array([R[0, 0], R[1, 1], R[2, 2], np.float32(0)], np.float32)
    dec[3] = dec[:3].sum(dtype=np.float32)
    choice = int(np.argmax(dec))
    q = np.empty(4, np.float32)
    if choice != 3:
        i = choice
        j = (i + 1) % 3
        k = (j + 1) % 3
        q[i] = np.float32(1) - dec[3] + np.float32(2) * R[i, i]
        q[j] = R[j, i] + R[i, j]
        q[k] = R[k, i] + R[i, k]
        q[3] = R[k, j] - R[j, k]
    else:
        q[0] = R[2, 1] - R[1, 2]
        q[1] = R[0, 2] - R[2, 0]
        q[2] = R[1, 0] - R[0, 1]
        q[3] = np.float32(1) + dec[3]
    return q / np.float32(np.sqrt((q * q).sum(dtype=np.float32)))


def sensor_pose_from_c2w(T_to_world) -> SensorPose3D:
    rt = _world_to_view(np.asarray(T_to_world).reshape(4, 4))
    tq = np.concatenate([rt[:3, 3], so3_matrix_to_quat_xyzw(rt[:3, :3])]).astype(np.float32)
    return SensorPose3D(T_world_sensors=[tq, tq.copy()], timestamps_us=[0, 1])

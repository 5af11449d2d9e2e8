"""Generate golden vectors for the host-side pose math of the 3DGUT tracer boundary.

Runs ONLY in the build container (needs /root/reference).  It imports the reference's
`threedgut_tracer/tracer.py` by file path with `omegaconf` and the `threedgrut.datasets`
package shell stubbed in sys.modules (they are not installed here and are not used by the
class under test), then drives `SensorPose3DModel(R, T).get_sensor_pose()` exactly the way
`Tracer.__create_camera_parameters` does (tracer.py:373-383) on seeded random camera-to-world
matrices.  Output: tests/golden/pose_golden.npz  (inputs C2W [K,4,4] f32, outputs tquat [K,7] f32).

The fixture is data (inputs + expected outputs); no reference source is stored.
"""
import importlib.util
import sys
import types
import os

import numpy as np

REF = "/root/reference/threedgut_tracer/tracer.py"


def _load_reference_tracer():
    om = types.ModuleType("omegaconf")
    om.OmegaConf = type("OmegaConf", (), {})
    sys.modules.setdefault("omegaconf", om)
    for name in ("threedgrut", "threedgrut.datasets", "threedgrut.datasets.protocols"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["threedgrut.datasets.protocols"].Batch = object
    spec = importlib.util.spec_from_file_location("_ref_tracer", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def random_c2w(rng, k):
    out = np.zeros((k, 4, 4), dtype=np.float32)
    for i in range(k):
        a = rng.normal(size=(3, 3))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        out[i, :3, :3] = q.astype(np.float32)
        out[i, :3, 3] = rng.uniform(-5, 5, size=3).astype(np.float32)
        out[i, 3, 3] = 1.0
    # hand-picked cases that exercise every branch of the matrix->quaternion conversion
    special = []
    for axis, ang in (((1, 0, 0), np.pi), ((0, 1, 0), np.pi), ((0, 0, 1), np.pi),
                      ((1, 0, 0), 3.0), ((0, 1, 0), 3.0), ((0, 0, 1), 3.0), ((1, 1, 1), 0.0)):
        ax = np.asarray(axis, dtype=np.float64)
        ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = R.astype(np.float32)
        m[:3, 3] = (1.0, -2.0, 3.0)
        special.append(m)
    return np.concatenate([out, np.stack(special)], axis=0)


def main():
    mod = _load_reference_tracer()
    rng = np.random.default_rng(1234)
    c2w = random_c2w(rng, 25)
    res = []
    for m in c2w:
        C2W = np.concatenate((m[:3, :4], np.zeros((1, 4))))
        C2W[3, 3] = 1.0
        W2C = np.linalg.inv(C2W)
        R = np.transpose(W2C[:3, :3])
        T = W2C[:3, 3]
        pose = mod.SensorPose3DModel(R=R, T=T).get_sensor_pose()
        assert pose.timestamps_us == [0, 1]
        a, b = pose.T_world_sensors
        assert (a == b).all()
        res.append(a.numpy().astype(np.float32))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_golden.npz")
    np.savez(out, c2w=c2w, tquat=np.stack(res))
    print("wrote", out, np.stack(res).shape)


if __name__ == "__main__":
    main()

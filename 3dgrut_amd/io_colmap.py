"""COLMAP scene input for the 3DGUT path ("next" row N4 of SURVEY §8f): sparse-model readers, the per-view `Batch`
(camera-space rays + intrinsics dictionary + pose) and the initial Gaussians from the SfM points.

What it mirrors (behaviour, not code):
  * threedgrut/datasets/utils.py:258-566 — `cameras.{bin,txt}`, `images.{bin,txt}`, `points3D.{bin,txt}` of COLMAP's
    sparse model (the binary layout is COLMAP's documented one: little-endian, `uint64` counts, `double` parameters);
  * threedgrut/datasets/dataset_colmap.py:104-283 — SIMPLE_PINHOLE / PINHOLE / OPENCV_FISHEYE only (undistorted
    datasets), intrinsics divided by the down-sampling factor, principal point at the image centre for pinhole
    cameras, fisheye `max_angle` from the farthest image corner, `C2W = inv([R|t])`, every `test_split_interval`-th
    frame held out for testing, `cameras_extent = 1.1 * max ||centre_i - mean centre||`;
  * threedgrut/model/model.py:207-248, 438-483 — Gaussians from the SfM points: scale = 0.01 * distance to the nearest
    camera (`use_observation_points`) or the RMS distance to the 3 nearest points, density 0.1, SH dc from the point
    colour, random rotations.
Images are optional (there are none in the build environment): `batch()` attaches `rgb_gt` only if the file exists.
"""
import os
import struct
from dataclasses import dataclass

import numpy as np
import torch

from . import cameras as cams
from .protocols import Batch

# COLMAP camera models: id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_PARAMS = {name: n for name, n in CAMERA_MODELS.values()}


@dataclass
class ColmapCamera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray


@dataclass
class ColmapImage:
    id: int
    qvec: np.ndarray   # (w, x, y, z), world -> camera
    tvec: np.ndarray
    camera_id: int
    name: str


def _unpack(f, fmt):
    size = struct.calcsize("<" + fmt)
    data = f.read(size)
    if len(data) != size:
        raise EOFError("truncated COLMAP file")
    return struct.unpack("<" + fmt, data)


def read_cameras_binary(path):
    out = {}
    with open(path, "rb") as f:
        (n,) = _unpack(f, "Q")
        for _ in range(n):
            cam_id, model_id, w, h = _unpack(f, "iiQQ")
            name, npar = CAMERA_MODELS[model_id]
            out[cam_id] = ColmapCamera(cam_id, name, int(w), int(h), np.array(_unpack(f, "d" * npar), np.float64))
    return out


def read_images_binary(path):
    out = []
    with open(path, "rb") as f:
        (n,) = _unpack(f, "Q")
        for _ in range(n):
            vals = _unpack(f, "idddddddi")
            name = bytearray()
            while True:
                c = f.read(1)
                if c in (b"\x00", b""):
                    break
                name += c
            (n2d,) = _unpack(f, "Q")
            f.seek(24 * n2d, os.SEEK_CUR)  # (x, y, point3D_id) per observation: not needed
            out.append(ColmapImage(vals[0], np.array(vals[1:5], np.float64), np.array(vals[5:8], np.float64), vals[8],
                                   name.decode("utf-8")))
    # sorted by image name, as the reference's readers return them (datasets/utils.py:500,565): the every-n-th
    # train/test split indexes this order (pinned by tests/golden/host_golden.json)
    return sorted(out, key=lambda im: im.name)


def read_points3D_binary(path):
    with open(path, "rb") as f:
        (n,) = _unpack(f, "Q")
        xyz = np.zeros((n, 3), np.float64)
        rgb = np.zeros((n, 3), np.uint8)
        err = np.zeros(n, np.float64)
        for i in range(n):
            v = _unpack(f, "QdddBBBd")
            xyz[i], rgb[i], err[i] = v[1:4], v[4:7], v[7]
            (track,) = _unpack(f, "Q")
            f.seek(8 * track, os.SEEK_CUR)
    return xyz, rgb, err


def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line


def read_cameras_text(path):
    out = {}
    for line in _data_lines(path):
        t = line.split()
        out[int(t[0])] = ColmapCamera(int(t[0]), t[1], int(t[2]), int(t[3]), np.array([float(x) for x in t[4:]], np.float64))
    return out


def read_images_text(path):
    out = []
    lines = list(_data_lines_keep_pairs(path))
    for head in lines:
        t = head.split()
        out.append(ColmapImage(int(t[0]), np.array([float(x) for x in t[1:5]]), np.array([float(x) for x in t[5:8]]), int(t[8]),
                               " ".join(t[9:])))
    return sorted(out, key=lambda im: im.name)


def _data_lines_keep_pairs(path):
    """images.txt has two lines per image (header, 2-D points; the second may be empty): yield the headers."""
    with open(path, "r") as f:
        rows = [ln.rstrip("\n") for ln in f if not ln.startswith("#")]
    while rows and not rows[-1].strip():
        rows.pop()
    for k in range(0, len(rows), 2):
        if rows[k].strip():
            yield rows[k].strip()


def read_points3D_text(path):
    xyz, rgb, err = [], [], []
    for line in _data_lines(path):
        t = line.split()
        xyz.append([float(x) for x in t[1:4]]); rgb.append([int(x) for x in t[4:7]]); err.append(float(t[7]))
    return np.array(xyz, np.float64).reshape(-1, 3), np.array(rgb, np.uint8).reshape(-1, 3), np.array(err, np.float64)


def write_model_binary(sparse_dir, cameras, images, xyz, rgb):
    """Writer for the three sparse-model files (tests and synthetic round trips)."""
    os.makedirs(sparse_dir, exist_ok=True)
    ids = {name: k for k, (name, _) in CAMERA_MODELS.items()}
    with open(os.path.join(sparse_dir, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras.values():
            f.write(struct.pack("<iiQQ", c.id, ids[c.model], c.width, c.height))
            f.write(struct.pack("<" + "d" * len(c.params), *[float(x) for x in c.params]))
    with open(os.path.join(sparse_dir, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images:
            f.write(struct.pack("<idddddddi", im.id, *[float(x) for x in im.qvec], *[float(x) for x in im.tvec], im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", 0))
    with open(os.path.join(sparse_dir, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(xyz)))
        for i in range(len(xyz)):
            f.write(struct.pack("<QdddBBBd", i + 1, *[float(x) for x in xyz[i]], *[int(x) for x in rgb[i]], 0.5))
            f.write(struct.pack("<Q", 0))


def qvec_to_rotation(q):
    w, x, y, z = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], np.float64)


def rotation_to_qvec(R):
    """(w, x, y, z) of a rotation matrix (used by the synthetic-model writer of the tests)."""
    t = np.trace(R)
    if t > 0:
        s = 2.0 * np.sqrt(1.0 + t)
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


class ColmapScene:
    """One split of a COLMAP scene directory (`<root>/sparse/0/*`, images in `<root>/images[_<factor>]/`)."""

    def __init__(self, root, split="train", downsample_factor=1, test_split_interval=8):
        self.root, self.split, self.downsample_factor = root, split, int(downsample_factor)
        sparse = os.path.join(root, "sparse", "0")
        if os.path.isfile(os.path.join(sparse, "images.bin")):
            images = read_images_binary(os.path.join(sparse, "images.bin"))
            self.cameras = read_cameras_binary(os.path.join(sparse, "cameras.bin"))
        else:
            images = read_images_text(os.path.join(sparse, "images.txt"))
            self.cameras = read_cameras_text(os.path.join(sparse, "cameras.txt"))
        for c in self.cameras.values():
            if c.model not in ("SIMPLE_PINHOLE", "PINHOLE", "OPENCV_FISHEYE"):
                raise ValueError(f"Colmap camera model '{c.model}' not handled: only undistorted datasets "
                                 "(PINHOLE, SIMPLE_PINHOLE or OPENCV_FISHEYE cameras) are supported")
        idx = np.arange(len(images))
        if test_split_interval > 0:
            keep = (idx % test_split_interval != 0) if split == "train" else (idx % test_split_interval == 0)
            images = [images[i] for i in idx[keep]]
        self.images = images
        poses = []
        for im in images:
            w2c = np.eye(4)
            w2c[:3, :3] = qvec_to_rotation(im.qvec)
            w2c[:3, 3] = im.tvec
            poses.append(np.linalg.inv(w2c))
        self.poses = np.stack(poses).astype(np.float32) if poses else np.zeros((0, 4, 4), np.float32)
        self.camera_centers = self.poses[:, :3, 3].astype(np.float64)
        centre = self.camera_centers.mean(0) if len(images) else np.zeros(3)
        self.center = centre
        dist = np.linalg.norm(self.camera_centers - centre, axis=1) if len(images) else np.zeros(1)
        self.cameras_extent = float(dist.max() * 1.1)    # trainer's scene extent (learning-rate scale of the positions)
        self.length_scale = float(dist.mean())
        self._ray_cache = {}

    def __len__(self):
        return len(self.images)

    def images_folder(self):
        return "images" if self.downsample_factor == 1 else f"images_{self.downsample_factor}"

    def resolution(self, cam: ColmapCamera):
        f = self.downsample_factor
        return int(round(cam.width / f)), int(round(cam.height / f))

    def camera_rays(self, camera_id):
        """(rays_ori, rays_dir [1,H,W,3] float32, intrinsics key, intrinsics dict) of one COLMAP camera."""
        if camera_id in self._ray_cache:
            return self._ray_cache[camera_id]
        cam = self.cameras[camera_id]
        W, H = self.resolution(cam)
        s = cam.height / H  # the factor the intrinsics are divided by (dataset_colmap.py:213-231)
        if cam.model == "SIMPLE_PINHOLE":
            fx = fy = cam.params[0] / s
        elif cam.model == "PINHOLE":
            fx, fy = cam.params[0] / s, cam.params[1] / s
        if cam.model in ("SIMPLE_PINHOLE", "PINHOLE"):
            ro, rd = cams.pinhole_rays(W, H, fx, fy)
            out = (ro, rd, "intrinsics_OpenCVPinholeCameraModelParameters", cams.pinhole_intrinsics_dict(W, H, fx, fy))
        else:
            fx, fy, cx, cy = [p / s for p in cam.params[:4]]
            radial = cam.params[4:8]
            ro, rd = cams.fisheye_rays(W, H, fx, fy, cx, cy, radial=radial)
            K = cams.fisheye_intrinsics_dict(W, H, fx, fy, cx, cy)
            K["radial_coeffs"] = np.asarray(radial, np.float32)
            out = (ro, rd, "intrinsics_OpenCVFisheyeCameraModelParameters", K)
        self._ray_cache[camera_id] = out
        return out

    def batch(self, i, device="cuda", pose_on_host=True):
        im = self.images[i]
        ro, rd, key, K = self.camera_rays(im.camera_id)
        pose = torch.as_tensor(self.poses[i])[None]
        kw = {key: K}
        rgb = self.load_image(i)
        return Batch(rays_ori=torch.as_tensor(ro, device=device), rays_dir=torch.as_tensor(rd, device=device),
                     T_to_world=pose if pose_on_host else pose.to(device),
                     rgb_gt=None if rgb is None else torch.as_tensor(rgb, device=device)[None], **kw)

    def load_image(self, i):
        path = os.path.join(self.root, self.images_folder(), self.images[i].name)
        if not os.path.isfile(path):
            return None
        from PIL import Image
        with Image.open(path) as img:
            return np.asarray(img.convert("RGB"), np.float32) / 255.0

    def points(self):
        sparse = os.path.join(self.root, "sparse", "0")
        if os.path.isfile(os.path.join(sparse, "points3D.bin")):
            xyz, rgb, _ = read_points3D_binary(os.path.join(sparse, "points3D.bin"))
        else:
            xyz, rgb, _ = read_points3D_text(os.path.join(sparse, "points3D.txt"))
        return xyz.astype(np.float32), rgb

    def initial_gaussians(self, use_observation_points=True, observation_scale_factor=0.01, default_density=0.1,
                          default_scale_factor=1.0, seed=0):
        """Scene dictionary (activated parameters, layout of scenes.py) from the SfM points."""
        from scipy.spatial import cKDTree
        xyz, rgb = self.points()
        n = xyz.shape[0]
        if use_observation_points:
            d, _ = cKDTree(self.camera_centers).query(xyz.astype(np.float64), k=1)
            scale = np.maximum(d, 1e-7) * observation_scale_factor
        else:
            d, _ = cKDTree(xyz).query(xyz, k=min(4, n))
            scale = np.sqrt((d[:, 1:] ** 2).mean(-1))
        scale = np.maximum(scale * default_scale_factor, 1e-9)
        rng = np.random.default_rng(seed)
        q = rng.uniform(0.0, 1.0, size=(n, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        feats = np.zeros((n, 48), np.float32)
        feats[:, 0:3] = (rgb.astype(np.float32) / 255.0 - 0.5) / 0.28209479177387814   # RGB2SH
        return dict(positions=xyz.astype(np.float32), rotation=q.astype(np.float32),
                    scale=np.repeat(scale[:, None], 3, 1).astype(np.float32),
                    density=np.full((n, 1), default_density, np.float32), features=feats)

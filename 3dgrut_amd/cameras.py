"""Synthetic cameras for the 3DGUT path: poses, camera-space rays and intrinsics dictionaries in the
formats the reference's datasets hand to the tracer (threedgrut/datasets/protocols.py:23-34).

Ray generation follows threedgrut/datasets/utils.py:39-59 (pinhole) and
threedgrut/datasets/camera_models.py:156-235 (OpenCV fisheye, zero radial coefficients as the
ScanNet++ loader uses, dataset_scannetpp.py:43-45; max_angle rule dataset_colmap.py:167-172).
Camera convention: "right-down-front" (+x right, +y down, +z forward), protocols.py:79-86.
"""
import enum
import math

import numpy as np


@enum.unique
class ShutterType(enum.IntEnum):
    """The DATASET-side shutter enum (threedgrut/datasets/camera_models.py:29-36: `IntEnum` with `auto()`, i.e. 1..5).
    This is what the reference's `*CameraModelParameters.to_dict()` puts under "shutter_type"; the tracer maps it by
    name to the plugin enum 0..4 (threedgut_tracer/tracer.py:365-371)."""
    ROLLING_TOP_TO_BOTTOM = 1
    ROLLING_LEFT_TO_RIGHT = 2
    ROLLING_BOTTOM_TO_TOP = 3
    ROLLING_RIGHT_TO_LEFT = 4
    GLOBAL = 5


def look_at_c2w(eye, target, up=(0.0, -1.0, 0.0)):
    """Camera-to-world 4x4 (float32) for a right-down-front camera at `eye` looking at `target`.
    `up` is the world direction that should appear up in the image."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    upv = np.asarray(up, np.float64)
    r = np.cross(f, upv)
    if np.linalg.norm(r) < 1e-8:
        r = np.cross(f, np.array([1.0, 0.0, 0.0]))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)  # down
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, d, f, eye
    return m.astype(np.float32)


def pinhole_rays(W, H, fx, fy):
    """(rays_ori, rays_dir) float32 [1,H,W,3]; pixel centres, principal point at the image centre."""
    x, y = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="xy")
    xs = ((x + 0.5) - 0.5 * W) / fx
    ys = ((y + 0.5) - 0.5 * H) / fy
    d = np.stack((xs, ys, np.ones_like(xs)), axis=-1)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    return np.zeros((1, H, W, 3), np.float32), d.astype(np.float32)[None]


def pinhole_intrinsics_dict(W, H, fx, fy, cx=None, cy=None):
    return dict(
        resolution=np.array([W, H], np.int64), shutter_type=ShutterType.GLOBAL,
        principal_point=np.array([W / 2 if cx is None else cx, H / 2 if cy is None else cy], np.float32),
        focal_length=np.array([fx, fy], np.float32), radial_coeffs=np.zeros(6, np.float32),
        tangential_coeffs=np.zeros(2, np.float32), thin_prism_coeffs=np.zeros(4, np.float32),
    )


def fisheye_max_angle(W, H, fx, fy, cx, cy):
    mx = max(cx, W - cx)
    my = max(cy, H - cy)
    max_radius = math.sqrt(mx * mx + my * my)
    return max(2.0 * max_radius / fx, 2.0 * max_radius / fy) / 2.0


def fisheye_rays(W, H, fx, fy, cx=None, cy=None, radial=None, newton_iterations=3):
    """OpenCV fisheye rays.  radial = (k1..k4) of the forward polynomial delta = theta (1 + k1 theta^2 + ... + k4 theta^8);
    it is inverted per pixel with Newton steps from the linear (equidistant) initial guess, as
    camera_models.py:156-235 does (3 iterations).  With zero coefficients theta = delta exactly."""
    cx = W / 2 if cx is None else cx
    cy = H / 2 if cy is None else cy
    x, y = np.meshgrid(np.arange(W, dtype=np.float64) + 0.5, np.arange(H, dtype=np.float64) + 0.5, indexing="xy")
    nx, ny = (x - cx) / fx, (y - cy) / fy
    delta = np.sqrt(nx * nx + ny * ny)
    theta = delta
    if radial is not None and np.any(np.asarray(radial, np.float64) != 0.0):
        k1, k2, k3, k4 = [float(k) for k in np.asarray(radial, np.float64)[:4]]
        max_angle = fisheye_max_angle(W, H, fx, fy, cx, cy)
        max_norm_dist = max(W / 2 / fx, H / 2 / fy)
        theta = delta * (max_angle / max_norm_dist)
        for _ in range(newton_iterations):
            t2 = theta * theta
            f = theta * (1.0 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4)))) - delta
            df = 1.0 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)))
            theta = theta - f / df
    s = np.sin(theta) / np.maximum(delta, 1e-6)
    d = np.stack((s * nx, s * ny, np.cos(theta)), axis=-1)
    d[delta < 1e-6] = (0.0, 0.0, 1.0)
    return np.zeros((1, H, W, 3), np.float32), d.astype(np.float32)[None]


def fisheye_intrinsics_dict(W, H, fx, fy, cx=None, cy=None, radial=None, max_angle=None):
    """radial: (k1..k4) of the forward polynomial (zeros when None, as the ScanNet++ loader passes them); max_angle: the
    dataset rule (dataset_colmap.py:167-172) when None."""
    cx = W / 2 if cx is None else cx
    cy = H / 2 if cy is None else cy
    return dict(
        resolution=np.array([W, H], np.int64), shutter_type=ShutterType.GLOBAL,
        principal_point=np.array([cx, cy], np.float32), focal_length=np.array([fx, fy], np.float32),
        radial_coeffs=np.zeros(4, np.float32) if radial is None else np.asarray(radial, np.float32)[:4].copy(),
        max_angle=float(fisheye_max_angle(W, H, fx, fy, cx, cy) if max_angle is None else max_angle),
    )


def orbit_c2w(radius, azimuth_deg, elevation_deg, target=(0.0, 0.0, 0.0)):
    """Camera on a sphere around `target`; world up is +z... here -y is up (right-down-front world)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    eye = np.array([radius * math.cos(el) * math.sin(az), -radius * math.sin(el), -radius * math.cos(el) * math.cos(az)])
    return look_at_c2w(eye + np.asarray(target), target)

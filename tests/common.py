"""Shared builders for the parity tests: seeded scenes + cameras in both forms (oracle dict / tracer Batch)."""
import importlib
import math

import numpy as np

cams = importlib.import_module("3dgrut_amd.cameras")
scenes = importlib.import_module("3dgrut_amd.scenes")
pose = importlib.import_module("3dgrut_amd.pose")


def make_view(kind, W, H, c2w, fx=None, fy=None, distortion=None):
    """Returns dict(W,H,c2w,ro,rd,oracle_cam,intrinsics_kw) for kind in {'pinhole','pinhole_list','fisheye'}."""
    tq = pose.sensor_pose_from_c2w(c2w).T_world_sensors[0]
    if kind in ("pinhole", "pinhole_list"):
        fx = fx or 1.0 * W
        fy = fy or fx
        ro, rd = cams.pinhole_rays(W, H, fx, fy)
        K = cams.pinhole_intrinsics_dict(W, H, fx, fy)
        if distortion:
            K["radial_coeffs"] = np.asarray(distortion["radial"], np.float32)
            K["tangential_coeffs"] = np.asarray(distortion["tangential"], np.float32)
            K["thin_prism_coeffs"] = np.asarray(distortion["thin_prism"], np.float32)
        ocam = dict(model="pinhole", principal_point=K["principal_point"], focal_length=K["focal_length"],
                    radial=K["radial_coeffs"], tangential=K["tangential_coeffs"], thin_prism=K["thin_prism_coeffs"],
                    pose_start=tq)
        if kind == "pinhole_list":
            # [fx,fy,cx,cy] path: focal -> fov -> focal round trip and orig_w=int(2cx) (tracer.py:386-403)
            w2, h2 = int(2 * (W / 2)), int(2 * (H / 2))
            fx2 = w2 / (2.0 * math.tan(0.5 * (2 * math.atan(w2 / (2 * fx)))))
            fy2 = h2 / (2.0 * math.tan(0.5 * (2 * math.atan(h2 / (2 * fy)))))
            ocam["focal_length"] = np.array([fx2, fy2], np.float32)
            ocam["principal_point"] = np.array([w2, h2], np.float32) / 2
            kw = dict(intrinsics=[fx, fy, W / 2, H / 2])
        else:
            kw = dict(intrinsics_OpenCVPinholeCameraModelParameters=K)
    elif kind == "fisheye":
        fx = fx or 0.45 * W
        fy = fy or fx
        ro, rd = cams.fisheye_rays(W, H, fx, fy)
        K = cams.fisheye_intrinsics_dict(W, H, fx, fy)
        ocam = dict(model="fisheye", principal_point=K["principal_point"], focal_length=K["focal_length"],
                    radial=list(K["radial_coeffs"]), max_angle=K["max_angle"], pose_start=tq)
        kw = dict(intrinsics_OpenCVFisheyeCameraModelParameters=K)
    else:
        raise ValueError(kind)
    return dict(W=W, H=H, c2w=c2w, ro=ro, rd=rd, oracle_cam=ocam, intrinsics_kw=kw, tq=tq)


def to_batch(view, device):
    import torch
    gut = importlib.import_module("3dgrut_amd")
    return gut.Batch(rays_ori=torch.as_tensor(view["ro"], device=device), rays_dir=torch.as_tensor(view["rd"], device=device),
                     T_to_world=torch.as_tensor(view["c2w"], device=device)[None], **view["intrinsics_kw"])


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


# A colour difference above the fp32 tolerance is only acceptable where the oracle itself says a hit/no-hit (or early
# termination) decision along that ray sat within this many fp32-noise widths of its threshold (oracle_render_margins):
FLIP_MARGIN_BOUND = 4.0


def check_colour_outliers(rgba_gpu, hits_gpu, ref, margins, tol=2e-4, bound=FLIP_MARGIN_BOUND, label="", max_prone=0.05):
    """Earns the "threshold flip" allowance instead of asserting it: every pixel whose colour differs from the oracle by
    more than `tol`, or whose hit count differs, must be flip-prone (decision margin < bound); every pixel that is not
    flip-prone must be within `tol` and have the oracle's hit count.  Returns a small report (fractions, worst margin)."""
    H, W = ref["rgba"].shape[:2]
    diff = np.abs(np.asarray(rgba_gpu).reshape(H, W, 4) - ref["rgba"]).max(-1)
    hdiff = np.asarray(hits_gpu).reshape(H, W) != ref["hits"].reshape(H, W)
    m = margins.min(-1)
    prone = m < bound
    out = (diff > tol) | hdiff
    rep = dict(outliers=int(out.sum()), flip_prone=int(prone.sum()), pixels=int(H * W), max_diff=float(diff.max()),
               max_diff_not_prone=float(diff[~prone].max()) if (~prone).any() else 0.0,
               worst_outlier_margin=float(m[out].max()) if out.any() else 0.0)
    print(f"[outliers {label}] {rep}")
    assert not (out & ~prone).any(), f"{label}: {(out & ~prone).sum()} pixels differ although no decision is near a threshold: {rep}"
    assert diff.max() <= 2.5e-2, rep                  # a flipped hit moves a pixel by at most ~alpha*T*colour of that hit
    assert prone.mean() <= max_prone, rep             # the allowance stays a small part of the image
    assert out.mean() <= 2e-3, rep
    return rep

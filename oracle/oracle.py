"""ctypes front-end of the C oracle (oracle/gut_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  "parity unpinned" for the device kernels (the
reference ships no fixtures for this path); see gut_oracle.c header.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleParams(C.Structure):
    _fields_ = [
        ("alpha_threshold", C.c_float), ("max_alpha", C.c_float), ("min_kernel_density", C.c_float),
        ("min_transmittance", C.c_float), ("min_sensor_z", C.c_float), ("cov_dilation", C.c_float),
        ("ut_alpha", C.c_float), ("ut_beta", C.c_float), ("ut_kappa", C.c_float), ("ut_margin", C.c_float),
        ("rect_bounding", C.c_int32), ("tight_opacity_bounding", C.c_int32), ("tile_culling", C.c_int32),
        ("global_z_order", C.c_int32),
        ("kernel_degree", C.c_int32), ("rolling_shutter_iterations", C.c_int32), ("enable_hitcounts", C.c_int32),
    ]


class OracleCamera(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("shutter", C.c_int32),
        ("principal_point", C.c_float * 2), ("focal_length", C.c_float * 2),
        ("radial", C.c_float * 6), ("tangential", C.c_float * 2), ("thin_prism", C.c_float * 4),
        ("max_angle", C.c_float), ("pose_start", C.c_float * 7), ("pose_end", C.c_float * 7),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libgut_oracle.so")
    src = os.path.join(_HERE, "gut_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgut_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_scan.restype = C.c_uint32
        _LIB.oracle_higher_msb.restype = C.c_uint32
        _LIB.oracle_det_logf.restype = C.c_float
        _LIB.oracle_det_logf.argtypes = [C.c_float]
        _LIB.oracle_det_atan2f.restype = C.c_float
        _LIB.oracle_det_atan2f.argtypes = [C.c_float, C.c_float]
        _LIB.oracle_tile_min_power.restype = C.c_float
        _LIB.oracle_tile_min_power.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    return _LIB


class variant:
    """`with oracle.variant(1): ...` — everything inside evaluates the per-(ray, Gaussian) formulas of the compositing passes in the
    second fp32 form (gut_oracle.c: eval_hit_fused: FMA, pre-scaled rows, one reciprocal, exp2).  Projection and binning are
    unaffected (they carry the bit-exact contract).  Tests only: the measured basis of the GPU tolerances."""

    def __init__(self, v):
        self.v = int(v)

    def __enter__(self):
        self.prev = int(lib().oracle_get_variant())
        lib().oracle_set_variant(C.c_int(self.v))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_variant(C.c_int(self.prev))
        return False


def default_params():
    p = OracleParams()
    lib().oracle_default_params(C.byref(p))
    return p


def make_camera(cam: dict) -> OracleCamera:
    """cam: dict with keys model('pinhole'|'fisheye'), principal_point, focal_length, radial, tangential,
    thin_prism, max_angle, pose_start[7], pose_end[7] (world->sensor t, q xyzw)."""
    c = OracleCamera()
    c.model = 0 if cam["model"] == "pinhole" else 1
    c.shutter = int(cam.get("shutter", 4))
    c.principal_point[:] = [float(v) for v in cam["principal_point"]]
    c.focal_length[:] = [float(v) for v in cam["focal_length"]]
    rad = list(cam.get("radial", [])) + [0.0] * 6
    c.radial[:] = [float(v) for v in rad[:6]]
    c.tangential[:] = [float(v) for v in cam.get("tangential", [0.0, 0.0])]
    c.thin_prism[:] = [float(v) for v in cam.get("thin_prism", [0.0] * 4)]
    c.max_angle = float(cam.get("max_angle", 0.0))
    c.pose_start[:] = [float(v) for v in cam["pose_start"]]
    c.pose_end[:] = [float(v) for v in cam.get("pose_end", cam["pose_start"])]
    return c


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def higher_msb(n):
    return int(lib().oracle_higher_msb(C.c_uint32(n)))


def pose_matrices(cam: dict):
    c = make_camera(cam)
    a = np.zeros((3, 4), np.float32); b = np.zeros((3, 4), np.float32); d = np.zeros((3, 4), np.float32)
    lib().oracle_pose_matrices(C.byref(c), _p(a), _p(b), _p(d))
    return a, b, d


def forward(cam: dict, W, H, density12, sph48, ray_ori, ray_dir, sh_degree=3, params=None, threads=None):
    """Full forward. Returns dict of every intermediate buffer of SURVEY §8a plus outputs."""
    L = lib()
    prm = params or default_params()
    c = make_camera(cam)
    d12 = _f32(density12); sph = _f32(sph48)
    ro = _f32(ray_ori).reshape(-1, 3); rd = _f32(ray_dir).reshape(-1, 3)
    N = d12.shape[0]
    assert d12.shape == (N, 12) and sph.shape == (N, 48) and ro.shape[0] == W * H
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    out = dict(
        tiles_count=np.zeros(N, np.uint32), proj_pos=np.zeros((N, 2), np.float32),
        conic_opacity=np.zeros((N, 4), np.float32), extent=np.zeros((N, 2), np.float32),
        depth=np.zeros(N, np.float32), feat=np.zeros((N, 3), np.float32), visibility=np.zeros(N, np.int32),
    )
    L.oracle_project(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), C.c_uint32(N), C.c_int(sh_degree),
                     _p(d12), _p(sph), _p(out["tiles_count"]), _p(out["proj_pos"]), _p(out["conic_opacity"]),
                     _p(out["extent"]), _p(out["depth"]), _p(out["feat"]), _p(out["visibility"]))
    out["tiles_offset"] = np.zeros(N, np.uint32)
    M = int(L.oracle_scan(C.c_uint32(N), _p(out["tiles_count"]), _p(out["tiles_offset"]))) if N else 0
    out["M"] = M
    keys = np.zeros(M, np.uint64); ids = np.zeros(M, np.uint32)
    if M:
        L.oracle_expand(C.byref(prm), C.c_int(W), C.c_int(H), C.c_uint32(N), _p(out["tiles_offset"]),
                        _p(out["proj_pos"]), _p(out["conic_opacity"]), _p(out["extent"]), _p(out["depth"]),
                        _p(keys), _p(ids))
    out["unsorted_keys"], out["unsorted_ids"] = keys, ids
    skeys = np.zeros(M, np.uint64); sids = np.zeros(M, np.uint32)
    end_bit = 32 + higher_msb(T)
    out["end_bit"] = end_bit
    if M:
        L.oracle_sort_pairs(C.c_uint32(M), C.c_int(end_bit), _p(keys), _p(ids), _p(skeys), _p(sids))
    out["sorted_keys"], out["sorted_ids"] = skeys, sids
    ranges = np.zeros((T, 2), np.uint32)
    if M:
        L.oracle_tile_ranges(C.c_uint32(M), C.c_uint32(T), _p(skeys), _p(ranges))
    out["tile_ranges"] = ranges
    # output initial values as allocated by the reference (splatRaster.cpp:196-199)
    rgba = np.zeros((H, W, 4), np.float32); dist = np.full((H, W, 1), 1e6, np.float32)
    hits = np.zeros((H, W, 1), np.float32)
    trav = C.c_uint64(0)
    tile_trav = np.zeros(T, np.uint32)
    if M:  # zero intersections => early return with untouched outputs (gutRenderer.cu:323-325)
        L.oracle_set_tile_traversed_out(_p(tile_trav))
        L.oracle_render(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), _p(d12), _p(out["feat"]),
                        _p(ro), _p(rd), _p(ranges), _p(sids), _p(rgba), _p(dist), _p(hits), C.byref(trav))
        L.oracle_set_tile_traversed_out(None)
    out.update(rgba=rgba, dist=dist, hits=hits, traversed_fwd=int(trav.value), tile_traversed_fwd=tile_trav,
               _inputs=(d12, sph, ro, rd, sh_degree, W, H))
    return out


def debug_ray(cam: dict, fwd: dict, px, py, params=None, max_entries=4096):
    """[n,8] float64: id, d2, response, alpha, noise estimate nu, accepted, T after, |gro| of the entries pixel (px, py) walks."""
    L = lib()
    prm = params or default_params()
    c = make_camera(cam)
    d12, sph, ro, rd, sh_degree, W, H = fwd["_inputs"]
    out = np.zeros((max_entries, 8), np.float64)
    L.oracle_debug_ray.restype = C.c_int
    n = L.oracle_debug_ray(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), _p(d12), _p(ro), _p(rd), _p(fwd["tile_ranges"]), _p(fwd["sorted_ids"]),
                           C.c_int(int(px)), C.c_int(int(py)), _p(out), C.c_int(max_entries))
    return out[:n]


def render_margins(cam: dict, fwd: dict, params=None, budget_bound=None):
    """budget_bound (e.g. 12.0): returns (margins, budget) with budget [H,W,2] float32 = the per-pixel flip budget (how far decisions
    within that many noise widths of a threshold may move the pixel's colour; by how many hits its count may differ), see gut_oracle.c.
    Per-pixel decision margins of a forward() result's compositing (oracle_render_margins): [H,W,2] float32,
    channel 0 = hit/no-hit thresholds (min_response, min_alpha), channel 1 = early termination (min_transmittance),
    both in units of the estimated fp32 noise of the compared quantity (< ~1: a different but equally valid fp32 evaluation
    of the same formula may decide differently)."""
    L = lib()
    prm = params or default_params()
    c = make_camera(cam)
    d12, sph, ro, rd, sh_degree, W, H = fwd["_inputs"]
    rgba = np.zeros((H, W, 4), np.float32); dist = np.full((H, W, 1), 1e6, np.float32); hits = np.zeros((H, W, 1), np.float32)
    margins = np.full((H, W, 2), np.finfo(np.float32).max, np.float32)
    budget = np.zeros((H, W, 2), np.float32) if budget_bound is not None else None
    if fwd["M"]:
        if budget is not None:
            L.oracle_set_pixel_budget_out(_p(budget), C.c_float(float(budget_bound)))
        L.oracle_render_margins(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), _p(d12), _p(fwd["feat"]), _p(ro), _p(rd),
                                _p(fwd["tile_ranges"]), _p(fwd["sorted_ids"]), _p(rgba), _p(dist), _p(hits), _p(margins))
        L.oracle_set_pixel_budget_out(None, C.c_float(0.0))
        assert np.array_equal(rgba, fwd["rgba"])
    return margins if budget is None else (margins, budget)


def render_kbuffer(cam: dict, fwd: dict, K=16, params=None, max_order=0):
    """Sorted-variant compositing (k_buffer_size=K) on top of a forward() result's tile lists.  Returns rgba, dist, hits and,
    if max_order > 0, the per-pixel composited particle order (order_ids [P,max_order] int32 -1 padded, order_count [P])."""
    L = lib()
    prm = params or default_params()
    c = make_camera(cam)
    d12, sph, ro, rd, sh_degree, W, H = fwd["_inputs"]
    rgba = np.zeros((H, W, 4), np.float32); dist = np.full((H, W, 1), 1e6, np.float32); hits = np.zeros((H, W, 1), np.float32)
    oc = np.zeros(W * H, np.int32)
    oi = np.full((W * H, max(1, max_order)), -1, np.int32)
    if fwd["M"]:
        L.oracle_render_kbuffer(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), C.c_int(K), _p(d12), _p(fwd["feat"]), _p(ro), _p(rd),
                                _p(fwd["tile_ranges"]), _p(fwd["sorted_ids"]), _p(rgba), _p(dist), _p(hits),
                                _p(oi) if max_order else None, _p(oc), C.c_int(max_order))
    return dict(rgba=rgba, dist=dist, hits=hits, order_ids=oi, order_count=oc)


def backward(cam: dict, fwd: dict, rgba_grad, dist_grad, params=None, flip_bound=None):
    """Backward for a forward() result. Returns (density_grad [N,12] f64, sph_grad [N,48] f64, feat_grad [N,3] f64).
    flip_bound (e.g. 6.0): additionally returns, as a 4th element, [N,10] f64: columns 0..4 the per-Gaussian flip budget
    (positions, density, rotation, scale, colour) — how far a different but equally valid fp32 evaluation may move each gradient
    row because a hit / no-hit decision within `flip_bound` noise widths of its threshold flips — and columns 5..9 the fp32
    conditioning of the same rows, sum over hits of eps * nu * |contribution| (gut_oracle.c: render_bwd_impl)."""
    L = lib()
    prm = params or default_params()
    c = make_camera(cam)
    d12, sph, ro, rd, sh_degree, W, H = fwd["_inputs"]
    N = d12.shape[0]
    rg = _f32(rgba_grad).reshape(H, W, 4); dg = _f32(dist_grad).reshape(H, W, 1)
    dens_g = np.zeros((N, 12), np.float64); feat_g = np.zeros((N, 3), np.float64)
    sph_g = np.zeros((N, 48), np.float64)
    trav = C.c_uint64(0)
    budget = np.zeros((N, 10), np.float64) if flip_bound is not None else None
    tile_trav = np.zeros(fwd["tile_ranges"].shape[0], np.uint32)
    if fwd["M"]:
        L.oracle_set_tile_traversed_out(_p(tile_trav))
    if fwd["M"] and budget is not None:
        L.oracle_render_bwd_budget(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), _p(d12), _p(fwd["feat"]),
                                   _p(ro), _p(rd), _p(fwd["tile_ranges"]), _p(fwd["sorted_ids"]),
                                   _p(fwd["rgba"]), _p(rg), _p(fwd["dist"]), _p(dg), _p(dens_g), _p(feat_g), C.byref(trav),
                                   _p(budget), C.c_float(float(flip_bound)))
    elif fwd["M"]:
        L.oracle_render_bwd(C.byref(prm), C.byref(c), C.c_int(W), C.c_int(H), _p(d12), _p(fwd["feat"]),
                            _p(ro), _p(rd), _p(fwd["tile_ranges"]), _p(fwd["sorted_ids"]),
                            _p(fwd["rgba"]), _p(rg), _p(fwd["dist"]), _p(dg), _p(dens_g), _p(feat_g), C.byref(trav))
    if fwd["M"]:
        L.oracle_project_bwd(C.byref(c), C.c_uint32(N), C.c_int(sh_degree), _p(d12), _p(fwd["tiles_count"]),
                             _p(fwd["feat"]), _p(feat_g), _p(sph_g))
    L.oracle_set_tile_traversed_out(None)
    fwd["traversed_bwd"] = int(trav.value)
    fwd["tile_traversed_bwd"] = tile_trav
    if budget is not None:
        return dens_g, sph_g, feat_g, budget
    return dens_g, sph_g, feat_g

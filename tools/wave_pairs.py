"""Dev probe (CPU, oracle): how many (wave, list entry) pairs of the bench frame have at least one hit lane when a wave owns a
16x4 strip vs an 8x8 block of the tile (oracle_count_wave_pairs).  Decided the compositors' pixel layout."""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import make_view, cams, scenes  # noqa: E402
import bench  # noqa: E402
oracle = importlib.import_module("oracle.oracle")
name = sys.argv[1] if len(sys.argv) > 1 else "bicycle_like_6M_1237x822"
fn, kw, W, H, fx, radius, elev, extent = bench.WORKLOADS[name]
sc = getattr(scenes, fn)(**kw)
view = make_view("fisheye" if "fisheye" in name else "pinhole", W, H, cams.orbit_c2w(radius, 7.0, elev), fx=fx, fy=fx)
d12 = scenes.pack_density(sc)
ref = oracle.forward(view["oracle_cam"], W, H, d12, sc["features"], view["ro"], view["rd"])
L, prm, cam = oracle.lib(), oracle.default_params(), oracle.make_camera(view["oracle_cam"])
out = np.zeros(8, np.uint64)
f32 = lambda a: np.ascontiguousarray(a, np.float32)
ro, rd = f32(view["ro"]).reshape(-1, 3), f32(view["rd"]).reshape(-1, 3)
t = time.time()
L.oracle_count_wave_pairs(C.byref(prm), C.byref(cam), C.c_int(W), C.c_int(H), oracle._p(f32(d12)), oracle._p(ro), oracle._p(rd),
                          oracle._p(ref["tile_ranges"]), oracle._p(ref["sorted_ids"]), oracle._p(out), oracle._p(f32(ref["proj_pos"])),
                          oracle._p(f32(ref["extent"])))
print(f"{name}: M {ref['M']}, walked {int(out[3])}, hit (pixel, entry) pairs {int(out[2])}")
print(f"(wave, entry) pairs with a hit lane: 16x4 strips {int(out[0])}, 8x8 blocks {int(out[1])}  ({out[1] / out[0] - 1:+.1%}); "
      f"hit lanes per pair {out[2] / out[0]:.1f} -> {out[2] / out[1]:.1f}   [{time.time() - t:.1f} s]")
print(f"(wave, entry) pairs in which the wave still has an alive ray (no culling): strips {int(out[4])}, blocks {int(out[5])}")
print(f"8x8 blocks, screen-rectangle proxies of the cull test: static block rectangle {int(out[6])}, rectangle of the alive pixels {int(out[7])}")

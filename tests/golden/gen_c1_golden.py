"""Writes tests/golden/c1_golden.npz from the CPU oracle on BASELINE configs[0] (seeded 1k Gaussians, 128x128).
The reference ships no fixtures for this path (parity unpinned), so this fixture pins the ORACLE
against accidental drift; it does not pin the oracle to the reference."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.common import cams, make_view, scenes
oracle = importlib.import_module("oracle.oracle")
sc = scenes.scene_c1(1000, 0)
view = make_view("pinhole", 128, 128, cams.look_at_c2w((0, 0, -4), (0, 0, 0)), fx=128.0)
o = oracle.forward(view["oracle_cam"], 128, 128, scenes.pack_density(sc), sc["features"], view["ro"], view["rd"])
rg = np.random.default_rng(42).normal(size=(128, 128, 4)).astype(np.float32)
dg, sg, _ = oracle.backward(view["oracle_cam"], o, rg, np.zeros((128, 128, 1), np.float32))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_golden.npz"), M=o["M"], tiles_count=o["tiles_count"],
                    sorted_ids=o["sorted_ids"], sorted_keys=o["sorted_keys"], tile_ranges=o["tile_ranges"], rgba_sub=o["rgba"][::4, ::4],
                    rgba_grad=rg, density_grad_colsum=dg.sum(0))
print("ok", o["M"])

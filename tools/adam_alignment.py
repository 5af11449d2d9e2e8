"""Dev probe: does the relative placement of the optimiser's six streams (p/m/v x [N,12], [N,48]) change the time of the fused
optimiser kernel?  Carves the arrays out of one arena with chosen byte offsets and times gut_sh_adam_step (k_sh_adam<false>)."""
import ctypes as C, importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("3dgrut_amd._capi")
lib = capi.load()
dev = "cuda:0"
N = 6_000_000
MB = 1 << 20


def carve(arena, off, shape):
    n = int(np.prod(shape))
    assert off % 16 == 0
    return arena[off // 4: off // 4 + n].view(*shape), off + n * 4


def run(pads, reps=12):
    """pads: byte paddings inserted before each of the 9 arrays (p12,m12,v12,act,g12,mrgb,p48,m48,v48)."""
    total = 4 * N * (12 * 5 + 3 + 48 * 3) + sum(pads) + 64 * MB
    arena = torch.zeros(total // 4, dtype=torch.float32, device=dev)
    base = arena.data_ptr()
    off = (-base) % (2 * MB)          # start 2 MiB aligned
    arrs = []
    for pad, shape in zip(pads, [(N, 12)] * 5 + [(N, 3)] + [(N, 48)] * 3):
        off += pad
        a, off = carve(arena, off, shape)
        arrs.append(a)
        off = (off + 2 * MB - 1) // (2 * MB) * (2 * MB)   # next array 2 MiB aligned again (what torch's allocator gives)
    p12, m12, v12, act, g12, mrgb, p48, m48, v48 = arrs
    p12.normal_(); p48.normal_(); g12.normal_(); mrgb.normal_()
    cam = torch.zeros((1, 3), device=dev)
    lr12 = (C.c_float * 12)(*([1e-3] * 12)); lr48 = (C.c_float * 48)(*([1e-3] * 48))
    st = torch.cuda.current_stream().cuda_stream
    def launch(step):
        rc = lib.gut_sh_adam_step(C.c_void_p(st), N, 3, 1, cam.data_ptr(), mrgb.data_ptr(), g12.data_ptr(), 1.0, p12.data_ptr(), m12.data_ptr(),
                                  v12.data_ptr(), p48.data_ptr(), m48.data_ptr(), v48.data_ptr(), lr12, lr48, 0.9, 0.999, 1e-15, step, None,
                                  act.data_ptr(), N)
        assert rc == 0
    for s in range(3):
        launch(s + 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in range(reps):
        launch(s + 4)
    e1.record(); torch.cuda.synchronize()
    del arena
    return e0.elapsed_time(e1) / reps


K = 1024
configs = {
    "all 2MiB-aligned": [0] * 9,
    "m48 +4K, v48 +8K": [0, 0, 0, 0, 0, 0, 0, 4 * K, 8 * K],
    "m48 +64K, v48 +128K": [0, 0, 0, 0, 0, 0, 0, 64 * K, 128 * K],
    "m48 +256K, v48 +512K": [0, 0, 0, 0, 0, 0, 0, 256 * K, 512 * K],
    "m48 +680K, v48 +1360K": [0, 0, 0, 0, 0, 0, 0, 680 * K, 1360 * K],
    "all nine staggered by 200K": [i * 200 * K for i in range(9)],
}
for rnd in range(2):
    for name, pads in configs.items():
        print(f"round {rnd}  {name:32s} {run(pads):.3f} ms", flush=True)

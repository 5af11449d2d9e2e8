"""Dev probe (GPU): which of the seven state tensors decides the plateau of the no-op optimiser pass?  Starting from seven fresh
allocations, one tensor at a time is moved to a fresh allocation (the old one is kept alive) and the pass is timed again."""
import ctypes as C, importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("3dgrut_amd._capi")
lib = capi.load()
dev = "cuda:0"
N = 6_000_000
nw = (N + 63) // 64
flags = torch.zeros(nw, dtype=torch.uint8, device=dev)
z12, z48 = (C.c_float * 12)(), (C.c_float * 48)()
st = torch.cuda.current_stream()
names = ["raw", "m12", "v12", "sh", "m48", "v48", "act"]
cols = [12, 12, 12, 48, 48, 48, 12]
t = [torch.ones((N, c), dtype=torch.float32, device=dev) for c in cols]
keep = []

def timed(reps=4):
    def call():
        rc = lib.gut_adam_unwalked_waves_ex(C.c_void_p(st.cuda_stream), N, flags.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                            t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), z12, z48, 1.0, 1.0, 1e-15, 0, t[6].data_ptr(), None)
        assert rc == 0
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

print(f"start: {timed():.3f} ms", " ".join(f"{n}={t[i].data_ptr() >> 21:x}" for i, n in enumerate(names)), flush=True)
for rnd in range(3):
    for k in (3, 4, 5, 0, 1, 2, 6):
        keep.append(t[k])
        t[k] = t[k].clone()
        print(f"round {rnd}: moved {names[k]:4s} -> {t[k].data_ptr() >> 21:x}: {timed():.3f} ms", flush=True)
# subsets: which streams cost what (only raw-sized / only SH-sized tensors aliased onto one buffer is not possible: time reads via torch)

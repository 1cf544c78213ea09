"""Sustained bursts (30 back-to-back launches, the last 20 timed as one interval) of the embed modes, bursts alternated."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import batch, native
from svsdct.native import Planes
F, H, W = 200, 2160, 3840
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["noise", "natural"]
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
planes = Planes.contiguous(F, H, W)
cap = batch.capacity_bits(F, H, W, n); nbytes = (cap + 7) // 8
gray = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
out = torch.empty_like(gray)
pay = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.svs_fill_bits_dev(pay.data_ptr(), cap, 1, 0, st)
def burst(mode, reps=20, warm=10):
    for _ in range(warm):
        batch.embed_device(gray.data_ptr(), out.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode=mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        batch.embed_device(gray.data_ptr(), out.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode=mode)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for kind in kinds:
    lib.svs_fill_synthetic_dev(gray.data_ptr(), C.byref(planes), 7, 0, 16, 224, st)
    if kind == "natural":
        g = gray[:, ::8, ::8].float()
        up = torch.nn.functional.interpolate(g[:, None], size=(H, W), mode="bilinear", align_corners=False)[:, 0]
        tex = (gray.float() - 128.0) * 0.06
        ramp = torch.linspace(-30, 30, W, device=dev)[None, None, :]
        gray.copy_((up * 0.6 + 50 + tex + ramp).clamp(0, 255).to(torch.uint8)); del g, up, tex
    elif kind == "flat128":
        gray.fill_(128)
    torch.cuda.synchronize()
    res = {m: [] for m in ("exact", "guarded", "fast")}
    for rnd in range(4):
        for m in res:
            res[m].append(burst(m))
    print(f"n={n} delta={delta:g} {kind:8s} " + "  ".join(f"{m} " + "/".join(f"{t:.3f}" for t in v) for m, v in res.items()))

"""Is the fast/guarded gap of guarded_probe.py an ordering artefact?  Same launch timed in alternation."""
import ctypes as C, os, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import batch, native
from svsdct.native import Planes
F, H, W, n, delta = 200, 2160, 3840, 3, 8.0
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
planes = Planes.contiguous(F, H, W)
cap = batch.capacity_bits(F, H, W, n); nbytes = (cap + 7) // 8
gray = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
out = torch.empty_like(gray); out2 = torch.empty_like(gray)
pay = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.svs_fill_bits_dev(pay.data_ptr(), cap, 1, 0, st)
lib.svs_fill_synthetic_dev(gray.data_ptr(), C.byref(planes), 7, 0, 16, 224, st)
def timed(mode, o):
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        batch.embed_device(gray.data_ptr(), o.data_ptr(), planes, delta, n, pay.data_ptr(), 0, cap, st, mode=mode)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts[2:])
for rnd in range(3):
    for mode, o in (("guarded", out), ("fast", out), ("guarded", out2), ("fast", out2), ("exact", out), ("fast", out), ("guarded", out)):
        print(rnd, mode, "out" if o is out else "out2", f"{timed(mode, o):.4f}")
    print("equal", torch.equal(out, out2))

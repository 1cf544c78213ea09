import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "secure-video-steganography-using-ecc-and-dct_amd"))
import torch
from svsdct import native
from svsdct.native import Planes
lib = native.load(); native.ensure_device(0)
dev = torch.device("cuda", 0)
F, H, W = 200, 2160, 3840
planes = Planes.contiguous(F, H, W)
a = torch.empty((F, H, W), dtype=torch.uint8, device=dev); b = torch.empty_like(a)
st = torch.cuda.current_stream().cuda_stream
lib.svs_fill_synthetic_dev(a.data_ptr(), C.byref(planes), 1, 0, 16, 224, st)
lib.svs_fill_synthetic_dev(b.data_ptr(), C.byref(planes), 2, 0, 16, 224, st)
work = torch.empty(int(lib.svs_ssim_workspace_bytes(C.byref(planes))) // 8 + 8, dtype=torch.float64, device=dev)
ssim = torch.empty(F, dtype=torch.float64, device=dev)
rng = torch.full((F,), 255.0, dtype=torch.float64, device=dev)
def timed(name, fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:28s} {ms:8.3f} ms  {F*H*W/ms/1e6:7.0f} Gpixel/s")
timed("ssim, data_range given", lambda: native.check(lib.svs_frame_ssim_dev(a.data_ptr(), b.data_ptr(), C.byref(planes), rng.data_ptr(), ssim.data_ptr(), work.data_ptr(), st), "x"))
timed("ssim, data_range = max-min", lambda: native.check(lib.svs_frame_ssim_dev(a.data_ptr(), b.data_ptr(), C.byref(planes), None, ssim.data_ptr(), work.data_ptr(), st), "x"))
print(ssim[:2].tolist())

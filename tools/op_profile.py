#!/usr/bin/env python3
"""Where the time of one call of the drop-in operator goes (4K frame, n = 10, delta = 20; median of 40): the pieces of
config_and_setup.proses_frame_qim_dct timed one by one next to the whole call."""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
import config_and_setup as cs
from svsdct import batch, hostmem, native, synth
from svsdct.native import Planes

lib = native.load(); native.ensure_device(0)


def t(f, reps=40):
    for _ in range(3):
        f()
    xs = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); xs.append(time.perf_counter() - t0)
    return np.median(xs) * 1e3


for (h, w) in ((2160, 3840), (1080, 1920)):
    n_ac, delta = 10, 20
    frame = synth.synthetic_frames(1, h, w, seed=3)[0]
    cap = batch.capacity_bits(1, h, w, n_ac)
    payload = batch.bits_to_str(synth.synthetic_bits(cap, seed=4)) * 3      # "the whole remaining payload": three frames' worth
    planes = Planes.contiguous(1, h, w)
    src_pin, dst_pin = hostmem.pinned_copy(frame), hostmem.pinned_empty(frame.shape)
    dst_page = np.empty_like(frame)
    addr, nch = batch._ascii_address(payload)
    used = C.c_uint64()
    print(f"== {w}x{h}, n = {n_ac}, delta = {delta}  (ms, median of 40)")
    print(f"frame.copy() (fresh pageable array)            {t(lambda: frame.copy()):.3f}")
    print(f"hostmem.pinned_copy(frame) (pooled)            {t(lambda: hostmem.pinned_copy(frame)):.3f}")
    print(f"hostmem.pinned_empty(shape)                    {t(lambda: hostmem.pinned_empty(frame.shape)):.3f}")
    print(f"_ascii_address(payload)                        {t(lambda: batch._ascii_address(payload)):.4f}")
    for name, s_, d_ in (("page-locked in, page-locked out", src_pin, dst_pin), ("pageable in,    page-locked out", frame, dst_pin),
                         ("pageable in,    pageable out   ", frame, dst_page)):
        print(f"svs_embed_str raw, {name}  {t(lambda: lib.svs_embed_str(s_.ctypes.data, None, d_.ctypes.data, C.byref(planes), float(delta), n_ac, addr, nch, 2, C.byref(used))):.3f}")
    ref_page = np.empty_like(frame)
    print(f"svs_embed_str raw + gray_ref_out, pageable in, page-locked out  {t(lambda: lib.svs_embed_str(frame.ctypes.data, ref_page.ctypes.data, dst_pin.ctypes.data, C.byref(planes), float(delta), n_ac, addr, nch, 2, C.byref(used))):.3f}")
    pk = batch.pack_bits(batch.str_to_bits(payload, cap))
    print(f"svs_embed raw (packed bits), page-locked both  {t(lambda: lib.svs_embed(src_pin.ctypes.data, dst_pin.ctypes.data, C.byref(planes), float(delta), n_ac, pk.ctypes.data, 0, cap, 2, C.byref(used))):.3f}")
    print(f"batch.embed_frames_str(pageable frame)         {t(lambda: batch.embed_frames_str(frame, delta, n_ac, payload)):.3f}")
    print(f"batch.embed_frames_str(page-locked frame)      {t(lambda: batch.embed_frames_str(src_pin, delta, n_ac, payload)):.3f}")
    print(f"proses_frame_qim_dct embed                     {t(lambda: cs.proses_frame_qim_dct(frame, 'embed', delta, payload, num_ac_coeffs_to_use=n_ac)):.3f}")
    _, stego, _ = cs.proses_frame_qim_dct(frame, "embed", delta, payload, num_ac_coeffs_to_use=n_ac)
    out = hostmem.pinned_empty(cap)
    got = C.c_uint64()
    print(f"svs_extract_str raw, page-locked both          {t(lambda: lib.svs_extract_str(stego.ctypes.data, C.byref(planes), float(delta), n_ac, out.ctypes.data, cap, 2, C.byref(got))):.3f}")
    outp = np.empty(cap, np.uint8)
    spage = np.array(stego)
    print(f"svs_extract_str raw, pageable both             {t(lambda: lib.svs_extract_str(spage.ctypes.data, C.byref(planes), float(delta), n_ac, outp.ctypes.data, cap, 2, C.byref(got))):.3f}")
    print(f"str(memoryview(out), 'ascii')                  {t(lambda: str(memoryview(out)[:cap], 'ascii')):.3f}")
    pkx = np.zeros((cap + 7) // 8 + 8, np.uint8)
    print(f"svs_extract raw (packed), page-locked in       {t(lambda: lib.svs_extract(stego.ctypes.data, C.byref(planes), float(delta), n_ac, pkx.ctypes.data, pkx.size, 2, C.byref(got))):.3f}")
    print(f"batch.unpack_to_str(packed)                    {t(lambda: batch.unpack_to_str(pkx, cap)):.3f}")
    print(f"batch.extract_frames_str(page-locked frame)    {t(lambda: batch.extract_frames_str(stego, delta, n_ac)):.3f}")
    print(f"proses_frame_qim_dct extract                   {t(lambda: cs.proses_frame_qim_dct(stego, 'extract', delta, num_ac_coeffs_to_use=n_ac)):.3f}")

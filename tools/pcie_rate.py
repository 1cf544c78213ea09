#!/usr/bin/env python3
"""PCIe-inclusive rates of the HOST-pointer boundary (svs_embed / svs_extract and the per-frame drop-in operator) next to the
box's own link ceilings.  Reported in DESIGN.md / profiles/r05_pcie_rate.txt only; never bench.py's `value`.

  1. ceilings: page-locked H2D alone, D2H alone, both at once on two streams (what a full-duplex pipeline can reach)
  2. svs_embed / svs_extract on a batch of 32 4K frames: pageable or page-locked input x pageable or page-locked output
  3. the drop-in operator config_and_setup.proses_frame_qim_dct frame by frame (the call shape of the reference's loops,
     embed_process.py:117-121, extract_process.py:64-68) at 640x480 / 1080p / 4K, n = 10, delta = 20 (the GUI's defaults)
  4. the overlapped FramePipeline (pinned slots, one stream per slot) for comparison
"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np

import config_and_setup as cs
from svsdct import batch, hostmem, native, synth
from svsdct.native import Planes

lib = native.load()
native.ensure_device(0)
print(f"library: {native.LIB_PATH}")


def med(xs):
    return float(np.median(xs))


# ---- 1. link ceilings -----------------------------------------------------------------------------------------------
def ceilings(nbytes, reps=9):
    h_a, h_b = hostmem.pinned_empty(nbytes), hostmem.pinned_empty(nbytes)
    h_a[:] = 1
    h_b[:] = 2
    d_a, d_b = C.c_void_p(), C.c_void_p()
    native.check(lib.svs_malloc(C.byref(d_a), nbytes), "malloc")
    native.check(lib.svs_malloc(C.byref(d_b), nbytes), "malloc")
    s1, s2 = C.c_void_p(), C.c_void_p()
    native.check(lib.svs_stream_create(C.byref(s1)), "stream")
    native.check(lib.svs_stream_create(C.byref(s2)), "stream")
    up, down, both = [], [], []
    for _ in range(reps):
        t = time.perf_counter()
        lib.svs_memcpy_h2d(d_a, h_a.ctypes.data, nbytes, s1)
        lib.svs_stream_synchronize(s1)
        up.append(time.perf_counter() - t)
        t = time.perf_counter()
        lib.svs_memcpy_d2h(h_b.ctypes.data, d_b, nbytes, s2)
        lib.svs_stream_synchronize(s2)
        down.append(time.perf_counter() - t)
        t = time.perf_counter()
        lib.svs_memcpy_h2d(d_a, h_a.ctypes.data, nbytes, s1)
        lib.svs_memcpy_d2h(h_b.ctypes.data, d_b, nbytes, s2)
        lib.svs_stream_synchronize(s1)
        lib.svs_stream_synchronize(s2)
        both.append(time.perf_counter() - t)
    for p in (d_a, d_b):
        lib.svs_free(p)
    for s in (s1, s2):
        lib.svs_stream_destroy(s)
    return nbytes / med(up) / 1e9, nbytes / med(down) / 1e9, nbytes / med(both) / 1e9, med(up), med(down), med(both)


print("\n== 1. link ceilings (page-locked memory, one hipMemcpyAsync + sync; GB/s, median of 9)")
ceil = {}
for label, nb in (("8.3 MB (one 4K frame)", 2160 * 3840), ("2.1 MB (one 1080p frame)", 1080 * 1920), ("265 MB (32 4K frames)", 32 * 2160 * 3840)):
    u, d, b, tu, td, tb = ceilings(nb)
    ceil[nb] = (u, d, b)
    print(f"{label:26s} H2D {u:6.1f} ({tu*1e3:.3f} ms)  D2H {d:6.1f} ({td*1e3:.3f} ms)  both at once {b:6.1f} each way ({tb*1e3:.3f} ms)")

# ---- 2. batch entry points ------------------------------------------------------------------------------------------
F, H, W, n, d = 32, 2160, 3840, 3, 8
px = F * H * W
frames_pageable = np.ascontiguousarray(np.broadcast_to(synth.synthetic_frames(1, H, W), (F, H, W)))
frames_pinned = hostmem.pinned_copy(frames_pageable)
bits = synth.synthetic_bits(batch.capacity_bits(F, H, W, n))
packed = batch.pack_bits(bits)
planes = Planes.contiguous(F, H, W)
up_c, down_c, both_c = ceil[px]
print(f"\n== 2. svs_embed / svs_extract, {F} x 4K frames ({px/1e6:.0f} MB each way), guarded mode, median of 7 calls")
print(f"   (serial ceiling of upload + download: {1/(1/up_c+1/down_c):.1f} Gpix/s; full-duplex ceiling: {both_c:.1f} Gpix/s)")
for in_name, src in (("pageable", frames_pageable), ("page-locked", frames_pinned)):
    for out_name, dst in (("pageable", np.empty_like(frames_pageable)), ("page-locked", hostmem.pinned_empty(frames_pageable.shape))):
        used = C.c_uint64()
        te = []
        for _ in range(8):
            t = time.perf_counter()
            rc = lib.svs_embed(src.ctypes.data, dst.ctypes.data, C.byref(planes), float(d), n, packed.ctypes.data, 0, bits.size,
                               native.SVS_EXACT_GUARDED, C.byref(used))
            te.append(time.perf_counter() - t)
            assert rc == 0 and used.value == bits.size
        te = med(te[1:])
        out = np.zeros((bits.size + 7) // 8 + 8, np.uint8)
        got = C.c_uint64()
        tx = []
        for _ in range(8):
            t = time.perf_counter()
            rc = lib.svs_extract(dst.ctypes.data, C.byref(planes), float(d), n, out.ctypes.data, out.size, native.SVS_EXACT_GUARDED, C.byref(got))
            tx.append(time.perf_counter() - t)
            assert rc == 0
        tx = med(tx[1:])
        ok = np.array_equal(np.unpackbits(out, count=bits.size), bits)
        print(f"in {in_name:11s} out {out_name:11s}: embed {te*1e3:6.1f} ms = {px/te/1e9:5.1f} Gpix/s ({px/te/1e9/both_c*100:3.0f} % of the duplex ceiling, "
              f"{px/te/1e9*(1/up_c+1/down_c)*100:3.0f} % of the serial one); extract {tx*1e3:6.1f} ms = {px/tx/1e9:5.1f} Gpix/s "
              f"({px/tx/1e9/up_c*100:3.0f} % of H2D); payload ok={ok}")
t = time.perf_counter(); stego, used = batch.embed_frames(frames_pageable, d, n, bits); te = time.perf_counter() - t
t = time.perf_counter(); pk, nb = batch.extract_frames(stego, d, n); tx = time.perf_counter() - t
print(f"svsdct.batch.embed_frames (NumPy in, pooled page-locked NumPy out, bit packing included): embed {te*1e3:.1f} ms = {px/te/1e9:.1f} Gpix/s, "
      f"extract_frames {tx*1e3:.1f} ms = {px/tx/1e9:.1f} Gpix/s, round trip {px/(te+tx)/1e9:.1f} Gpix/s")
del stego, frames_pinned

# ---- 3. the per-frame operator --------------------------------------------------------------------------------------
print("\n== 3. drop-in operator proses_frame_qim_dct frame by frame (NumPy frame + '0'/'1' string in, arrays / string out; n = 10, delta = 20; median of 30)")
for (h, w) in ((480, 640), (1080, 1920), (2160, 3840)):
    frame = synth.synthetic_frames(1, h, w, seed=3)[0]
    n_ac, delta = 10, 20
    cap = batch.capacity_bits(1, h, w, n_ac)
    pbits = synth.synthetic_bits(cap, seed=4)
    payload = batch.bits_to_str(pbits)
    for _ in range(3):
        g, s, used = cs.proses_frame_qim_dct(frame, "embed", delta, payload, num_ac_coeffs_to_use=n_ac)
    t_e, t_x, t_b = [], [], []
    for _ in range(30):
        t0 = time.perf_counter(); g, s, used = cs.proses_frame_qim_dct(frame, "embed", delta, payload, num_ac_coeffs_to_use=n_ac); t_e.append(time.perf_counter() - t0)
    for _ in range(30):
        t0 = time.perf_counter(); out = cs.proses_frame_qim_dct(s, "extract", delta, num_ac_coeffs_to_use=n_ac); t_x.append(time.perf_counter() - t0)
    assert out == payload[:used]
    for _ in range(30):
        t0 = time.perf_counter(); batch.embed_frames(frame[None], delta, n_ac, pbits); t_b.append(time.perf_counter() - t0)
    nb = h * w
    u, dn, _ = ceil.get(nb, (None, None, None)) if nb in ceil else ceilings(nb)[:3]
    link = nb / u / 1e9 + nb / dn / 1e9
    print(f"{w}x{h}: embed {med(t_e)*1e3:.3f} ms, extract {med(t_x)*1e3:.3f} ms; batch.embed_frames (packed bits) {med(t_b)*1e3:.3f} ms; "
          f"pure link time of the frame, up + down: {link*1e3:.3f} ms")

# ---- 4. overlapped pipeline -----------------------------------------------------------------------------------------
from svsdct.pipeline import FramePipeline
B, NB = 8, 12
clip = np.ascontiguousarray(np.broadcast_to(synth.synthetic_frames(1, H, W), (B, H, W)))
bits_all = synth.synthetic_bits(batch.capacity_bits(B * NB, H, W, n))
with FramePipeline(H, W, B, d, n, depth=3) as pipe:
    pipe.set_payload(bits_all)
    for rep in range(2):
        t = time.perf_counter()
        for k in range(NB + pipe.depth):
            slot = k % pipe.depth
            if k >= pipe.depth:
                pipe.embed_result(slot)
            if k < NB:
                np.copyto(pipe.input(slot), clip)            # stands for the decoder writing into pinned memory
                pipe.submit_embed(slot, B, bit_offset=k * pipe.batch_capacity)
        te = time.perf_counter() - t
    t = time.perf_counter()
    for k in range(NB + pipe.depth):
        slot = k % pipe.depth
        if k >= pipe.depth:
            pipe.embed_result(slot)
        if k < NB:
            pipe.submit_embed(slot, B, bit_offset=k * pipe.batch_capacity)   # producer already in pinned memory
    tn = time.perf_counter() - t
print(f"\n== 4. FramePipeline ({pipe.mode}), {NB} batches x {B} 4K frames: embed incl. host copy into pinned memory "
      f"{B*NB*H*W/te/1e9:.2f} Gpix/s; staging + kernel only {B*NB*H*W/tn/1e9:.2f} Gpix/s")

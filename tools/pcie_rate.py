#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-pointer entry points (svs_embed / svs_extract: pageable host memory ->
hipMemcpy -> kernel -> hipMemcpy).  Reported in DESIGN.md only; never bench.py's `value`."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
from svsdct import batch, synth
F, H, W, n, d = 32, 2160, 3840, 3, 8
frames = np.ascontiguousarray(np.broadcast_to(synth.synthetic_frames(1, H, W), (F, H, W)))
bits = synth.synthetic_bits(batch.capacity_bits(F, H, W, n))
for mode in ("fast", "exact"):
    batch.embed_frames(frames[:2], d, n, bits, mode=mode)                      # warm-up
    t = time.perf_counter(); stego, used = batch.embed_frames(frames, d, n, bits, mode=mode); te = time.perf_counter() - t
    t = time.perf_counter(); packed, nb = batch.extract_frames(stego, d, n, mode=mode); tx = time.perf_counter() - t
    ok = np.array_equal(np.unpackbits(packed, count=nb), bits)
    print(f"{mode}: {F} x 4K frames through host pointers: embed {te*1e3:.1f} ms ({F*H*W/te/1e9:.2f} Gpix/s), "
          f"extract {tx*1e3:.1f} ms ({F*H*W/tx/1e9:.2f} Gpix/s), round trip {F*H*W/(te+tx)/1e9:.2f} Gpix/s, payload ok={ok}")

# overlapped staging (svsdct/pipeline.py): pinned buffers, one stream per slot
from svsdct.pipeline import FramePipeline
B, NB = 8, 12
clip = np.ascontiguousarray(np.broadcast_to(synth.synthetic_frames(1, H, W), (B, H, W)))
bits_all = synth.synthetic_bits(batch.capacity_bits(B * NB, H, W, n))
for mode in ("fast", "exact"):
    with FramePipeline(H, W, B, d, n, depth=3, mode=mode) as pipe:
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
    print(f"{mode}: overlapped pipeline, {NB} batches x {B} 4K frames: embed incl. host copy into pinned memory "
          f"{B*NB*H*W/te/1e9:.2f} Gpix/s; staging + kernel only {B*NB*H*W/tn/1e9:.2f} Gpix/s")

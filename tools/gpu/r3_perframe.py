"""Cost of the per-frame drop-in call (config_and_setup.proses_frame_qim_dct) and of the host-pointer C entry points."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "secure-video-steganography-using-ecc-and-dct_amd"))
import numpy as np
import config_and_setup as cs
from svsdct import batch, synth
for (h, w) in ((480, 640), (1080, 1920), (2160, 3840)):
    frame = synth.synthetic_frames(1, h, w, seed=3)[0]
    n_ac, delta = 10, 20
    cap = batch.capacity_bits(1, h, w, n_ac)
    bits = synth.synthetic_bits(cap, seed=4)
    payload = "".join("1" if b else "0" for b in bits)
    cs.proses_frame_qim_dct(frame, "embed", delta, payload, num_ac_coeffs_to_use=n_ac)     # warm-up
    t = []
    for _ in range(20):
        t0 = time.perf_counter(); g, s, used = cs.proses_frame_qim_dct(frame, "embed", delta, payload, num_ac_coeffs_to_use=n_ac); t.append(time.perf_counter() - t0)
    te = []
    for _ in range(20):
        t0 = time.perf_counter(); out = cs.proses_frame_qim_dct(s, "extract", delta, num_ac_coeffs_to_use=n_ac); te.append(time.perf_counter() - t0)
    assert out == payload[:used]
    tb = []
    for _ in range(20):
        t0 = time.perf_counter(); batch.embed_frames(frame[None], delta, n_ac, bits); tb.append(time.perf_counter() - t0)
    print(f"{w}x{h}: operator embed {np.median(t)*1e3:.3f} ms, extract {np.median(te)*1e3:.3f} ms (bit strings in / out); batch.embed_frames on the same frame {np.median(tb)*1e3:.3f} ms")

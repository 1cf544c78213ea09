set -e
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/guard_small.log 2>&1
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'secure-video-steganography-using-ecc-and-dct_amd')
import numpy as np
from svsdct import batch, native, synth
from oracle import qim_dct_oracle as orc
native.ensure_device(0)
for (n, d) in [(3,8),(7,8),(1,8),(3,20),(5,0.5),(3,7.3),(3,100)]:
    frames = synth.synthetic_frames(3, 136, 264, seed=11)
    frames[0, :16, :48] = 128
    frames[1, 40:80, :] = frames[1, 40:41, :]
    cap = batch.capacity_bits(3, 136, 264, n)
    payload = synth.synthetic_bits(cap - 100, seed=11)
    ref, used = orc.batch_embed(frames, d, payload, n)
    g, u2 = batch.embed_frames(frames, d, n, payload, mode="guarded")
    x, u3 = batch.embed_frames(frames, d, n, payload, mode="exact")
    print(n, d, "guarded==oracle", np.array_equal(g, ref), "exact==oracle", np.array_equal(x, ref), used, u2, u3, int((g!=ref).sum()))
PY
cat gpurun_out/guard_small.log
timeout -k 10 300 python tools/guarded_probe.py --frames 200 --json gpurun_out/guarded_probe_n3.json > gpurun_out/guarded_probe_n3.log 2>&1
cat gpurun_out/guarded_probe_n3.log

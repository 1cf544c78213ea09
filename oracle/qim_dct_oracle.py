"""CPU oracle for the 8x8 block-DCT / QIM frame operator.

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement of the reference's
`proses_frame_qim_dct` (reference `config_and_setup.py:106-174`).  It exists so
that tests, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg have
something to check the HIP path against.  Nothing in the product package may
import it: the product path fails loudly when the HIP library is missing.

Parity status: PINNED.  `tests/golden/make_golden.py` imported the reference's
own function in the build container and wrote its inputs/outputs to
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this restatement
against every one of those vectors bit for bit (stego pixels, bit counts and
extracted bits).

Third-party arithmetic the reference relies on (not under /root/reference):
`scipy.fftpack.dct/idct(norm='ortho')` (pocketfft, single precision for
float32 input).  scipy is unpinned in the reference (`requirements.txt:2`); the
vectors were produced with scipy 1.15.3 / numpy 2.2.6.  The restatement calls
the same scipy entry points, batched over all blocks of a frame; pocketfft's
per-line arithmetic does not depend on the batch, so results are bit-identical
to the per-block calls of the reference (checked by the golden tests).

Two forms are provided:
  * `frame_embed` / `frame_extract`  - vectorised over (H/8, W/8, 8, 8); the
    form that is timed as the CPU baseline.
  * `frame_operator_loops`           - a block-by-block, coefficient-by-
    coefficient walk (pure Python; small inputs only) kept as an independent
    cross-check of the vectorised form's bookkeeping.
"""
from __future__ import annotations

import numpy as np
from scipy.fftpack import dct, idct

BLOCK = 8
MAX_AC = BLOCK * BLOCK - 1  # reference clamps to len(flat)-1, config_and_setup.py:138


# --------------------------------------------------------------------------
# payload helpers (bit strings <-> 0/1 arrays <-> MSB-first packed bytes)
# --------------------------------------------------------------------------
def bits_from_any(payload) -> np.ndarray:
    """Accept the reference's '0'/'1' `str`, None, or a 0/1 array -> uint8 0/1 array."""
    if payload is None:
        return np.zeros(0, np.uint8)
    if isinstance(payload, str):
        return np.frombuffer(payload.encode("ascii"), np.uint8) - np.uint8(48)
    return np.asarray(payload, np.uint8)


def bits_to_str(bits: np.ndarray) -> str:
    return (np.asarray(bits, np.uint8) + np.uint8(48)).tobytes().decode("ascii")


def _blocks_view(plane_f32: np.ndarray) -> np.ndarray:
    """(H, W) -> (H/8, W/8, 8, 8) copy, block raster order (rows outer), as the
    reference's `for r_start ... for c_start ...` scan, config_and_setup.py:129-134."""
    h, w = plane_f32.shape
    return np.ascontiguousarray(
        plane_f32.reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK).transpose(0, 2, 1, 3))


def _check_plane(gray: np.ndarray) -> None:
    if gray.ndim != 2:
        # the 3-channel branch needs cv2.cvtColor (config_and_setup.py:112); the oracle
        # covers the gray-plane branch (:113-114) only - BGR->gray parity is unpinned.
        raise ValueError("oracle handles 2-D gray planes only")
    if gray.shape[0] % BLOCK or gray.shape[1] % BLOCK:
        raise ValueError("frame dimensions must be multiples of 8")


def _fwd(blocks: np.ndarray) -> np.ndarray:
    # config_and_setup.py:135  dct(dct(block, axis=0, 'ortho'), axis=1, 'ortho')
    return dct(dct(blocks, axis=2, norm="ortho"), axis=3, norm="ortho")


def _inv(coeffs: np.ndarray) -> np.ndarray:
    # config_and_setup.py:168  idct(idct(D, axis=0, 'ortho'), axis=1, 'ortho')
    return idct(idct(coeffs, axis=2, norm="ortho"), axis=3, norm="ortho")


def _quant_index(c: np.ndarray, delta) -> np.ndarray:
    """`int(round(c / delta))` with c float32 (config_and_setup.py:148,160).

    NumPy >= 2: float32 / python-number divides in float32 (the python scalar is
    weak); `round()` of a numpy float32 rounds half to even."""
    return np.rint(c / np.float32(delta)).astype(np.int64)


def _requantised(q: np.ndarray, delta) -> np.ndarray:
    """`float(q * delta)` stored into a float32 array (config_and_setup.py:156):
    python-int * python-number (exact for int delta, one double rounding for float
    delta), then rounded to float32 by the array store."""
    if isinstance(delta, (int, np.integer)):
        return (q * int(delta)).astype(np.float32)
    return (q.astype(np.float64) * float(delta)).astype(np.float32)


# --------------------------------------------------------------------------
# vectorised restatement
# --------------------------------------------------------------------------
def frame_embed(gray: np.ndarray, delta, payload, n_ac: int = MAX_AC):
    """Embed mode of config_and_setup.py:106-172 on a 2-D uint8 plane.

    Returns (gray_copy, stego_uint8, bits_consumed)."""
    _check_plane(gray)
    gray = np.ascontiguousarray(gray, np.uint8)
    bits = bits_from_any(payload)
    n_use = max(0, min(int(n_ac), MAX_AC))                      # :138
    h, w = gray.shape
    n_blocks = (h // BLOCK) * (w // BLOCK)
    budget = int(bits.size)                                      # :124-126
    out_f = np.float32(gray)                                     # :117,120

    if budget == 0:
        # loops break before the first block (:130) -> nothing is transformed
        return gray.copy(), gray.copy(), 0

    if delta <= 0 or n_use == 0:
        # every coefficient is skipped without consuming a bit (:143-145), so the
        # budget never runs out and every block is DCT->IDCT round-tripped (:166-169)
        touched = n_blocks
        consumed = 0
    else:
        touched = min(n_blocks, -(-budget // n_use))            # blocks entered before :130/:132 break
        consumed = min(budget, n_blocks * n_use)

    blk = _blocks_view(out_f).reshape(n_blocks, BLOCK, BLOCK)[:touched]
    coef = _fwd(blk.reshape(1, touched, BLOCK, BLOCK)).reshape(touched, BLOCK * BLOCK)

    if consumed:
        use = bits[:consumed].astype(np.int64)
        # bit i of the frame belongs to block i // n, flat coefficient 1 + i % n  (:139-140)
        bi = np.arange(consumed) // n_use
        ki = 1 + np.arange(consumed) % n_use
        c = coef[bi, ki]
        q = _quant_index(c, delta)                               # :148
        par = q & 1                                              # python % 2 is non-negative (:150)
        q = np.where(par != use, np.where(use == 1, q + 1, q - 1), q)   # :151-155
        coef[bi, ki] = _requantised(q, delta)                    # :156

    rec = _inv(coef.reshape(1, touched, BLOCK, BLOCK)).reshape(touched, BLOCK, BLOCK)
    full = _blocks_view(out_f).reshape(n_blocks, BLOCK, BLOCK)
    full[:touched] = rec
    out_f = full.reshape(h // BLOCK, w // BLOCK, BLOCK, BLOCK).transpose(0, 2, 1, 3).reshape(h, w)
    stego = np.uint8(np.clip(out_f, 0, 255))                     # :171 clip, then C truncation
    return gray.copy(), stego, int(consumed)


def frame_extract_bits(gray: np.ndarray, delta, n_ac: int = MAX_AC) -> np.ndarray:
    """Extract mode (config_and_setup.py:159-165,173-174) -> uint8 0/1 array of
    length (H/8)(W/8)*min(n_ac, 63), block raster order, coefficients 1..n inside a block."""
    _check_plane(gray)
    n_use = max(0, min(int(n_ac), MAX_AC))
    h, w = gray.shape
    n_blocks = (h // BLOCK) * (w // BLOCK)
    if n_use == 0:
        return np.zeros(0, np.uint8)
    if delta <= 0:
        return np.zeros(n_blocks * n_use, np.uint8)              # :143-145 emits '0'
    blk = _blocks_view(np.float32(gray)).reshape(1, n_blocks, BLOCK, BLOCK)
    coef = _fwd(blk).reshape(n_blocks, BLOCK * BLOCK)[:, 1:1 + n_use]
    q = _quant_index(coef, delta)                                # :160
    return (q & 1).astype(np.uint8).reshape(-1)                  # :161


def frame_extract(gray: np.ndarray, delta, n_ac: int = MAX_AC) -> str:
    return bits_to_str(frame_extract_bits(gray, delta, n_ac))


def frame_operator(frame, mode, delta, payload=None, n_ac=MAX_AC):
    """Same call shape as the reference operator for 2-D input."""
    if mode == "embed":
        return frame_embed(frame, delta, payload, n_ac)
    if mode == "extract":
        return frame_extract(frame, delta, n_ac)
    return None  # the reference falls off the end for unknown modes


# --------------------------------------------------------------------------
# batch helpers used by tests / bench (frames share one running bit stream)
# --------------------------------------------------------------------------
def batch_embed(frames: np.ndarray, delta, bits: np.ndarray, n_ac: int):
    """Frame loop of embed_process.py:108-128 over an array [F,H,W]: frame k gets
    bits [k*cap, (k+1)*cap) of the stream.  Returns (stego[F,H,W], consumed)."""
    bits = bits_from_any(bits)
    out = np.empty_like(frames)
    pos = 0
    for k in range(frames.shape[0]):
        if pos < bits.size:
            _, out[k], used = frame_embed(frames[k], delta, bits[pos:], n_ac)
            pos += used
            if delta <= 0:
                # reference would loop forever consuming nothing; callers never do this
                pass
        else:
            out[k] = frames[k]
    return out, pos


def batch_extract_bits(frames: np.ndarray, delta, n_ac: int) -> np.ndarray:
    """Concatenation of extract_process.py:64-76 over frames in order."""
    return np.concatenate([frame_extract_bits(f, delta, n_ac) for f in frames])


# --------------------------------------------------------------------------
# literal per-block walk (small inputs; independent bookkeeping check)
# --------------------------------------------------------------------------
def frame_operator_loops(gray: np.ndarray, mode: str, delta, payload=None, n_ac: int = MAX_AC):
    _check_plane(gray)
    bits = bits_from_any(payload)
    pix = np.float32(gray)
    out = pix.copy()
    h, w = pix.shape
    taken = 0
    limit = int(bits.size) if mode == "embed" else 0
    emitted = []
    done = False
    for y0 in range(0, h, BLOCK):
        if done:
            break
        for x0 in range(0, w, BLOCK):
            if mode == "embed" and taken >= limit:               # :130,:132
                done = True
                break
            tile = pix[y0:y0 + BLOCK, x0:x0 + BLOCK]
            flat = dct(dct(tile, axis=0, norm="ortho"), axis=1, norm="ortho").flatten()
            new = flat.copy()
            for k in range(1, min(int(n_ac), MAX_AC) + 1):
                if mode == "embed" and taken >= limit:           # :141
                    break
                if delta <= 0:                                   # :143-145
                    if mode == "extract":
                        emitted.append(0)
                    continue
                qi = int(round(flat[k] / delta))
                if mode == "embed":
                    want = int(bits[taken])
                    if qi % 2 != want:
                        qi += 1 if want == 1 else -1
                    new[k] = float(qi * delta)
                    taken += 1
                else:
                    emitted.append(qi % 2)
            if mode == "embed":
                out[y0:y0 + BLOCK, x0:x0 + BLOCK] = idct(
                    idct(new.reshape(BLOCK, BLOCK), axis=0, norm="ortho"), axis=1, norm="ortho")
    if mode == "embed":
        return gray.copy(), np.uint8(np.clip(out, 0, 255)), taken
    return bits_to_str(np.array(emitted, np.uint8))


# --------------------------------------------------------------------------
# metrics used by the parity harness
# --------------------------------------------------------------------------
def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    """10*log10(255^2 / MSE) as cv2.PSNR computes it (embed_process.py:205, app.py:342);
    exact integer MSE, inf for identical inputs."""
    d = a.astype(np.int64) - b.astype(np.int64)
    sse = int((d * d).sum())
    if sse == 0:
        return float("inf")
    return 10.0 * np.log10(255.0 * 255.0 * d.size / sse)

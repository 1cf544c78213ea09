"""Test doubles for the third-party libraries the host glue talks to and that are absent from the
build image (cv2, cryptography).  They implement just the calls the drop-in pipelines make, over
in-memory videos, so the frame-loop / framing logic can be exercised here; nothing in the product
imports this file."""
import hashlib
import time
import types

import numpy as np

WRITE_DELAY = [0.0]   # seconds every VideoWriter.write sleeps (tests of the decode / encode overlap set it)
VIDEOS = {}      # path -> {"frames": [BGR uint8 arrays], "fps": float, "size": (w, h), "fourcc": int}


def make_fake_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT, cv2.CAP_PROP_FPS = 3, 4, 5
    cv2.COLOR_BGR2GRAY, cv2.COLOR_GRAY2BGR = 6, 8
    cv2.IMREAD_GRAYSCALE = 0

    class VideoCapture:
        def __init__(self, path):
            self.video = VIDEOS.get(path)
            self.pos = 0

        def isOpened(self):
            return self.video is not None

        def get(self, prop):
            h, w = self.video["frames"][0].shape[:2]
            return {3: float(w), 4: float(h), 5: self.video["fps"]}[prop]

        def read(self):
            if self.video is None or self.pos >= len(self.video["frames"]):
                return False, None
            if self.video.get("read_delay"):                       # a slow decoder
                time.sleep(self.video["read_delay"])
            self.video.setdefault("read_times", []).append(time.perf_counter())
            self.pos += 1
            return True, self.video["frames"][self.pos - 1].copy()

        def release(self):
            self.video = None

    class VideoWriter:
        def __init__(self, path, fourcc, fps, size, isColor=True):
            self.path, self.open = path, True
            VIDEOS[path] = {"frames": [], "fps": fps, "size": size, "fourcc": fourcc, "color": isColor}

        def isOpened(self):
            return self.open

        def write(self, frame):
            w, h = VIDEOS[self.path]["size"]
            assert frame.shape == (h, w, 3) and frame.dtype == np.uint8, frame.shape
            if WRITE_DELAY[0]:                                      # a slow encoder
                time.sleep(WRITE_DELAY[0])
            VIDEOS[self.path].setdefault("write_times", []).append(time.perf_counter())
            VIDEOS[self.path]["frames"].append(frame.copy())

        def release(self):
            self.open = False

    def cvtColor(img, code):
        if code == cv2.COLOR_BGR2GRAY:       # OpenCV's fixed-point BT.601
            b, g, r = (img[..., i].astype(np.uint32) for i in range(3))
            return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)
        if code == cv2.COLOR_GRAY2BGR:
            return np.repeat(img[..., None], 3, axis=2)
        raise NotImplementedError(code)

    def psnr(a, b):
        d = a.astype(np.float64) - b.astype(np.float64)
        mse = (d * d).mean()
        return float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)

    cv2.VideoCapture, cv2.VideoWriter, cv2.cvtColor, cv2.PSNR = VideoCapture, VideoWriter, cvtColor, psnr
    cv2.VideoWriter_fourcc = lambda *c: sum(ord(ch) << (8 * i) for i, ch in enumerate(c))
    return cv2


# ---- stand-in crypto with the same call shapes as config_and_setup's wrappers ---------------------
class FakeKey:
    def __init__(self, secret: bytes):
        self.secret = secret

    def public(self):
        return FakePub(hashlib.sha256(b"pub" + self.secret).digest())


class FakePub:
    def __init__(self, ident: bytes):
        self.ident = ident


_counter = [0]


def buat_pasangan_kunci_ecc():
    _counter[0] += 1
    k = FakeKey(hashlib.sha256(b"eph%d" % _counter[0]).digest())
    return k, k.public()


def serialisasi_kunci_publik_ecc_compressed(pub):
    return b"\x02" + pub.ident                       # 33 bytes like a compressed P-256 point


def deserialisasi_kunci_publik_ecc_compressed(data, kurva=None):
    if len(data) != 33:
        raise ValueError("bad point")
    return FakePub(data[1:])


def buat_shared_secret_ecdh(priv, pub):
    # symmetric in (priv, pub): hash of the sorted pair of public identities
    return hashlib.sha256(b"".join(sorted([priv.public().ident, pub.ident]))).digest()


def derive_kunci_aes_dari_shared_secret(secret, salt=None, panjang=32):
    return hashlib.sha256(b"hkdf" + (salt or b"") + secret).digest()[:panjang]


def _keystream(key, nonce, n):
    out, i = b"", 0
    while len(out) < n:
        out += hashlib.sha256(key + nonce + i.to_bytes(4, "big")).digest()
        i += 1
    return out[:n]


def enkripsi_aes_gcm(data, key):
    nonce = hashlib.sha256(b"nonce" + data[:8]).digest()[:12]
    ct = bytes(a ^ b for a, b in zip(data, _keystream(key, nonce, len(data))))
    return ct, nonce, hashlib.sha256(key + nonce + ct).digest()[:16]


def dekripsi_aes_gcm(ct, key, nonce, tag):
    if hashlib.sha256(key + nonce + ct).digest()[:16] != tag:
        print("Error Dekripsi AES: Tag autentikasi tidak valid.")
        return None
    return bytes(a ^ b for a, b in zip(ct, _keystream(key, nonce, len(ct))))


CRYPTO_NAMES = ["buat_pasangan_kunci_ecc", "serialisasi_kunci_publik_ecc_compressed",
                "deserialisasi_kunci_publik_ecc_compressed", "buat_shared_secret_ecdh",
                "derive_kunci_aes_dari_shared_secret", "enkripsi_aes_gcm", "dekripsi_aes_gcm"]


# ---- stand-in for svsdct.pipeline.FramePipeline over the CPU build of the kernel header (CPU tier only) -------------
class EmuFramePipeline:
    """Same interface and batch bookkeeping as svsdct.pipeline.FramePipeline; the work is done synchronously by
    tests/hostemu at submit time.  Lets the CPU tier run the drop-in frame loops without a GPU."""

    def __init__(self, height, width, batch_frames, delta, n_ac, depth=3, mode=None, device=0):
        from svsdct import batch
        self.h, self.w, self.batch, self.depth = height, width, batch_frames, depth
        self.delta, self.n_ac, self.exact = delta, n_ac, (mode or "fast") == "exact"
        self.frame_capacity = batch.capacity_bits(1, height, width, n_ac)
        self.batch_capacity = self.frame_capacity * batch_frames
        self._in = [np.zeros((batch_frames, height, width), np.uint8) for _ in range(depth)]
        self._out = [None] * depth
        self._payload = np.zeros(0, np.uint8)

    def bind_thread(self):
        pass

    def set_payload(self, bits):
        self._payload = np.asarray(bits, np.uint8).copy()

    def input(self, slot):
        return self._in[slot]

    def submit_embed(self, slot, n_frames, bit_offset):
        from testlib import emu_embed
        stego, used = emu_embed(self._in[slot][:n_frames].copy(), self.delta, self.n_ac, self._payload,
                                bit_offset=min(bit_offset, self._payload.size), exact=self.exact)
        self._out[slot] = stego
        return used

    def embed_result(self, slot):
        return self._out[slot]

    def submit_extract(self, slot, n_frames):
        from testlib import emu_extract
        flags = emu_extract(self._in[slot][:n_frames].copy(), self.delta, self.n_ac, exact=self.exact)
        self._out[slot] = (np.packbits(flags), int(flags.size))
        return int(flags.size)

    def extract_result(self, slot):
        return self._out[slot]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

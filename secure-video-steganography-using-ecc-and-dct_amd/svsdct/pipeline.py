"""Overlapped host <-> GPU staging for frame producers / consumers that live on the host
(the video loops of embed_process.py:89-146 / extract_process.py:30-86).

`FramePipeline` owns `depth` slots; each slot has a pinned host input buffer, a pinned host output
buffer, device buffers and its own HIP stream.  While the GPU works on slot k (H2D copy -> kernel ->
D2H copy, all asynchronous on the slot's stream) the host fills slot k+1 - e.g. with decoded frames -
and drains slot k-1.  Copies from pinned memory run at link rate, and H2D / D2H of different slots
overlap on the two DMA directions.

    pipe = FramePipeline(height, width, batch_frames, delta, n_ac, depth=3)
    pipe.set_payload(bits)                       # once: the whole stream, indexed by bit offset per batch
    for k, frames in enumerate(batches):         # frames: uint8 [B, H, W]
        slot = k % pipe.depth
        if k >= pipe.depth:
            consume(pipe.embed_result(slot))     # stego of batch k - depth (waits for that slot only)
        np.copyto(pipe.input(slot)[:len(frames)], frames)
        pipe.submit_embed(slot, len(frames), bit_offset=k * pipe.batch_capacity)

With a host producer that blocks outside the interpreter lock (cv2 decoding) the loop above still serialises decode and
encode; `SlotFeeder` moves the producer half (decode into the slot's pinned buffer + submit) to a thread of its own, so
that decoding, the GPU and the consumer's encoding all run at once, and `read_ahead` does the same for a plain
decode -> encode copy loop.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading

import numpy as np

from . import batch, native
from .native import Planes


def _pinned(nbytes: int):
    ptr = C.c_void_p()
    native.check(native.load().svs_host_alloc(C.byref(ptr), nbytes), "svs_host_alloc")
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,))
    return ptr, arr


def _device(nbytes: int):
    ptr = C.c_void_p()
    native.check(native.load().svs_malloc(C.byref(ptr), nbytes), "svs_malloc")
    return ptr


class FramePipeline:
    def __init__(self, height: int, width: int, batch_frames: int, delta, n_ac, depth: int = 3,
                 mode: str | None = None, device: int = 0):
        if height % 8 or width % 8:
            raise ValueError("frame height and width must be multiples of 8")
        native.ensure_device(device)
        self.lib = native.load()
        self.device = device
        self.h, self.w, self.batch, self.depth = height, width, batch_frames, depth
        self.delta, self.n_ac, self.mode = delta, n_ac, batch.resolve_mode(mode)
        self.frame_capacity = batch.capacity_bits(1, height, width, n_ac)
        self.batch_capacity = self.frame_capacity * batch_frames
        self._bits_bytes = (self.batch_capacity + 7) // 8 + 8
        nbytes = batch_frames * height * width
        self._slots = []
        for _ in range(depth):
            st = C.c_void_p()
            native.check(self.lib.svs_stream_create(C.byref(st)), "svs_stream_create")
            hin_p, hin = _pinned(nbytes)
            hout_p, hout = _pinned(nbytes)
            hbits_p, hbits = _pinned(self._bits_bytes)
            self._slots.append(dict(stream=st, hin_p=hin_p, hin=hin.reshape(batch_frames, height, width),
                                    hout_p=hout_p, hout=hout.reshape(batch_frames, height, width),
                                    hbits_p=hbits_p, hbits=hbits, d_frames=_device(nbytes),
                                    d_bits=_device(self._bits_bytes), frames=0, bits=0))
        self._d_payload = None
        self._payload_bits = 0

    def bind_thread(self) -> None:
        """Call once on any other thread that is going to use this pipeline (HIP's current device is per thread)."""
        native.ensure_device(self.device)

    # ---- payload (embed) ------------------------------------------------------------------------
    def set_payload(self, bits: np.ndarray) -> None:
        """Upload the whole 0/1 stream once; batches index it by bit offset."""
        packed = batch.pack_bits(np.asarray(bits, np.uint8))
        if self._d_payload is not None:
            self.lib.svs_free(self._d_payload)
        self._d_payload = _device(packed.size + 8)
        native.check(self.lib.svs_memcpy_h2d(self._d_payload, packed.ctypes.data, packed.size, None), "svs_memcpy_h2d")
        native.check(self.lib.svs_stream_synchronize(None), "svs_stream_synchronize")
        self._payload_bits = int(np.asarray(bits).size)

    # ---- slots -------------------------------------------------------------------------------------
    def input(self, slot: int) -> np.ndarray:
        """Pinned uint8 [batch, H, W] buffer to fill with gray frames (only after the slot's previous result
        has been collected)."""
        return self._slots[slot]["hin"]

    def _planes(self, n_frames: int) -> Planes:
        return Planes.contiguous(n_frames, self.h, self.w)

    def submit_embed(self, slot: int, n_frames: int, bit_offset: int) -> int:
        """Enqueue H2D -> embed -> D2H for the first n_frames frames of the slot; returns the bits this batch
        will carry.  Does not wait."""
        s = self._slots[slot]
        nbytes = n_frames * self.h * self.w
        left = max(0, self._payload_bits - bit_offset)
        native.check(self.lib.svs_memcpy_h2d(s["d_frames"], s["hin_p"], nbytes, s["stream"]), "svs_memcpy_h2d")
        used = batch.embed_device(s["d_frames"].value, s["d_frames"].value, self._planes(n_frames), self.delta, self.n_ac,
                                  self._d_payload.value if self._d_payload else 0, bit_offset, left,
                                  stream=s["stream"].value, mode=self.mode)
        native.check(self.lib.svs_memcpy_d2h(s["hout_p"], s["d_frames"], nbytes, s["stream"]), "svs_memcpy_d2h")
        s["frames"], s["bits"] = n_frames, used
        return used

    def embed_result(self, slot: int) -> np.ndarray:
        """Wait for the slot's stream and return the pinned stego frames [n_frames, H, W] (valid until the slot is
        submitted again)."""
        s = self._slots[slot]
        native.check(self.lib.svs_stream_synchronize(s["stream"]), "svs_stream_synchronize")
        return s["hout"][: s["frames"]]

    def submit_extract(self, slot: int, n_frames: int) -> int:
        s = self._slots[slot]
        nbytes = n_frames * self.h * self.w
        native.check(self.lib.svs_memcpy_h2d(s["d_frames"], s["hin_p"], nbytes, s["stream"]), "svs_memcpy_h2d")
        got = batch.extract_device(s["d_frames"].value, self._planes(n_frames), self.delta, self.n_ac, s["d_bits"].value,
                                   self._bits_bytes, stream=s["stream"].value, mode=self.mode)
        native.check(self.lib.svs_memcpy_d2h(s["hbits_p"], s["d_bits"], (got + 7) // 8, s["stream"]), "svs_memcpy_d2h")
        s["frames"], s["bits"] = n_frames, got
        return got

    def extract_result(self, slot: int) -> tuple[np.ndarray, int]:
        """-> (packed bits of the batch (pinned view), n_bits)"""
        s = self._slots[slot]
        native.check(self.lib.svs_stream_synchronize(s["stream"]), "svs_stream_synchronize")
        return s["hbits"][: (s["bits"] + 7) // 8], s["bits"]

    def close(self) -> None:
        for s in self._slots:
            self.lib.svs_stream_synchronize(s["stream"])
            self.lib.svs_free(s["d_frames"])
            self.lib.svs_free(s["d_bits"])
            for key in ("hin_p", "hout_p", "hbits_p"):
                self.lib.svs_host_free(s[key])
            self.lib.svs_stream_destroy(s["stream"])
        self._slots = []
        if self._d_payload is not None:
            self.lib.svs_free(self._d_payload)
            self._d_payload = None
        native.release_thread_context()      # the per-frame operator calls of this thread (first-frame PSNR pair, ...) grew one

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class SlotFeeder:
    """Producer half of an overlapped frame loop, on a thread of its own.

    The thread takes a free slot, calls `fill(slot)` - which puts up to a batch of frames into `pipe.input(slot)` and returns
    how many (0 = end of input) -, calls `submit(slot, k, n)` - which enqueues the slot's GPU work (`submit_embed` /
    `submit_extract` return at once) and returns whatever the consumer wants to know about batch k -, and passes
    `(slot, k, n, info)` on, in order.  The consumer iterates the feeder, collects the slot's result (`*_result` waits for
    that slot's stream only), uses it, and hands the slot back with `release(slot)`; only then is it filled again, so
    `pipe.input(slot)` and the result stay valid for as long as the consumer holds the slot.

        with SlotFeeder(pipe, fill, submit) as feeder:
            for slot, k, n, info in feeder:
                consume(pipe.embed_result(slot))
                feeder.release(slot)

    With `depth` slots, batch k + 1 is decoded while batch k is on the GPU and batch k - 1 is being consumed (encoded):
    decoders and encoders such as cv2's release the interpreter lock, so the three really run at the same time.
    An exception on the thread is raised again in the consumer; leaving the `with` block stops the thread."""

    _END = object()

    def __init__(self, pipe, fill, submit):
        self._free: queue.Queue = queue.Queue()
        for slot in range(pipe.depth):
            self._free.put(slot)
        self._ready: queue.Queue = queue.Queue()
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, args=(pipe, fill, submit), name="svs-slot-feeder", daemon=True)
        self._thread.start()

    def _run(self, pipe, fill, submit):
        try:
            bind = getattr(pipe, "bind_thread", None)
            if bind is not None:
                bind()
            k = 0
            while True:
                slot = self._free.get()
                if slot is None or self._stop.is_set():
                    break
                n = fill(slot)
                if n <= 0:
                    break
                self._ready.put((slot, k, n, submit(slot, k, n)))
                k += 1
            self._ready.put(self._END)
        except BaseException as exc:       # handed to the consumer
            self._ready.put(exc)

    def __iter__(self):
        while True:
            item = self._ready.get()
            if item is self._END:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def release(self, slot: int) -> None:
        self._free.put(slot)

    def close(self) -> None:
        self._stop.set()
        self._free.put(None)
        self._thread.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_ahead(read, depth: int = 8):
    """Generator over the frames of `read() -> (ok, frame)` (cv2.VideoCapture.read), decoded up to `depth` frames ahead on a
    thread: the decoder runs while the consumer does something else with the previous frames (e.g. encodes them)."""
    frames: queue.Queue = queue.Queue(maxsize=max(1, depth))
    stop = threading.Event()
    end = object()

    def put(item):
        while not stop.is_set():
            try:
                frames.put(item, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def run():
        try:
            while not stop.is_set():
                ok, frame = read()
                if not ok:
                    break
                if not put(frame):
                    return
            put(end)
        except BaseException as exc:
            put(exc)

    thread = threading.Thread(target=run, name="svs-read-ahead", daemon=True)
    thread.start()
    try:
        while True:
            item = frames.get()
            if item is end:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        thread.join()

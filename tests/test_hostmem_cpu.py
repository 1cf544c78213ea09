"""CPU tier: host-side plumbing of the round-5 boundary that needs no GPU - the recycling pool of page-locked NumPy arrays
(svsdct/hostmem.py; the allocator is replaced by a stand-in here, the real one is hipHostMalloc behind svs_host_alloc) and the
zero-copy view of a '0'/'1' payload string (svsdct.batch._ascii_address)."""
import ctypes as C
import gc
import threading

import numpy as np
import pytest

from svsdct import batch, hostmem, native


class _FakeLib:
    """svs_host_alloc / svs_host_free over ordinary memory, counting what is outstanding"""

    def __init__(self):
        self.live = {}
        self.allocs = self.frees = 0
        self.refuse = False

    def svs_host_alloc(self, ref, size):
        if self.refuse:
            return native.SVS_ERR_HIP
        buf = C.create_string_buffer(size)
        ref._obj.value = C.addressof(buf)
        self.live[ref._obj.value] = buf
        self.allocs += 1
        return 0

    def svs_host_free(self, ptr):
        del self.live[ptr.value]
        self.frees += 1
        return 0


@pytest.fixture
def fake(monkeypatch):
    lib = _FakeLib()
    monkeypatch.setattr(native, "load", lambda: lib)
    hostmem.trim()
    for k in hostmem.stats:
        hostmem.stats[k] = 0
    assert hostmem._live_bytes == 0
    yield lib
    gc.collect()
    hostmem.trim()
    assert not lib.live            # everything handed out went back to the allocator


def test_arrays_are_ordinary_numpy_arrays_and_buffers_are_recycled(fake):
    a = hostmem.pinned_empty((3, 40, 72))
    assert a.shape == (3, 40, 72) and a.dtype == np.uint8 and a.flags.c_contiguous and a.flags.writeable
    a[:] = 7
    first = a.ctypes.data
    view = a[1, 5:9]
    del a
    gc.collect()
    assert fake.allocs == 1 and hostmem.stats["reused"] == 0
    b = hostmem.pinned_empty((3, 40, 72))                  # a view of the first array is alive: its buffer must not be handed out
    assert b.ctypes.data != first and fake.allocs == 2 and (view == 7).all()
    del view, b
    gc.collect()
    c = hostmem.pinned_empty((3, 40, 72))                  # both buffers are free now: no new allocation
    d = hostmem.pinned_empty(3 * 40 * 72 - 100)            # same 64 KB size class
    assert fake.allocs == 2 and hostmem.stats["reused"] == 2 and {c.ctypes.data, d.ctypes.data} >= {first}
    e = hostmem.pinned_empty((8, 8), np.float64)
    assert e.dtype == np.float64 and e.nbytes == 512
    src = np.arange(24, dtype=np.uint8).reshape(2, 3, 4)[:, ::2]      # not contiguous
    f = hostmem.pinned_copy(src)
    assert f.flags.c_contiguous and np.array_equal(f, src)
    assert hostmem.pinned_empty(0).size == 0


def test_a_frame_loop_cycles_through_a_few_buffers(fake):
    """what the reference's loop does with the operator's results (embed_process.py:117-128): every iteration drops the
    previous arrays - 200 iterations must not allocate 200 times"""
    keep = None
    for k in range(200):
        gray, stego = hostmem.pinned_empty((1080, 1920)), hostmem.pinned_empty((1080, 1920))
        if k == 0:
            keep = stego.copy()                                       # :123-124 keeps copies of the first frame only
        del gray, stego
    assert fake.allocs <= 4 and hostmem.stats["reused"] >= 396 and keep is not None


def test_the_pool_is_thread_safe_and_bounded(fake, monkeypatch):
    monkeypatch.setattr(hostmem, "_KEEP_BYTES", 1 << 20)
    errors = []

    def work():
        try:
            for _ in range(300):
                x = hostmem.pinned_empty(200_000)
                x[:10] = 1
        except Exception as exc:      # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    gc.collect()
    assert not errors
    assert sum(len(v) for v in hostmem._free.values()) * 262144 <= 1 << 20      # beyond the cap, buffers go back to the allocator
    assert fake.allocs == len(fake.live) + fake.frees


def test_ascii_address_is_the_strings_own_buffer():
    text = "0110" * 1000
    addr, n = batch._ascii_address(text)
    assert n == 4000 and C.string_at(addr, 16) == b"0110011001100110"
    again, _ = batch._ascii_address(text)
    assert again == addr                                                # no copy is made per call
    for bad in ("01€1", "0é1"):
        with pytest.raises(ValueError):
            batch._ascii_address(bad)


def test_beyond_the_cap_or_when_the_driver_refuses_arrays_are_pageable(fake, monkeypatch):
    """a caller that keeps every result (a list of all stego frames of a clip) must not exhaust page-locked memory: beyond
    _MAX_LIVE_BYTES, and whenever svs_host_alloc fails, pinned_empty hands out ordinary arrays - slower downloads, same results"""
    monkeypatch.setattr(hostmem, "_MAX_LIVE_BYTES", 3 * 65536)
    kept = [hostmem.pinned_empty(60000) for _ in range(5)]
    assert fake.allocs == 3 and hostmem.stats["pageable"] == 2
    assert all(a.shape == (60000,) and a.dtype == np.uint8 for a in kept)
    del kept
    gc.collect()
    again = hostmem.pinned_empty(60000)
    assert hostmem.stats["reused"] == 1 and fake.allocs == 3
    del again
    fake.refuse = True
    big = hostmem.pinned_empty((4, 70000))               # a size class with nothing in the pool: the driver says no
    assert big.shape == (4, 70000) and hostmem.stats["pageable"] == 3
    fake.refuse = False
    gc.collect()
    hostmem.trim()
    assert hostmem._live_bytes == 0

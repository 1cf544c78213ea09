"""Wire format of the payload inside the video (host side).

Layout written by the reference at embed_process.py:62-74 and parsed at
extract_process.py:89-165 - all integers big-endian, bits MSB first:

    16 b image width | 16 b image height
     8 b len | ephemeral public key (33 B compressed SECP256R1 point)
     8 b len | HKDF salt (16 B)
     8 b len | SHA3-256 of the plaintext (32 B)
     8 b len | AES-GCM nonce (12 B)
     8 b len | AES-GCM tag (16 B)
    32 b len | ciphertext

With the reference's field sizes the part before the ciphertext is 976 bits
(extract_process.py:51-53).  Here the stream is handled as a NumPy 0/1 array (what the
kernels' packed buffers unpack to) instead of a Python str of '0'/'1' characters.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

DIM_BITS = 16
HEADER_BITS_STANDARD = 2 * DIM_BITS + (8 + 264) + (8 + 128) + (8 + 256) + (8 + 96) + (8 + 128) + 32   # = 976


def _int_bits(value: int, width: int) -> np.ndarray:
    if value < 0 or value >= (1 << width):
        raise ValueError(f"Nilai {value} di luar jangkauan untuk {width} bit.")   # as int_ke_bitstream
    return np.array([(value >> (width - 1 - i)) & 1 for i in range(width)], np.uint8)


def _bytes_bits(data: bytes) -> np.ndarray:
    return np.unpackbits(np.frombuffer(bytes(data), np.uint8))


def build_payload_bits(width: int, height: int, eph_pub: bytes, salt: bytes, digest: bytes, nonce: bytes,
                       tag: bytes, ciphertext: bytes, dim_bits: int = DIM_BITS) -> np.ndarray:
    """Concatenate the fields in the reference's order (embed_process.py:69-74)."""
    if width >= (1 << dim_bits) or height >= (1 << dim_bits) or width < 0 or height < 0:
        raise ValueError(f"Dimensi gambar (lebar={width}, tinggi={height}) di luar jangkauan untuk {dim_bits}-bit.")
    parts = [_int_bits(width, dim_bits), _int_bits(height, dim_bits)]
    for field in (eph_pub, salt, digest, nonce, tag):
        parts += [_int_bits(len(field), 8), _bytes_bits(field)]
    parts += [_int_bits(len(ciphertext), 32), _bytes_bits(ciphertext)]
    return np.concatenate(parts)


@dataclass
class Header:
    width: int
    height: int
    eph_pub: bytes
    salt: bytes
    digest: bytes
    nonce: bytes
    tag: bytes
    ciphertext_len: int
    bits: int            # number of stream bits the header occupies (ciphertext starts here)


class HeaderIncomplete(ValueError):
    """Not enough bits yet for field `what`; `needed` = bits required to get past it."""

    def __init__(self, what: str, needed: int):
        super().__init__(f"Bit tidak cukup untuk {what}.")
        self.what = what
        self.needed = needed


class _Reader:
    def __init__(self, bits: np.ndarray):
        self.bits = np.asarray(bits, np.uint8)
        self.pos = 0

    def take(self, count: int, what: str) -> np.ndarray:
        if self.pos + count > self.bits.size:
            raise HeaderIncomplete(what, self.pos + count)
        out = self.bits[self.pos:self.pos + count]
        self.pos += count
        return out

    def integer(self, width: int, what: str) -> int:
        value = 0
        for b in self.take(width, what):
            value = (value << 1) | int(b)
        return value

    def blob(self, what: str) -> bytes:
        n = self.integer(8, f"panjang {what}")
        return np.packbits(self.take(8 * n, what)).tobytes()


def parse_header(bits: np.ndarray, dim_bits: int = DIM_BITS) -> Header:
    """Sequential parse in the order of extract_process.py:89-165.  Raises HeaderIncomplete when the
    stream is too short (the caller extracts more frames and retries) and ValueError for a 0x0 image."""
    rd = _Reader(bits)
    width = rd.integer(dim_bits, "metadata gambar")
    height = rd.integer(dim_bits, "metadata gambar")
    if width == 0 or height == 0:
        raise ValueError("Error: Metadata gambar 0x0.")
    eph_pub = rd.blob("kunci publik ECC pengirim")
    salt = rd.blob("salt HKDF")
    digest = rd.blob("hash gambar")
    nonce = rd.blob("nonce")
    tag = rd.blob("tag")
    ct_len = rd.integer(32, "panjang ciphertext")
    return Header(width, height, eph_pub, salt, digest, nonce, tag, ct_len, rd.pos)

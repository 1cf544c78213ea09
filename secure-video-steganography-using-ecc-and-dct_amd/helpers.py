"""Drop-in module: the reference's `helpers.py` (image <-> bit-string, 16+16-bit size metadata,
.avi path helper) with the same public names, return values and messages.  Host-side glue
(SURVEY section 2 row 4); conversions are vectorised with NumPy instead of per-pixel format()."""
from __future__ import annotations

import os

import numpy as np


def _bits_to_str(bits: np.ndarray) -> str:
    return (np.asarray(bits, np.uint8) + np.uint8(48)).tobytes().decode("ascii")


def gambar_ke_bitstream(path_gambar):
    """Open an image, convert to 8-bit gray ('L'), return (width, height, '0101...' of 8 bits per pixel,
    row-major, MSB first) - or (None, None, None) after printing the error (reference helpers.py:5-42)."""
    from PIL import Image
    try:
        gray = Image.open(path_gambar).convert("L")
        lebar, tinggi = gray.size
        bitstream = _bits_to_str(np.unpackbits(np.asarray(gray, np.uint8).reshape(-1)))
        print(f"Gambar '{path_gambar}' ({lebar}x{tinggi}) berhasil diubah jadi bitstream ({len(bitstream)} bits).")
        return lebar, tinggi, bitstream
    except FileNotFoundError:
        print(f"Error: File gambar '{path_gambar}' tidak ditemukan.")
    except Exception as exc:
        print(f"Error saat memproses gambar '{path_gambar}': {exc}")
    return None, None, None


def bitstream_ke_gambar(bitstream_gambar, lebar, tinggi):
    """'0101...' of exactly lebar*tinggi*8 bits -> PIL 'L' image, or None after printing the error
    (reference helpers.py:44-82)."""
    from PIL import Image
    try:
        expected = lebar * tinggi * 8
        if len(bitstream_gambar) != expected:
            print(f"Error: Panjang bitstream ({len(bitstream_gambar)}) tidak sesuai dengan dimensi yang diharapkan "
                  f"({expected} untuk {lebar}x{tinggi}x8bit).")
            return None
        bits = np.frombuffer(bitstream_gambar.encode("ascii"), np.uint8) - np.uint8(48)
        if bits.size and bits.max() > 1:
            raise ValueError("bitstream berisi karakter selain 0/1")
        pixels = np.packbits(bits).reshape((tinggi, lebar))
        image = Image.fromarray(pixels, mode="L")
        print(f"Bitstream berhasil diubah kembali menjadi gambar ({lebar}x{tinggi}).")
        return image
    except Exception as exc:
        print(f"Error saat mengubah bitstream menjadi gambar: {exc}")
        return None


def buat_metadata_bitstream(lebar, tinggi, bits_untuk_dimensi=16):
    """width then height, each as a fixed-width big-endian bit string (reference helpers.py:86-105)."""
    limit = 2 ** bits_untuk_dimensi
    if lebar >= limit or tinggi >= limit or lebar < 0 or tinggi < 0:
        raise ValueError(f"Dimensi gambar (lebar={lebar}, tinggi={tinggi}) di luar jangkauan untuk "
                         f"{bits_untuk_dimensi}-bit.")
    return format(lebar, f"0{bits_untuk_dimensi}b") + format(tinggi, f"0{bits_untuk_dimensi}b")


def parse_metadata_bitstream(bitstream_metadata, bits_untuk_dimensi=16):
    """Inverse of buat_metadata_bitstream; extra trailing bits are ignored (reference helpers.py:107-126)."""
    need = 2 * bits_untuk_dimensi
    if len(bitstream_metadata) < need:
        raise ValueError(f"Bitstream metadata terlalu pendek ({len(bitstream_metadata)} bits). "
                         f"Butuh minimal {need} bits.")
    return (int(bitstream_metadata[:bits_untuk_dimensi], 2),
            int(bitstream_metadata[bits_untuk_dimensi:need], 2))


def get_avi_path(base_path_or_full_path):
    """Output name the embed pipeline actually writes: extension replaced by .avi (reference helpers.py:184-187)."""
    return os.path.splitext(base_path_or_full_path)[0] + ".avi"

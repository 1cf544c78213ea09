"""Drop-in module: the evaluation helpers `app.py` imports (reference evaluation.py:9-26) plus the
frame / image comparison wrappers (:28-91).  Host-side; PSNR keeps the reference's uint8 arithmetic
(the squared difference wraps modulo 256 for uint8 inputs) so the GUI shows the numbers
it always showed; the parity harness uses the exact integer PSNR instead (cv2.PSNR definition)."""
from __future__ import annotations

import math

import numpy as np


def psnr(original, compressed):
    """20*log10(255 / sqrt(mean((a-b)**2))) evaluated in the inputs' dtype, inf for identical inputs."""
    mse = np.mean((original - compressed) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def calc_ssim(original, compressed):
    """skimage structural_similarity with data_range = max - min of the second image (reference :26)."""
    from skimage.metrics import structural_similarity
    return structural_similarity(original, compressed, data_range=compressed.max() - compressed.min())


def _nilai(psnr_val, batas_baik, label):
    if psnr_val > 30:
        return f"    {label}: {batas_baik} (PSNR > 30dB)"
    if psnr_val > 20:
        return f"    {label}: {'CUKUP' if batas_baik == 'BAIK' else 'BAIK'} (PSNR > 20dB)"
    return f"    {label}: KURANG (PSNR <= 20dB)"


def bandingkan_frame_video(frame_original, frame_stego):
    print("\n  [Evaluasi Kualitas Frame Video]")
    p, s = psnr(frame_original, frame_stego), calc_ssim(frame_original, frame_stego)
    print(f"    PSNR: {p:.2f} dB")
    print(f"    SSIM: {s:.4f}")
    print(_nilai(p, "BAIK", "Kualitas frame stego"))
    return p, s


def bandingkan_gambar(path_gambar_asli, path_gambar_ekstraksi):
    import cv2
    try:
        asli = cv2.imread(path_gambar_asli, cv2.IMREAD_GRAYSCALE)
        if asli is None:
            print(f"  Error: Tidak bisa membaca gambar asli '{path_gambar_asli}'")
            return None, None
        hasil = cv2.imread(path_gambar_ekstraksi, cv2.IMREAD_GRAYSCALE)
        if hasil is None:
            print(f"  Error: Tidak bisa membaca gambar ekstraksi '{path_gambar_ekstraksi}'")
            return None, None
        if asli.shape != hasil.shape:
            print(f"  Warning: Ukuran gambar berbeda. Asli {asli.shape}, Ekstraksi {hasil.shape}")
            hasil = cv2.resize(hasil, (asli.shape[1], asli.shape[0]))
        print("\n  [Evaluasi Kualitas Gambar Ekstraksi]")
        p, s = psnr(asli, hasil), calc_ssim(asli, hasil)
        print(f"    PSNR: {p:.2f} dB")
        print(f"    SSIM: {s:.4f}")
        print(_nilai(p, "SANGAT BAIK", "Kualitas ekstraksi"))
        return p, s
    except Exception as exc:
        print(f"  Error saat membandingkan gambar: {exc}")
        return None, None

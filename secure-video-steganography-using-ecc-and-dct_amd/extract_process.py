"""Drop-in module: `ekstraksi_gambar_video_final` with the reference's signature and return value
(reference extract_process.py:22-216); its frame loops (:55-86, :173-182) run on the GPU.

Phase 1 reads frames until at least 976 stream bits are available (the reference's header size,
:51-53,81), phase 2 parses the header on the host, phase 3 extracts - as ONE batched launch - exactly
the further frames the ciphertext needs.  Decryption, hash check and image reconstruction stay on the
host as in the reference (:186-214).
"""
from __future__ import annotations

import os

import numpy as np

import helpers as steg_helpers
from config_and_setup import (bytes_ke_bitstream, buat_shared_secret_ecdh, dekripsi_aes_gcm,
                              derive_kunci_aes_dari_shared_secret, deserialisasi_kunci_publik_ecc_compressed,
                              hitung_sha3_256, setup_kunci_ecc)  # noqa: F401
from svsdct import batch as _batch
from svsdct import framing as _framing
from svsdct.pipeline import FramePipeline, SlotFeeder

BATCH_FRAMES = int(os.environ.get("SVS_BATCH_FRAMES", "32"))
PIPELINE_DEPTH = int(os.environ.get("SVS_PIPELINE_DEPTH", "3"))     # batches in flight between decode and the kernels
_MIN_HEADER_BITS = _framing.HEADER_BITS_STANDARD        # 976 (extract_process.py:51-53)
# SVS_FUSED_COLOUR=1: bits are extracted straight from the colour frames (svs_extract_bgr_dev) when the device
# BGR -> gray reproduces this machine's cv2 (svsdct.colour); see embed_process.py
FUSED_COLOUR = os.environ.get("SVS_FUSED_COLOUR", "0") == "1"


def _cv2():
    import cv2
    return cv2


def _gagal(pesan, cap=None):
    print(f"  Error Kritis Ekstraksi: {pesan}")
    if cap is not None and cap.isOpened():
        cap.release()
    return False


def _extract_frames(frames, delta, n_ac, tabel_warna=None):
    if tabel_warna:                                            # frames are colour: convert + extract in one kernel
        packed, n_bits = _batch.extract_bgr_frames(np.stack(frames), delta, n_ac, weights=tabel_warna)
    else:
        packed, n_bits = _batch.extract_frames(np.stack(frames), delta, n_ac)
    return np.unpackbits(packed, count=n_bits)


def ekstraksi_gambar_video_final(path_stego_video, path_gambar_output,
                                 delta_kuantisasi, num_ac_coeffs,
                                 kunci_privat_ecc_penerima,
                                 bits_untuk_dimensi=16):
    """Recover, decrypt and save the secret image carried by a stego video.  -> True on success."""
    print("\n=== MEMULAI PROSES EKSTRAKSI GAMBAR DARI VIDEO ===")
    print(f"  Stego Video: '{path_stego_video}'")
    print(f"  Parameter: DELTA={delta_kuantisasi}, Koefisien AC per Blok={num_ac_coeffs}")
    cv2 = _cv2()
    cap = cv2.VideoCapture(path_stego_video)
    if not cap.isOpened():
        print(f"  Error: Tidak bisa membuka stego-video '{path_stego_video}'.")
        return False
    w_in, h_in = int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    w, h = (w_in // 8) * 8, (h_in // 8) * 8
    if w == 0 or h == 0:
        print("  Error: Dimensi video terlalu kecil.")
        cap.release()
        return False
    per_frame = _batch.capacity_bits(1, h, w, num_ac_coeffs)
    tabel_warna = None
    if FUSED_COLOUR:
        from svsdct import colour as _colour
        try:
            tabel_warna = _colour.weights_matching_cv2(cv2)
        except _colour.ColourMismatch as exc:
            print(f"  Info: jalur warna terfusi tidak dipakai ({exc}).")

    def baca_gray():
        ok, frame = cap.read()
        if not ok:
            return None
        return frame[0:h, 0:w] if tabel_warna else cv2.cvtColor(frame[0:h, 0:w], cv2.COLOR_BGR2GRAY)

    print("\n  [Tahap Ekstraksi 1: Membaca Bit Awal dari Video]")
    stream = np.zeros(0, np.uint8)
    frame_num = 0
    while stream.size < _MIN_HEADER_BITS:
        gray = baca_gray()
        frame_num += 1
        if gray is None:
            print(f"  Error: Video habis sebelum cukup bit diekstrak (setelah {frame_num - 1} frame).")
            cap.release()
            return False
        print(f"    Mengekstrak bit dari frame video ke-{frame_num}...")
        bits = _extract_frames([gray], delta_kuantisasi, num_ac_coeffs, tabel_warna)
        if bits.size == 0:
            print(f"  Error: Tidak ada bit diekstrak dari frame ke-{frame_num}.")
            cap.release()
            return False
        stream = np.concatenate([stream, bits])
        print(f"      Bit dari frame ini: {bits.size}. Total bit terkumpul: {stream.size}")

    print("\n  [Tahap Ekstraksi 2: Parsing Metadata dan Kunci]")
    try:
        hdr = _framing.parse_header(stream, bits_untuk_dimensi)
    except ValueError as exc:                                   # includes HeaderIncomplete
        return _gagal(str(exc), cap)
    print(f"    Metadata gambar diurai: Lebar={hdr.width}, Tinggi={hdr.height}")
    print(f"    Kunci Publik ECC Pengirim ({len(hdr.eph_pub)} bytes) diekstrak.")
    print(f"    Salt HKDF ({len(hdr.salt)} bytes) diekstrak.")
    try:
        pengirim = deserialisasi_kunci_publik_ecc_compressed(hdr.eph_pub)
        rahasia = buat_shared_secret_ecdh(kunci_privat_ecc_penerima, pengirim)
        kunci_aes = derive_kunci_aes_dari_shared_secret(rahasia, hdr.salt, 32)
        print("    Shared secret dan kunci AES berhasil diderivasi oleh penerima.")
    except Exception as exc:
        return _gagal(f"Error saat ECDH atau derivasi kunci AES penerima: {exc}", cap)
    print(f"    Hash SHA3-256 gambar dari stego ({len(hdr.digest)} bytes) diekstrak.")
    print(f"    Panjang Ciphertext diharapkan: {hdr.ciphertext_len} bytes.")

    butuh = 8 * hdr.ciphertext_len
    ct_bits = stream[hdr.bits:]
    if ct_bits.size < butuh:
        print(f"    Ciphertext belum lengkap ({ct_bits.size}/{butuh} bits). Melanjutkan ke frame berikutnya...")
        lagi = -(-(butuh - ct_bits.size) // max(per_frame, 1))      # frames still needed: known from the header
        pieces = [ct_bits]
        if tabel_warna:
            # fused colour path (opt-in): synchronous batches through svs_extract_bgr
            while lagi > 0:
                grays = []
                while len(grays) < min(BATCH_FRAMES, lagi):
                    gray = baca_gray()
                    if gray is None:
                        break
                    grays.append(gray)
                if not grays:
                    print("    Warning: Video selesai sebelum semua ciphertext diekstrak.")
                    break
                pieces.append(_extract_frames(grays, delta_kuantisasi, num_ac_coeffs, tabel_warna))
                lagi -= len(grays)
                frame_num += len(grays)
                print(f"    Sisa ciphertext diekstrak sampai frame {frame_num}. "
                      f"Total bit ciphertext terkumpul: {sum(p.size for p in pieces)}")
        else:
            # overlapped staging: batch k+1 is decoded (feeder thread) while batch k is copied to the GPU, extracted and
            # copied back and the bits of batch k-1 are unpacked here
            per_batch = max(1, min(BATCH_FRAMES, lagi))
            with FramePipeline(h, w, per_batch, delta_kuantisasi, num_ac_coeffs,
                               depth=max(1, min(PIPELINE_DEPTH, -(-lagi // per_batch))),
                               mode=_batch.host_level_mode()) as pipe:
                rencana = {"lagi": lagi}

                def isi(slot):
                    masukan, n = pipe.input(slot), 0
                    while n < min(per_batch, rencana["lagi"]):
                        gray = baca_gray()
                        if gray is None:
                            break
                        masukan[n] = gray
                        n += 1
                    rencana["lagi"] -= n
                    return n

                with SlotFeeder(pipe, isi, lambda slot, k, n: pipe.submit_extract(slot, n)) as feeder:
                    for slot, k, n, _ in feeder:
                        packed, n_bits = pipe.extract_result(slot)
                        pieces.append(np.unpackbits(packed, count=n_bits))
                        feeder.release(slot)
                        frame_num += n
                        print(f"    Sisa ciphertext diekstrak sampai frame {frame_num}. "
                              f"Total bit ciphertext terkumpul: {sum(p.size for p in pieces)}")
                if rencana["lagi"] > 0:                              # the reference warns when a read fails (:177)
                    print("    Warning: Video selesai sebelum semua ciphertext diekstrak.")
        ct_bits = np.concatenate(pieces)
    if ct_bits.size < butuh:
        print("  Ekstraksi GAGAL: Ciphertext tidak lengkap.")
        cap.release()
        return False
    ciphertext = np.packbits(ct_bits[:butuh]).tobytes()

    print("\n  [Tahap Ekstraksi 3: Dekripsi dan Verifikasi]")
    plaintext = dekripsi_aes_gcm(ciphertext, kunci_aes, hdr.nonce, hdr.tag)
    if plaintext is None:
        print("    Dekripsi GAGAL.")
        cap.release()
        return False
    print("    Dekripsi berhasil.")
    if hitung_sha3_256(plaintext) == hdr.digest:
        print("    Verifikasi Hash SHA3-256 BERHASIL: Gambar tidak korup.")
    else:                                                        # the reference only warns here (:201-202)
        print("    Verifikasi Hash SHA3-256 GAGAL: Gambar mungkin korup atau telah diubah!")

    print("\n  [Tahap Ekstraksi 4: Rekonstruksi Gambar]")
    gambar = steg_helpers.bitstream_ke_gambar(bytes_ke_bitstream(plaintext), hdr.width, hdr.height)
    if gambar is None:
        return _gagal("Gagal merekonstruksi gambar.", cap)
    try:
        gambar.save(path_gambar_output)
        print(f"    Gambar (SHA3-ECC-AES) berhasil diekstrak dan disimpan sebagai '{path_gambar_output}'.")
    except Exception as exc:
        return _gagal(f"Error simpan gambar: {exc}", cap)
    cap.release()
    print("--- Proses Ekstraksi (SHA3-ECC-AES) Selesai ---")
    return True


if __name__ == "__main__":
    # same hard-coded demo as the reference's __main__ (extract_process.py:219-272)
    stego = os.path.join("media/output", "stego_video_final.avi")
    keluaran = os.path.join("media/output", "gambar_hasil_ekstraksi.png")
    priv, _ = setup_kunci_ecc()
    if priv is not None and os.path.exists(stego):
        ok = ekstraksi_gambar_video_final(stego, keluaran, 20, 10, priv)
        print("PROSES EKSTRAKSI " + ("BERHASIL" if ok else "GAGAL"))
    else:
        print(f"Stego video '{stego}' atau kunci privat tidak tersedia.")

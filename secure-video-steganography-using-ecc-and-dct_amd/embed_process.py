"""Drop-in module: `embed_gambar_ke_video_final` with the reference's signature and return values
(reference embed_process.py:17-152), its frame loop (:108-144) run as batched GPU launches.

Host work kept on the host (north-star): secret image -> bits (helpers), SHA3 / ECDH / HKDF /
AES-GCM (config_and_setup), OpenCV decode + FFV1 encode.  GPU work: every frame that carries payload
goes through `svsdct.batch.embed_frames` in batches of SVS_BATCH_FRAMES frames - frame k of the clip
takes stream bits [k*cap, (k+1)*cap), exactly what the reference's per-frame calls hand out
(:116-128), so the stego frames are the ones a frame-by-frame loop would produce.
"""
from __future__ import annotations

import os

import numpy as np

import helpers as steg_helpers
from config_and_setup import (bitstream_ke_bytes, buat_pasangan_kunci_ecc, buat_shared_secret_ecdh,
                              derive_kunci_aes_dari_shared_secret, deserialisasi_kunci_publik_ecc_compressed,
                              enkripsi_aes_gcm, hitung_sha3_256, persiapkan_file_input,  # noqa: F401
                              serialisasi_kunci_publik_ecc_compressed, setup_kunci_ecc)  # noqa: F401
from svsdct import batch as _batch
from svsdct import framing as _framing
from svsdct.pipeline import FramePipeline, SlotFeeder, read_ahead

BATCH_FRAMES = int(os.environ.get("SVS_BATCH_FRAMES", "32"))
PIPELINE_DEPTH = int(os.environ.get("SVS_PIPELINE_DEPTH", "3"))     # batches in flight between decode and encode
# SVS_FUSED_COLOUR=1: colour frames go to the GPU as they are and BGR -> gray -> embed -> BGR runs as ONE kernel
# (svs_embed_bgr_dev) instead of cv2.cvtColor on the host either side of the operator (:117-126).  Only used when the
# device conversion reproduces this machine's cv2 bit for bit (svsdct.colour); otherwise the host conversion stays.
FUSED_COLOUR = os.environ.get("SVS_FUSED_COLOUR", "0") == "1"


def _cv2():
    import cv2
    return cv2


def _siapkan_payload(path_gambar_rahasia, kunci_publik_penerima_compressed):
    """Secret image -> framed, encrypted payload bits (reference :25-74).  Returns None after printing the
    reason when a stage fails, like the reference's early `return False, None, None` exits."""
    lebar, tinggi, bit_gambar = steg_helpers.gambar_ke_bitstream(path_gambar_rahasia)
    if bit_gambar is None:
        return None
    try:
        plaintext = bitstream_ke_bytes(bit_gambar)
    except ValueError as exc:
        print(f"  Error: Konversi bitstream gambar ke bytes gagal: {exc}")
        return None
    print("\n  [Tahap Embedding 1: Persiapan Kriptografi]")
    digest = hitung_sha3_256(plaintext)
    print(f"      Hash SHA3-256 ({len(digest)} bytes) dibuat.")
    try:
        eph_priv, eph_pub = buat_pasangan_kunci_ecc()
        penerima = deserialisasi_kunci_publik_ecc_compressed(kunci_publik_penerima_compressed)
        salt = os.urandom(16)
        kunci_aes = derive_kunci_aes_dari_shared_secret(buat_shared_secret_ecdh(eph_priv, penerima), salt, 32)
        eph_pub_bytes = serialisasi_kunci_publik_ecc_compressed(eph_pub)
        print("      Kunci AES berhasil diderivasi dari shared secret ECC.")
    except Exception as exc:
        print(f"    Error: Setup ECC atau derivasi kunci AES gagal: {exc}")
        return None
    try:
        ciphertext, nonce, tag = enkripsi_aes_gcm(plaintext, kunci_aes)
        print("      Gambar berhasil dienkripsi.")
    except Exception as exc:
        print(f"    Error: Enkripsi AES gagal: {exc}")
        return None
    print("\n  [Tahap Embedding 2: Membuat Payload Lengkap]")
    try:
        bits = _framing.build_payload_bits(lebar, tinggi, eph_pub_bytes, salt, digest, nonce, tag, ciphertext)
    except ValueError as exc:
        print(f"    Error: Gagal membuat payload: {exc}")
        return None
    print(f"    Total bit payload yang akan disisipkan: {bits.size} bits.")
    print(f"      - Metadata Gambar: {2 * _framing.DIM_BITS} bits (L:{lebar}, T:{tinggi})")
    print(f"      - Header kunci/salt/hash/nonce/tag: {_framing.HEADER_BITS_STANDARD - 2 * _framing.DIM_BITS - 32} bits")
    print(f"      - Info Ciphertext: {32 + 8 * len(ciphertext)} bits")
    return bits


def embed_gambar_ke_video_final(path_video_input, path_gambar_rahasia, path_video_output_base,
                                delta_kuantisasi, num_ac_coeffs,
                                kunci_publik_ecc_penerima_bytes_compressed):
    """Embed the (encrypted) secret image into the video.  -> (True, first_gray, first_stego) when the whole
    payload was embedded, else (False, None, None).  Output is always '<base>.avi', FFV1, colour frames."""
    print("\n=== MEMULAI PROSES EMBEDDING GAMBAR KE VIDEO ===")
    print(f"  Gambar Rahasia: '{path_gambar_rahasia}'")
    print(f"  Video Input: '{path_video_input}'")
    print(f"  Parameter: DELTA={delta_kuantisasi}, Koefisien AC per Blok={num_ac_coeffs}")

    payload = _siapkan_payload(path_gambar_rahasia, kunci_publik_ecc_penerima_bytes_compressed)
    if payload is None:
        return False, None, None
    total_bits = int(payload.size)

    print("\n  [Tahap Embedding 3: Menyisipkan Payload ke Frame Video]")
    cv2 = _cv2()
    cap = cv2.VideoCapture(path_video_input)
    if not cap.isOpened():
        print(f"    Error: Video input '{path_video_input}' tidak bisa dibuka.")
        return False, None, None
    w_in, h_in = int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
    fps = cap.get(cv2.CAP_PROP_FPS)
    out_w, out_h = (w_in // 8) * 8, (h_in // 8) * 8                  # crop to whole blocks (:94)
    if out_w == 0 or out_h == 0:
        print("    Error: Dimensi video terlalu kecil.")
        cap.release()
        return False, None, None
    path_out = steg_helpers.get_avi_path(path_video_output_base)
    writer = cv2.VideoWriter(path_out, cv2.VideoWriter_fourcc(*"FFV1"), fps, (out_w, out_h), isColor=True)
    if not writer.isOpened():
        print(f"    ERROR: Gagal VideoWriter FFV1 '{path_out}'.")
        cap.release()
        return False, None, None
    print(f"    Video output akan disimpan sebagai '{path_out}' (Codec: FFV1).")

    tabel_warna = None
    if FUSED_COLOUR:
        from svsdct import colour as _colour
        try:
            tabel_warna = _colour.weights_matching_cv2(cv2)
        except _colour.ColourMismatch as exc:
            print(f"    Info: jalur warna terfusi tidak dipakai ({exc}).")
    per_frame = _batch.capacity_bits(1, out_h, out_w, num_ac_coeffs)
    usable = per_frame if delta_kuantisasi > 0 else 0              # nothing can be embedded otherwise (:143-145)
    state = {"disisipkan": 0, "frame_num": 0, "first": None}

    def baca(n):
        """decode up to n frames, cropped; gray unless the fused colour path takes them as they are"""
        frames = []
        while len(frames) < n:
            ok, frame_bgr = cap.read()
            if not ok:
                break
            potong = frame_bgr[0:out_h, 0:out_w]
            frames.append(potong if tabel_warna else cv2.cvtColor(potong, cv2.COLOR_BGR2GRAY))
        return frames

    def tulis(gray_stack, stego_stack, stego_bgr=None):
        """encode one finished batch, one log line per frame as the reference prints them (:129)"""
        for k in range(len(stego_stack)):
            state["frame_num"] += 1
            bits_frame = min(usable, total_bits - state["disisipkan"])
            if state["frame_num"] == 1:
                state["first"] = (np.array(gray_stack[0]), np.array(stego_stack[0]))
            writer.write(stego_bgr[k] if stego_bgr is not None else cv2.cvtColor(stego_stack[k], cv2.COLOR_GRAY2BGR))
            state["disisipkan"] += bits_frame
            print(f"    Frame {state['frame_num']}: {bits_frame} bits disisipkan. "
                  f"Total disisipkan: {state['disisipkan']}/{total_bits}")

    # frames that carry payload: frame k takes stream bits [k*cap, (k+1)*cap) (:116-128), so their number is known now
    carrying = -(-total_bits // usable) if usable else None        # None: every frame is entered, nothing is consumed
    habis = False
    if tabel_warna:
        # fused colour path (opt-in): synchronous batches through svs_embed_bgr
        sisa = carrying
        while not habis and (sisa is None or sisa > 0):
            frames = baca(BATCH_FRAMES if sisa is None else min(BATCH_FRAMES, sisa))
            if not frames:
                habis = True
                break
            stego_bgr, gray, used = _batch.embed_bgr_frames(np.stack(frames), delta_kuantisasi, num_ac_coeffs, payload,
                                                            bit_offset=state["disisipkan"],
                                                            n_bits=total_bits - state["disisipkan"], weights=tabel_warna)
            expect = min(len(frames) * usable, total_bits - state["disisipkan"])
            if used != expect:
                raise RuntimeError(f"embed kernel consumed {used} bits, expected {expect}")
            tulis(gray, stego_bgr[..., 0], stego_bgr)
            if sisa is not None:
                sisa -= len(frames)
    else:
        # Overlapped staging: batch k+1 is decoded (feeder thread) while batch k is on the GPU (H2D copy,
        # kernel and D2H copy run asynchronously on the slot's stream) and batch k-1 is encoded (this thread); the payload is
        # uploaded once.
        per_batch = BATCH_FRAMES if carrying is None else max(1, min(BATCH_FRAMES, carrying))
        n_batches = PIPELINE_DEPTH if carrying is None else -(-carrying // per_batch)
        with FramePipeline(out_h, out_w, per_batch, delta_kuantisasi, num_ac_coeffs,
                           depth=max(1, min(PIPELINE_DEPTH, n_batches)), mode=_batch.host_level_mode()) as pipe:
            pipe.set_payload(payload)
            rencana = {"sisa": carrying}

            def isi(slot):
                """decode the next batch straight into the slot's pinned input (feeder thread)"""
                mau = per_batch if rencana["sisa"] is None else min(per_batch, rencana["sisa"])
                masukan, n = pipe.input(slot), 0
                while n < mau:
                    ok, frame_bgr = cap.read()
                    if not ok:
                        break
                    masukan[n] = cv2.cvtColor(frame_bgr[0:out_h, 0:out_w], cv2.COLOR_BGR2GRAY)
                    n += 1
                if rencana["sisa"] is not None:
                    rencana["sisa"] -= n
                return n

            def kirim(slot, k, n):
                offset = k * per_batch * usable
                used = pipe.submit_embed(slot, n, bit_offset=min(offset, total_bits))
                return used, min(n * usable, max(0, total_bits - offset))

            with SlotFeeder(pipe, isi, kirim) as feeder:
                for slot, k, n, (used, expect) in feeder:
                    if used != expect:
                        raise RuntimeError(f"embed kernel consumed {used} bits, expected {expect}")
                    tulis(pipe.input(slot)[:n], pipe.embed_result(slot))
                    feeder.release(slot)
    disisipkan = state["disisipkan"]
    selesai = usable > 0 and disisipkan >= total_bits
    if selesai:
        print("    Semua payload (SHA3-ECC-AES) berhasil disisipkan!")
        for frame_bgr in read_ahead(cap.read):                         # remaining frames: copied, in colour (:134-139);
            writer.write(frame_bgr[0:out_h, 0:out_w])                  # decoded on a thread while this one encodes
    else:
        print(f"    Warning: Video selesai sebelum semua payload ({total_bits} bits) disisipkan.")
    first_gray, first_stego = state["first"] if state["first"] else (None, None)
    cap.release()
    writer.release()
    if selesai:
        print(f"  Proses embedding (SHA3-ECC-AES) selesai. Video output: '{path_out}'.")
        return True, first_gray, first_stego
    print("  Proses embedding (SHA3-ECC-AES) selesai, namun TIDAK semua data berhasil disisipkan.")
    return False, None, None


if __name__ == "__main__":
    # same hard-coded demo as the reference's __main__ (embed_process.py:155-217)
    input_dir, output_dir = "media/input", "media/output"
    os.makedirs(output_dir, exist_ok=True)
    video_in = os.path.join(input_dir, "cover.mp4")
    gambar = os.path.join(input_dir, "ini_adalah_rahasia_grayscale.png")
    video_out = os.path.join(output_dir, "stego_video_final")
    DELTA, N_AC = 20, 10
    siap = persiapkan_file_input(input_dir, video_in, gambar)
    priv, pub_bytes = setup_kunci_ecc()
    if siap and pub_bytes:
        ok, g0, s0 = embed_gambar_ke_video_final(video_in, gambar, video_out, DELTA, N_AC, pub_bytes)
        if ok and g0 is not None:
            print(f"  PSNR Frame Pertama (Asli vs. Stego): {_cv2().PSNR(g0, s0):.2f} dB")
        print("PROSES EMBEDDING " + ("BERHASIL" if ok else "GAGAL"))

"""Drop-in module: same public names and call signatures as the reference's
`config_and_setup.py`, with the frame operator running on the MI355X.

What is GPU work here: `proses_frame_qim_dct` (reference config_and_setup.py:106-174) calls the
fused HIP kernels through `svsdct.batch` / libsvsdct.so.  Everything else in this module is host
glue the north-star keeps on the host (bit-string codecs :22-41, AES-GCM :44-70, ECDH/HKDF
:73-96, SHA3 :99-103, key files :177-216); it is written against the same third-party APIs
(`cryptography`, `PIL`, `cv2`) and imports them lazily so the operator works without them.
"""
from __future__ import annotations

import os

import numpy as np

from svsdct import batch as _batch

_HKDF_INFO = b"kunci aes untuk steganografi video"  # reference config_and_setup.py:94
_GCM_TAG_BYTES = 16


# ------------------------------------------------------------------------------------------
# bit-string codecs (reference :22-41) - vectorised, same results and same exceptions
# ------------------------------------------------------------------------------------------
def bytes_ke_bitstream(data_bytes) -> str:
    arr = np.frombuffer(bytes(data_bytes), np.uint8)
    return _batch.bits_to_str(np.unpackbits(arr))


def bitstream_ke_bytes(bitstream_data: str) -> bytes:
    whole = len(bitstream_data) - len(bitstream_data) % 8
    if whole != len(bitstream_data):
        bitstream_data = bitstream_data[:whole]
        if not bitstream_data:
            raise ValueError("Bitstream kosong setelah dipotong.")
    if not bitstream_data:
        return b""
    try:
        bits = _batch.str_to_bits(bitstream_data)
        clean = bits.max() <= 1
    except UnicodeEncodeError:
        clean = False
    if clean:
        return np.packbits(bits).tobytes()
    # something other than '0'/'1' in the string: byte by byte through int(), whose rules decide what is accepted
    # (' 0101010' is) and what the ValueError says (it quotes the whole 8-character group)
    out = bytearray()
    for start in range(0, len(bitstream_data), 8):
        out.append(int(bitstream_data[start:start + 8], 2))
    return bytes(out)


def int_ke_bitstream(nilai_int: int, jumlah_bit: int) -> str:
    if nilai_int < 0 or nilai_int >= (1 << jumlah_bit):
        raise ValueError(f"Nilai {nilai_int} di luar jangkauan untuk {jumlah_bit} bit.")
    return format(nilai_int, f"0{jumlah_bit}b")


def bitstream_ke_int(bitstream_nilai: str, jumlah_bit_diharapkan=None) -> int:
    if jumlah_bit_diharapkan and len(bitstream_nilai) != jumlah_bit_diharapkan:
        raise ValueError(f"Panjang bitstream {len(bitstream_nilai)} tidak sesuai.")
    if not bitstream_nilai:
        raise ValueError("String bit kosong.")
    return int(bitstream_nilai, 2)


# ------------------------------------------------------------------------------------------
# the frame operator (reference :106-174) - GPU
# ------------------------------------------------------------------------------------------
def _bgr_to_gray(frame_bgr: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(frame, COLOR_BGR2GRAY) (reference :112).  Uses OpenCV when it is installed so
    that the gray plane is the reference's; otherwise OpenCV 4's fixed-point BT.601 table
    (B 3735, G 19235, R 9798, + 2^14, >> 15 - the default of svs_bgr_to_gray_dev as well).  Parity of
    this branch is unpinned because cv2 is absent from the build image."""
    try:
        import cv2  # noqa: WPS433 (lazy on purpose)
    except ImportError:
        b = frame_bgr[..., 0].astype(np.uint32)
        g = frame_bgr[..., 1].astype(np.uint32)
        r = frame_bgr[..., 2].astype(np.uint32)
        return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)
    return cv2.cvtColor(frame_bgr, cv2.COLOR_BGR2GRAY)


def proses_frame_qim_dct(frame_bgr_input, mode, delta,
                         bit_payload_segment=None,
                         enable_debug_prints_extract=False,
                         num_ac_coeffs_to_use=63):
    """QIM embed / extract on the 8x8 block DCT of one frame, on the GPU.

    Same contract as the reference operator: 'embed' returns (gray_ref uint8[H,W],
    stego uint8[H,W], bits_embedded); 'extract' returns the '0'/'1' string of
    (H/8)(W/8)*min(n,63) bits; any other mode returns None; a frame that is neither 2-D nor
    3-channel raises ValueError("Format frame input tidak didukung.").
    Frame sides must be multiples of 8 (the reference's callers crop before calling)."""
    frame = np.asarray(frame_bgr_input)
    if frame.ndim == 3 and frame.shape[2] == 3:
        gray = _bgr_to_gray(frame)                      # a fresh array (:112)
    elif frame.ndim == 2:
        gray = frame                                    # the reference's `.copy()` (:114) is made below, where it is needed
    else:
        raise ValueError("Format frame input tidak didukung.")
    src = np.ascontiguousarray(gray, np.uint8)          # what the kernels read; the caller's own memory when it can be
    if mode == "embed":
        # the operator receives the whole remaining payload as a '0'/'1' string and reads at most the capacity; the library
        # takes the string as it is (svs_embed_str).  None / "" = nothing to embed: the frame is copied (:124-126).
        # The first return value is the gray frame BEFORE embedding as an array of its own (:113-114,172): for a 2-D input a
        # copy of the caller's frame, which the library makes while the GPU works - the upload reads the caller's own memory
        fresh = not (src is frame or np.may_share_memory(src, frame))
        if fresh:
            stego, used = _batch.embed_frames_str(src, delta, num_ac_coeffs_to_use, bit_payload_segment)
            return src, stego[0], used
        gray_ref, stego, used = _batch.embed_frames_str(src, delta, num_ac_coeffs_to_use, bit_payload_segment, want_gray=True)
        return gray_ref[0], stego[0], used
    if mode == "extract":
        return _batch.extract_frames_str(src, delta, num_ac_coeffs_to_use)
    return None


# ------------------------------------------------------------------------------------------
# host crypto glue (reference :44-103): thin wrappers over `cryptography`, imported lazily
# ------------------------------------------------------------------------------------------
def _crypto():
    from cryptography.exceptions import InvalidTag
    from cryptography.hazmat.primitives import hashes, serialization
    from cryptography.hazmat.primitives.asymmetric import ec
    from cryptography.hazmat.primitives.ciphers.aead import AESGCM
    from cryptography.hazmat.primitives.kdf.hkdf import HKDF
    return InvalidTag, hashes, serialization, ec, AESGCM, HKDF


def _check_aes_key(kunci):
    if len(kunci) not in (16, 24, 32):
        raise ValueError("Kunci AES harus 16, 24, atau 32 byte.")


def enkripsi_aes_gcm(data_bytes, kunci_aes_derived):
    """-> (ciphertext, 12-byte nonce, 16-byte tag)"""
    _check_aes_key(kunci_aes_derived)
    AESGCM = _crypto()[4]
    nonce = os.urandom(12)
    sealed = AESGCM(kunci_aes_derived).encrypt(nonce, data_bytes, None)
    if len(sealed) < _GCM_TAG_BYTES:
        raise ValueError("Hasil enkripsi AESGCM terlalu pendek.")
    return sealed[:-_GCM_TAG_BYTES], nonce, sealed[-_GCM_TAG_BYTES:]


def dekripsi_aes_gcm(ciphertext_bytes, kunci_aes_derived, nonce_bytes, tag_bytes):
    """-> plaintext, or None (after printing why) when authentication fails"""
    _check_aes_key(kunci_aes_derived)
    InvalidTag, _, _, _, AESGCM, _ = _crypto()
    try:
        return AESGCM(kunci_aes_derived).decrypt(nonce_bytes, ciphertext_bytes + tag_bytes, None)
    except InvalidTag:
        print("Error Dekripsi AES: Tag autentikasi tidak valid.")
    except Exception as exc:  # same catch-all as the reference
        print(f"Error Dekripsi AES lainnya: {exc}")
    return None


def buat_pasangan_kunci_ecc():
    ec = _crypto()[3]
    priv = ec.generate_private_key(ec.SECP256R1())
    return priv, priv.public_key()


def serialisasi_kunci_publik_ecc_compressed(public_key_ecc):
    serialization = _crypto()[2]
    return public_key_ecc.public_bytes(encoding=serialization.Encoding.X962,
                                       format=serialization.PublicFormat.CompressedPoint)


def deserialisasi_kunci_publik_ecc_compressed(public_key_bytes_compressed, kurva=None):
    ec = _crypto()[3]
    return ec.EllipticCurvePublicKey.from_encoded_point(kurva or ec.SECP256R1(), public_key_bytes_compressed)


def buat_shared_secret_ecdh(private_key_lokal_ecc, public_key_remote_ecc):
    ec = _crypto()[3]
    return private_key_lokal_ecc.exchange(ec.ECDH(), public_key_remote_ecc)


def derive_kunci_aes_dari_shared_secret(shared_secret_bytes, salt_bytes=None, panjang_kunci_aes_bytes=32):
    _, hashes, _, _, _, HKDF = _crypto()
    return HKDF(algorithm=hashes.SHA256(), length=panjang_kunci_aes_bytes, salt=salt_bytes,
                info=_HKDF_INFO).derive(shared_secret_bytes)


def hitung_sha3_256(data_bytes):
    import hashlib
    return hashlib.sha3_256(data_bytes).digest()


# ------------------------------------------------------------------------------------------
# key files and dummy inputs (reference :177-238)
# ------------------------------------------------------------------------------------------
_PRIV_PEM, _PUB_PEM = "bob_private_key.pem", "bob_public_key.pem"


def setup_kunci_ecc():
    """Load (or create on first use) the receiver's key pair in the working directory.
    -> (private_key, compressed_public_bytes) or (None, None)"""
    serialization = _crypto()[2]
    print("=" * 70)
    print("SETUP KUNCI ECC UNTUK STEGANOGRAFI VIDEO (SHA3-ECC-AES)")
    print("=" * 70)
    print("\n--- SETUP KUNCI ECC PENERIMA (BOB) ---")
    if not (os.path.exists(_PRIV_PEM) and os.path.exists(_PUB_PEM)):
        print("  Membuat pasangan kunci ECC baru untuk Penerima (Bob)...")
        priv, pub = buat_pasangan_kunci_ecc()
        try:
            with open(_PRIV_PEM, "wb") as fh:
                fh.write(priv.private_bytes(encoding=serialization.Encoding.PEM,
                                            format=serialization.PrivateFormat.PKCS8,
                                            encryption_algorithm=serialization.NoEncryption()))
            with open(_PUB_PEM, "wb") as fh:
                fh.write(pub.public_bytes(encoding=serialization.Encoding.PEM,
                                          format=serialization.PublicFormat.SubjectPublicKeyInfo))
            print(f"  Kunci ECC Bob berhasil dibuat dan disimpan ke '{_PRIV_PEM}' dan '{_PUB_PEM}'.")
        except Exception as exc:
            print(f"  Error saat menyimpan kunci ECC Bob: {exc}")
            return None, None
    else:
        print("  Menggunakan kunci ECC Bob yang sudah ada dari file.")
    try:
        with open(_PRIV_PEM, "rb") as fh:
            priv = serialization.load_pem_private_key(fh.read(), password=None)
        with open(_PUB_PEM, "rb") as fh:
            pub = serialization.load_pem_public_key(fh.read())
        print("  Kunci ECC Bob berhasil dimuat.")
        return priv, serialisasi_kunci_publik_ecc_compressed(pub)
    except Exception as exc:
        print(f"  Error saat memuat kunci ECC Bob: {exc}")
        return None, None


def persiapkan_file_input(input_dir, video_input_path, gambar_rahasia_path):
    """Create the reference's dummy inputs when missing: 32x32 'lightgray' L-mode PNG and a
    640x480, 24 fps, 120-frame uniform-noise mp4v clip.  -> True when both files exist."""
    os.makedirs(input_dir, exist_ok=True)
    if not os.path.exists(gambar_rahasia_path):
        try:
            from PIL import Image
            Image.new("L", (32, 32), color="lightgray").save(gambar_rahasia_path)
            print(f"  INFO: Gambar dummy '{gambar_rahasia_path}' (32x32) dibuat.")
        except Exception as exc:
            print(f"  ERROR: Gagal buat gambar dummy: {exc}")
    if not os.path.exists(video_input_path):
        try:
            import cv2
            writer = cv2.VideoWriter(video_input_path, cv2.VideoWriter_fourcc(*"mp4v"), 24.0, (640, 480))
            for _ in range(24 * 5):
                writer.write(np.random.randint(0, 256, (480, 640, 3), dtype=np.uint8))
            writer.release()
            print(f"  INFO: Video dummy '{video_input_path}' dibuat. Jalankan lagi.")
        except Exception as exc:
            print(f"  ERROR: Gagal buat video dummy: {exc}")
    return os.path.exists(video_input_path) and os.path.exists(gambar_rahasia_path)

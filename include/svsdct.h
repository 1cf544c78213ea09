/*
 * svsdct.h - C ABI of libsvsdct.so: the MI355X (gfx950) implementation of the per-frame
 * 8x8 block-DCT / QIM embed-and-extract operator.
 *
 * What it replaces in the reference (erc-a/Secure-Video-Steganography-using-ECC-and-DCT):
 *   proses_frame_qim_dct(frame, mode, delta, bit_payload_segment, ..., num_ac_coeffs_to_use)
 *       config_and_setup.py:106-174      (the operator: mode 'embed' -> svs_embed*, 'extract' -> svs_extract*)
 *   its two frame loops, batched:
 *       embed_process.py:108-128         (frame k takes bits [k*cap, (k+1)*cap) of the stream)
 *       extract_process.py:55-86,173-182 (per-frame bit strings concatenated in frame order)
 * The reference has no FFI of its own (it is pure Python); the binding a maintainer adds is the
 * ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns SVS_OK (0) or a negative SVS_ERR_* code and
 *     never throws; svs_last_error() gives the message for the calling thread.
 *   - the caller owns every buffer.  `*_dev` entry points take DEVICE pointers and a
 *     hipStream_t (as void*; NULL = the null stream), enqueue work and return without
 *     synchronising.  The entry points without the suffix take HOST pointers, stage through
 *     device memory and return when the result is in the host buffers: the frames travel in
 *     chunks over the upload and the download stream of a per-thread staging context (upload of
 *     chunk k+1 beside the download of chunk k: the link is full duplex), nothing is allocated per call once the context has grown to the
 *     largest call, and a page-locked caller buffer (svs_host_alloc) is the DMA's own
 *     source / target - pageable memory goes through the HIP runtime's staging (uploads at
 *     the same rate, downloads at about half of it: prefer page-locked OUTPUT buffers).
 *   - frames are gray uint8 planes, H and W multiples of 8 (the reference's callers crop:
 *     embed_process.py:94,113; extract_process.py:34,62), laid out [frame][row][col] with byte
 *     pitches given in svs_planes.
 *   - payload bits are packed MSB-first (numpy.packbits order): stream bit i is bit 7-(i%8) of
 *     byte i/8.  Stream bit i of a batch belongs to global block i / n_ac (frames in order,
 *     blocks in raster order inside a frame) and flat row-major coefficient 1 + i % n_ac of
 *     that block (config_and_setup.py:139-140) - so numpy.unpackbits of the extract output is
 *     the reference's '0'/'1' string.
 *   - n_ac is clamped to [0, 63] as the reference does (config_and_setup.py:138).
 *   - delta is passed as double: the quantiser divides in float32 by (float)delta and
 *     requantises with q*delta rounded once to float32, which is what the reference's
 *     `int(round(c / delta))` / `float(q * delta)` do for python int/float delta under
 *     NumPy >= 2 (config_and_setup.py:148,156,160).
 */
#ifndef SVSDCT_H
#define SVSDCT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVS_ABI_VERSION 4

#define SVS_OK 0
#define SVS_ERR_INVALID_ARG (-1)  /* bad geometry / NULL pointer / size overflow */
#define SVS_ERR_HIP (-2)          /* a HIP runtime call failed; see svs_last_error() */
#define SVS_ERR_NO_DEVICE (-3)    /* no usable AMD GPU */
#define SVS_ERR_CAPACITY (-4)     /* an output buffer is too small */

/* `flags` of the embed / extract entry points.  EVERY value gives the reference's stego pixels and extracted bits, bit for
 * bit; the flags only choose between two kernel families.
 *   0 / SVS_EXACT_GUARDED  (identical in behaviour; SVS_EXACT_GUARDED is kept as a named value for ABI compatibility and is
 *                      what the Python layer passes by default.)  Embedding with n_ac <= 15 and 0.25 <= delta <= 4096 runs the
 *                      STREAMING kernel (csrc/svs_device.hpp): every block goes HBM -> registers -> HBM once; the kernel
 *                      computes the payload coefficients exactly as pocketfft does (every quantiser decision is the
 *                      reference's), predicts each stego pixel from the sparse inverse of the coefficient changes, and keeps
 *                      the prediction only where a RIGOROUS per-block bound on the reference's own float32 round-trip noise
 *                      (tools/guard_bound.py: running error analysis of every pocketfft operation; BETA = u (17.0 mean +
 *                      31.05 ||block - mean||_2 + KD (1.5 delta + 0.01)) + 2^-20, KD = 19.6 for n_ac <= 7, 54.8 for
 *                      n_ac <= 15, and at least 2^-14 for n_ac = 8..15) proves the truncation cannot differ
 *                      (config_and_setup.py:166-171 transforms every block forth and back and truncates: x - 1e-5 becomes
 *                      x - 1).  The blocks it cannot decide (n_ac <= 7: 8 tests per block, 0.05 - 1.7 % of the blocks, 12.5 %
 *                      of flat ones at n = 3; n_ac = 8..15: 64 tests with position-dependent bounds, 1.5 - 13 %) are redone
 *                      INSIDE the launch with the pocketfft-identical arithmetic (eight lanes per block, LDS worklist
 *                      private to the wave) - no second launch, no scratch memory.  Every other embed call (n_ac >= 16, delta
 *                      outside the range) runs the SVS_EXACT_POCKETFFT kernel.  Extraction: n_ac <= 7 the
 *                      pocketfft-identical forward transform; n_ac >= 8 an FMA-factored transform, a block with a quantiser
 *                      input within a proven error bound of a rounding tie being recomputed with the pocketfft-identical
 *                      one - the reference's bits for ANY input frame.
 *   SVS_EXACT_POCKETFFT  every float32 operation of scipy.fftpack.dct/idct(norm='ortho') (pocketfft) is replayed in order, on
 *                      all 64 coefficients of every block, one lane per block.  About 5x the arithmetic of the streaming
 *                      kernel: VALU-bound (0.33 - 0.42 of the HBM roofline).  The yardstick the other kernels are tested
 *                      against.
 * The *_dev entry points keep no state at all; the host-pointer entry points keep a per-thread staging context (below).
 * Every entry point is re-entrant and thread-safe.  The library reads no environment variable. */
#define SVS_EXACT_POCKETFFT 1u
#define SVS_EXACT_GUARDED 2u

/* Geometry of a batch of gray planes. */
typedef struct svs_planes {
    int32_t n_frames;
    int32_t height;      /* multiple of 8 */
    int32_t width;       /* multiple of 8 */
    int32_t reserved;    /* set to 0 */
    int64_t row_pitch;   /* bytes between rows, >= width, multiple of 8 */
    int64_t frame_pitch; /* bytes between frames, >= height*row_pitch, multiple of 8 */
} svs_planes;

/* ---- library / device ------------------------------------------------------------------ */
int svs_abi_version(void);
const char *svs_last_error(void);
int svs_device_count(int *count);
/* Select the device for the calling thread (hipSetDevice) and check it is usable. */
int svs_init(int device);
/* Name of the architecture the device reports, e.g. "gfx950". */
int svs_device_arch(int device, char *buf, size_t buf_len);
/* Releases the CALLING THREAD's staging context of the host-pointer entry points (svs_embed, svs_extract, svs_embed_bgr,
 * svs_extract_bgr, svs_embed_str, svs_extract_str): two streams and device buffers sized by the calls the thread makes (a
 * buffer above 64 MB of which eight calls in a row used less than a quarter is given back).  A thread's context is also
 * released when the thread exits; calling any host-pointer entry point afterwards simply builds a new one.  The *_dev entry
 * points keep nothing. */
int svs_shutdown(void);

/* ---- device memory / stream helpers for callers that do not bring their own ------------- */
int svs_malloc(void **dev_ptr, size_t bytes);
int svs_free(void *dev_ptr);
int svs_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int svs_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int svs_memset(void *dev_dst, int value, size_t bytes, void *stream);
int svs_stream_synchronize(void *stream);
/* Streams and pinned (page-locked) host memory for callers that overlap frame I/O with the kernels
 * (svsdct/pipeline.py): copies to/from pinned memory run asynchronously and at full link rate. */
int svs_stream_create(void **stream);
int svs_stream_destroy(void *stream);
int svs_host_alloc(void **host_ptr, size_t bytes);
int svs_host_free(void *host_ptr);

/* ---- capacity arithmetic ------------------------------------------------------------------ */
/* bits one batch carries: n_frames * (H/8) * (W/8) * clamp(n_ac, 0, 63) */
uint64_t svs_capacity_bits(const svs_planes *p, int n_ac);
/* bytes svs_extract* writes for that many bits: ceil(bits / 8) */
uint64_t svs_packed_bytes(uint64_t n_bits);

/* ---- the operator: embed -----------------------------------------------------------------
 * Replaces mode 'embed' of proses_frame_qim_dct (config_and_setup.py:117-172) for a whole batch
 * and the offset bookkeeping of embed_process.py:116-128.
 *   gray / stego : [n_frames] planes described by `planes` (same geometry for both); may alias.
 *   bits_packed  : MSB-first packed payload; the first stream bit used is `bit_offset`
 *                  (lets every rank index one shared buffer by its frame offset); the buffer
 *                  must be 4-byte aligned and readable up to a multiple of 4 bytes.
 *   n_bits       : bits available from bit_offset on.  min(n_bits, capacity) are embedded.
 *                  Blocks past the budget are copied byte-identically; a block the budget ends
 *                  in has only its first coefficients modified (config_and_setup.py:130,132,141).
 *   n_embedded   : (host) receives min(n_bits, capacity); 0 when delta <= 0 or n_ac <= 0.
 * delta <= 0 or n_ac <= 0: nothing can be embedded; as in the reference every block is still transformed forth and
 * back when n_bits > 0 (config_and_setup.py:143-145,166-169) - both modes run the exact arithmetic for that.
 * No scratch memory and no state between calls: concurrent calls from several host threads or on several streams are
 * independent (work on one stream is ordered as usual).
 */
int svs_embed_dev(const uint8_t *d_gray, uint8_t *d_stego, const svs_planes *planes,
                  double delta, int n_ac,
                  const uint8_t *d_bits_packed, uint64_t bit_offset, uint64_t n_bits,
                  uint32_t flags, uint64_t *n_embedded, void *stream);

int svs_embed(const uint8_t *gray, uint8_t *stego, const svs_planes *planes,
              double delta, int n_ac,
              const uint8_t *bits_packed, uint64_t bit_offset, uint64_t n_bits,
              uint32_t flags, uint64_t *n_embedded);

/* The same call in the reference operator's own types (config_and_setup.py:106-109,172): the payload is `bit_payload_segment`,
 * a string of '0' / '1' characters (one character per bit, no terminator needed), and the operator's first return value - the
 * gray frame before embedding, as an array of its own (:113-114) - is produced as well.
 *   bits_ascii, n_chars : n_chars characters are available, min(n_chars, capacity) are read (each must be '0' or '1':
 *                         SVS_ERR_INVALID_ARG otherwise - other characters have no pinned meaning in the reference) - a frame loop may hand over the
 *                         whole remaining payload as the reference does (embed_process.py:116-121) - and are packed on the
 *                         device.  n_chars > 0 with nothing embeddable (delta <= 0, n_ac <= 0) still round-trips every block,
 *                         n_chars = 0 (or NULL) copies the frames, as in the reference (:124-126).
 *   gray_ref_out        : NULL, or a buffer of the planes' geometry that receives a copy of `gray` (pixel bytes only): the
 *                         frames BEFORE embedding, also when stego aliases gray (in-place embedding - the copy is then made
 *                         before the first download; otherwise the calling thread makes it while the GPU works).  It may be
 *                         `gray` itself (no copy); it must not overlap `stego`, nor overlap `gray` partially
 *                         (SVS_ERR_INVALID_ARG).  The up / down overlap of the call needs page-locked stego memory
 *                         (svs_host_alloc): a pageable download blocks the calling thread. */
int svs_embed_str(const uint8_t *gray, uint8_t *gray_ref_out, uint8_t *stego, const svs_planes *planes,
                  double delta, int n_ac, const char *bits_ascii, uint64_t n_chars,
                  uint32_t flags, uint64_t *n_embedded);

/* ---- the operator: extract ----------------------------------------------------------------
 * Replaces mode 'extract' (config_and_setup.py:159-165,173-174) for a whole batch and the
 * concatenation of extract_process.py:76,181.
 *   bits_packed_out : receives svs_packed_bytes(capacity) bytes (the last byte zero padded);
 *                     must be 4-byte aligned; out_capacity_bytes is its size.
 *   n_bits_out      : (host) receives the capacity in bits.
 * delta <= 0: every bit is 0 (config_and_setup.py:143-145).
 */
int svs_extract_dev(const uint8_t *d_gray, const svs_planes *planes, double delta, int n_ac,
                    uint8_t *d_bits_packed_out, uint64_t out_capacity_bytes,
                    uint32_t flags, uint64_t *n_bits_out, void *stream);

int svs_extract(const uint8_t *gray, const svs_planes *planes, double delta, int n_ac,
                uint8_t *bits_packed_out, uint64_t out_capacity_bytes, uint32_t flags,
                uint64_t *n_bits_out);

/* Extraction into the reference operator's own return type: the '0' / '1' string of config_and_setup.py:173-174, one
 * character per bit (no terminator is written).  bits_ascii_out receives capacity characters; out_capacity_chars is its size. */
int svs_extract_str(const uint8_t *gray, const svs_planes *planes, double delta, int n_ac,
                    char *bits_ascii_out, uint64_t out_capacity_chars, uint32_t flags,
                    uint64_t *n_bits_out);

/* ---- colour plumbing around the operator (device resident) --------------------------------------------
 * Interleaved 8-bit BGR frames [frame][row][col][3] <-> gray planes.  bgr_row_pitch / bgr_frame_pitch in bytes,
 * multiples of 4; BGR base pointers 4-byte aligned.
 * svs_bgr_to_gray_dev replaces cv2.cvtColor(frame, COLOR_BGR2GRAY) (config_and_setup.py:112): OpenCV's fixed-point
 * (B*wb + G*wg + R*wr + 2^(shift-1)) >> shift.  weights = {wb, wg, wr, shift}; NULL = {3735, 19235, 9798, 15}
 * (OpenCV 4's 15-bit table; older builds use {1868, 9617, 4899, 14}).  Parity with cv2 is UNPINNED (cv2 is not
 * available in the build image) - a caller that has cv2 should compare once at start-up.
 * svs_gray_to_bgr_dev replaces cv2.cvtColor(stego, COLOR_GRAY2BGR) (embed_process.py:126): B = G = R = gray. */
int svs_bgr_to_gray_dev(const uint8_t *d_bgr, int64_t bgr_row_pitch, int64_t bgr_frame_pitch, uint8_t *d_gray,
                        const svs_planes *planes, const uint32_t *weights, void *stream);
int svs_gray_to_bgr_dev(const uint8_t *d_gray, const svs_planes *planes, uint8_t *d_bgr, int64_t bgr_row_pitch,
                        int64_t bgr_frame_pitch, void *stream);

/* Fused colour path: BGR frames in, stego BGR frames out, one pass (3 B/pixel read + 3 B/pixel written; the frame
 * loop's cvtColor -> operator -> cvtColor at embed_process.py:117-127 for the frames that carry payload).
 *   d_bgr_in / d_bgr_out : interleaved 8-bit BGR [frame][row][col][3]; pitches in bytes (multiples of 8; base
 *                          pointers 8-byte aligned); may alias when the pitches are equal.
 *   d_gray_ref           : optional (NULL to skip) gray planes, geometry `planes`: the gray frame BEFORE embedding,
 *                          i.e. the operator's first return value (config_and_setup.py:112,172).
 *   planes               : n_frames / height / width of the clip and the pitches of d_gray_ref.
 *   weights, flags, bits : as svs_bgr_to_gray_dev / svs_embed_dev.  Every block of these frames is written (gray
 *                          replicated into B, G, R); pass only the frames that carry payload - the reference copies
 *                          the remaining frames in colour (embed_process.py:134-139). */
int svs_embed_bgr_dev(const uint8_t *d_bgr_in, int64_t in_row_pitch, int64_t in_frame_pitch,
                      uint8_t *d_bgr_out, int64_t out_row_pitch, int64_t out_frame_pitch,
                      uint8_t *d_gray_ref, const svs_planes *planes, const uint32_t *weights,
                      double delta, int n_ac, const uint8_t *d_bits_packed, uint64_t bit_offset, uint64_t n_bits,
                      uint32_t flags, uint64_t *n_embedded, void *stream);
/* Extract straight from interleaved BGR frames (gray computed on the fly, pocketfft-identical forward transform:
 * the bits equal svs_extract_dev(..., SVS_EXACT_POCKETFFT) of svs_bgr_to_gray_dev's output). */
int svs_extract_bgr_dev(const uint8_t *d_bgr, int64_t bgr_row_pitch, int64_t bgr_frame_pitch,
                        const svs_planes *planes, const uint32_t *weights, double delta, int n_ac,
                        uint8_t *d_bits_packed_out, uint64_t out_capacity_bytes, uint64_t *n_bits_out, void *stream);

/* Host-pointer forms of the two calls above for tightly packed frames (BGR row pitch 3*width, gray row pitch width):
 * frames and payload are staged through device memory inside the call, which returns when the results are in the
 * caller's buffers.  bgr_out / gray_ref_out: [n_frames][height][width][3] resp. [n_frames][height][width];
 * gray_ref_out may be NULL.  `planes` must describe tightly packed gray planes (pitches width and height*width). */
int svs_embed_bgr(const uint8_t *bgr, uint8_t *bgr_out, uint8_t *gray_ref_out, const svs_planes *planes,
                  const uint32_t *weights, double delta, int n_ac, const uint8_t *bits_packed, uint64_t bit_offset,
                  uint64_t n_bits, uint32_t flags, uint64_t *n_embedded);
int svs_extract_bgr(const uint8_t *bgr, const svs_planes *planes, const uint32_t *weights, double delta, int n_ac,
                    uint8_t *bits_packed_out, uint64_t out_capacity_bytes, uint64_t *n_bits_out);

/* ---- measurement helpers (synthetic inputs and on-device checks for bench.py / tests) ------ */
/* value = lo + hash32(seed, first_frame + f, y, x) % span  - same hash as svsdct/synth.py */
int svs_fill_synthetic_dev(uint8_t *d_frames, const svs_planes *planes, uint32_t seed,
                           uint32_t first_frame, uint32_t lo, uint32_t span, void *stream);
/* packed Bernoulli(1/2) stream, bit i = lowbias32(seed*0x632BE5AB + first_bit + i) >> 31;
 * writes ceil(n_bits/8) bytes rounded up to a multiple of 4 (buffer must be that large). */
int svs_fill_bits_dev(uint8_t *d_bits_packed, uint64_t n_bits, uint32_t seed,
                      uint64_t first_bit, void *stream);
/* per-frame sum of squared differences (exact integers) -> d_sse[n_frames] (uint64, device);
 * PSNR = 10 log10(255^2 H W / sse)   (cv2.PSNR as used at embed_process.py:205, app.py:342) */
int svs_frame_sse_dev(const uint8_t *d_a, const uint8_t *d_b, const svs_planes *planes,
                      uint64_t *d_sse, void *stream);
/* mean SSIM per frame -> d_ssim[n_frames] (double, device), as skimage.metrics.structural_similarity with its
 * defaults for 2-D uint8 input (7x7 uniform window, K1 .01, K2 .03, sample covariance, float64) - the call
 * behind the reference's evaluation.calc_ssim (evaluation.py:21-26).  d_data_range[n_frames] (double, device)
 * gives skimage's data_range per frame; pass NULL to use the reference's quirk, max - min of frame b.
 * d_workspace: at least svs_ssim_workspace_bytes(planes) bytes of device memory (8-byte aligned).  H, W >= 7. */
uint64_t svs_ssim_workspace_bytes(const svs_planes *planes);
int svs_frame_ssim_dev(const uint8_t *d_a, const uint8_t *d_b, const svs_planes *planes,
                       const double *d_data_range, double *d_ssim, void *d_workspace, void *stream);
/* number of differing bits among the first n_bits of two packed streams -> *d_count (uint64,
 * device, overwritten) */
int svs_bit_errors_dev(const uint8_t *d_a_packed, const uint8_t *d_b_packed, uint64_t n_bits,
                       uint64_t *d_count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SVSDCT_H */

/*
 * include/sushi_hip.h -- C ABI of libsushi_hip.so (MI355X / gfx950 only)
 *
 * The reference (tp7/Sushi) has no FFI on this path: `wav.WavStream.find_substream`
 * (wav.py:177-188) calls `cv2.matchTemplate(..., cv2.TM_SQDIFF_NORMED)` (wav.py:185) and
 * `ndarray.argmin` (wav.py:186) in-process.  These entry points are what a ctypes binding
 * inside a drop-in `wav.py` binds instead of `import cv2`:
 *
 *   sushi_hip_prepare_stream   replaces the per-call work cv2 redoes on `self.data`
 *                              (CV_64F integral of the search image, templmatch.cpp
 *                              common_matchTemplate) by doing it once per WavStream
 *                              (wav.py:108-162 builds `self.data`; this runs right after).
 *   sushi_hip_match_batch      replaces wav.py:185-186 for a whole batch of
 *                              (pattern, window) pairs; one launch, one (index, score) per pair.
 *                              Direct form: exact-f32 MFMA sliding dot product, O(P*M).
 *   sushi_hip_prepare_spectra  block DFTs of the search stream, once per WavStream (cv2's crossCorr
 *                              recomputes them inside every matchTemplate call).
 *   sushi_hip_match_batch_fft  the same contract as sushi_hip_match_batch through overlap-save FFT
 *                              (what cv2's crossCorr does): f32 FFT scores for every position, then
 *                              the exact float64 evaluation of every position within `delta` of the
 *                              minimum; searches with too many near-ties are finished by a kernel that evaluates
 *                              every position (float64 for float32 streams, the MFMA kernel for uint8).
 *
 * Conventions: extern "C", plain pointers and sizes, no C++ or torch types.  Every pointer
 * named *_dev is a device (HBM) pointer owned by the caller for the duration of the call's
 * execution on `hip_stream`; nothing is retained.  `hip_stream` is a hipStream_t passed as
 * void* (NULL = the default stream).  Calls are asynchronous with respect to the host and
 * return 0 or a negative SUSHI_HIP_E* code; no exception crosses the boundary.
 */
#ifndef SUSHI_HIP_H
#define SUSHI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUSHI_HIP_ABI_VERSION 5

#if defined(__GNUC__)
#define SUSHI_HIP_API __attribute__((visibility("default")))
#else
#define SUSHI_HIP_API
#endif

/* error codes */
#define SUSHI_HIP_OK 0
#define SUSHI_HIP_EINVAL (-1)    /* bad argument (null pointer, negative size, unknown dtype/method/variant) */
#define SUSHI_HIP_EALIGN (-2)    /* a device pointer is not aligned as documented */
#define SUSHI_HIP_ELAUNCH (-3)   /* the HIP runtime rejected a launch / memset (see hipGetLastError) */
#define SUSHI_HIP_ENOSPACE (-4)  /* workspace too small */
#define SUSHI_HIP_ENODEV (-5)    /* no gfx950 device visible */

/* sample types of WavStream.data (wav.py:109: 'uint8' or 'float32') */
#define SUSHI_HIP_U8 0
#define SUSHI_HIP_F32 1

/* matching methods.  0 is what the reference uses (wav.py:185). */
#define SUSHI_HIP_SQDIFF_NORMED 0

/* One search = one call of WavStream.find_substream (wav.py:177-188) after its window
 * arithmetic: `pattern` is src.data[0, tmpl_off : tmpl_off + tmpl_len] and `search_source`
 * is dst.data[0, win_start : win_start + n_pos + tmpl_len - 1]; n_pos = result.shape[1].
 * first_tile = sum over previous searches of ceil(n_pos / tile_positions(variant)). */
typedef struct SushiHipSearch {
    int64_t tmpl_off;
    int64_t win_start;
    int32_t tmpl_len;
    int32_t n_pos;
    int32_t first_tile;   /* direct path: tiles of the searches before this one */
    int32_t first_pair;   /* FFT path: block pairs of the searches before this one (sushi_hip_fft_layout) */
    int32_t first_seg;    /* FFT path: template segments of the searches before this one */
    int32_t reserved;
} SushiHipSearch;         /* 40 bytes */

SUSHI_HIP_API int sushi_hip_abi_version(void);
SUSHI_HIP_API const char* sushi_hip_strerror(int code);

/* 0 if a gfx950 device is current, SUSHI_HIP_ENODEV otherwise. */
SUSHI_HIP_API int sushi_hip_device_ok(void);

/* Kernel variants differ only in how many result positions one workgroup owns. */
SUSHI_HIP_API int sushi_hip_variant_count(void);
SUSHI_HIP_API int sushi_hip_variant_tile_positions(int variant);

/* Stream preparation.  raw_dev: n samples of `dtype` (the row WavStream.data[0]).
 * Outputs: xc_dev[n] float32 = sample - centre (centre = 0.5 for float32 data in [0,1],
 * 128 for uint8: what the direct MFMA kernel multiplies); s1_dev[n+1] / s2_dev[n+1] float64 exclusive
 * prefix sums of the samples and their squares AS THEY ARE (the CV_64F integral cv2 builds per call; exact
 * for uint8); and the window energies in the cheap form the FFT path's scoring reads: urel_dev[n+1]
 * float32 and base_dev[0 .. nb] float64, nb = ceil(n / B), B = sushi_hip_fft_hop():
 *     s2[e] = sum_{e' < e} sample[e']^2 = base[e / B] + urel[e]          (e = 0 .. n)
 * (base_dev holds 2 * (nb + 1) doubles; the second half is scratch of this call.)
 * raw_dev stays the caller's: the FFT path reads it again (spectra, template spectra, exact refinement).
 * xc_dev must be 16-byte aligned; base_bytes >= sushi_hip_prepare_base_bytes(n). */
SUSHI_HIP_API size_t sushi_hip_prepare_base_bytes(int64_t n);
SUSHI_HIP_API double sushi_hip_centre(int dtype);
SUSHI_HIP_API int sushi_hip_prepare_stream(const void* raw_dev, int dtype, int64_t n,
                             float* xc_dev, double* s1_dev, double* s2_dev,
                             float* urel_dev, double* base_dev, size_t base_bytes, void* hip_stream);

/* Batched template match + arg-minimum.
 *   dst_* : prepared search stream (the WavStream find_substream is called on), dst_len samples
 *   src_* : prepared stream the patterns are slices of, src_len samples
 *   centre: the value subtracted by sushi_hip_prepare_stream for these streams' dtype
 *           (the cross terms are accumulated in float32 over xc = sample - centre: exact for uint8; for
 *           float32 the rounding is relative to sum |T - 0.5| |I - 0.5|, which is what makes the kernel at
 *           least as accurate as cv2 on WavStream data -- silence sits at the mid level -- but not on
 *           windows of samples near 0, where sum T I is small itself and float32(sample - 0.5) drops low
 *           bits; sushi_hip_match_batch_fft reads the samples themselves and has no such limit)
 *   searches_dev[n_search], ordered by first_tile; n_tiles = total tile count
 *   keys_ws_dev[n_search] : uint64 scratch
 *   out_idx_dev[n_search]   = result.argmin(axis=1)[0]        (wav.py:186)
 *   out_score_dev[n_search] = result[0][min_idx], float32     (wav.py:188)
 * Preconditions checked by the caller: 1 <= tmpl_len, 1 <= n_pos, tmpl_off + tmpl_len <= src_len,
 * win_start + n_pos + tmpl_len - 1 <= dst_len. */
SUSHI_HIP_API int sushi_hip_match_batch(const float* dst_xc_dev, const double* dst_s1_dev, const double* dst_s2_dev, int64_t dst_len,
                          const float* src_xc_dev, const double* src_s1_dev, const double* src_s2_dev, int64_t src_len,
                          double centre, int method,
                          const SushiHipSearch* searches_dev, int n_search, int n_tiles, int variant,
                          uint64_t* keys_ws_dev, int32_t* out_idx_dev, float* out_score_dev,
                          void* hip_stream);

/* ---- overlap-save FFT path ------------------------------------------------------------------
 * The destination stream is cut into blocks of sushi_hip_fft_hop() = B samples; block j is
 * stored as the 2B-point complex DFT of x[jB .. jB+2B) + i * x[(j+1)B .. (j+3)B) (x = the samples as they are; zeros past
 * the end), 2B complex float32 each, followed by one all-zero block:
 * sushi_hip_spectra_bytes(n) = (ceil(n/B) + 1) * 2B * 8 bytes.
 * A search covers the blocks floor(win_start/B) .. floor((win_start+n_pos-1)/B), two per
 * "pair", and its template is cut into ceil(tmpl_len/B) segments. */
SUSHI_HIP_API int sushi_hip_fft_hop(void);
SUSHI_HIP_API int64_t sushi_hip_spectra_blocks(int64_t n);
SUSHI_HIP_API size_t sushi_hip_spectra_bytes(int64_t n);
SUSHI_HIP_API int sushi_hip_fft_layout(int64_t win_start, int32_t n_pos, int32_t tmpl_len,
                                       int32_t* n_pairs, int32_t* n_seg);
/* Workspace needed to run n_search searches with n_pairs block pairs and n_seg template segments in
 * total as ONE sub-batch (the call splits a batch into sub-batches that fit the workspace it is given;
 * more workspace = fewer, larger launches; the minimum is what the largest single search needs). */
SUSHI_HIP_API size_t sushi_hip_fft_workspace_bytes(int64_t n_pairs, int64_t n_seg, int64_t n_search);

/* How many sub-batches sushi_hip_match_batch_fft cuts these searches into for a workspace of ws_bytes
 * (>= 1), or a negative SUSHI_HIP_E* code (ENOSPACE: some search does not fit on its own). */
SUSHI_HIP_API int sushi_hip_fft_sub_batches(const SushiHipSearch* searches_host, int n_search, size_t ws_bytes);

/* Optional L2-friendly schedule of the inverse-transform workgroups: order_host[total pairs of the batch]
 * receives, per sub-batch (as cut for a workspace of ws_bytes), a permutation of the sub-batch's pairs in
 * which pairs scoring the same region of the destination stream run back to back on one XCD.  Host-side,
 * computed once per batch; upload it and pass it as pair_order_dev (or pass NULL: workgroup = pair). */
SUSHI_HIP_API int sushi_hip_fft_pair_order(const SushiHipSearch* searches_host, int n_search, size_t ws_bytes,
                             int32_t* order_host, int64_t order_len);

/* raw_dev: the stream's samples as they are (uint8 or float32, as given to sushi_hip_prepare_stream);
 * spec_dev: output (16-byte aligned). */
SUSHI_HIP_API int sushi_hip_prepare_spectra(const void* raw_dev, int dtype, int64_t n, void* spec_dev, size_t spec_bytes,
                                            void* hip_stream);

/* Same results as sushi_hip_match_batch.  Additional arguments:
 *   dst_raw_dev / src_raw_dev / dtype : the two streams' samples as they were given to
 *                       sushi_hip_prepare_stream (template spectra and the exact refinement read them)
 *   dst_urel_dev / dst_base_dev : the relative window-energy prefix and its block bases of the dst stream
 *   dst_spec_dev      : sushi_hip_prepare_spectra output for the dst stream
 *   searches_host     : the same n_search descriptors in host memory (read during the call only);
 *                       first_tile must be laid out for variant sushi_hip_variant_count()-1,
 *                       first_pair / first_seg as running sums of sushi_hip_fft_layout()
 *   delta             : score margin, > 2x the error of the f32 FFT scores.  |d corr| <~ eps * |T| * |B|, B the
 *                       2*hop-sample blocks the window lies in: in score units a few float32 epsilons times
 *                       |B| / |window|.  For WavStream data (samples in [0,1] around the mid level) that ratio is
 *                       <= 2 * sqrt(max(1, 2*hop / tmpl_len)); patterns shorter than 2048 samples, where it
 *                       grows, are finished by the fallback kernel (flags_dev) whatever the FFT stage says.  2e-5 leaves a
 *                       margin of ~8x on BASELINE-shaped batches (measured per search: keys_ws_dev below)
 *   ws_dev / ws_bytes : scratch, >= sushi_hip_fft_workspace_bytes(pairs, segments, 1) of the largest search
 *   keys_ws_dev       : uint64[2 * n_search] scratch
 *   pair_order_dev    : int32[total pairs] from sushi_hip_fft_pair_order for the SAME ws_bytes, or NULL
 *   keys_ws_dev       : on completion the float32 at byte offset 8 * (n_search + k) is |FFT score - exact score|
 *                       of search k's result position (0 for searches finished by a fallback kernel): the
 *                       measured error of the ranking stage, to be compared with delta / 2
 *   flags_dev         : int32[2 * n_search + 2] scratch; on completion flags[k] = 1 if search k had too
 *                       many near-ties (or a pattern shorter than 2048 samples) and was finished by a fallback
 *                       kernel that evaluates every position: float32 streams exactly in float64 from the samples
 *                       themselves, uint8 streams by the MFMA kernel (exact on integers); flags[n_search] = how many */
SUSHI_HIP_API int sushi_hip_match_batch_fft(const float* dst_xc_dev, const double* dst_s1_dev, const double* dst_s2_dev, int64_t dst_len,
                              const float* dst_urel_dev, const double* dst_base_dev, const void* dst_spec_dev,
                              const float* src_xc_dev, const double* src_s1_dev, const double* src_s2_dev, int64_t src_len,
                              const void* dst_raw_dev, const void* src_raw_dev, int dtype, int method,
                              const SushiHipSearch* searches_dev, const SushiHipSearch* searches_host,
                              int n_search, double delta,
                              void* ws_dev, size_t ws_bytes,
                              uint64_t* keys_ws_dev, int32_t* flags_dev, const int32_t* pair_order_dev,
                              int32_t* out_idx_dev, float* out_score_dev, void* hip_stream);

/* ---- WavStream.__init__ value pipeline on the GPU (wav.py:113-156 after the host-side RIFF decode) ----
 * sushi_hip_load_resample : data[pad + u] = nearest-neighbour decimation of the one-second chunks of the
 *     downmixed frames (cv2.resize INTER_NEAREST, wav.py:125-137: sx = min(floor(x * scale), chunk - 1) in
 *     float64), zeros where the reference never writes, and both pads filled with the nearest inner sample
 *     (wav.py:140-141).  n_full chunks of `chunk` frames become nl_full samples each (scale_full =
 *     1 / (nl_full / chunk)); a last partial chunk of `rest` frames becomes nl_rest samples.
 * sushi_hip_load_histogram: 256-bin histogram (uint64) of key byte `shift/8` over the samples >= 0
 *     (side 0; key = bit pattern) or <= 0 (side 1; key = bit pattern of -x) whose key & mask == prefix:
 *     the building block of the exact radix select behind np.median(data[data >= 0]) (wav.py:145-146).
 * sushi_hip_load_normalise: in place clip to [lo, hi], subtract lo, divide by range (wav.py:148-151);
 *     if u8_dev != NULL also v * 255 + 0.5 truncated to uint8 (wav.py:153-156). */
SUSHI_HIP_API int sushi_hip_load_resample(const float* raw_dev, int64_t n_raw, int32_t chunk, int32_t nl_full, double scale_full,
                            int64_t n_full, int32_t rest, int32_t nl_rest, double scale_rest,
                            int64_t pad, int64_t total, float* data_dev, void* hip_stream);
SUSHI_HIP_API int sushi_hip_load_histogram(const float* data_dev, int64_t n, int side, uint32_t prefix, uint32_t mask, int shift,
                             uint64_t* hist_dev, void* hip_stream);
SUSHI_HIP_API int sushi_hip_load_normalise(float* data_dev, int64_t n, float lo, float hi, float range, uint8_t* u8_dev,
                             void* hip_stream);

/* Optional per-stage timing of sushi_hip_match_batch_fft with HIP events recorded on the launch
 * stream (bench.py's roofline figures).  Between _begin and _end every call records its stage
 * boundaries; _end waits for them and writes, per call, the milliseconds spent in each stage
 * (summed over the call's sub-batches) into stage_ms[call][SUSHI_HIP_NSTAGES].  Not thread safe. */
#define SUSHI_HIP_NSTAGES 5
#define SUSHI_HIP_STAGE_TSPEC 0    /* template-segment DFTs                          */
#define SUSHI_HIP_STAGE_MAC 1      /* frequency-domain multiply-accumulate           */
#define SUSHI_HIP_STAGE_IFFT 2     /* inverse DFTs + scoring epilogue                */
#define SUSHI_HIP_STAGE_REFINE 3   /* exact float64 evaluation of the candidates     */
#define SUSHI_HIP_STAGE_FINISH 4   /* direct-kernel fallback (if any) + unpack       */
SUSHI_HIP_API int sushi_hip_profile_begin(void);
SUSHI_HIP_API int sushi_hip_profile_end(float* stage_ms, int max_calls, int* n_calls);

#ifdef __cplusplus
}
#endif
#endif /* SUSHI_HIP_H */

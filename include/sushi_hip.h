/*
 * include/sushi_hip.h -- C ABI of libsushi_hip.so (MI355X / gfx950 only)
 *
 * The reference (tp7/Sushi) has no FFI on this path: `wav.WavStream.find_substream`
 * (wav.py:177-188) calls `cv2.matchTemplate(..., cv2.TM_SQDIFF_NORMED)` (wav.py:185) and
 * `ndarray.argmin` (wav.py:186) in-process.  These entry points are what a ctypes binding
 * inside a drop-in `wav.py` binds instead of `import cv2`:
 *
 *   sushi_hip_stream_create    once per WavStream, right after wav.py:108-162 has built `self.data`:
 *                              what cv2 redoes inside every matchTemplate call on the search image
 *                              (the CV_64F integral of common_matchTemplate, the block DFTs of crossCorr)
 *                              is done once per stream and kept in HBM.
 *   sushi_hip_batch_create     a batch of find_substream calls after their window arithmetic
 *                              (wav.py:178-184): one SushiHipRequest per call.
 *   sushi_hip_batch_run        replaces wav.py:185-186 for the whole batch: one (index, score) per request.
 *
 * Conventions: extern "C", plain pointers and sizes, no C++ or torch types.  Every pointer named *_dev is a
 * device (HBM) pointer owned by the caller; *_host pointers are host memory read during the call only.
 * The library allocates NO device memory: a handle is built inside a buffer the caller provides
 * (sushi_hip_*_bytes tells its size) and the caller keeps that buffer -- and, for a stream, the samples --
 * alive until the handle is destroyed.  `hip_stream` is a hipStream_t passed as void* (NULL = the default
 * stream).  Calls are asynchronous with respect to the host and return 0 or a negative SUSHI_HIP_E* code;
 * no exception crosses the boundary.  A handle may be used from one host thread at a time.
 */
#ifndef SUSHI_HIP_H
#define SUSHI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUSHI_HIP_ABI_VERSION 13  /* 10: band-split exclusion (low-band rows + row norms behind the spectra, SUSHI_HIP_EXCLUDE_BAND / _WHOLE,
                                     SushiHipBatchDiag.excluded_audited / .max_slb_ratio_excluded / .slb_violations / .band / .band_votes);
                                     11: SushiHipBatchDiag.second_look_audited (appended);
                                     12: sushi_hip_batch_set_bound_model (the excluded side of the pair exclusion is a worst-case bound by
                                     default), the band is |f| < N/8 strictly (bin 7N/8 of a low row is zero and counted with the rest), sushi_hip_batch_reset,
                                     sushi_hip_batch_workspace_view, SUSHI_HIP_ENOMEM / _EINTERNAL;
                                     13: SushiHipBatchInfo.lanes (appended): a large batch's sub-batches run side by side on HIP streams
                                     of the library's own, forked off and joined back into the stream a run is given;
                                     sushi_hip_batch_set_early_output, sushi_hip_device_prepare */

#if defined(__GNUC__)
#define SUSHI_HIP_API __attribute__((visibility("default")))
#else
#define SUSHI_HIP_API
#endif

/* error codes */
#define SUSHI_HIP_OK 0
#define SUSHI_HIP_EINVAL (-1)    /* bad argument (null pointer, negative size, unknown dtype/path/variant, request outside its stream) */
#define SUSHI_HIP_EALIGN (-2)    /* a device pointer is not aligned as documented */
#define SUSHI_HIP_ELAUNCH (-3)   /* the HIP runtime rejected a launch / copy / memset (see hipGetLastError) */
#define SUSHI_HIP_ENOSPACE (-4)  /* buffer or workspace too small */
#define SUSHI_HIP_ENODEV (-5)    /* no gfx950 device visible */
#define SUSHI_HIP_ENOMEM (-6)    /* the host ran out of memory while a plan was built (std::bad_alloc caught at the boundary) */
#define SUSHI_HIP_EINTERNAL (-7) /* any other C++ exception caught at the boundary: a bug, never a property of the input */

/* sample types of WavStream.data (wav.py:109: 'uint8' or 'float32') */
#define SUSHI_HIP_U8 0
#define SUSHI_HIP_F32 1

/* how a batch is matched.  Both give the same results (tests/test_gpu_parity.py). */
#define SUSHI_HIP_PATH_FFT 0     /* overlap-save FFT ranking + exact float64 evaluation of the near-minimum positions */
#define SUSHI_HIP_PATH_DIRECT 1  /* exact-f32 MFMA sliding dot product, O(P*M) */

/* what cv2.matchTemplate's `method` argument selects (and which extremum the caller takes) */
#define SUSHI_HIP_METHOD_SQDIFF_NORMED 0  /* cv2.TM_SQDIFF_NORMED + argmin: what wav.py:185-186 does; both paths */
#define SUSHI_HIP_METHOD_CCOEFF_NORMED 1  /* cv2.TM_CCOEFF_NORMED + argmax (BASELINE.json's wording); both paths */

SUSHI_HIP_API int sushi_hip_abi_version(void);
SUSHI_HIP_API const char* sushi_hip_strerror(int code);

/* 0 if a gfx950 device is current, SUSHI_HIP_ENODEV otherwise. */
SUSHI_HIP_API int sushi_hip_device_ok(void);
/* What a batch's first run would otherwise create on the current device, made NOW (v13): the HIP streams the lanes of large batches
 * run on (a stream is a hardware queue: ~5 ms each) and the pinned words a run's counts come back through.  Optional -- a process
 * that calls it while it is busy elsewhere (sushi_amd.device.warm_up does, behind the demux) takes 11 ms off its first large batch. */
SUSHI_HIP_API int sushi_hip_device_prepare(void);

/* ---- streams --------------------------------------------------------------------------------------------
 * raw_dev: the n samples of the row WavStream.data[0] (`dtype`), already in HBM; they stay the caller's and
 * are read again by every batch (pattern spectra, exact evaluation).  Built into mem_dev (256-byte aligned,
 * >= sushi_hip_stream_bytes(n, dtype, searchable) bytes):
 *   xc[n]      float32  sample - centre (0.5 for float32 data in [0,1], 128 for uint8): the direct path's operand
 *   s1[n+1], s2[n+1]  float64 exclusive prefix sums of the samples and of their squares AS THEY ARE (the CV_64F
 *              integral cv2 builds per call; exact for uint8)
 *   urel[n+1]  float32 + base[nb+1] float64, nb = ceil(n / B), B = sushi_hip_fft_block():
 *              s2[e] = base[e / B] + urel[e]   (the window energies in the cheap form the FFT scoring reads)
 *   usrel[n+1][2] float32 + base1[nb+1] float64: (urel[e], srel[e]) interleaved, srel = the same for s1 (the window sums of
 *              TM_CCOEFF_NORMED, which reads both prefixes with one 8-byte load per window end)
 *   spectra    (searchable streams only; sushi_hip_stream_add_spectra attaches them later)  for every block
 *              j = 0 .. nb-1 the N-point complex DFT, N = sushi_hip_fft_size(), H = N - B, of
 *                  (x - mean)[jB .. jB+N) + i * (x - mean)[jB+H .. jB+H+N)         (zeros past the end)
 *              as packed halves (float16 re, float16 im: 4 bytes per bin) times one power of two per stream, bin f at
 *              sushi_hip_fft_slot_of_bin(f), followed by one all-zero block:
 *              (nb + 1) * N * 4 bytes; behind them, for the band-split form of the pair exclusion:
 *              the LOW BAND (bins |f| < N/8; the slot of bin 7N/8 is zero) of every block spectrum once more, (nb + 1) * N bytes, in the order
 *              the bound's transform loads it, and float32[nb + 1]: the norm of each block spectrum's stored halves
 *              OUTSIDE that band.  sushi_hip_stream_spectra_bytes(n) is the sum.
 * A stream that is only a source of patterns does not need spectra. */
typedef struct SushiHipStream SushiHipStream;

SUSHI_HIP_API int sushi_hip_fft_size(void);      /* N: complex points per transform */
SUSHI_HIP_API int sushi_hip_fft_block(void);     /* B: samples per block = per pattern segment */
/* Where bin f (0 <= f < N) of a block spectrum sits inside its N stored complex values (4 bytes each): spectra are kept
 * in the order the inverse transform loads them (coalesced 16-byte loads), not in natural order.  -1 for an invalid bin. */
SUSHI_HIP_API int sushi_hip_fft_slot_of_bin(int bin);
/* The same for the low-band rows behind the spectra (bins f < N/8 and f >= 7N/8 only; N/4 complex values of 4 bytes per row):
 * their order is the one the band-split bound's transform loads them in.  -1 for a bin outside the band. */
SUSHI_HIP_API int sushi_hip_fft_low_slot_of_bin(int bin);
SUSHI_HIP_API double sushi_hip_centre(int dtype);
SUSHI_HIP_API size_t sushi_hip_stream_bytes(int64_t n, int dtype, int searchable);
SUSHI_HIP_API size_t sushi_hip_stream_spectra_bytes(int64_t n);
SUSHI_HIP_API int sushi_hip_stream_create(const void* raw_dev, int dtype, int64_t n, int searchable,
                                          void* mem_dev, size_t mem_bytes, void* hip_stream, SushiHipStream** out);
SUSHI_HIP_API int sushi_hip_stream_add_spectra(SushiHipStream* stream, void* mem_dev, size_t mem_bytes, void* hip_stream);
/* Where a part lives (for tests and diagnostics): which = SUSHI_HIP_VIEW_*; SPECTRA gives NULL / 0 before they exist. */
#define SUSHI_HIP_VIEW_XC 0
#define SUSHI_HIP_VIEW_S1 1
#define SUSHI_HIP_VIEW_S2 2
#define SUSHI_HIP_VIEW_UREL 3
#define SUSHI_HIP_VIEW_BASE 4
#define SUSHI_HIP_VIEW_SPECTRA 5
#define SUSHI_HIP_VIEW_USREL 6    /* float32[n+1][2]: (urel[e], srel[e]), s1[e] = base1[e / B] + srel[e] (TM_CCOEFF_NORMED's window sums on the FFT path) */
#define SUSHI_HIP_VIEW_BASE1 7    /* float64[nb+1] */
#define SUSHI_HIP_VIEW_COARSE 8   /* float64[2][n / 256 + 2]: s2, then s1, at every 256th sample (entries past the end: the totals) --
                                    what the FFT path's lower bound of a block pair's window energies reads */
#define SUSHI_HIP_VIEW_SPECTRA_LOW 9   /* the low-band rows behind the spectra (packed halves, N bytes per block) */
#define SUSHI_HIP_VIEW_ZNORM_REST 10   /* float32[nb + 1]: norm of each block spectrum outside the band */
SUSHI_HIP_API int sushi_hip_stream_view(const SushiHipStream* stream, int which, const void** ptr_dev, size_t* bytes);
SUSHI_HIP_API void sushi_hip_stream_destroy(SushiHipStream* stream);

/* ---- batches --------------------------------------------------------------------------------------------
 * One request = one call of WavStream.find_substream (wav.py:177-188) after its window arithmetic: `pattern` is
 * src.data[0, tmpl_off : tmpl_off + tmpl_len] and `search_source` is
 * dst.data[0, win_start : win_start + n_pos + tmpl_len - 1]; n_pos = result.shape[1]. */
typedef struct SushiHipRequest {
    int64_t tmpl_off;
    int64_t win_start;
    int32_t tmpl_len;
    int32_t n_pos;
} SushiHipRequest;        /* 24 bytes */

typedef struct SushiHipBatchInfo {
    int32_t n_search;
    int32_t path;
    int32_t variant;          /* direct path: tile-size variant in use */
    int32_t sub_batches;      /* FFT path: launches of each kernel per run (the batch is cut to fit the workspace, or -- a large batch --
                                 into parts that run side by side: `lanes`) */
    int64_t direct_tiles;
    int64_t fft_pairs;        /* FFT path: block pairs (inverse transforms) of the whole batch */
    int64_t fft_segments;     /* FFT path: pattern segments (forward transforms) of the whole batch */
    uint64_t workspace_bytes; /* of mem_bytes: scratch reused by the sub-batches */
    uint64_t mem_bytes;       /* what sushi_hip_batch_bytes returned */
    double flops;             /* 2 * P * M summed over the requests (the direct form's work) */
    double algorithmic_bytes; /* every search and pattern sample once + 8 bytes out per request */
    int32_t lanes;            /* FFT path: HIP streams the sub-batches of a run are spread over (1: all on the stream the run is given).
                                 Lane 0 IS that stream; the others are the library's own (one set per device, made by the first run
                                 that needs them or by sushi_hip_device_prepare), start behind the run's first launch and are joined
                                 before its last, so the caller sees one stream's ordering.  The stages of the path are bound by
                                 different things (stores, instruction issue, LDS, HBM reads): side by side they fill each other's gaps.
                                 SUSHI_HIP_LANES="subs:lanes" in the environment when the batch is created overrides the choice. */
    int32_t reserved;
} SushiHipBatchInfo;

typedef struct SushiHipBatchDiag {
    int32_t flagged;          /* searches with more near-minimum positions than the per-pair / per-search lists hold */
    int32_t all_positions;    /* of those: searches evaluated at every position (a candidate violated its error bound) */
    int64_t tiles_dense;      /* 1024-position tiles evaluated exactly at every position */
    int64_t tiles_sparse;     /* tiles evaluated exactly at listed candidate positions only */
    int64_t candidates;       /* listed candidate positions */
    float max_bound_ratio;    /* max over the exactly evaluated candidates of |f32 score - exact score| / the pair's
                                 modelled error bound (without the delta/2 floor); < 1 or the search went to all_positions */
    float max_bound_ratio_noncandidate; /* the same over the AUDITED positions: per search the audit runs (4 consecutive
                                 positions at a hashed place) of 4 hashed pairs spread over the window -- up to 16 positions that
                                 were NOT selected, evaluated exactly and held to the bound they were ranked with: the error
                                 model checked where it was not already believed.  A violation there sends the search to
                                 all_positions too.  The bound is a statistical model (8 standard deviations of the packed-half
                                 roundings), not a worst-case proof: this field is what watches it. */
    int64_t audited;          /* audited non-candidate positions of the run */
    int64_t pairs_transformed; /* FFT path: block pairs whose inverse transform was run and scored; the other
                                 info.fft_pairs - this were excluded by a lower bound of their scores */
    int64_t excluded_audited; /* of those: pairs the bound HAD excluded, transformed all the same -- per run one hashed pair of every
                                 audited search (all of them by default), a different one every run -- so that the lower bound is held
                                 to what such a pair really scores, not only to the pairs it let through */
    float max_slb_ratio_excluded; /* max over the audited excluded pairs of (the pair's lower bound) / (an upper bound of the exact
                                 score of its best position); < 1 or the search went to all_positions */
    int32_t slb_violations;   /* transformed pairs (audited or not) whose lower bound was found above a real score: 0 */
    int32_t band;             /* form of the exclusion the last run used (its last sub-batch that went through it): 0 whole rows,
                                 1 band-split, -1 none */
    int32_t suspended;        /* 1: AUTO left the exclusion out of the last run -- an earlier run of this batch had excluded next to
                                 nothing (searches without a match anywhere: no bound can help), so the passes that compute the bounds
                                 would only be overhead; every 64th run looks again */
    int32_t band_votes[2];    /* what AUTO / ALWAYS decided the form from (first run of a batch): block pairs looked at (all of them; those of
                                 the first sub-batch where there are several), and those whose
                                 bound -- with nothing but the rows' norms outside the band -- already leaves room to exclude; the
                                 band-split form is taken when that is >= 75 % */
    int64_t second_look_audited; /* of excluded_audited: pairs the FIRST bound had let through and the second look (band-split form:
                                 the low band's samples themselves) then excluded -- a hashed 1/32 of them per run
                                 (1/16 with SUSHI_HIP_AUDIT_EVERY=1, none with 0), other ones every run -- transformed all the same and
                                 held to what they really score, like the others */
} SushiHipBatchDiag;

typedef struct SushiHipBatch SushiHipBatch;

/* Bytes of device memory a batch of these requests needs: descriptors, schedules, result keys and a workspace of
 * at most workspace_cap_bytes (0 = as much as one sub-batch for all requests needs; the minimum is what the most
 * demanding single request needs -- the batch is cut into sub-batches that fit).  variant: -1 = choose. */
SUSHI_HIP_API size_t sushi_hip_batch_bytes(const SushiHipRequest* req_host, int n, int path, int variant,
                                           size_t workspace_cap_bytes);
/* dst: the stream find_substream is called on (searchable for the FFT path); src: the stream the patterns are slices
 * of (may be dst itself); same dtype.  Requests must lie inside their streams (EINVAL otherwise: the reference gets a
 * cv2.error for a pattern longer than the window).  mem_dev: 256-byte aligned, >= sushi_hip_batch_bytes(...) with the
 * same arguments.  The handle keeps pointers to dst and src: destroy it before them. */
SUSHI_HIP_API int sushi_hip_batch_create(const SushiHipStream* dst, const SushiHipStream* src,
                                         const SushiHipRequest* req_host, int n, int path, int variant,
                                         size_t workspace_cap_bytes, void* mem_dev, size_t mem_bytes,
                                         void* hip_stream, SushiHipBatch** out);
/* The same handle for OTHER requests (equally many, same streams, path and settings): descriptors, plan and schedule are redone
 * in place and uploaded in one copy -- what a caller that issues one small batch after another (sushi.calculate_shifts: one
 * find_substream call at a time, sushi.py:432,450-452) does instead of destroy + create.  ENOSPACE if the new requests need more
 * than the memory the batch was created in (the handle is then unchanged and still runs its old requests); a run still in flight on
 * another stream than `hip_stream` must have finished.  What the batch had learnt about its searches (exclusion form, suspension) is reset. */
SUSHI_HIP_API int sushi_hip_batch_reset(SushiHipBatch* batch, const SushiHipRequest* req_host, int n, void* hip_stream);
SUSHI_HIP_API int sushi_hip_batch_info(const SushiHipBatch* batch, SushiHipBatchInfo* info);
/* Matching method of the following runs (default after create: SUSHI_HIP_METHOD_SQDIFF_NORMED).
 * SUSHI_HIP_METHOD_CCOEFF_NORMED: out_idx = first index of the MAXIMUM of cv2.matchTemplate(..., TM_CCOEFF_NORMED),
 * out_score = that float32. */
SUSHI_HIP_API int sushi_hip_batch_set_method(SushiHipBatch* batch, int method);
/* FFT path: whether block pairs are excluded by a lower bound of their scores before they are transformed.
 * Results are the same either way; what differs is time.  AUTO (default after create): a sub-batch uses the exclusion when it has
 * enough pairs for the bound pass and its extra launches to pay (more than 3000 + 2 x its searches: one find_substream call
 * over a small window is a dozen pairs and a dozen launches -- 0.17 ms without, 0.23 with); ALWAYS / NEVER: for tests and measurements. */
#define SUSHI_HIP_EXCLUDE_AUTO 0
#define SUSHI_HIP_EXCLUDE_ALWAYS 1
#define SUSHI_HIP_EXCLUDE_NEVER 2
/* The exclusion has two forms with the same results.  WHOLE: the bound is taken from the products of whole spectra (every pair's
 * 64 KB row written and read back once).  BAND: only the low band (|f| < N/8, a quarter of the bins) is multiplied, stored and
 * transformed; what the other bins can add is bounded from the norms of the rows that meet (Cauchy-Schwarz), and whole rows are
 * formed only for the few pairs that are transformed after all.  BAND needs streams that keep most of their energy in the band
 * (audio does): AUTO and ALWAYS decide per batch, from the streams' own norms, which form to use; these two force one. */
#define SUSHI_HIP_EXCLUDE_BAND 3
#define SUSHI_HIP_EXCLUDE_WHOLE 4
SUSHI_HIP_API int sushi_hip_batch_set_exclusion(SushiHipBatch* batch, int mode);
/* How the roundings of the stored spectra and products (packed halves) enter the LOWER BOUND by which a block pair is excluded
 * without being transformed.  WORST_CASE (default after create): every rounding at its largest, all of them in phase -- triangle
 * inequality over the bins, Cauchy-Schwarz, a proven error bound of the float32 transforms; no independence is assumed, so a
 * pair is excluded only if NO position of it can reach the search's minimum.  STATISTICAL: round 5's model (8
 * standard deviations of independent roundings), kept for A/B measurements.  Results are the same either way wherever the
 * model holds; what differs is how many pairs are left to transform. */
#define SUSHI_HIP_BOUND_WORST_CASE 0
#define SUSHI_HIP_BOUND_STATISTICAL 1
SUSHI_HIP_API int sushi_hip_batch_set_bound_model(SushiHipBatch* batch, int model);
/* Multi-GPU callers: every following run ALSO writes its results as n 8-byte records (int32 index, float32 score bits) to
 * out_packed_dev (8-byte aligned, n records; NULL = off) -- the block a rank contributes to the one all-gather of the path, written
 * by the kernel that writes out_idx / out_score instead of by two copies afterwards. */
SUSHI_HIP_API int sushi_hip_batch_set_packed_output(SushiHipBatch* batch, int32_t* out_packed_dev);
/* The answer as soon as it exists (FFT path; v13).  early: n 16-byte records {int32 index, float32 score bits, int32 ready, int32
 * flagged} in memory BOTH sides can touch -- pinned host memory mapped to the device (16-byte aligned; NULL = off).  The kernel that
 * finishes a search from its candidate lists (refine_kernel) writes the search's record with ONE 16-byte store: ready = 1 and either
 * the final (index, score bits) -- the very values out_idx / out_score receive -- or flagged = 1: the search goes on to the tile stage
 * and its answer is in out_idx / out_score when the run's stream has drained.  A caller that sets ready = 0 in every record before a
 * run can poll the records instead of synchronising the stream: a drop-in find_substream call then does not wait for the three
 * launches behind refine_kernel (collection pass, exact tiles, unpack), which find nothing to do for it (~30 of ~140 us a call).
 * The direct path writes no early records. */
SUSHI_HIP_API int sushi_hip_batch_set_early_output(SushiHipBatch* batch, int32_t* early);
/* One pass of the hot path over the batch.  Asynchronous, with ONE exception: the first run of a batch (and the first after its
 * method changed) that goes through the pair exclusion in AUTO or ALWAYS mode reads 8 bytes back to decide the exclusion's form and
 * synchronises `hip_stream` once for that (not capturable in a hipGraph; BAND, WHOLE and NEVER never synchronise, nor does a
 * batch too small for AUTO to use the exclusion -- a drop-in find_substream call).  Environment variables are read when a batch is
 * created, never here.  A batch on lanes (SushiHipBatchInfo.lanes > 1) launches its sub-batches on streams of the library's own between the
 * run's first launch and its last on `hip_stream`: what is ordered behind the run on `hip_stream` is ordered behind all of it.  The
 * first run of such a batch that forms whole rows for every pair (whole-row form, no exclusion) builds the plan's one-sub-batch
 * cut on the host first (milliseconds, once).
 *   out_idx_dev[n]   = result.argmin(axis=1)[0]        (wav.py:186)
 *   out_score_dev[n] = result[0][min_idx], float32     (wav.py:188)
 * delta (FFT path; > 0, <= 1): floor of the score margin inside which positions are re-evaluated exactly -- every
 * position whose f32 FFT score minus its error bound is not above the smallest (score plus bound) of the search.
 * The bound of a pair is max(delta / 2, modelled f32 error); 2e-5 is the default of the Python layer. */
SUSHI_HIP_API int sushi_hip_batch_run(SushiHipBatch* batch, double delta, int32_t* out_idx_dev, float* out_score_dev,
                                      void* hip_stream);
/* What the last run did (FFT path).  Waits for that run.  ranking_err_host (n floats or NULL): |f32 FFT score - exact
 * score| at every request's result position (0 where a tile kernel found it); flagged_host (n int32 or NULL): 0 = the
 * candidate list sufficed, 1 = tiles, 2 = every position. */
SUSHI_HIP_API int sushi_hip_batch_diagnostics(SushiHipBatch* batch, SushiHipBatchDiag* diag, float* ranking_err_host,
                                              int32_t* flagged_host);
/* FFT path, after a run: the lower bound of the scores of every block pair of the LAST sub-batch (slb_host[pairs]; -inf = no
 * bound) and what it was put together from (acc_host[pairs][2], in the units of the stored products.  Whole-row form: the bound of
 * the cross term's magnitude, the largest row energy of a wave; band-split form: the SIGNED upper bound of the low band's part of the
 * cross term -- sqrt(2) and bin 0 in it --, the low row's energy; the energies only under SUSHI_HIP_BOUND_STATISTICAL, whose term they
 * feed: the worst-case bound does not form them) -- for tests and tools that look at how sharp the exclusion is.  *n_pairs
 * in: the capacity of the arrays, out: how many pairs there are.  Synchronises. */
SUSHI_HIP_API int sushi_hip_batch_pair_bounds(SushiHipBatch* batch, float* slb_host, float* acc_host, int64_t* n_pairs);
/* FFT path, after a run, for tests and tools: where the LAST sub-batch's pattern spectra and products live in the batch's workspace
 * (packed halves, 4 bytes per bin, bin f of a whole row at sushi_hip_fft_slot_of_bin(f), of a low row at
 * sushi_hip_fft_low_slot_of_bin(f)): TSPEC [segments][N], Y [block pairs][N] -- whole rows exist only for the pairs that were
 * transformed --, TSPEC_LOW [segments][N/4], Y_LOW [block pairs][N/4] (band-split form only).  Synchronises; the memory is the
 * caller's own batch buffer, valid until the next run. */
#define SUSHI_HIP_WS_TSPEC 0
#define SUSHI_HIP_WS_Y 1
#define SUSHI_HIP_WS_TSPEC_LOW 2
#define SUSHI_HIP_WS_Y_LOW 3
SUSHI_HIP_API int sushi_hip_batch_workspace_view(SushiHipBatch* batch, int which, void** ptr_dev, size_t* bytes);
SUSHI_HIP_API void sushi_hip_batch_destroy(SushiHipBatch* batch);

/* FFT path geometry of one request: block pairs (inverse transforms) and pattern segments (forward transforms). */
SUSHI_HIP_API int sushi_hip_fft_layout(int64_t win_start, int32_t n_pos, int32_t tmpl_len,
                                       int32_t* n_pairs, int32_t* n_seg);

/* ---- WavStream.__init__ on the GPU (wav.py:64-91 decode + downmix, wav.py:113-156 value pipeline) ----
 * sushi_hip_load_decode   : interleaved little-endian PCM frames (sample_width 2 or 3 bytes, `channels` per frame)
 *     -> float32 mono: the int16 (for 24-bit: the top two bytes, wav.py:70-74) of every channel converted to
 *     float32, summed left to right in float32 and divided by float(channels) (wav.py:78-91).  Frames
 *     [0, n_frames) of pcm_dev go to mono_dev[0 .. n_frames): callers upload and decode a file in bounded chunks.
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
SUSHI_HIP_API int sushi_hip_load_decode(const void* pcm_dev, int64_t n_frames, int32_t channels, int32_t sample_width,
                            float* mono_dev, void* hip_stream);
SUSHI_HIP_API int sushi_hip_load_resample(const float* raw_dev, int64_t n_raw, int32_t chunk, int32_t nl_full, double scale_full,
                            int64_t n_full, int32_t rest, int32_t nl_rest, double scale_rest,
                            int64_t pad, int64_t total, float* data_dev, void* hip_stream);
SUSHI_HIP_API int sushi_hip_load_histogram(const float* data_dev, int64_t n, int side, uint32_t prefix, uint32_t mask, int shift,
                             uint64_t* hist_dev, void* hip_stream);
SUSHI_HIP_API int sushi_hip_load_normalise(float* data_dev, int64_t n, float lo, float hi, float range, uint8_t* u8_dev,
                             void* hip_stream);

/* Optional per-stage timing of the FFT path with HIP events recorded on the launch stream (bench.py's roofline
 * figures).  Between _begin and _end every sushi_hip_batch_run records its stage boundaries; _end waits for them and
 * writes, per run, the milliseconds spent in each stage (summed over the run's sub-batches) into
 * stage_ms[run][SUSHI_HIP_NSTAGES].  Not thread safe. */
#define SUSHI_HIP_NSTAGES 6
#define SUSHI_HIP_STAGE_TSPEC 0    /* pattern-segment DFTs                                          */
#define SUSHI_HIP_STAGE_MAC 1      /* frequency-domain multiply-accumulate                          */
#define SUSHI_HIP_STAGE_IFFT 2     /* inverse DFTs + scoring epilogue                               */
#define SUSHI_HIP_STAGE_REFINE 3   /* exact float64 evaluation of the listed candidates             */
#define SUSHI_HIP_STAGE_FINISH 4   /* candidate collection + exact tiles of flagged searches, unpack */
#define SUSHI_HIP_STAGE_BOUND 5    /* lower bounds of the block pairs' scores (three of the inverse transform's four passes)  */
SUSHI_HIP_API int sushi_hip_profile_begin(void);
SUSHI_HIP_API int sushi_hip_profile_end(float* stage_ms, int max_calls, int* n_calls);

#ifdef __cplusplus
}
#endif
#endif /* SUSHI_HIP_H */

// sushi_amd/csrc/sushi_internal.hpp -- types and launchers shared between the translation units of
// libsushi_hip.so (hidden visibility: not part of the C ABI).
#ifndef SUSHI_INTERNAL_HPP
#define SUSHI_INTERNAL_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sushi_hip.h"

// The opaque stream handle of the C ABI: where the parts of a prepared stream live (all inside the caller's buffer).
struct SushiHipStream {
    const void* raw;          // the samples as they are (uint8 / float32), caller-owned
    int dtype;
    int64_t n;
    float* xc;                // [n]      sample - centre
    double* s1;               // [n + 1]  prefix sums of the samples
    double* s2;               // [n + 1]  prefix sums of their squares
    float* urel;              // [n + 1]  s2 relative to the block base
    float* usrel;             // [n + 1][2]  (urel[e], s1 relative to the block base): TM_CCOEFF_NORMED on the FFT path, one 8-byte load per window end
    double* base;             // [nb + 1] block bases of s2, then [nb + 1] block bases of s1, then `stats`
    double* stats;            // [2] FFT path: largest centred energy of seven consecutive blocks; the centring constant
    double* coarse;           // [2][nc] s2 and s1 at every COARSE_G-th sample (nc = n / COARSE_G + 2; entries past the end hold the
                              //         totals): bound_kernel's lower bound of a block pair's window energies
    int64_t nc;
    size_t base_bytes;
    void* spec;               // [(nb + 1) * N] block spectra as packed halves, or null; behind them:
    void* spec_low;           // [(nb + 1) * N / 4] the low band (|f| < N / 8) of every block spectrum again, in bound_low_kernel's order
    float* znorm_rest;        // [3][norm_stride] norms over the bins OUTSIDE the band of a block spectrum's stored halves: of Z itself, of
                              // the spectrum of its real block at j B, of the one at j B + H (sushi_fft.hip real_block_rest_norms)
    int64_t norm_stride;
    size_t spec_bytes;
    int64_t blocks;           // nb
};

namespace sushi {

// One search on the device: a SushiHipRequest plus the running sums that let a workgroup find its work.
struct SearchDesc {
    int64_t tmpl_off;
    int64_t win_start;
    int32_t tmpl_len;
    int32_t n_pos;
    int32_t first_tile;   // direct path: tiles of the searches before this one
    int32_t first_pair;   // FFT path: block pairs of the searches before this one
    int32_t first_seg;    // FFT path: pattern segments of the searches before this one
    int32_t reserved;
};
static_assert(sizeof(SearchDesc) == 40, "SearchDesc layout");

struct StreamRefs {
    const float* dst_xc; const double* dst_s1; const double* dst_s2; int64_t dst_len;
    const float* src_xc; const double* src_s1; const double* src_s2; int64_t src_len;
    double centre;
    const void* dst_raw; const void* src_raw; int dtype;     // the samples as they are (exact evaluation)
};

// One exact-evaluation work item: TILE consecutive positions of one search, aligned to the absolute grid.
constexpr int SPARSE_TILE_MAX = 256;   // candidates per tile up to which collect_kernel lists them; beyond: every position of the tile
constexpr int SPARSE_UNIT = 64;        // a listed tile is handed to exact_tiles_kernel in entries of at most this many candidates
struct TileDesc {
    int32_t search;       // global search index
    int32_t p0;           // first position of the tile relative to the search's window (may be < 0 for the first tile)
    int32_t off;          // sparse: first entry of the tile's candidate list in the candidate buffer
    int32_t cnt;          // sparse: candidates of this entry (<= SPARSE_UNIT: a tile of more is several entries); dense (every valid position of the tile): -1
};

// Counters of one run, in device memory (zeroed at the start of a run).
// What the exact stages of ONE sub-batch count with: every sub-batch of a plan has its own (sub-batches of a batch may run side by
// side on several HIP streams: sushi_fft.hip "lanes"); cleared by the run's first launch.
struct SubCounters {
    int32_t n_tiles;          // entries of the tile list
    int32_t tile_next;        // exact_tiles_kernel's queue: the next entry to hand out
    int32_t n_cand;           // entries of the candidate buffer
    int32_t sub_flagged;      // searches of this sub-batch refine_kernel flagged (entries of its part of the flag list)
};

struct RunCounters {
    int32_t n_flagged;        // searches refine_kernel could not finish from the per-pair lists
    int32_t n_all_positions;  // of those: every position (bound violated)
    unsigned long long tiles_dense, tiles_sparse, candidates;    // totals of the run
    uint32_t max_ratio_bits;  // float bits of SushiHipBatchDiag.max_bound_ratio
    uint32_t max_ratio_audit_bits;  // float bits of SushiHipBatchDiag.max_bound_ratio_noncandidate
    unsigned long long audited;     // non-candidate positions evaluated exactly (SushiHipBatchDiag.audited)
    unsigned long long pairs_transformed;   // block pairs whose inverse transform was run (the others were excluded by bound_kernel's bound)
    unsigned long long excluded_audited;    // of those: pairs the bound HAD excluded, transformed as a check of the bound
    uint32_t max_slb_ratio_bits;            // float bits: largest (lower bound / upper bound of the pair's real best score) over the audited excluded pairs
    int32_t slb_violations;                 // pairs whose lower bound turned out above a real score (their searches go to every position)
    unsigned long long second_look_audited; // of excluded_audited: pairs the second look had excluded (survivor2_kernel's sample)
};

int direct_variant_count();
int direct_variant_tile(int variant);

// direct path: one launch of the MFMA kernel + unpack
int launch_direct(const StreamRefs& r, const SearchDesc* searches_dev, int n_search, int n_tiles, int variant, int method,
                  unsigned long long* keys_dev, int32_t* out_idx_dev, float* out_score_dev, int32_t* out_packed_dev, hipStream_t st);

// FFT path, exact stages (sushi_hip.hip):
// refine: exact float64 evaluation of the listed candidates of searches [first_search, first_search + n_sub).
// flags_dev[s] = 0 done / 1 needs tiles / 2 every position; flag_list_dev receives the flagged searches of this sub-batch.
struct RefineParams {
    StreamRefs r;
    const SearchDesc* searches;       // all searches of the batch
    int first_search, n_sub, sub_first_pair;
    const unsigned long long* cand;   // [pairs of the sub-batch][FFT_ROW]
    const float* pair_lb;             // [pairs of the sub-batch] smallest lower bound of each pair
    unsigned long long* gkeys;        // [all searches] in: min over pairs of (f32 score + bound); out: |f32 - exact| bits
    unsigned long long* keys;         // [all searches] result keys
    int* flags;                       // [all searches]
    int* flag_list;                   // [n_sub] flagged searches of this sub-batch (global indices)
    SubCounters* sub;                 // this sub-batch's counters (sub_flagged: how many)
    int* citems;                      // [pairs of the sub-batch] out: the pairs of flagged searches collect_kernel has to look at
    int* n_citems;                    // [1] how many
    RunCounters* counters;
    float delta;
    int method;                       // SUSHI_HIP_METHOD_*
    const int* viol;                  // [all searches] or NULL: 1 = a pair's lower bound was found above a real score (ifft_kernel's audit)
    int4* early;                      // [all searches] or NULL: sushi_hip_batch_set_early_output's records (host memory mapped to the device)
};
int launch_refine(const RefineParams& p, hipStream_t st);

// Up to FILL_RANGES ranges of 32-bit words set to a value each, in ONE launch: what a run clears before its first kernel (result
// keys, flags, counters, the candidate rows of a small batch) used to be half a dozen hipMemsetAsync calls -- a third of the
// launches of a drop-in find_substream call, whose cost IS its launches.
constexpr int FILL_RANGES = 8;
struct FillArgs { uint32_t* p[FILL_RANGES]; uint32_t words[FILL_RANGES]; uint32_t value[FILL_RANGES]; int n; };
int launch_fill(const FillArgs& a, hipStream_t st);

struct TileParams {
    StreamRefs r;
    const SearchDesc* searches;
    const TileDesc* tiles;
    const int32_t* cand;              // candidate positions (relative to the search window)
    unsigned long long* keys;
    RunCounters* counters;
    SubCounters* sub;                 // this sub-batch's tile list length and queue head (read / advanced on the device)
    int method;                       // SUSHI_HIP_METHOD_*
};
int launch_tiles(const TileParams& p, hipStream_t st);
int launch_unpack(const unsigned long long* keys_dev, int n, int method, int32_t* out_idx_dev, float* out_score_dev,
                  int32_t* out_packed_dev, hipStream_t st);

}  // namespace sushi
#endif

// sushi_amd/csrc/sushi_fft.hip -- overlap-save FFT form of Sushi's template match for gfx950 (MI355X),
// and the batch handle of the C ABI.
//
// For every (pattern, window) request the float32 result.argmin and result[argmin] of
//   wav.py:185  result = cv2.matchTemplate(search_source, pattern, cv2.TM_SQDIFF_NORMED)
//   wav.py:186  min_idx = result.argmin(axis=1)[0]
// cv2's crossCorr() computes the sliding dot product by block DFT; so does this path, which turns
// the O(P*M) FLOP-bound search into an O(P log N) HBM-bound one (DESIGN.md "FFT path").  With N-point
// complex transforms, pattern segments of B = FFT_SEG samples and H = N - B valid positions per real block:
//
//   spectra_kernel   once per destination stream, for every block j (hop B):
//                    Z_j = DFT_N( x[jB .. jB+N) + i * x[jB+H .. jB+H+N) )
//                    (+ the low band of Z_j once more and its norms outside the band: the band-split exclusion below)
//   tspec_kernel     per search, per pattern segment s (B samples, zero padded to N):
//                    Tt_s = conj(DFT_N(t_s)) / N                                  (+ its low band and its norm outside the band)
//   mac_kernel       per search, per frequency bin f, per pair I of the absolute pair grid (pair I starts at
//                    block FFT_STEP * I, FFT_STEP = 2H / B):
//                    Y_I(f) = sum_s Tt_s(f) * Z_{FFT_STEP*I+s}(f)       (a 1-D Toeplitz product along j)
//                    -- over every pair on the LOW rows (a quarter of the bins: the band-split form of the exclusion), or on whole
//                    rows (the whole-row form, no exclusion, the dense fall-back); mac_list_kernel / mac_rows_kernel form the
//                    whole rows of LISTED pairs only
//   bound_low_kernel / bound_kernel + slb_kernel, pilot_kernel, survivor_kernel
//                    a LOWER bound of every pair's scores without scoring it; the pair with the smallest bound of every search
//                    is transformed first, then only the pairs whose bound is not above what the search has found
//   ifft_kernel      y_I = IDFT_N(Y_I): Re y_I[r] / Im y_I[r] (r < H) are the cross terms of positions
//                    FFT_STEP*I*B + r and FFT_STEP*I*B + H + r.  Fused epilogue: window energies, normalised f32
//                    score, the pair's error bound, arg-min, and the list of positions that can still be the
//                    minimum (candidates); the pair's lower bound held to what it really scores (the exclusion's audit).
//   refine_kernel    (sushi_hip.hip) exact float64 re-evaluation of the candidates -> final (index, score);
//   collect + tiles  searches with more candidates than the lists hold: the inverse transforms of their pairs
//                    are redone with the search's final threshold, every candidate goes to a per-tile list
//                    (a tile = 1024 positions) and exact_tiles_kernel (sushi_hip.hip) evaluates those exactly.
//
// Pairing two real blocks as one complex block makes every N-point complex DFT produce 2H useful
// results and needs no real-FFT untangling pass: the pattern is real, so correlation is linear
// over the real and imaginary parts.
//
// gfx950 only: wave64, 160 KiB LDS/CU.  No CUDA compatibility paths.
//
// ONE translation unit, cut by stage (every part is included below, inside this file's anonymous namespace):
//   sushi_fft_store.inc    packed-half storage: scales, stored bin order, the low band of a row and the norms outside it
//   sushi_fft_spectra.inc  spectra_kernel, tspec_kernel
//   sushi_fft_mac.inc      mac_kernel / mac_long_kernel / mac_list_kernel
//   sushi_fft_ifft.inc     ifft_kernel / ifft_list_kernel: transform, scoring epilogue, error model, candidates
//   sushi_fft_bound.inc    the pair exclusion: bound_kernel / bound_low_kernel / slb_kernel / pilot / survivor / second look / mac_rows_kernel
//   sushi_fft_collect.inc  collect_kernel
//   sushi_fft_plan.inc     host: workspace layout, stage timing, the plan of a batch
//   (this file)            the batch handle and the C ABI's entry points

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/sushi_hip.h"
#include "sushi_common.hpp"
#include "sushi_internal.hpp"
#include "fft_core.hpp"
#include "mac_core.hpp"

namespace {

using namespace sushi;
using sushi_fft::cpx;

constexpr int FN = FFT_N;                              // complex points per transform
constexpr int FT = sushi_fft::Plan<FFT_LOGN>::NT;      // threads per transform workgroup
constexpr int FH = FFT_H;                              // result positions per half of a pair
constexpr int HPT = FH / FT;                           // result positions per thread and half
constexpr int RPB = FFT_SEG / FT;                      // of which per block
static_assert(FH % FT == 0 && FFT_SEG % FT == 0 && HPT == FFT_VB * RPB, "position layout");
static_assert(HPT <= sushi_fft::PER, "a thread's valid outputs are a prefix of its transform outputs");
constexpr int LDS_FLOATS = sushi_fft::lds_floats<FFT_LOGN>();

#include "sushi_fft_store.inc"
#include "sushi_fft_spectra.inc"
#include "sushi_fft_mac.inc"
#include "sushi_fft_ifft.inc"
#include "sushi_fft_bound.inc"
#include "sushi_fft_collect.inc"
#include "sushi_fft_plan.inc"

}  // namespace

// What a batch's FIRST run used to allocate -- 16 bytes of pinned host memory for the counts that come back behind an event, the
// lanes' HIP streams -- cost that run 9 - 13 ms of host time (hipHostMalloc maps into every device's page tables; a stream is a
// hardware queue): both now come from pools of the process's own, made once (tools/first_run_probe.py: the first run of a batch of
// BASELINE configs[2] spent 9.4 of its 21.6 ms inside sushi_hip_batch_run on the host).  Pool slots are handed out under a mutex; a handle is used by one host thread at a time.
struct HostSlots {
    static constexpr int SLOTS = 1024, WORDS = 2;
    std::mutex mu;
    unsigned long long* base = nullptr;
    std::vector<int> free_list;
    unsigned long long* take() {
        std::lock_guard<std::mutex> g(mu);
        if (!base) {
            if (hipHostMalloc((void**)&base, (size_t)SLOTS * WORDS * sizeof(unsigned long long), hipHostMallocPortable) != hipSuccess) { base = nullptr; return nullptr; }
            for (int k = SLOTS - 1; k >= 0; --k) free_list.push_back(k);
        }
        if (free_list.empty()) return nullptr;
        const int k = free_list.back(); free_list.pop_back();
        return base + (size_t)k * WORDS;
    }
    void give(unsigned long long* p) {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu);
        free_list.push_back((int)((p - base) / WORDS));
    }
};
static HostSlots g_host_slots;

// The lanes' streams, per device: shared by every batch on that device (batches that run at the same time on different caller
// streams then share them too -- ordered by their own events, side by side no longer; one batch at a time is the product's use).
struct LanePool {
    static constexpr int MAX_DEVICES = 16;
    std::mutex mu;
    hipStream_t st[MAX_DEVICES][MAX_LANES] = {};
    hipStream_t get(int lane) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
        std::lock_guard<std::mutex> g(mu);
        if (!st[dev][lane] && hipStreamCreateWithFlags(&st[dev][lane], hipStreamNonBlocking) != hipSuccess) st[dev][lane] = nullptr;
        return st[dev][lane];
    }
};
static LanePool g_lane_pool;

// The opaque batch handle of the C ABI.
struct SushiHipBatch {
    const SushiHipStream* dst;
    const SushiHipStream* src;
    int n, path, variant, method, exclusion;
    int band;                           // the exclusion's form in AUTO / ALWAYS: -1 not decided yet, 0 whole rows (bound_kernel), 1 band-split
    int band_decided_method;            // ... which was decided for this method (the pattern spectra differ)
    int band_votes[2];                  // what the decision was taken from: pairs looked at, pairs whose bound leaves room
    unsigned run_seq;                   // runs so far: rotates which excluded pairs are audited
    int audit_every;                    // one search in this many has one excluded pair transformed as a check, per run
    int bound_model;                    // SUSHI_HIP_BOUND_WORST_CASE (default) / _STATISTICAL: how the excluded side's roundings enter slb
    int last_band;                      // form of the exclusion the last run used (its last sub-batch that went through it; -1: none did)
    bool last_whole_cut;                // the last run took the plan's one-sub-batch cut (Plan::subs_whole)
    // AUTO learns from its own runs: a batch whose exclusion excluded next to nothing (searches without a match anywhere) runs
    // without it from then on, looking again every 64th run.  The last run's counts come back through 16 bytes of pinned host memory
    // behind an event that is only ever QUERIED: a run never waits for an earlier one.
    unsigned long long* host_stats;     // [2] pairs transformed, excluded pairs audited
    hipEvent_t stats_ready;
    bool stats_pending;
    unsigned long long last_transformed; // pairs the last finished run transformed (0: not known): sizes the next run's one-workgroup-per-slot launch
    int suspended;                      // 1: the exclusion is left out (AUTO)
    unsigned suspended_at;              // run_seq of the run that showed it
    int last_suspended;                 // whether the last run was one of those
    int32_t* packed_out;                // NULL, or where every run ALSO leaves its results as 8-byte (index, score bits) records
    int32_t* early_out;                 // NULL, or sushi_hip_batch_set_early_output's 16-byte records (memory host and device both touch)
    int64_t n_tiles;
    int64_t direct_pairs;               // pairs of the last run's sub-batches that were transformed without the exclusion
    std::vector<SearchDesc> descs;
    Plan plan;
    BatchLayout lay;
    std::vector<char> upload;           // descriptors | schedule | work items as they lie in `mem`: one copy per (re)plan
    size_t mem_bytes, ws_cap;           // what the caller gave: a re-plan (sushi_hip_batch_reset) must fit it
    char* mem;
    double flops, algorithmic_bytes;
    hipStream_t last_stream;
    bool ran;
    hipEvent_t uploaded;                // recorded on the create-time stream behind the descriptor / plan uploads
    // lanes (sushi_fft_plan.inc): lane 0 is the stream a run is given; the others are the batch's own, forked off it behind the
    // run's first launch and joined before its last
    hipStream_t lane_stream[MAX_LANES];
    hipEvent_t lane_done[MAX_LANES];
    hipEvent_t fork;
    ~SushiHipBatch() {
        for (int l = 1; l < MAX_LANES; ++l) {
            if (lane_stream[l]) (void)hipStreamSynchronize(lane_stream[l]);          // (the pool's: this batch's work on it has to be through)
            if (lane_done[l]) (void)hipEventDestroy(lane_done[l]);
        }
        if (fork) (void)hipEventDestroy(fork);
        if (uploaded) { (void)hipEventSynchronize(uploaded); (void)hipEventDestroy(uploaded); }    // (an upload may still read the handle's host buffers)
        if (stats_pending && stats_ready) (void)hipEventSynchronize(stats_ready);       // the last run's counts may still be on their way
        if (stats_ready) (void)hipEventDestroy(stats_ready);
        g_host_slots.give(host_stats);
    }
};

extern "C" {

int sushi_hip_device_prepare(void) {
    for (int l = 1; l < MAX_LANES; ++l)
        if (!g_lane_pool.get(l)) return SUSHI_HIP_ELAUNCH;
    unsigned long long* p = g_host_slots.take();
    if (!p) return SUSHI_HIP_ENOMEM;
    g_host_slots.give(p);
    return SUSHI_HIP_OK;
}

int sushi_hip_fft_size(void) { return FN; }

int sushi_hip_fft_block(void) { return FFT_SEG; }

int sushi_hip_fft_slot_of_bin(int bin) { return (bin < 0 || bin >= FN) ? -1 : sushi_fft::mslot_of_bin(bin); }

int sushi_hip_fft_low_slot_of_bin(int bin) { return sushi_fft::lslot_of_bin(bin); }

size_t sushi_hip_stream_spectra_bytes(int64_t n) {
    if (n <= 0) return 0;
    const size_t rows = (size_t)((n + FFT_SEG - 1) / FFT_SEG + 1);
    return rows * ROW_BYTES + rows * LROW_BYTES + 3 * align_up(rows * sizeof(float), 256);   // norms outside the band: of Z, of its two real blocks
}

int sushi_hip_fft_layout(int64_t win_start, int32_t n_pos, int32_t tmpl_len, int32_t* n_pairs, int32_t* n_seg) {
    if (win_start < 0 || n_pos < 1 || tmpl_len < 1 || !n_pairs || !n_seg) return SUSHI_HIP_EINVAL;
    const FftLayout l = fft_layout(win_start, n_pos, tmpl_len);
    *n_pairs = l.n_pairs;
    *n_seg = l.n_seg;
    return SUSHI_HIP_OK;
}

int sushi_hip_stream_add_spectra(SushiHipStream* s, void* mem_dev, size_t mem_bytes, void* hip_stream) {
    if (!s || !mem_dev) return SUSHI_HIP_EINVAL;
    if ((uintptr_t)mem_dev & 255) return SUSHI_HIP_EALIGN;
    const size_t need = sushi_hip_stream_spectra_bytes(s->n);
    if (mem_bytes < need) return SUSHI_HIP_ENOSPACE;
    if (s->blocks >= 0x7fffffff) return SUSHI_HIP_EINVAL;
    // one block more than the stream has: its samples are all past the end, so its spectrum is zero
    const size_t rows = (size_t)s->blocks + 1;
    uint4* low = (uint4*)((char*)mem_dev + rows * ROW_BYTES);
    float* zn = (float*)((char*)mem_dev + rows * ROW_BYTES + rows * LROW_BYTES);
    if (s->dtype == SUSHI_HIP_F32)
        hipLaunchKernelGGL(spectra_kernel<float>, dim3((unsigned)s->blocks + 1), dim3(FT), 0, (hipStream_t)hip_stream,
                           (const float*)s->raw, s->n, (uint32_t*)mem_dev, (const double*)s->stats, low, zn,
                           (int64_t)(align_up(rows * sizeof(float), 256) / sizeof(float)));
    else
        hipLaunchKernelGGL(spectra_kernel<uint8_t>, dim3((unsigned)s->blocks + 1), dim3(FT), 0, (hipStream_t)hip_stream,
                           (const uint8_t*)s->raw, s->n, (uint32_t*)mem_dev, (const double*)s->stats, low, zn,
                           (int64_t)(align_up(rows * sizeof(float), 256) / sizeof(float)));
    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    s->spec = mem_dev;
    s->spec_low = low;
    s->znorm_rest = zn;
    s->norm_stride = (int64_t)(align_up(rows * sizeof(float), 256) / sizeof(float));
    s->spec_bytes = rows * ROW_BYTES;
    return SUSHI_HIP_OK;
}

// A caller sizes a batch (sushi_hip_batch_bytes) and then creates it from the same requests: the plan made for the first call is
// kept for the second (per host thread; compared request by request, so a changed request list simply plans again).
struct PlanCache {
    bool valid = false;
    size_t cap = 0;
    std::string lanes_env;
    std::vector<SushiHipRequest> req;
    Plan plan;
};
static thread_local PlanCache g_plan_cache;
static std::string lanes_env_now() { const char* e = getenv("SUSHI_HIP_LANES"); return e ? std::string(e) : std::string(); }

size_t sushi_hip_batch_bytes(const SushiHipRequest* req_host, int n, int path, int variant, size_t workspace_cap_bytes) try {
    if (!req_host || n <= 0 || (path != SUSHI_HIP_PATH_FFT && path != SUSHI_HIP_PATH_DIRECT)) return 0;
    if (path == SUSHI_HIP_PATH_FFT) variant = direct_variant_count() - 1;
    else if (variant < 0) variant = choose_direct_variant(req_host, n);
    if (variant >= direct_variant_count()) return 0;
    std::vector<SearchDesc> descs;
    int64_t tiles;
    if (make_descs(req_host, n, variant, descs, &tiles) != SUSHI_HIP_OK) return 0;
    if (path == SUSHI_HIP_PATH_DIRECT) return batch_layout(n, path, 0, 0, 0, 0, 0).total;
    Plan plan;
    if (make_plan(descs, workspace_cap_bytes, plan) != SUSHI_HIP_OK) return 0;
    const size_t total = batch_layout(n, path, plan.order.size(), plan.items.size(), plan.ws_bytes, plan.subs.size(), plan.segs).total;
    PlanCache& pc = g_plan_cache;
    pc.valid = true; pc.cap = workspace_cap_bytes; pc.lanes_env = lanes_env_now();
    pc.req.assign(req_host, req_host + n);
    pc.plan = std::move(plan);
    return total;
} catch (...) { return 0; }        // std::bad_alloc etc.: nothing crosses the C boundary

// requests -> descriptors, plan and layout of `b` (all three replaced together or not at all), uploaded on `st`.
// ENOSPACE: they do not fit the memory the batch was created in (nothing is changed then).
static int plan_and_upload(SushiHipBatch* b, const SushiHipRequest* req_host, int n, hipStream_t st) {
    std::vector<SearchDesc> descs;
    int64_t n_tiles = 0;
    int rc = make_descs(req_host, n, b->variant, descs, &n_tiles);
    if (rc != SUSHI_HIP_OK) return rc;
    double flops = 0.0, abytes = 0.0;
    const double width = b->dst->dtype == SUSHI_HIP_F32 ? 4.0 : 1.0;
    for (int k = 0; k < n; ++k) {
        const SushiHipRequest& r = req_host[k];
        if (r.tmpl_off + r.tmpl_len > b->src->n || r.win_start + (int64_t)r.n_pos + r.tmpl_len - 1 > b->dst->n) return SUSHI_HIP_EINVAL;
        flops += 2.0 * (double)r.n_pos * (double)r.tmpl_len;
        abytes += width * ((double)r.n_pos + r.tmpl_len - 1) + width * r.tmpl_len + 8.0;
    }
    Plan plan;
    if (b->path == SUSHI_HIP_PATH_FFT) {
        PlanCache& pc = g_plan_cache;
        if (pc.valid && pc.cap == b->ws_cap && (int)pc.req.size() == n && memcmp(pc.req.data(), req_host, (size_t)n * sizeof(SushiHipRequest)) == 0 &&
            pc.lanes_env == lanes_env_now()) {
            plan = std::move(pc.plan);                      // (the plan sushi_hip_batch_bytes made for these very requests)
            pc.valid = false;
        } else {
            rc = make_plan(descs, b->ws_cap, plan);
            if (rc != SUSHI_HIP_OK) return rc;
        }
    }
    const BatchLayout lay = batch_layout(n, b->path, plan.order.size(), plan.items.size(), plan.ws_bytes, plan.subs.size(), plan.segs);
    if (b->mem_bytes < lay.total) return SUSHI_HIP_ENOSPACE;
    // an earlier plan's upload reads the handle's host buffer until its event has passed
    if (b->uploaded && hipEventSynchronize(b->uploaded) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    b->descs.swap(descs); b->plan = std::move(plan); b->lay = lay; b->n_tiles = n_tiles;
    b->flops = flops; b->algorithmic_bytes = abytes;
    const size_t up_bytes = b->path == SUSHI_HIP_PATH_FFT ? lay.items + align_up(b->plan.items.size() * sizeof(int32_t), 256) - lay.desc
                                                          : align_up((size_t)n * sizeof(SearchDesc), 256);
    b->upload.assign(up_bytes, 0);
    memcpy(b->upload.data(), b->descs.data(), (size_t)n * sizeof(SearchDesc));
    if (!b->plan.order.empty()) memcpy(b->upload.data() + (lay.order - lay.desc), b->plan.order.data(), b->plan.order.size() * sizeof(int32_t));
    if (!b->plan.items.empty()) memcpy(b->upload.data() + (lay.items - lay.desc), b->plan.items.data(), b->plan.items.size() * sizeof(int32_t));
    // (the host buffer lives in the handle: the copy may still be in flight when this returns)
    if (hipMemcpyAsync(b->mem + lay.desc, b->upload.data(), up_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    // a run may be launched on another stream than this one: it waits for this event first
    if (!b->uploaded && hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    if (hipEventRecord(b->uploaded, st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    // what the batch had learnt about its searches is about other searches now
    b->band = -1; b->last_band = -1; b->band_decided_method = -1; b->band_votes[0] = b->band_votes[1] = 0;
    b->suspended = 0; b->suspended_at = 0; b->last_suspended = 0; b->last_transformed = 0; b->ran = false; b->direct_pairs = 0;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_create(const SushiHipStream* dst, const SushiHipStream* src, const SushiHipRequest* req_host, int n,
                           int path, int variant, size_t workspace_cap_bytes, void* mem_dev, size_t mem_bytes,
                           void* hip_stream, SushiHipBatch** out) try {
    if (!dst || !src || !req_host || !mem_dev || !out || n <= 0) return SUSHI_HIP_EINVAL;
    if (path != SUSHI_HIP_PATH_FFT && path != SUSHI_HIP_PATH_DIRECT) return SUSHI_HIP_EINVAL;
    if (dst->dtype != src->dtype) return SUSHI_HIP_EINVAL;       // cv2.matchTemplate asserts equal types
    if ((uintptr_t)mem_dev & 255) return SUSHI_HIP_EALIGN;
    if (path == SUSHI_HIP_PATH_FFT) {
        if (!dst->spec) return SUSHI_HIP_EINVAL;                 // not searchable: sushi_hip_stream_add_spectra first
        variant = direct_variant_count() - 1;
    } else if (variant < 0) {
        variant = choose_direct_variant(req_host, n);
    }
    if (variant >= direct_variant_count()) return SUSHI_HIP_EINVAL;
    SushiHipBatch* b = new (std::nothrow) SushiHipBatch();
    if (!b) return SUSHI_HIP_ENOMEM;
    std::unique_ptr<SushiHipBatch> guard(b);                     // freed on every early return and on an exception
    b->dst = dst; b->src = src; b->n = n; b->path = path; b->variant = variant; b->method = SUSHI_HIP_METHOD_SQDIFF_NORMED;
    b->exclusion = SUSHI_HIP_EXCLUDE_AUTO;
    b->packed_out = nullptr;
    b->early_out = nullptr;
    b->last_transformed = 0;
    b->host_stats = nullptr; b->stats_ready = nullptr; b->stats_pending = false; b->suspended = 0; b->suspended_at = 0; b->last_suspended = 0;
    b->band = -1; b->last_band = -1; b->band_decided_method = -1; b->band_votes[0] = b->band_votes[1] = 0; b->run_seq = 0; b->audit_every = 2;
    b->bound_model = SUSHI_HIP_BOUND_WORST_CASE;
    {
        // (measurements only, read once per batch: 0 = no excluded pair is audited; "statistical" = round 5's error model)
        const char* e = getenv("SUSHI_HIP_AUDIT_EVERY");
        if (e && *e) { const int v = atoi(e); b->audit_every = v < 0 ? 0 : v; }
        const char* m = getenv("SUSHI_HIP_BOUND_MODEL");
        if (m && !strcmp(m, "statistical")) b->bound_model = SUSHI_HIP_BOUND_STATISTICAL;
    }
    b->mem = (char*)mem_dev; b->mem_bytes = mem_bytes; b->ws_cap = workspace_cap_bytes;
    b->last_stream = nullptr; b->ran = false; b->uploaded = nullptr; b->n_tiles = 0; b->direct_pairs = 0; b->last_whole_cut = false;
    for (int l = 0; l < MAX_LANES; ++l) { b->lane_stream[l] = nullptr; b->lane_done[l] = nullptr; }
    b->fork = nullptr;
    const int rc = plan_and_upload(b, req_host, n, (hipStream_t)hip_stream);
    if (rc != SUSHI_HIP_OK) return rc;
    *out = guard.release();
    return SUSHI_HIP_OK;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

int sushi_hip_batch_reset(SushiHipBatch* b, const SushiHipRequest* req_host, int n, void* hip_stream) try {
    if (!b || !req_host || n != b->n) return SUSHI_HIP_EINVAL;
    // the last run's counts may still be on their way into the handle's pinned words
    if (b->stats_pending && b->stats_ready) { (void)hipEventSynchronize(b->stats_ready); b->stats_pending = false; }
    return plan_and_upload(b, req_host, n, (hipStream_t)hip_stream);
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }

int sushi_hip_batch_info(const SushiHipBatch* b, SushiHipBatchInfo* info) {
    if (!b || !info) return SUSHI_HIP_EINVAL;
    memset(info, 0, sizeof(*info));
    info->n_search = b->n; info->path = b->path; info->variant = b->variant;
    info->sub_batches = b->path == SUSHI_HIP_PATH_FFT ? (int32_t)b->plan.subs.size() : 1;
    info->direct_tiles = b->n_tiles;
    info->fft_pairs = b->plan.pairs; info->fft_segments = b->plan.segs;
    info->workspace_bytes = b->plan.ws_bytes; info->mem_bytes = b->lay.total;
    info->flops = b->flops; info->algorithmic_bytes = b->algorithmic_bytes;
    info->lanes = b->path == SUSHI_HIP_PATH_FFT ? b->plan.lanes : 1;
    return SUSHI_HIP_OK;
}

void sushi_hip_batch_destroy(SushiHipBatch* b) { delete b; }

int sushi_hip_batch_set_method(SushiHipBatch* b, int method) {
    if (!b) return SUSHI_HIP_EINVAL;
    if (method != SUSHI_HIP_METHOD_SQDIFF_NORMED && method != SUSHI_HIP_METHOD_CCOEFF_NORMED) return SUSHI_HIP_EINVAL;
    b->method = method;                                          // both paths compute both methods
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_packed_output(SushiHipBatch* b, int32_t* out_packed_dev) {
    if (!b) return SUSHI_HIP_EINVAL;
    if ((uintptr_t)out_packed_dev & 7) return SUSHI_HIP_EALIGN;
    b->packed_out = out_packed_dev;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_early_output(SushiHipBatch* b, int32_t* early) {
    if (!b) return SUSHI_HIP_EINVAL;
    if ((uintptr_t)early & 15) return SUSHI_HIP_EALIGN;
    b->early_out = early;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_exclusion(SushiHipBatch* b, int mode) {
    if (!b || mode < SUSHI_HIP_EXCLUDE_AUTO || mode > SUSHI_HIP_EXCLUDE_WHOLE) return SUSHI_HIP_EINVAL;
    b->exclusion = mode;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_set_bound_model(SushiHipBatch* b, int model) {
    if (!b || (model != SUSHI_HIP_BOUND_WORST_CASE && model != SUSHI_HIP_BOUND_STATISTICAL)) return SUSHI_HIP_EINVAL;
    b->bound_model = model;
    return SUSHI_HIP_OK;
}

int sushi_hip_batch_run(SushiHipBatch* b, double delta, int32_t* out_idx_dev, float* out_score_dev, void* hip_stream) try {
    if (!b || !out_idx_dev || !out_score_dev) return SUSHI_HIP_EINVAL;
    const hipStream_t st0 = (hipStream_t)hip_stream;
    const SushiHipStream* dst = b->dst;
    const SushiHipStream* src = b->src;
    const int n_search = b->n;
    StreamRefs r;
    r.dst_xc = dst->xc; r.dst_s1 = dst->s1; r.dst_s2 = dst->s2; r.dst_len = dst->n;
    r.src_xc = src->xc; r.src_s1 = src->s1; r.src_s2 = src->s2; r.src_len = src->n;
    r.centre = sushi_hip_centre(dst->dtype);
    r.dst_raw = dst->raw; r.src_raw = src->raw; r.dtype = dst->dtype;
    const SearchDesc* searches_dev = (const SearchDesc*)(b->mem + b->lay.desc);
    unsigned long long* keys = (unsigned long long*)(b->mem + b->lay.keys);
    b->last_stream = st0; b->ran = true; b->direct_pairs = 0;
    if (hipStreamWaitEvent(st0, b->uploaded, 0) != hipSuccess) return SUSHI_HIP_ELAUNCH;   // descriptors and plan have landed
    if (b->path == SUSHI_HIP_PATH_DIRECT)
        return launch_direct(r, searches_dev, n_search, (int)b->n_tiles, b->variant, b->method, keys, out_idx_dev,
                             out_score_dev, b->packed_out, st0);

    if (!(delta >= 3.8e-6) || delta > 1.0) return SUSHI_HIP_EINVAL;      // the floor covers the scoring arithmetic's own rounding
    unsigned long long* gkeys = keys + n_search;
    int* flags = (int*)(b->mem + b->lay.flags);
    int* flag_list = (int*)(b->mem + b->lay.flag_list);
    char* subc = b->mem + b->lay.subc;                            // SUBC_BYTES per sub-batch: SubCounters, then its scount words
    float* tnorm_all = (float*)(b->mem + b->lay.tnorm);           // [all segments] squared norms of the pattern rows outside the band
    RunCounters* counters = (RunCounters*)(b->mem + b->lay.counters);
    const int32_t* order = (const int32_t*)(b->mem + b->lay.order);
    const int32_t* items = (const int32_t*)(b->mem + b->lay.items);
    ProfCall* pc = nullptr;
    if (g_prof_on) { g_prof.emplace_back(); pc = &g_prof.back(); }
    int* viol = (int*)(b->mem + b->lay.viol);
    // Everything a run clears before its first kernel, in ONE launch: result keys (all ones), flags / violation marks / flag list /
    // every sub-batch's small counters / the pattern rows' norm accumulators / run counters (one contiguous zero span of the batch's
    // layout), and -- a batch of one sub-batch, while they are small -- its candidate rows.
    bool cand_filled = false;
    {
        FillArgs fa;
        memset(&fa, 0, sizeof(fa));
        auto add = [&](void* p, size_t bytes, uint32_t v) { fa.p[fa.n] = (uint32_t*)p; fa.words[fa.n] = (uint32_t)(bytes / 4); fa.value[fa.n] = v; ++fa.n; };
        add(keys, (size_t)2 * n_search * sizeof(uint64_t), 0xffffffffu);
        add(flags, b->lay.counters + align_up(sizeof(RunCounters), 256) - b->lay.flags, 0u);
        if (b->plan.subs.size() == 1) {
            const SubBatch& s0 = b->plan.subs[0];
            const WsLayout w0 = ws_layout(s0.pairs, s0.segs, s0.b0 - s0.a0);
            const size_t cand_bytes = (size_t)s0.pairs * FFT_ROW * sizeof(unsigned long long);
            if (cand_bytes <= ((size_t)8 << 20)) { add(b->mem + b->lay.ws + w0.cand, cand_bytes, 0xffffffffu); cand_filled = true; }
        }
        if (launch_fill(fa, st0) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
    }
    const unsigned run_seq = b->run_seq++;
    const bool ccm = b->method == SUSHI_HIP_METHOD_CCOEFF_NORMED;
    if (b->stats_pending && hipEventQuery(b->stats_ready) == hipSuccess) {
        b->stats_pending = false;
        const unsigned long long left = b->host_stats[0] - b->host_stats[1];
        b->last_transformed = b->host_stats[0];
        if (b->exclusion == SUSHI_HIP_EXCLUDE_AUTO) {
            // (three quarters: a batch HALF of whose searches find nothing -- a dub -- still gains from the exclusion on the other half)
            if ((double)left > 0.75 * (double)b->plan.pairs) { if (!b->suspended) b->suspended_at = run_seq; b->suspended = 1; }
            else b->suspended = 0;
        }
    }
    // (suspended: every 64th run looks again)
    const bool suspended_now = b->exclusion == SUSHI_HIP_EXCLUDE_AUTO && b->suspended && ((run_seq - b->suspended_at) & 63u) != 63u;
    b->last_suspended = suspended_now ? 1 : 0;
    bool excluded_any = false;
    // The lanes: the batch's own streams start behind the fill, the caller's stream goes on behind them (below).  Side by side pays
    // where the stages differ in what bounds them -- the band-split form; whole rows for every pair (the whole-row form, no
    // exclusion at all) are HBM traffic from the first kernel to the last and only contend: those runs keep their sub-batches on
    // the caller's stream, one after the other (measured at BASELINE configs[2]: unrelated audio 25.1 ms on one stream, 26.8 side by side).
    const bool whole_rows_throughout = suspended_now || b->exclusion == SUSHI_HIP_EXCLUDE_NEVER || b->exclusion == SUSHI_HIP_EXCLUDE_WHOLE ||
                                       ((b->exclusion == SUSHI_HIP_EXCLUDE_AUTO || b->exclusion == SUSHI_HIP_EXCLUDE_ALWAYS) && b->band == 0 &&
                                        b->band_decided_method == b->method);
    if (whole_rows_throughout && b->plan.whole_pending) {
        // the first run that wants the one-sub-batch cut makes it (host) and uploads its schedule and items behind the fill
        if (complete_whole_cut(b->descs, b->plan)) {
            const size_t o0 = b->plan.whole_order_first, o1 = b->plan.order.size(), i0 = b->plan.whole_items_first, i1 = b->plan.items.size();
            if (hipMemcpyAsync(b->mem + b->lay.order + o0 * sizeof(int32_t), b->plan.order.data() + o0, (o1 - o0) * sizeof(int32_t), hipMemcpyHostToDevice, st0) != hipSuccess ||
                hipMemcpyAsync(b->mem + b->lay.items + i0 * sizeof(int32_t), b->plan.items.data() + i0, (i1 - i0) * sizeof(int32_t), hipMemcpyHostToDevice, st0) != hipSuccess)
                return SUSHI_HIP_ELAUNCH;
            // (the copies read the handle's own vectors: a re-plan and the destructor wait for this event before they touch them)
            if (hipEventRecord(b->uploaded, st0) != hipSuccess) return SUSHI_HIP_ELAUNCH;
        } else {
            b->plan.whole_pending = false;                       // (cannot happen: the room was sized for it; the parts run one after the other then)
        }
    }
    const bool whole_cut = whole_rows_throughout && !b->plan.subs_whole.empty();
    const std::vector<SubBatch>& subs = whole_cut ? b->plan.subs_whole : b->plan.subs;
    const int lanes = whole_rows_throughout ? 1 : b->plan.lanes;
    b->last_whole_cut = whole_cut;
    hipStream_t lane_st[MAX_LANES] = {st0, st0, st0, st0};
    if (lanes > 1) {
        if (!b->fork && hipEventCreateWithFlags(&b->fork, hipEventDisableTiming) != hipSuccess) return SUSHI_HIP_ELAUNCH;
        if (hipEventRecord(b->fork, st0) != hipSuccess) return SUSHI_HIP_ELAUNCH;
        for (int l = 1; l < lanes; ++l) {
            // (stream priorities for the lanes -- the batch's own below the caller's, above it, one of each -- and an occupancy cap on
            // mac_kernel<1024> were measured flat: 8.31 - 8.55 ms whatever the setting, tools/experiments/README.md)
            if (!b->lane_stream[l] && !(b->lane_stream[l] = g_lane_pool.get(l))) return SUSHI_HIP_ELAUNCH;
            if (!b->lane_done[l] && hipEventCreateWithFlags(&b->lane_done[l], hipEventDisableTiming) != hipSuccess) return SUSHI_HIP_ELAUNCH;
            if (hipStreamWaitEvent(b->lane_stream[l], b->fork, 0) != hipSuccess) return SUSHI_HIP_ELAUNCH;
            lane_st[l] = b->lane_stream[l];
        }
    }

    b->last_band = -1;                                           // (the form of the last sub-batch of this run that went through the exclusion)
    // Sub-batches of a plan on lanes run side by side (sushi_fft_plan.inc "Lanes"); the others one after the other.
    for (size_t si = 0; si < subs.size(); ++si) {
        const SubBatch& sbt = subs[si];
        const hipStream_t st = lane_st[sbt.lane];
        const int n_sub = sbt.b0 - sbt.a0;
        const WsLayout wl = ws_layout(sbt.pairs, sbt.segs, n_sub);
        char* wsp = b->mem + b->lay.ws + (size_t)sbt.lane * b->plan.ws_lane;
        SubCounters* subcnt = (SubCounters*)(subc + si * SUBC_BYTES);
        uint32_t* tspec = (uint32_t*)(wsp + wl.tspec);
        uint4* y = (uint4*)(wsp + wl.y);
        unsigned long long* cand = (unsigned long long*)(wsp + wl.cand);
        int* pairmap = (int*)(wsp + wl.pairmap);
        float* pair_lb = (float*)(wsp + wl.pair_lb);
        TemplConsts* tconst = (TemplConsts*)(wsp + wl.tconst);
        TileDesc* tiles = (TileDesc*)(wsp + wl.tiles);
        int32_t* candbuf = (int32_t*)(wsp + wl.candbuf);
        uint4* tspec_low = (uint4*)(wsp + wl.tspec_low);
        uint4* ylow = (uint4*)(wsp + wl.ylow);
        float* tnorm_rest = tnorm_all + sbt.first_seg;
        int* scount = (int*)subcnt + SUBC_SCOUNT;

        hipEvent_t t0 = prof_begin(pc, st);
        TspecArgs ta;
        ta.src_raw = src->raw; ta.searches = searches_dev + sbt.a0; ta.n_sub = n_sub; ta.sub_first_seg = sbt.first_seg;
        ta.tspec = tspec; ta.sub_first_pair = sbt.first_pair; ta.pairmap = pairmap; ta.tconst = tconst;
        ta.src_s1 = src->s1; ta.src_s2 = src->s2; ta.centre = r.centre; ta.dst_stats = dst->stats; ta.method = b->method;
        ta.tspec_low = tspec_low; ta.tnorm_rest = tnorm_rest;
        if (src->dtype == SUSHI_HIP_F32) hipLaunchKernelGGL(tspec_kernel<float>, dim3((unsigned)sbt.segs), dim3(FT), 0, st, ta);
        else hipLaunchKernelGGL(tspec_kernel<uint8_t>, dim3((unsigned)sbt.segs), dim3(FT), 0, st, ta);
        if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
        prof_end(pc, t0, SUSHI_HIP_STAGE_TSPEC, st);

        // The exclusion costs a pass over Y (~16 ns per pair) and half a dozen launches (~60 us); transforming a pair ~37 ns:
        // it pays from ~3000 pairs on, plus two per search (the pairs transformed first are transformed either way).
        const bool exclude = b->exclusion == SUSHI_HIP_EXCLUDE_ALWAYS || b->exclusion == SUSHI_HIP_EXCLUDE_BAND ||
                             b->exclusion == SUSHI_HIP_EXCLUDE_WHOLE ||
                             (b->exclusion == SUSHI_HIP_EXCLUDE_AUTO && !suspended_now && sbt.pairs > 3000 + 2 * (int64_t)n_sub);
        excluded_any = excluded_any || exclude;
        BoundArgs ba;
        memset(&ba, 0, sizeof(ba));
        ba.dst_stats = dst->stats; ba.searches = searches_dev + sbt.a0; ba.sub_first_pair = sbt.first_pair;
        ba.first_search = sbt.a0; ba.dst_len = dst->n; ba.pairmap = pairmap; ba.tconst = tconst; ba.ubase = dst->base;
        ba.sbase = dst->base + (dst->blocks + 1); ba.nb = dst->blocks; ba.coarse = dst->coarse; ba.nc = dst->nc;
        ba.slb = (float*)(wsp + wl.slb); ba.n_sub = n_sub; ba.n_pairs = (int)sbt.pairs; ba.plist = (int*)(wsp + wl.plist);
        ba.slist = (int*)(wsp + wl.slist); ba.scount = scount; ba.order = order + sbt.order_first;
        ba.gkeys = gkeys; ba.pair_lb = pair_lb; ba.counters = counters;
        ba.acc = (float*)(wsp + wl.acc);
        ba.sub_first_seg = sbt.first_seg; ba.tnorm_rest = tnorm_rest; ba.znorm_rest = dst->znorm_rest; ba.norm_stride = dst->norm_stride;
        ba.band_votes = (int*)(wsp + wl.votes);
        ba.audit_mark = (unsigned char*)(wsp + wl.audit_mark); ba.audit_seq = run_seq; ba.audit_every = b->audit_every;
        // What a packed-half transform output (bound_low_kernel / bound_kernel) may be off by, in units of the largest pass-1 value:
        // every output is a sum of 64 pass-1 values through ROUNDING LEVELS of 2^-11 each -- a level at which the partial sums hold m
        // terms each costs 2^-11 m per value and 64 / m values meet in an output: 2^-11 64 per level whatever m.  Worst path: pass 1's
        // own result 1, its half-precision matrix (2^-12 sqrt 2 per entry, sum |inputs| <= 4 max |output| by Parseval) 2.8, pass 2's
        // twiddle 2 + its radix-16 butterflies 1 + 1 + 2 + 3 (h_bfly_root32: q = 0 / 8 one rounding, 4 / 12 two, others three on the
        // e - w o side), pass 3's twiddle 2 + radix 4: 1 + 1, four levels of half-precision twiddle constants at 2^-12 each = 2:
        // 18.8 levels = 0.59.  (Round 5's 0.29 counted 9: about right for independent roundings, not a worst case.)
        ba.worst_case = b->bound_model == SUSHI_HIP_BOUND_WORST_CASE ? 1 : 0;
        ba.half_err = ba.worst_case ? 0.6f : 0.29f;
        auto launch_slb = [&](const BoundArgs& x) {
            if (ccm) hipLaunchKernelGGL(slb_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3((unsigned)((sbt.pairs + 3) / 4)), dim3(256), 0, st, x);
            else hipLaunchKernelGGL(slb_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3((unsigned)((sbt.pairs + 3) / 4)), dim3(256), 0, st, x);
            return launch_ok();
        };

        // Which form of the exclusion: the band-split form multiplies, stores and transforms only the low band of
        // every spectrum and bounds the rest by the rows' norms -- a quarter of the bytes and a third of the instructions, IF the
        // streams keep most of their energy in the band (audio does; white noise does not).  Decided once per batch and method, on
        // the device's own numbers: with nothing at all from the low band, does the rest alone leave the bound room to exclude?
        // (One small kernel over the first excluded sub-batch's pairs and one 8 KB read-back, in the first run only.)
        int band = 0;
        if (exclude) {
            if (b->exclusion == SUSHI_HIP_EXCLUDE_BAND) band = 1;
            else if (b->exclusion == SUSHI_HIP_EXCLUDE_WHOLE) band = 0;
            else {
                if (b->band < 0 || b->band_decided_method != b->method) {
                    static_assert(VOTE_SLOTS * VOTE_STRIDE == 64 * 32, "ws_layout keeps room for the prediction's counters");
                    int slots[VOTE_SLOTS * VOTE_STRIDE];
                    if (hipMemsetAsync(ba.band_votes, 0, sizeof(slots), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
                    BoundArgs bp = ba;
                    bp.band = 2;
                    if (launch_slb(bp) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                    if (hipMemcpyAsync(slots, ba.band_votes, sizeof(slots), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess)
                        return SUSHI_HIP_ELAUNCH;
                    b->band_votes[0] = b->band_votes[1] = 0;
                    for (int v = 0; v < VOTE_SLOTS; ++v) { b->band_votes[0] += slots[v * VOTE_STRIDE]; b->band_votes[1] += slots[v * VOTE_STRIDE + 1]; }
                    // (measured at BASELINE configs[2]: 97 % of the pairs vote for it at 12 dB of noise on the source -- 9.7 ms against 17.5 for
                    // the whole-row form --, 87 % at 6 dB -- 12.5 against 17.5 --, 14 % at 0 dB -- 28.7 against 18.7)
                    b->band = b->band_votes[0] > 0 && (double)b->band_votes[1] >= 0.75 * (double)b->band_votes[0] ? 1 : 0;
                    b->band_decided_method = b->method;
                }
                band = b->band;
            }
            b->last_band = band;
        }

        // the multiply-accumulate over ALL pairs: of the low rows (band-split form) or of whole rows; `enable`: a device flag that
        // may call the launch off (the whole-row launch queued behind the survivors' list, below)
        auto launch_mac = [&](const bool low, const int* enable, const int* items_of_the_launch) {
            MacArgs ma;
            ma.spec_blocks = dst->blocks;
            ma.searches = searches_dev + sbt.a0; ma.tconst = tconst; ma.sub_first_seg = sbt.first_seg;
            ma.sub_first_pair = sbt.first_pair;
            ma.dummy = (uint4*)(wsp + wl.dummy);
            ma.enable = enable;
            if (low) { ma.spec = (const uint4*)dst->spec_low; ma.tspec = tspec_low; ma.y = ylow; }
            else { ma.spec = (const uint4*)dst->spec; ma.tspec = (const uint4*)tspec; ma.y = y; }
            for (int kern = 0; kern < 2; ++kern) {
                if (sbt.item_count[kern] == 0) continue;
                ma.items = items_of_the_launch + (size_t)(sbt.item_first[kern] - sbt.item_first[0]) * (1 + MAC_SPW);
                ma.n_items = sbt.item_count[kern];
                const int chunks = (low ? LROWE : ROWE) / MAC_BW;
                ma.chunk_group = std::min(sbt.chunk_group[kern], chunks / 8);
                const dim3 grid((unsigned)chunks * (unsigned)ma.n_items);
                if (low) {
                    if (kern == 0) hipLaunchKernelGGL(mac_kernel<LROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                    else hipLaunchKernelGGL(mac_long_kernel<LROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                } else {
                    if (kern == 0) hipLaunchKernelGGL(mac_kernel<ROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                    else hipLaunchKernelGGL(mac_long_kernel<ROWE>, grid, dim3(MAC_THREADS), 0, st, ma);
                }
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
            return SUSHI_HIP_OK;
        };
        t0 = prof_begin(pc, st);
        const int32_t* sub_items = items + (size_t)sbt.item_first[0] * (1 + MAC_SPW);         // (the sub-batch's two item lists lie next to each other)
        if (launch_mac(band != 0, nullptr, sub_items) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
        prof_end(pc, t0, SUSHI_HIP_STAGE_MAC, st);

        t0 = prof_begin(pc, st);
        // the candidate rows start empty: ifft_kernel writes only the entries that exist
        if (!cand_filled && hipMemsetAsync(cand, 0xff, (size_t)sbt.pairs * FFT_ROW * sizeof(unsigned long long), st) != hipSuccess)
            return SUSHI_HIP_ELAUNCH;
        IfftArgs ia;
        memset(&ia, 0, sizeof(ia));
        ia.y = (const uint2*)y; ia.dst_stats = dst->stats; ia.searches = searches_dev + sbt.a0; ia.n_sub = n_sub; ia.first_search = sbt.a0;
        ia.sub_first_pair = sbt.first_pair; ia.dst_len = dst->n; ia.delta = (float)delta; ia.cand = cand; ia.pair_lb = pair_lb; ia.gkeys = gkeys;
        ia.pairmap = pairmap; ia.tconst = tconst; ia.order = order + sbt.order_first;
        ia.urel = dst->urel; ia.nb = dst->blocks; ia.ubase = dst->base;
        ia.usrel = dst->usrel; ia.sbase = dst->base + (dst->blocks + 1);
        ia.flags = flags; ia.flag_list = flag_list + sbt.a0; ia.sub = subcnt; ia.tiles = tiles; ia.candbuf = candbuf;
        ia.cand_cap = (int)cand_capacity(sbt.pairs); ia.counters = counters;
        ia.viol = viol;
        auto launch_ifft = [&](const IfftArgs& x, unsigned grid) {
            if (x.count && !x.list_direct) {                         // a list whose length only the device knows: a fixed grid strides over its tail
                if (ccm) hipLaunchKernelGGL(ifft_list_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
                else hipLaunchKernelGGL(ifft_list_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
            } else {
                if (ccm) hipLaunchKernelGGL(ifft_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
                else hipLaunchKernelGGL(ifft_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(grid), dim3(FT), 0, st, x);
            }
            return launch_ok();
        };
        // the whole rows of LISTED pairs (band-split form: nothing but the low band exists until a pair is to be transformed)
        auto launch_mac_list = [&](const int* list, const int* count, int n_list, const int* dense_search, int long_only) {
            MacListArgs la;
            la.dense_search = dense_search; la.long_only = long_only;
            la.spec = (const uint4*)dst->spec; la.spec_blocks = dst->blocks; la.tspec = (const uint4*)tspec; la.y = y;
            la.searches = searches_dev + sbt.a0; la.tconst = tconst; la.pairmap = pairmap; la.list = list; la.count = count;
            la.n_list = n_list; la.sub_first_seg = sbt.first_seg; la.sub_first_pair = sbt.first_pair;
            const int64_t want = (int64_t)n_list * MACL_PARTS;
            hipLaunchKernelGGL(mac_list_kernel, dim3((unsigned)std::min<int64_t>(want, 256 * 32)), dim3(MACL_THREADS), 0, st, la);
            return launch_ok();
        };
        if (!exclude) {
            // every pair, in the L2-friendly schedule (what round 3 did for every batch)
            if (launch_ifft(ia, (unsigned)sbt.pairs) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            b->direct_pairs += sbt.pairs;
        } else {
            // A lower bound of every pair's scores first (three of the transform's four passes, no scoring); then the most
            // promising pair of every search, which leaves the search's threshold; then whatever the bound could not exclude
            // (header of bound_kernel)
            ba.band = band;
            ba.y = band ? (const uint2*)ylow : (const uint2*)y;
            // (bound_kernel adds to the pairs' accumulators; bound_low_kernel -- a wave per pair -- stores them)
            if (!band && hipMemsetAsync(ba.acc, 0, (size_t)sbt.pairs * 2 * sizeof(float), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
            {
                // waves striding over the pairs: a small multiple of what is resident at the kernel's registers (four workgroups of four
                // waves per CU, three for bound_low_kernel), fewer for a small batch.  Exactly what is resident -- persistent waves --
                // leaves the hardware nothing to balance with: bound_low_kernel alone 2.41 - 2.44 ms at BASELINE configs[2] with 1 x,
                // 2.36 with 2 x, 2.30 - 2.32 with 4 x and 8 x; next to other lanes' kernels 2 x is as good as it gets (8 x and more:
                // the per-workgroup prologue shows)
                const int per_pair = band ? 1 : 16;                  // bound_low_kernel: a wave per pair
                const int64_t want = (sbt.pairs * per_pair + BOUND_THREADS / 64 - 1) / (BOUND_THREADS / 64);
                const unsigned grid = (unsigned)std::min<int64_t>(want, 256 * (band ? 3 * (lanes > 1 ? 2 : 4) : 4));
                // (the row energies are the statistical model's: the worst case -- the default -- does without them)
                if (band) {
                    if (ba.worst_case) hipLaunchKernelGGL(bound_low_kernel<false>, dim3(grid), dim3(BOUND_THREADS), 0, st, ba);
                    else hipLaunchKernelGGL(bound_low_kernel<true>, dim3(grid), dim3(BOUND_THREADS), 0, st, ba);
                } else {
                    if (ba.worst_case) hipLaunchKernelGGL(bound_kernel<false>, dim3(grid), dim3(BOUND_THREADS), 0, st, ba);
                    else hipLaunchKernelGGL(bound_kernel<true>, dim3(grid), dim3(BOUND_THREADS), 0, st, ba);
                }
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                if (launch_slb(ba) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
            prof_end(pc, t0, SUSHI_HIP_STAGE_BOUND, st);
            t0 = prof_begin(pc, st);
            hipLaunchKernelGGL(pilot_kernel, dim3((unsigned)n_sub), dim3(64), 0, st, ba);
            if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            IfftArgs ip = ia;
            ip.slb = ba.slb; ip.audit_mark = nullptr;              // (the pairs transformed first are nobody's excluded pairs)
            ip.order = ba.plist; ip.count = nullptr;
            if (band && launch_mac_list(ba.plist, nullptr, n_sub, nullptr, 0) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            if (launch_ifft(ip, (unsigned)n_sub) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            hipLaunchKernelGGL(survivor_kernel, dim3((unsigned)((sbt.pairs + 255) / 256)), dim3(256), 0, st, ba);
            if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            const int* final_list = ba.slist;
            const int* final_count = ba.scount;
            if (band) {
                // the second look at what the bound left (header of bound_low_exact_kernel): sharper bound, shorter list
                ba.list = ba.slist; ba.list_count = ba.scount;
                ba.list2 = (int*)(wsp + wl.slist2); ba.list2_count = scount + 5;
                hipLaunchKernelGGL(bound_low_exact_kernel, dim3(256 * 4), dim3(BLE_T), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                if (ccm) hipLaunchKernelGGL(slb_list_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(256 * 2), dim3(256), 0, st, ba);
                else hipLaunchKernelGGL(slb_list_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(256 * 2), dim3(256), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                hipLaunchKernelGGL(survivor2_kernel, dim3(256), dim3(256), 0, st, ba);
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                final_list = ba.list2; final_count = ba.list2_count;
                // whole rows of what is left: pair by pair for the searches that left few (the usual case), by the dense
                // multiply-accumulate for the searches the bound could exclude little of (no match anywhere) -- decided per search and
                // regrouped into items of their own, on the device (dense_search_kernel, dense_repack_kernel)
                int* any_dense = scount + 4;                         // (zero since the run's first launch)
                int* dense = (int*)(wsp + wl.dense_search);
                int* ditems = (int*)(wsp + wl.ditems);
                hipLaunchKernelGGL(dense_search_kernel, dim3((unsigned)n_sub), dim3(64), 0, st, ba, dense, scount + 6);
                {
                    const size_t second = (size_t)(sbt.item_first[1] - sbt.item_first[0]) * (1 + MAC_SPW);
                    hipLaunchKernelGGL(dense_repack_kernel, dim3(2), dim3(REPACK_THREADS), 0, st, sub_items, sbt.item_count[0], sub_items + second,
                                       sbt.item_count[1], dense, n_sub, ditems, ditems + second, scount + 6, (int)(sbt.pairs / 8), any_dense);
                }
                if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                {
                    MacRowsArgs ra;
                    ra.spec = (const uint4*)dst->spec; ra.spec_blocks = dst->blocks; ra.tspec = (const uint4*)tspec; ra.y = y;
                    ra.searches = searches_dev + sbt.a0; ra.tconst = tconst; ra.mark = ba.audit_mark; ra.n_sub = n_sub;
                    ra.sub_first_seg = sbt.first_seg; ra.sub_first_pair = sbt.first_pair; ra.dense_search = dense;
                    const int64_t want = (int64_t)n_sub * MACL_PARTS;
                    if (sbt.item_count[0] > 0) hipLaunchKernelGGL(mac_rows_kernel<0>, dim3((unsigned)std::min<int64_t>(want, 256 * 32)), dim3(MACL_THREADS), 0, st, ra);
                    if (sbt.item_count[1] > 0) hipLaunchKernelGGL(mac_rows_kernel<1>, dim3((unsigned)std::min<int64_t>(want, 256 * 16)), dim3(MACL_THREADS), 0, st, ra);
                    if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                }
                if (sbt.long_patterns && launch_mac_list(final_list, final_count, (int)sbt.pairs, dense, 1) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
                if (launch_mac(false, any_dense, ditems) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
            ip.order = final_list; ip.count = final_count; ip.audit_mark = ba.audit_mark;
            // One workgroup per list slot up to what the list usually holds (an eighth of the pairs: empty slots there cost a
            // workgroup's launch each, ~1 ns), and a fixed grid striding over whatever lies beyond: the striding form alone runs
            // at half the rate per pair (the loop costs it registers), one workgroup per POSSIBLE slot cost 0.3 ms of empty launches.
            // (A batch whose LAST run listed more than that -- searches without a match, a dub's own speech -- gets a workgroup per
            // possible slot instead: 0.3 ms of empty launches at most, against half the rate on everything behind the first eighth.
            // `ifft` took 31.6 ms at BASELINE configs[2] on a dub with TM_CCOEFF_NORMED, 160 k pairs listed: bench.py --source dub.)
            int64_t direct64 = std::min<int64_t>(sbt.pairs, std::max<int64_t>(4096, sbt.pairs / 8));
            if ((double)b->last_transformed * (double)sbt.pairs > (double)direct64 * (double)b->plan.pairs) direct64 = sbt.pairs;   // (this sub-batch's share of it)
            // (nothing known yet -- a batch's first run, which is all a one-shot job has --: half of the pairs get a workgroup each;
            // 0.15 ms of empty launches where there is a match everywhere, against half the rate on ten times as many pairs where
            // there is not: a dub's first run 29.7 ms)
            else if (b->last_transformed == 0) direct64 = std::min<int64_t>(sbt.pairs, std::max<int64_t>(4096, sbt.pairs / 2));
            const unsigned direct = (unsigned)direct64;
            ip.list_first = 0; ip.list_direct = 1;
            if (launch_ifft(ip, direct) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            if ((int64_t)direct < sbt.pairs) {
                ip.list_first = (int)direct; ip.list_direct = 0;
                if (launch_ifft(ip, (unsigned)std::min<int64_t>(sbt.pairs - direct, 1024)) != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
            }
        }
        prof_end(pc, t0, SUSHI_HIP_STAGE_IFFT, st);

        t0 = prof_begin(pc, st);
        RefineParams rp;
        rp.r = r; rp.searches = searches_dev; rp.first_search = sbt.a0; rp.n_sub = n_sub; rp.sub_first_pair = sbt.first_pair;
        rp.cand = cand; rp.pair_lb = pair_lb; rp.gkeys = gkeys; rp.keys = keys; rp.flags = flags; rp.flag_list = flag_list + sbt.a0;
        rp.sub = subcnt; rp.counters = counters; rp.delta = (float)delta; rp.method = b->method;
        rp.citems = (int*)(wsp + wl.citems); rp.n_citems = scount + 1;
        rp.viol = viol;
        rp.early = reinterpret_cast<int4*>(b->early_out);
        ia.citems = rp.citems; ia.n_citems = rp.n_citems;
        int rc = launch_refine(rp, st);
        if (rc != SUSHI_HIP_OK) return rc;
        prof_end(pc, t0, SUSHI_HIP_STAGE_REFINE, st);

        // searches the lists could not finish: collect their candidates per tile, evaluate those exactly
        // (both kernels leave after one load when nothing is flagged)
        t0 = prof_begin(pc, st);
        if (b->method == SUSHI_HIP_METHOD_CCOEFF_NORMED)
            hipLaunchKernelGGL(collect_kernel<SUSHI_HIP_METHOD_CCOEFF_NORMED>, dim3(COLLECT_GRID), dim3(FT), 0, st, ia);
        else
            hipLaunchKernelGGL(collect_kernel<SUSHI_HIP_METHOD_SQDIFF_NORMED>, dim3(COLLECT_GRID), dim3(FT), 0, st, ia);
        if (launch_ok() != SUSHI_HIP_OK) return SUSHI_HIP_ELAUNCH;
        TileParams tp;
        tp.r = r; tp.searches = searches_dev; tp.tiles = tiles; tp.cand = candbuf; tp.keys = keys; tp.counters = counters; tp.sub = subcnt;
        tp.method = b->method;
        rc = launch_tiles(tp, st);
        if (rc != SUSHI_HIP_OK) return rc;
        prof_end(pc, t0, SUSHI_HIP_STAGE_FINISH, st);
    }
    for (int l = 1; l < lanes; ++l)
        if (hipEventRecord(b->lane_done[l], lane_st[l]) != hipSuccess || hipStreamWaitEvent(st0, b->lane_done[l], 0) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    hipEvent_t t0 = prof_begin(pc, st0);
    const int rc = launch_unpack(keys, n_search, b->method, out_idx_dev, out_score_dev, b->packed_out, st0);
    prof_end(pc, t0, SUSHI_HIP_STAGE_FINISH, st0);
    if (rc == SUSHI_HIP_OK && excluded_any && !b->stats_pending) {
        // what this run's exclusion left, for the runs after it (never waited for: the event is queried)
        if (!b->host_stats) b->host_stats = g_host_slots.take();                 // (none left: this batch's AUTO does not learn)
        if (b->host_stats && !b->stats_ready && hipEventCreateWithFlags(&b->stats_ready, hipEventDisableTiming) != hipSuccess) b->stats_ready = nullptr;
        if (b->host_stats && b->stats_ready &&
            hipMemcpyAsync(b->host_stats, &counters->pairs_transformed, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st0) == hipSuccess &&
            hipEventRecord(b->stats_ready, st0) == hipSuccess)
            b->stats_pending = true;
    }
    return rc;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

int sushi_hip_batch_diagnostics(SushiHipBatch* b, SushiHipBatchDiag* diag, float* ranking_err_host, int32_t* flagged_host) try {
    if (!b || !diag) return SUSHI_HIP_EINVAL;
    memset(diag, 0, sizeof(*diag));
    if (!b->ran || b->path != SUSHI_HIP_PATH_FFT) {
        if (ranking_err_host) memset(ranking_err_host, 0, (size_t)b->n * sizeof(float));
        if (flagged_host) memset(flagged_host, 0, (size_t)b->n * sizeof(int32_t));
        return SUSHI_HIP_OK;
    }
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    RunCounters c;
    if (hipMemcpy(&c, b->mem + b->lay.counters, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    diag->flagged = c.n_flagged; diag->all_positions = c.n_all_positions;
    diag->tiles_dense = (int64_t)c.tiles_dense; diag->tiles_sparse = (int64_t)c.tiles_sparse;
    diag->candidates = (int64_t)c.candidates;
    memcpy(&diag->max_bound_ratio, &c.max_ratio_bits, sizeof(float));
    memcpy(&diag->max_bound_ratio_noncandidate, &c.max_ratio_audit_bits, sizeof(float));
    diag->audited = (int64_t)c.audited;
    diag->pairs_transformed = (int64_t)c.pairs_transformed + b->direct_pairs;
    diag->excluded_audited = (int64_t)c.excluded_audited;
    memcpy(&diag->max_slb_ratio_excluded, &c.max_slb_ratio_bits, sizeof(float));
    diag->slb_violations = c.slb_violations;
    diag->band = b->last_band;
    diag->suspended = b->last_suspended;
    diag->band_votes[0] = b->band_votes[0]; diag->band_votes[1] = b->band_votes[1];
    diag->second_look_audited = (int64_t)c.second_look_audited;
    std::vector<int32_t> fl((size_t)b->n);
    if (hipMemcpy(fl.data(), b->mem + b->lay.flags, (size_t)b->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return SUSHI_HIP_ELAUNCH;
    if (flagged_host) memcpy(flagged_host, fl.data(), (size_t)b->n * sizeof(int32_t));
    if (ranking_err_host) {
        std::vector<unsigned long long> g((size_t)b->n);
        if (hipMemcpy(g.data(), b->mem + b->lay.keys + (size_t)b->n * sizeof(unsigned long long),
                      (size_t)b->n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
            return SUSHI_HIP_ELAUNCH;
        for (int k = 0; k < b->n; ++k) {
            const uint32_t bits = fl[k] ? 0u : (uint32_t)(g[k] & 0xffffffffull);     // flagged searches keep their threshold there
            memcpy(&ranking_err_host[k], &bits, sizeof(float));
        }
    }
    return SUSHI_HIP_OK;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

int sushi_hip_batch_pair_bounds(SushiHipBatch* b, float* slb_host, float* acc_host, int64_t* n_pairs) try {
    if (!b || !n_pairs) return SUSHI_HIP_EINVAL;
    if (!b->ran || b->path != SUSHI_HIP_PATH_FFT || b->plan.subs.empty()) { *n_pairs = 0; return SUSHI_HIP_OK; }
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    const SubBatch& sbt = b->last_whole_cut ? b->plan.subs_whole.back() : b->plan.subs.back();
    const WsLayout wl = ws_layout(sbt.pairs, sbt.segs, sbt.b0 - sbt.a0);
    const int64_t cap = *n_pairs;
    *n_pairs = sbt.pairs;
    if (cap < sbt.pairs) return SUSHI_HIP_ENOSPACE;
    const char* wsp = b->mem + b->lay.ws + (size_t)sbt.lane * b->plan.ws_lane;
    if (slb_host && hipMemcpy(slb_host, wsp + wl.slb, (size_t)sbt.pairs * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    if (acc_host && hipMemcpy(acc_host, wsp + wl.acc, (size_t)sbt.pairs * 2 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    return SUSHI_HIP_OK;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }

int sushi_hip_batch_workspace_view(SushiHipBatch* b, int which, void** ptr_dev, size_t* bytes) {
    if (!b || !ptr_dev || !bytes) return SUSHI_HIP_EINVAL;
    *ptr_dev = nullptr; *bytes = 0;
    if (!b->ran || b->path != SUSHI_HIP_PATH_FFT || b->plan.subs.empty()) return SUSHI_HIP_OK;
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    const SubBatch& sbt = b->last_whole_cut ? b->plan.subs_whole.back() : b->plan.subs.back();
    const WsLayout wl = ws_layout(sbt.pairs, sbt.segs, sbt.b0 - sbt.a0);
    char* wsp = b->mem + b->lay.ws + (size_t)sbt.lane * b->plan.ws_lane;
    switch (which) {
        case SUSHI_HIP_WS_TSPEC: *ptr_dev = wsp + wl.tspec; *bytes = (size_t)sbt.segs * ROW_BYTES; break;
        case SUSHI_HIP_WS_Y: *ptr_dev = wsp + wl.y; *bytes = (size_t)sbt.pairs * ROW_BYTES; break;
        case SUSHI_HIP_WS_TSPEC_LOW: *ptr_dev = wsp + wl.tspec_low; *bytes = (size_t)sbt.segs * LROW_BYTES; break;
        case SUSHI_HIP_WS_Y_LOW: *ptr_dev = wsp + wl.ylow; *bytes = (size_t)sbt.pairs * LROW_BYTES; break;
        default: return SUSHI_HIP_EINVAL;
    }
    return SUSHI_HIP_OK;
}

int sushi_hip_profile_begin(void) {
    for (ProfCall& c : g_prof)
        for (ProfSpan& sp : c.spans) { (void)hipEventDestroy(sp.t0); (void)hipEventDestroy(sp.t1); }
    g_prof.clear();
    g_prof_on = true;
    return SUSHI_HIP_OK;
}

int sushi_hip_profile_end(float* stage_ms, int max_calls, int* n_calls) try {
    g_prof_on = false;
    if (!stage_ms || !n_calls || max_calls < 0) return SUSHI_HIP_EINVAL;
    int out = 0;
    int rc = SUSHI_HIP_OK;
    for (ProfCall& c : g_prof) {
        if (out < max_calls && !c.spans.empty()) {
            float* row = stage_ms + (size_t)out * SUSHI_HIP_NSTAGES;
            for (int k = 0; k < SUSHI_HIP_NSTAGES; ++k) row[k] = 0.f;
            for (ProfSpan& sp : c.spans) {
                float ms = 0.f;
                if (hipEventSynchronize(sp.t1) != hipSuccess || hipEventElapsedTime(&ms, sp.t0, sp.t1) != hipSuccess) {
                    rc = SUSHI_HIP_ELAUNCH;
                    break;
                }
                if (sp.stage >= 0 && sp.stage < SUSHI_HIP_NSTAGES) row[sp.stage] += ms;
            }
            ++out;
        }
        for (ProfSpan& sp : c.spans) { (void)hipEventDestroy(sp.t0); (void)hipEventDestroy(sp.t1); }
    }
    g_prof.clear();
    *n_calls = out;
    return rc;
} catch (const std::bad_alloc&) { return SUSHI_HIP_ENOMEM; } catch (...) { return SUSHI_HIP_EINTERNAL; }     // nothing crosses the C boundary

}  // extern "C"

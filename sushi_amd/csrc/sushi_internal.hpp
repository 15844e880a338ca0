// sushi_amd/csrc/sushi_internal.hpp -- launchers shared between the two translation units of
// libsushi_hip.so (hidden visibility: not part of the C ABI).
#ifndef SUSHI_INTERNAL_HPP
#define SUSHI_INTERNAL_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sushi_hip.h"

namespace sushi {

struct StreamRefs {
    const float* dst_xc; const double* dst_s1; const double* dst_s2; int64_t dst_len;
    const float* src_xc; const double* src_s1; const double* src_s2; int64_t src_len;
    double centre;
    const void* dst_raw; const void* src_raw; int dtype;     // the samples as they are (refine_kernel); may be null for the direct path
};

// Exact float64 evaluation of the candidates of searches [first_search, first_search + n_sub).
int launch_refine(const StreamRefs& r, const SushiHipSearch* searches_dev, int first_search, int n_sub,
                  int sub_first_pair, const unsigned long long* cand_dev, unsigned long long* gkeys_dev,
                  float delta, unsigned long long* keys_dev, int* flags_dev, int n_search, hipStream_t st);
// Direct (MFMA) kernel over the searches the refinement flagged.
int launch_flagged(const StreamRefs& r, const SushiHipSearch* searches_dev, int n_search,
                          unsigned long long* keys_dev, const int* flags_dev, hipStream_t st);
int launch_unpack(const unsigned long long* keys_dev, int n, int32_t* out_idx_dev, float* out_score_dev, hipStream_t st);

}  // namespace sushi
#endif

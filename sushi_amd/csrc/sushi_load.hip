// sushi_amd/csrc/sushi_load.hip -- WavStream.__init__'s value pipeline on the GPU (gfx950).
//
// Reference wav.py:64-91 (PCM decode + channel-mean downmix; only the RIFF header walk stays on the host) and
// wav.py:113-156: nearest-neighbour
// decimation of each one-second chunk (cv2.resize INTER_NEAREST, wav.py:125-137), edge-replicated
// padding (:140-141), 3 x median clipping of the positive and the negative side (:145-148), scaling
// to [0, 1] (:150-151) and optional quantisation to uint8 (:153-156).  Every float32 operation is the
// one NumPy performs, in the same order (this translation unit is compiled with -ffp-contract=off),
// so the resulting stream is bit-identical to the host pipeline in sushi_amd/wav.py::_build.
//
// All kernels are single passes over the stream: HBM-bound, 4 B in / 4 (or 1) B out per sample (decode: the
// file's bytes in, 4 B out per frame).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sushi_hip.h"

namespace {

inline int launch_ok() { return hipGetLastError() == hipSuccess ? SUSHI_HIP_OK : SUSHI_HIP_ELAUNCH; }

// wav.py:64-91 DownmixedWavFile.readframes for a run of frames: every channel's sample as int16 (24-bit samples keep
// their top two bytes, wav.py:70-74) -> float32; channels summed left to right in float32 and divided by
// float(channels) (the reference's reduce(a + b) then `data /= float(channels_count)`).  One thread per frame;
// a wave reads 64 consecutive frames = one contiguous run of bytes.
template <int WIDTH>
__global__ __launch_bounds__(256)
void decode_downmix_kernel(const uint8_t* __restrict__ pcm, int64_t n_frames, int channels, float* __restrict__ mono) {
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < n_frames; f += (int64_t)gridDim.x * 256) {
        const uint8_t* __restrict__ p = pcm + f * (int64_t)(channels * WIDTH) + (WIDTH - 2);
        float acc = (float)(int16_t)((uint16_t)p[0] | ((uint16_t)p[1] << 8));
        for (int c = 1; c < channels; ++c) {
            const uint8_t* __restrict__ q = p + c * WIDTH;
            acc += (float)(int16_t)((uint16_t)q[0] | ((uint16_t)q[1] << 8));
        }
        if (channels > 1) acc /= (float)channels;
        mono[f] = acc;
    }
}

struct ResampleArgs {
    const float* raw;       // downmixed frames, n_raw of them
    int64_t n_raw;
    int32_t chunk;          // frames per one-second chunk (wav.py:105,126)
    int32_t nl_full;        // samples a full chunk becomes: round(chunk * downsample_rate)
    double scale_full;      // 1 / (nl_full / chunk)   (resizeNN's scale_x)
    int64_t n_full;         // number of full chunks
    int32_t rest;           // frames of the last, partial chunk (0 if none)
    int32_t nl_rest;
    double scale_rest;
    int64_t pad;            // padding_size
    int64_t total;          // 2 * pad + sample_count
    float* data;
};

// value of the unpadded stream at position u (0 <= u < total - 2*pad); 0 where the reference's
// buffer is never written (the oracle zero-fills there, oracle.py load_wav_stream)
__device__ __forceinline__ float source_value(const ResampleArgs& a, int64_t u) {
    if (a.nl_full == a.chunk && a.nl_rest == a.rest) return u < a.n_raw ? a.raw[u] : 0.f;     // downsample_rate == 1
    const int64_t full_len = a.n_full * (int64_t)a.nl_full;
    if (u < full_len) {
        const int64_t c = u / a.nl_full;
        const int x = (int)(u - c * a.nl_full);
        int sx = (int)floor((double)x * a.scale_full);
        sx = sx < a.chunk - 1 ? sx : a.chunk - 1;
        return a.raw[c * (int64_t)a.chunk + sx];
    }
    const int64_t x = u - full_len;
    if (x < a.nl_rest) {
        int sx = (int)floor((double)x * a.scale_rest);
        sx = sx < a.rest - 1 ? sx : a.rest - 1;
        return a.raw[a.n_full * (int64_t)a.chunk + sx];
    }
    return 0.f;
}

__global__ __launch_bounds__(256)
void resample_pad_kernel(ResampleArgs a) {
    const int64_t inner = a.total - 2 * a.pad;
    for (int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x; pos < a.total; pos += (int64_t)gridDim.x * 256) {
        int64_t u = pos - a.pad;                       // wav.py:140-141: both pads replicate the nearest inner sample
        u = u < 0 ? 0 : (u > inner - 1 ? inner - 1 : u);
        a.data[pos] = source_value(a, u);
    }
}

// Radix-select support: 256-bin histogram of one key byte over the samples of one side of zero
// whose higher key bytes equal `prefix`.  side 0: samples >= 0, key = bit pattern; side 1: samples
// <= 0, key = bit pattern of -x; both map -0.0 and +0.0 to key 0 (NumPy compares them equal).
__device__ __forceinline__ bool side_key(float x, int side, uint32_t* key) {
    if (side == 0) {
        if (!(x >= 0.f)) return false;
        *key = x == 0.f ? 0u : __float_as_uint(x);
    } else {
        if (!(x <= 0.f)) return false;
        *key = x == 0.f ? 0u : __float_as_uint(-x);
    }
    return true;
}

__global__ __launch_bounds__(256)
void radix_hist_kernel(const float* __restrict__ data, int64_t n, int side, uint32_t prefix, uint32_t mask, int shift,
                       unsigned long long* __restrict__ hist) {
    __shared__ unsigned int local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        uint32_t key;
        if (side_key(data[e], side, &key) && (key & mask) == prefix) atomicAdd(&local[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}

__global__ __launch_bounds__(256)
void normalise_kernel(float* __restrict__ data, int64_t n, float lo, float hi, float range, uint8_t* __restrict__ u8) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        float v = data[e];
        v = v < lo ? lo : v;                 // np.clip(data, min_value, max_value)   wav.py:148
        v = v > hi ? hi : v;
        v = v - lo;                          // data -= min_value                      wav.py:150
        v = v / range;                       // data /= (max_value - min_value)        wav.py:151
        data[e] = v;
        if (u8) {
            float q = v * 255.0f;            // wav.py:155  (two roundings, as NumPy does them)
            q = q + 0.5f;
            u8[e] = (uint8_t)q;              // astype('uint8'): truncation
        }
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));    // ~32 blocks per CU, grid-stride the rest
}

}  // namespace

extern "C" {

int sushi_hip_load_decode(const void* pcm_dev, int64_t n_frames, int32_t channels, int32_t sample_width, float* mono_dev,
                          void* hip_stream) {
    if (!pcm_dev || !mono_dev || n_frames < 0 || channels < 1) return SUSHI_HIP_EINVAL;
    if (sample_width != 2 && sample_width != 3) return SUSHI_HIP_EINVAL;     // wav.py:75-76: 'Unsupported sample width'
    if ((uintptr_t)mono_dev & 3) return SUSHI_HIP_EALIGN;
    if (n_frames == 0) return SUSHI_HIP_OK;
    const int64_t want = (n_frames + 255) / 256;
    const unsigned grid = (unsigned)(want < 65536 ? want : 65536);
    if (sample_width == 2)
        hipLaunchKernelGGL(decode_downmix_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream,
                           (const uint8_t*)pcm_dev, n_frames, channels, mono_dev);
    else
        hipLaunchKernelGGL(decode_downmix_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream,
                           (const uint8_t*)pcm_dev, n_frames, channels, mono_dev);
    return launch_ok();
}

int sushi_hip_load_resample(const float* raw_dev, int64_t n_raw, int32_t chunk, int32_t nl_full, double scale_full,
                            int64_t n_full, int32_t rest, int32_t nl_rest, double scale_rest,
                            int64_t pad, int64_t total, float* data_dev, void* hip_stream) {
    if (!raw_dev || !data_dev || n_raw <= 0 || chunk <= 0 || nl_full < 0 || n_full < 0 || rest < 0 || nl_rest < 0 ||
        pad < 0 || total <= 2 * pad)
        return SUSHI_HIP_EINVAL;
    if (n_full * (int64_t)chunk + rest > n_raw) return SUSHI_HIP_EINVAL;
    ResampleArgs a;
    a.raw = raw_dev; a.n_raw = n_raw; a.chunk = chunk; a.nl_full = nl_full; a.scale_full = scale_full; a.n_full = n_full;
    a.rest = rest; a.nl_rest = nl_rest; a.scale_rest = scale_rest; a.pad = pad; a.total = total; a.data = data_dev;
    hipLaunchKernelGGL(resample_pad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)hip_stream, a);
    return launch_ok();
}

int sushi_hip_load_histogram(const float* data_dev, int64_t n, int side, uint32_t prefix, uint32_t mask, int shift,
                             uint64_t* hist_dev, void* hip_stream) {
    if (!data_dev || !hist_dev || n <= 0 || (side != 0 && side != 1) || shift < 0 || shift > 24) return SUSHI_HIP_EINVAL;
    hipStream_t st = (hipStream_t)hip_stream;
    if (hipMemsetAsync(hist_dev, 0, 256 * sizeof(uint64_t), st) != hipSuccess) return SUSHI_HIP_ELAUNCH;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(grid_for(n)), dim3(256), 0, st, data_dev, n, side, prefix, mask, shift,
                       (unsigned long long*)hist_dev);
    return launch_ok();
}

int sushi_hip_load_normalise(float* data_dev, int64_t n, float lo, float hi, float range, uint8_t* u8_dev,
                             void* hip_stream) {
    if (!data_dev || n <= 0) return SUSHI_HIP_EINVAL;
    hipLaunchKernelGGL(normalise_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)hip_stream, data_dev, n, lo, hi,
                       range, u8_dev);
    return launch_ok();
}

}  // extern "C"

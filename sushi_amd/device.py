"""Device-side state and launches.  PyTorch owns HBM allocations and HIP streams (plumbing);
all arithmetic is in libsushi_hip.so.

* ``DeviceStream``  -- the HBM mirror of one ``WavStream.data`` row: centred float32 samples
  and float64 prefix sums (built once by ``sushi_hip_prepare_stream``).
* ``SearchBatch``   -- a batch of (pattern, window) descriptors resident in HBM; ``run()`` is one
  pass of the hot path (memset + match kernel + unpack) and nothing else.
"""
import os

import numpy as np
import torch

from . import _native
from .common import SushiError

_DTYPE_CODE = {np.dtype(np.uint8): _native.U8, np.dtype(np.float32): _native.F32}


def _require_gpu(device=None):
    if not torch.cuda.is_available():
        raise SushiError("sushi_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is False "
                         "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def _raw_stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class DeviceStream(object):
    """HBM-resident, match-ready form of a 1-D sample row (uint8 or float32)."""

    def __init__(self, samples, device=None):
        """`samples`: a host array (1-D or (1, N), uint8 / float32) -- uploaded -- or a 1-D torch tensor
        of those dtypes that already lives on the GPU (used as is)."""
        on_device = isinstance(samples, torch.Tensor)
        if on_device:
            if samples.dim() != 1 or not samples.is_cuda or not samples.is_contiguous():
                raise SushiError("DeviceStream expects a contiguous 1-D CUDA tensor")
            np_dtype = {torch.uint8: np.dtype(np.uint8), torch.float32: np.dtype(np.float32)}.get(samples.dtype)
            if np_dtype is None:
                raise SushiError("Unknown sample type of WAV stream, must be uint8 or float32")
            device = samples.device
        else:
            samples = np.ascontiguousarray(samples)
            if samples.ndim == 2 and samples.shape[0] == 1:
                samples = samples[0]
            if samples.ndim != 1:
                raise SushiError("DeviceStream expects a 1-D (or (1, N)) sample array")
            if samples.dtype not in _DTYPE_CODE:
                raise SushiError("Unknown sample type of WAV stream, must be uint8 or float32")
            np_dtype = samples.dtype
        if samples.shape[0] < 1:
            raise SushiError("empty sample array")
        self.device = _require_gpu(device)
        self.dtype = np_dtype
        self.dtype_code = _DTYPE_CODE[np_dtype]
        self.n = int(samples.shape[0])
        L = _native.lib()
        _native.check(L.sushi_hip_device_ok(), "device check")
        self.centre = float(L.sushi_hip_centre(self.dtype_code))
        with torch.cuda.device(self.device):
            raw = samples if on_device else torch.from_numpy(samples).to(self.device, non_blocking=False)
            self.xc = torch.empty(self.n, dtype=torch.float32, device=self.device)
            self.s1 = torch.empty(self.n + 1, dtype=torch.float64, device=self.device)
            self.s2 = torch.empty(self.n + 1, dtype=torch.float64, device=self.device)
            # window energies for the FFT path's scoring: float32 prefix of the uncentred squares relative
            # to per-block float64 bases
            self.urel = torch.empty(self.n + 1, dtype=torch.float32, device=self.device)
            base_bytes = int(L.sushi_hip_prepare_base_bytes(self.n))
            self.base = torch.empty(base_bytes // 8, dtype=torch.float64, device=self.device)
            rc = L.sushi_hip_prepare_stream(raw.data_ptr(), self.dtype_code, self.n,
                                            self.xc.data_ptr(), self.s1.data_ptr(), self.s2.data_ptr(),
                                            self.urel.data_ptr(), self.base.data_ptr(), base_bytes,
                                            _raw_stream(self.device))
            _native.check(rc, "sushi_hip_prepare_stream")
        self.raw = raw             # the FFT path reads the samples themselves (spectra, exact refinement)
        self._spec = None

    def spectra(self):
        """Block DFTs of this stream for the FFT path (built once, on first use as a search target)."""
        if self._spec is None:
            L = _native.lib()
            nbytes = int(L.sushi_hip_spectra_bytes(self.n))
            with torch.cuda.device(self.device):
                spec = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
                rc = L.sushi_hip_prepare_spectra(self.raw.data_ptr(), self.dtype_code, self.n, spec.data_ptr(), nbytes,
                                                 _raw_stream(self.device))
                _native.check(rc, "sushi_hip_prepare_spectra")
            self._spec = spec
        return self._spec

    def nbytes(self):
        return self.raw.numel() * self.raw.element_size() + self.xc.numel() * 4 + (self.s1.numel() + self.s2.numel() + self.base.numel()) * 8 + self.urel.numel() * 4 + \
            (0 if self._spec is None else self._spec.numel() * 4)


def choose_variant(n_pos, total_waves_wanted=4096):
    """Largest tile whose grid still gives the chip (256 CUs x 4 SIMDs) a few waves per SIMD."""
    tiles = _native.variant_tiles()
    waves = [1, 4, 4]
    n_pos = np.asarray(n_pos, dtype=np.int64)
    best = 0
    for v, tp in enumerate(tiles):
        nt = int(((n_pos + tp - 1) // tp).sum())
        if nt * waves[v] >= total_waves_wanted:
            best = v
    return best


DEFAULT_DELTA = 2e-5            # FFT path: score margin for the exact re-evaluation
DEFAULT_FFT_WORKSPACE = 16 << 30  # bytes of scratch per batch at most (sub-batches are sized to fit; 288 GB HBM)


def default_path():
    p = os.environ.get("SUSHI_HIP_PATH", "fft")
    if p not in ("fft", "direct"):
        raise SushiError("SUSHI_HIP_PATH must be 'fft' or 'direct'")
    return p


class SearchBatch(object):
    """Descriptors of a batch of searches, resident in HBM, plus the output buffers.

    tmpl_off / tmpl_len : pattern = src row [tmpl_off, tmpl_off + tmpl_len)
    win_start / n_pos   : search_source = dst row [win_start, win_start + n_pos + tmpl_len - 1)
    (exactly the two arrays wav.py:184-185 hands to cv2.matchTemplate)

    path = 'fft' (default): overlap-save FFT scores + exact re-evaluation of the near-minimum
    positions (sushi_hip_match_batch_fft); path = 'direct': the exact-f32 MFMA sliding dot product
    (sushi_hip_match_batch, `variant` picks its tile size).  Same results either way.
    """

    def __init__(self, dst, src, tmpl_off, tmpl_len, win_start, n_pos, variant=None, path=None,
                 delta=DEFAULT_DELTA, workspace_bytes=None):
        if dst.device != src.device:
            raise SushiError("dst and src streams live on different devices")
        if dst.dtype != src.dtype:
            raise SushiError("pattern and stream sample types differ (cv2.matchTemplate asserts equal types)")
        self.dst, self.src = dst, src
        tmpl_off = np.asarray(tmpl_off, dtype=np.int64).reshape(-1)
        tmpl_len = np.asarray(tmpl_len, dtype=np.int64).reshape(-1)
        win_start = np.asarray(win_start, dtype=np.int64).reshape(-1)
        n_pos = np.asarray(n_pos, dtype=np.int64).reshape(-1)
        n = tmpl_off.shape[0]
        if not (tmpl_len.shape[0] == win_start.shape[0] == n_pos.shape[0] == n) or n == 0:
            raise SushiError("descriptor arrays must be non-empty and of equal length")
        if (tmpl_len < 1).any():
            raise SushiError("empty pattern")
        if (n_pos < 1).any():
            raise SushiError("pattern is longer than the search window (cv2.error in the reference)")
        if (tmpl_off < 0).any() or (tmpl_off + tmpl_len > src.n).any():
            raise SushiError("pattern slice outside the source stream")
        if (win_start < 0).any() or (win_start + n_pos + tmpl_len - 1 > dst.n).any():
            raise SushiError("search window outside the destination stream")
        if (n_pos > 0x7fffffff - 65536).any() or (tmpl_len > 0x7fffffff - 65536).any():
            raise SushiError("search too large")
        self.n = n
        self.sub_batches = 1
        self.path = default_path() if path is None else path
        if self.path not in ("fft", "direct"):
            raise SushiError("path must be 'fft' or 'direct'")
        if self.path == "fft":
            self.variant = len(_native.variant_tiles()) - 1      # the fallback kernel's tile size
        else:
            self.variant = choose_variant(n_pos) if variant is None else int(variant)
        tp = _native.variant_tiles()[self.variant]
        tiles = (n_pos + tp - 1) // tp
        first = np.zeros(n, dtype=np.int64)
        np.cumsum(tiles[:-1], out=first[1:])
        self.n_tiles = int(tiles.sum())
        if self.n_tiles > 0x7fffffff:
            raise SushiError("too many tiles in one batch")
        desc = np.zeros(n, dtype=_native.SEARCH_DTYPE)
        desc["tmpl_off"], desc["win_start"] = tmpl_off, win_start
        desc["tmpl_len"], desc["n_pos"], desc["first_tile"] = tmpl_len, n_pos, first
        if self.path == "fft":
            pairs, segs = _native.fft_layout(win_start, n_pos, tmpl_len)
            if pairs.sum() > 0x7fffffff or segs.sum() > 0x7fffffff:
                raise SushiError("too many blocks in one batch")
            desc["first_pair"][1:] = np.cumsum(pairs[:-1])
            desc["first_seg"][1:] = np.cumsum(segs[:-1])
            L = _native.lib()
            # what the most demanding single search needs (pairs and segments weigh differently, and the
            # alignment padding is per array: ask the library for every distinct (pairs, segments) shape)
            shapes = np.unique(np.stack([pairs, segs], axis=1), axis=0)
            need_one = max(int(L.sushi_hip_fft_workspace_bytes(int(p_), int(s_), 1)) for p_, s_ in shapes)
            need_all = int(L.sushi_hip_fft_workspace_bytes(int(pairs.sum()), int(segs.sum()), n))
            if workspace_bytes is None:
                workspace_bytes = int(os.environ.get("SUSHI_HIP_FFT_WS_MB", DEFAULT_FFT_WORKSPACE >> 20)) << 20
            self.ws_bytes = max(need_one, min(need_all, int(workspace_bytes)))
            self.delta = float(delta)
            self.fft_pairs, self.fft_segs = int(pairs.sum()), int(segs.sum())
            self.sub_batches = int(L.sushi_hip_fft_sub_batches(desc.ctypes.data, n, self.ws_bytes))
            if self.sub_batches < 1:
                _native.check(self.sub_batches, "sushi_hip_fft_sub_batches")
            # L2-friendly schedule of the inverse-transform workgroups (host side, once per batch)
            self.host_order = np.empty(self.fft_pairs, np.int32)
            _native.check(L.sushi_hip_fft_pair_order(desc.ctypes.data, n, self.ws_bytes, self.host_order.ctypes.data,
                                                     self.fft_pairs), "sushi_hip_fft_pair_order")
        self.host_desc = desc
        # algorithmic work of this batch: 2*P*M flop; every search and pattern sample read
        # once (4 bytes for float32 streams, 1 for uint8) and 8 bytes out per search
        self.flops = float((2.0 * n_pos.astype(np.float64) * tmpl_len.astype(np.float64)).sum())
        width = float(dst.dtype.itemsize)
        self.algorithmic_bytes = float((width * (n_pos + tmpl_len - 1) + width * tmpl_len + 8.0).sum())
        dev = dst.device
        with torch.cuda.device(dev):
            self.desc = torch.from_numpy(desc.view(np.uint8).reshape(-1)).to(dev)
            self.keys = torch.empty(2 * n, dtype=torch.int64, device=dev)
            self.out_idx = torch.empty(n, dtype=torch.int32, device=dev)
            self.out_score = torch.empty(n, dtype=torch.float32, device=dev)
            if self.path == "fft":
                self.flags = torch.zeros(2 * n + 2, dtype=torch.int32, device=dev)
                self.ws = torch.empty((self.ws_bytes + 255) // 256 * 64, dtype=torch.float32, device=dev)
                self.spec = dst.spectra()
                self.order = torch.from_numpy(self.host_order).to(dev)

    def run(self, hip_stream=None):
        """One pass of the hot path over this batch (asynchronous)."""
        L = _native.lib()
        dst, src = self.dst, self.src
        st = _raw_stream(dst.device) if hip_stream is None else hip_stream
        if self.path == "fft":
            rc = L.sushi_hip_match_batch_fft(dst.xc.data_ptr(), dst.s1.data_ptr(), dst.s2.data_ptr(), dst.n,
                                             dst.urel.data_ptr(), dst.base.data_ptr(), self.spec.data_ptr(),
                                             src.xc.data_ptr(), src.s1.data_ptr(), src.s2.data_ptr(), src.n,
                                             dst.raw.data_ptr(), src.raw.data_ptr(), dst.dtype_code,
                                             _native.SQDIFF_NORMED,
                                             self.desc.data_ptr(), self.host_desc.ctypes.data, self.n, self.delta,
                                             self.ws.data_ptr(), self.ws_bytes,
                                             self.keys.data_ptr(), self.flags.data_ptr(), self.order.data_ptr(),
                                             self.out_idx.data_ptr(), self.out_score.data_ptr(), st)
            _native.check(rc, "sushi_hip_match_batch_fft")
            return self.out_idx, self.out_score
        rc = L.sushi_hip_match_batch(dst.xc.data_ptr(), dst.s1.data_ptr(), dst.s2.data_ptr(), dst.n,
                                     src.xc.data_ptr(), src.s1.data_ptr(), src.s2.data_ptr(), src.n,
                                     dst.centre, _native.SQDIFF_NORMED,
                                     self.desc.data_ptr(), self.n, self.n_tiles, self.variant,
                                     self.keys.data_ptr(), self.out_idx.data_ptr(), self.out_score.data_ptr(), st)
        _native.check(rc, "sushi_hip_match_batch")
        return self.out_idx, self.out_score

    def results(self):
        """(idx int32 ndarray, score float32 ndarray) -- synchronises."""
        return self.out_idx.cpu().numpy(), self.out_score.cpu().numpy()

    def ranking_errors(self):
        """FFT path: |f32 FFT score - exact score| at every search's result position in the last run()
        (0 for searches a fallback kernel finished) -- to be compared with delta / 2."""
        if self.path != "fft":
            return np.zeros(self.n, np.float32)
        return self.keys[self.n:2 * self.n].cpu().numpy().astype(np.uint64).astype(np.uint32).view(np.float32)

    def fallback_count(self):
        """FFT path: how many searches of the last run() were finished by a fallback kernel (every position evaluated)."""
        return int(self.flags[self.n].item()) if self.path == "fft" else 0

"""Device-side state and launches.  PyTorch owns HBM allocations and HIP streams (plumbing);
all arithmetic is in libsushi_hip.so, behind the two handles of its C ABI (include/sushi_hip.h).

* ``DeviceStream``  -- the HBM form of one ``WavStream.data`` row (``SushiHipStream``): the samples, float64
  prefix sums, block-relative window energies and -- for streams that are searched -- block spectra, built
  once per stream.
* ``SearchBatch``   -- a batch of (pattern, window) requests resident in HBM (``SushiHipBatch``); ``run()`` is one
  pass of the hot path and nothing else.
* ``warm_up``       -- the process's one-time GPU start-up (context, code object), paid when the caller chooses.
"""
import ctypes
import os
import threading
import time

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


_warms = {}                    # device index -> {"thread", "done", "ms", "error"}: one start-up per device (it was one per process)
_warm_lock = threading.Lock()


def _warm_key(device):
    if device is None:
        return torch.cuda.current_device() if torch.cuda.is_available() else -1
    d = torch.device(device)
    return d.index if d.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else -1)


def warm_up(device=None, background=False):
    """Pay this process's one-time GPU start-up NOW: the HIP context's first real use and the load of the library's code
    object -- 0.09-0.15 s that otherwise sit in front of the first stream (tools/setup_probe.py: a 64 K-sample stream costs 93 ms
    as the first call of a process, a 2-h stream 8 ms as a later one).  A one-shot job calls it first thing, with
    background=True, and has it behind itself by the time its audio is demuxed / decoded (the reference spends seconds there,
    sushi.py:649-650 ffmpeg demux); a later call -- or the first DeviceStream on that device -- waits for it.  Idempotent per device;
    returns the milliseconds the start-up took (None while a background one is still running).  Do not fork after calling it."""
    key = _warm_key(device)
    with _warm_lock:
        w = _warms.setdefault(key, {"thread": None, "done": False, "ms": None, "error": None})
        if w["thread"] is None:
            def work():
                t0 = time.perf_counter()
                try:
                    dev = _require_gpu(device)
                    with torch.cuda.device(dev):
                        tiny = DeviceStream(np.linspace(0.0, 1.0, 65536, dtype=np.float32), device=dev, _wait_for_warm_up=False)
                        tiny.searchable()
                        # the lanes' HIP streams and the pinned words a run's counts come back through (a large batch's first run
                        # made them itself: 11 ms of it)
                        _native.check(_native.lib().sushi_hip_device_prepare(), "sushi_hip_device_prepare")
                        torch.cuda.synchronize(dev)
                except Exception as e:                  # (the real call that follows raises the same, where it can be handled)
                    w["error"] = e
                w["ms"] = (time.perf_counter() - t0) * 1e3
                w["done"] = True
            w["thread"] = threading.Thread(target=work, name="sushi_amd-warm-up-%s" % key, daemon=True)
            w["thread"].start()
        t = w["thread"]
    if background:
        return w["ms"] if w["done"] else None
    t.join()
    if w["error"] is not None:
        raise w["error"]
    return w["ms"]


def _join_warm_up(device=None):
    """A warm-up in flight on this device finishes before anything else touches it (its errors are the real call's to raise)."""
    w = _warms.get(_warm_key(device))
    if w is None:
        return
    t = w["thread"]
    if t is not None and not w["done"] and t is not threading.current_thread():
        t.join()


def _raw_stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _buffer(nbytes, device):
    """HBM for a handle (the library allocates nothing itself); the caching allocator's blocks are 512-byte aligned."""
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if buf.data_ptr() % 256:
        raise SushiError("device allocation is not 256-byte aligned")
    return buf


class DeviceStream(object):
    """HBM-resident, match-ready form of a 1-D sample row (uint8 or float32)."""

    def __init__(self, samples, device=None, _wait_for_warm_up=True):
        """`samples`: a host array (1-D or (1, N), uint8 / float32) -- uploaded -- or a 1-D torch tensor
        of those dtypes that already lives on the GPU (used as is)."""
        self._handle = None
        if _wait_for_warm_up:
            _join_warm_up(samples.device if isinstance(samples, torch.Tensor) and samples.is_cuda else device)
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
        self._spec_mem = None
        with torch.cuda.device(self.device):
            self.raw = samples if on_device else torch.from_numpy(samples).to(self.device, non_blocking=False)
            self._mem = _buffer(L.sushi_hip_stream_bytes(self.n, self.dtype_code, 0), self.device)
            h = ctypes.c_void_p()
            rc = L.sushi_hip_stream_create(self.raw.data_ptr(), self.dtype_code, self.n, 0, self._mem.data_ptr(),
                                           self._mem.numel(), _raw_stream(self.device), ctypes.byref(h))
            _native.check(rc, "sushi_hip_stream_create")
        self._handle = h

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _native.lib().sushi_hip_stream_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._handle

    def searchable(self):
        """Attach the block spectra (built once, on first use as a search target); returns the handle."""
        if self._spec_mem is None:
            L = _native.lib()
            with torch.cuda.device(self.device):
                mem = _buffer(L.sushi_hip_stream_spectra_bytes(self.n), self.device)
                _native.check(L.sushi_hip_stream_add_spectra(self._handle, mem.data_ptr(), mem.numel(),
                                                             _raw_stream(self.device)), "sushi_hip_stream_add_spectra")
            self._spec_mem = mem
        return self._handle

    def _view(self, which, dtype):
        p, nb = ctypes.c_void_p(), ctypes.c_size_t()
        _native.check(_native.lib().sushi_hip_stream_view(self._handle, which, ctypes.byref(p), ctypes.byref(nb)),
                      "sushi_hip_stream_view")
        if not p.value:
            return None
        owner = self._spec_mem if which in (_native.VIEW_SPECTRA, _native.VIEW_SPECTRA_LOW, _native.VIEW_ZNORM_REST) else self._mem
        off = p.value - owner.data_ptr()
        return owner[off:off + nb.value].view(dtype)

    # the parts of the prepared stream, as tensors over the handle's buffer (tests, diagnostics)
    xc = property(lambda self: self._view(_native.VIEW_XC, torch.float32))
    s1 = property(lambda self: self._view(_native.VIEW_S1, torch.float64))
    s2 = property(lambda self: self._view(_native.VIEW_S2, torch.float64))
    urel = property(lambda self: self._view(_native.VIEW_UREL, torch.float32))
    usrel = property(lambda self: self._view(_native.VIEW_USREL, torch.float32))      # [n + 1][2] flattened: (urel, srel) pairs
    base1 = property(lambda self: self._view(_native.VIEW_BASE1, torch.float64))
    base = property(lambda self: self._view(_native.VIEW_BASE, torch.float64))

    def spectra(self):
        self.searchable()
        return self._view(_native.VIEW_SPECTRA, torch.float16)      # (re, im) pairs, one power-of-two scale per stream

    def nbytes(self):
        return self.raw.numel() * self.raw.element_size() + self._mem.numel() + \
            (0 if self._spec_mem is None else self._spec_mem.numel())


DEFAULT_DELTA = 2e-5            # FFT path: floor of the score margin for the exact re-evaluation
DEFAULT_FFT_WORKSPACE = 16 << 30  # bytes of scratch per batch at most (sub-batches are sized to fit; 288 GB HBM)


from .distributed import fft_layout_host, search_work     # noqa: E402,F401  (host arithmetic; kept importable from here)


def default_path():
    p = os.environ.get("SUSHI_HIP_PATH", "fft")
    if p not in ("fft", "direct"):
        raise SushiError("SUSHI_HIP_PATH must be 'fft' or 'direct'")
    return p


def _checked_requests(dst, src, tmpl_off, tmpl_len, win_start, n_pos):
    """The SushiHipRequest array of a batch (include/sushi_hip.h), after the checks the reference leaves to NumPy and cv2."""
    tmpl_off = np.asarray(tmpl_off, dtype=np.int64).reshape(-1)
    tmpl_len = np.asarray(tmpl_len, dtype=np.int64).reshape(-1)
    win_start = np.asarray(win_start, dtype=np.int64).reshape(-1)
    n_pos = np.asarray(n_pos, dtype=np.int64).reshape(-1)
    n = tmpl_off.shape[0]
    if not (tmpl_len.shape[0] == win_start.shape[0] == n_pos.shape[0] == n) or n == 0:
        raise SushiError("descriptor arrays must be non-empty and of equal length")
    if n <= 4:
        # (a drop-in call is a batch of one, a triple one of three: a dozen NumPy reductions over one element each were a
        # tenth of such a call -- tools/call_breakdown.py; the same checks in the same order on plain ints)
        o_, m_, w_, p_ = tmpl_off.tolist(), tmpl_len.tolist(), win_start.tolist(), n_pos.tolist()
        empty = any(m < 1 for m in m_)
        too_long = any(p < 1 for p in p_)
        bad_src = any(o < 0 or o + m > src.n for o, m in zip(o_, m_))
        bad_dst = any(w < 0 or w + p + m - 1 > dst.n for w, p, m in zip(w_, p_, m_))
        too_large = any(p > 0x7fffffff - 65536 or m > 0x7fffffff - 65536 for p, m in zip(p_, m_))
    else:
        empty = bool((tmpl_len < 1).any())
        too_long = bool((n_pos < 1).any())
        bad_src = bool((tmpl_off < 0).any() or (tmpl_off + tmpl_len > src.n).any())
        bad_dst = bool((win_start < 0).any() or (win_start + n_pos + tmpl_len - 1 > dst.n).any())
        too_large = bool((n_pos > 0x7fffffff - 65536).any() or (tmpl_len > 0x7fffffff - 65536).any())
    if empty:
        raise SushiError("empty pattern")
    if too_long:
        raise SushiError("pattern is longer than the search window (cv2.error in the reference)")
    if bad_src:
        raise SushiError("pattern slice outside the source stream")
    if bad_dst:
        raise SushiError("search window outside the destination stream")
    if too_large:
        raise SushiError("search too large")
    req = np.zeros(n, dtype=_native.REQUEST_DTYPE)
    req["tmpl_off"], req["win_start"], req["tmpl_len"], req["n_pos"] = tmpl_off, win_start, tmpl_len, n_pos
    return req


class SearchBatch(object):
    """Requests of a batch of searches, resident in HBM, plus the output buffers.

    tmpl_off / tmpl_len : pattern = src row [tmpl_off, tmpl_off + tmpl_len)
    win_start / n_pos   : search_source = dst row [win_start, win_start + n_pos + tmpl_len - 1)
    (exactly the two arrays wav.py:184-185 hands to cv2.matchTemplate)

    path = 'fft' (default): overlap-save FFT scores + exact re-evaluation of the positions that can be the
    minimum; path = 'direct': the exact-f32 MFMA sliding dot product (`variant` picks its tile size).
    Same results either way.

    method = 'sqdiff_normed' (default; cv2.TM_SQDIFF_NORMED + argmin, what wav.py:185-186 does) or 'ccoeff_normed'
    (cv2.TM_CCOEFF_NORMED + argmax, the method BASELINE.json's wording names); both on either path.

    exclusion (FFT path) = 'auto' (default: the library excludes block pairs by a lower bound of their scores where a
    sub-batch is large enough for that to pay, in the form -- band-split or whole rows -- the streams' spectra allow), 'always'
    (that choice of form for a batch of any size), 'never', 'band' / 'whole' (one form forced) -- same results, different time; None reads
    SUSHI_HIP_EXCLUSION (the GPU test suite sets it to 'always' so that every edge case goes through the exclusion).
    """

    def __init__(self, dst, src, tmpl_off, tmpl_len, win_start, n_pos, variant=None, path=None,
                 delta=DEFAULT_DELTA, workspace_bytes=None, method="sqdiff_normed", exclusion=None, headroom=1.0):
        self._handle = None
        if method not in _native.METHODS:
            raise SushiError("method must be one of %s" % sorted(_native.METHODS))
        self.method = method
        if dst.device != src.device:
            raise SushiError("dst and src streams live on different devices")
        if dst.dtype != src.dtype:
            raise SushiError("pattern and stream sample types differ (cv2.matchTemplate asserts equal types)")
        self.dst, self.src = dst, src
        req = _checked_requests(dst, src, tmpl_off, tmpl_len, win_start, n_pos)
        n = req.shape[0]
        self.n = n
        self.path = default_path() if path is None else path
        if self.path not in ("fft", "direct"):
            raise SushiError("path must be 'fft' or 'direct'")
        path_code = _native.PATH_FFT if self.path == "fft" else _native.PATH_DIRECT
        self.requests = req
        if workspace_bytes is None:
            workspace_bytes = int(os.environ.get("SUSHI_HIP_FFT_WS_MB", DEFAULT_FFT_WORKSPACE >> 20)) << 20
        self.delta = float(delta)
        L = _native.lib()
        var = -1 if variant is None else int(variant)
        need = int(L.sushi_hip_batch_bytes(req.ctypes.data, n, path_code, var, int(workspace_bytes)))
        if need == 0:
            raise SushiError("batch does not fit: too many blocks in one batch, or an unknown kernel variant")
        need = int(need * max(1.0, float(headroom)) + 255) // 256 * 256        # (room for reset() to other, larger requests)
        dev = dst.device
        dst_handle = dst.searchable() if self.path == "fft" else dst.handle
        with torch.cuda.device(dev):
            self._mem = _buffer(need, dev)
            h = ctypes.c_void_p()
            rc = L.sushi_hip_batch_create(dst_handle, src.handle, req.ctypes.data, n, path_code, var, int(workspace_bytes),
                                          self._mem.data_ptr(), need, _raw_stream(dev), ctypes.byref(h))
            _native.check(rc, "sushi_hip_batch_create")
            self._handle = h
            _native.check(L.sushi_hip_batch_set_method(h, _native.METHODS[method]), "sushi_hip_batch_set_method")
            if exclusion is None:
                exclusion = os.environ.get("SUSHI_HIP_EXCLUSION", "auto")
            if exclusion not in _native.EXCLUSION:
                raise SushiError("exclusion must be one of %s" % sorted(_native.EXCLUSION))
            self.exclusion = exclusion
            _native.check(L.sushi_hip_batch_set_exclusion(h, _native.EXCLUSION[exclusion]), "sushi_hip_batch_set_exclusion")
            self.out_idx = torch.empty(n, dtype=torch.int32, device=dev)
            self.out_score = torch.empty(n, dtype=torch.float32, device=dev)
            # (index, score bits) records as well, written by the library's last kernel: results() is then ONE copy to the
            # host instead of two -- a fifth of what a drop-in call waits for its answer (tools/call_breakdown.py)
            self._packed = None
            self._host_rec = None
            self._early = self._early_np = None
            if n <= 4:
                # a drop-in call's answer is a handful of bytes: the library's last kernel writes the records straight into pinned
                # host memory (device-visible under unified addressing) -- results() is then a wait, not a copy
                self._host_rec = torch.empty((n, 2), dtype=torch.int32, pin_memory=True)
                _native.check(L.sushi_hip_batch_set_packed_output(h, self._host_rec.data_ptr()), "sushi_hip_batch_set_packed_output")
                if self.path == "fft":
                    # ... and the kernel that finishes a search from its candidate lists leaves the answer there the moment it has it
                    # (sushi_hip_batch_set_early_output: index, score bits, ready, flagged): results() polls these records and does
                    # not wait for the three launches behind that kernel, which find nothing to do for such a call
                    self._early = torch.zeros((n, 4), dtype=torch.int32, pin_memory=True)
                    self._early_np = self._early.numpy()
                    _native.check(L.sushi_hip_batch_set_early_output(h, self._early.data_ptr()), "sushi_hip_batch_set_early_output")
            else:
                self.set_packed_output(torch.empty((n, 2), dtype=torch.int32, device=dev))
        self._read_info()

    def _read_info(self):
        info = _native.BatchInfo()
        _native.check(_native.lib().sushi_hip_batch_info(self._handle, ctypes.byref(info)), "sushi_hip_batch_info")
        self.variant = int(info.variant)
        self.sub_batches = int(info.sub_batches)
        self.lanes = int(info.lanes)
        self.n_tiles = int(info.direct_tiles)
        self.fft_pairs, self.fft_segs = int(info.fft_pairs), int(info.fft_segments)
        self.ws_bytes = int(info.workspace_bytes)
        # algorithmic work of this batch: 2*P*M flop; every search and pattern sample read
        # once (4 bytes for float32 streams, 1 for uint8) and 8 bytes out per search
        self.flops = float(info.flops)
        self.algorithmic_bytes = float(info.algorithmic_bytes)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                _native.lib().sushi_hip_batch_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._handle

    def reset(self, tmpl_off, tmpl_len, win_start, n_pos):
        """The same batch for OTHER requests (equally many; same streams, path, method, exclusion): sushi_hip_batch_reset -- plan and
        descriptors redone in place, one upload, no allocation.  False if the new requests need more memory than this batch was
        created with (the batch is unchanged then): the caller makes a new one."""
        req = _checked_requests(self.dst, self.src, tmpl_off, tmpl_len, win_start, n_pos)
        if req.shape[0] != self.n:
            raise SushiError("reset: a batch keeps its number of searches")
        with torch.cuda.device(self.dst.device):
            rc = _native.lib().sushi_hip_batch_reset(self._handle, req.ctypes.data, self.n, _raw_stream(self.dst.device))
        if rc == -4:                                              # SUSHI_HIP_ENOSPACE
            return False
        _native.check(rc, "sushi_hip_batch_reset")
        self.requests = req
        self._read_info()
        return True

    def set_packed_output(self, packed):
        """Multi-GPU callers: every following run() also leaves its results as (index, score bits) int32 pairs in the first n rows
        of `packed` (a contiguous int32 CUDA tensor [>= n, 2]) -- a rank's block of the all-gather, written by the library's last
        kernel instead of by two copies afterwards (sushi_amd.distributed.ShardedSearch).  None turns it off."""
        self._host_rec = None
        if self._early is not None:
            _native.check(_native.lib().sushi_hip_batch_set_early_output(self._handle, None), "sushi_hip_batch_set_early_output")
            self._early = self._early_np = None
        if packed is None:
            _native.check(_native.lib().sushi_hip_batch_set_packed_output(self._handle, None), "sushi_hip_batch_set_packed_output")
            self._packed = None
            return
        if packed.dtype != torch.int32 or not packed.is_cuda or not packed.is_contiguous() or packed.dim() != 2 or \
                packed.shape[1] != 2 or packed.shape[0] < self.n:
            raise SushiError("packed output: a contiguous int32 CUDA tensor [>= n, 2]")
        _native.check(_native.lib().sushi_hip_batch_set_packed_output(self._handle, packed.data_ptr()), "sushi_hip_batch_set_packed_output")
        self._packed = packed           # kept alive with the batch

    def set_method(self, method):
        """Matching method of the following runs (sushi_hip_batch_set_method): 'sqdiff_normed' or 'ccoeff_normed'."""
        if method not in _native.METHODS:
            raise SushiError("method must be one of %s" % sorted(_native.METHODS))
        _native.check(_native.lib().sushi_hip_batch_set_method(self._handle, _native.METHODS[method]), "sushi_hip_batch_set_method")
        self.method = method

    def set_bound_model(self, model):
        """'worst_case' (default: the excluded side of the pair exclusion is a proof) or 'statistical' (round 5's model; A/B)."""
        if model not in _native.BOUND_MODEL:
            raise SushiError("bound model must be one of %s" % sorted(_native.BOUND_MODEL))
        _native.check(_native.lib().sushi_hip_batch_set_bound_model(self._handle, _native.BOUND_MODEL[model]),
                      "sushi_hip_batch_set_bound_model")

    def run(self, hip_stream=None):
        """One pass of the hot path over this batch (asynchronous)."""
        st = _raw_stream(self.dst.device) if hip_stream is None else hip_stream
        if self._early_np is not None:
            self._early_np[:, 2] = 0                      # not ready: results() polls
        rc = _native.lib().sushi_hip_batch_run(self._handle, self.delta, self.out_idx.data_ptr(),
                                               self.out_score.data_ptr(), st)
        _native.check(rc, "sushi_hip_batch_run")
        return self.out_idx, self.out_score

    def results(self):
        """(idx int32 ndarray, score float32 ndarray) -- synchronises."""
        if self._early_np is not None:
            # the answer as soon as the kernel that has it wrote it (a spin of a few dozen microseconds; a search that went on to
            # the tile stage, or a GPU that takes longer than any drop-in call does, falls through to the stream's end)
            e = self._early_np
            t_end = time.perf_counter() + 2e-3
            while not e[:, 2].all():
                if time.perf_counter() > t_end:
                    break
            else:
                if not e[:, 3].any():
                    return e[:, 0].copy(), e[:, 1].copy().view(np.float32)
        if self._host_rec is not None:
            torch.cuda.current_stream(self.dst.device).synchronize()
            rec = self._host_rec.numpy()
            return rec[:, 0].copy(), rec[:, 1].copy().view(np.float32)
        if self._packed is not None:
            rec = self._packed[:self.n].cpu().numpy()
            return np.ascontiguousarray(rec[:, 0]), np.ascontiguousarray(rec[:, 1]).view(np.float32)
        return self.out_idx.cpu().numpy(), self.out_score.cpu().numpy()

    def diagnostics(self, per_search=False):
        """What the last run() did (FFT path): a dict of the SushiHipBatchDiag fields, plus -- per_search=True --
        'ranking_err' (|f32 FFT score - exact score| at every result position) and 'flagged' (0 lists, 1 tiles,
        2 every position).  Synchronises."""
        d = _native.BatchDiag()
        err = np.zeros(self.n, np.float32) if per_search else None
        flg = np.zeros(self.n, np.int32) if per_search else None
        rc = _native.lib().sushi_hip_batch_diagnostics(self._handle, ctypes.byref(d),
                                                       err.ctypes.data if per_search else None,
                                                       flg.ctypes.data if per_search else None)
        _native.check(rc, "sushi_hip_batch_diagnostics")
        out = {k: getattr(d, k) for k, _ in _native.BatchDiag._fields_ if k != "reserved"}
        out["band_votes"] = [int(v) for v in d.band_votes]
        if per_search:
            out["ranking_err"], out["flagged_per_search"] = err, flg
        return out

    def pair_bounds(self):
        """FFT path, after run(): (slb float32[pairs], acc float32[pairs, 2]) of the last sub-batch -- every block pair's lower
        bound of its scores (-inf: none) and what it was made from (sushi_hip_batch_pair_bounds).  Only meaningful when that
        sub-batch went through the exclusion (exclusion='always', or 'auto' on a large batch).  Synchronises."""
        n = ctypes.c_int64(int(self.fft_pairs))
        slb = np.empty(int(self.fft_pairs), np.float32)
        acc = np.empty((int(self.fft_pairs), 2), np.float32)
        _native.check(_native.lib().sushi_hip_batch_pair_bounds(self._handle, slb.ctypes.data, acc.ctypes.data, ctypes.byref(n)),
                      "sushi_hip_batch_pair_bounds")
        return slb[:n.value], acc[:n.value]

    def workspace_view(self, which):
        """FFT path, after run(): the last sub-batch's pattern spectra / products as a float16 CUDA tensor of (re, im) pairs
        (sushi_hip_batch_workspace_view; which = _native.WS_*).  A view of this batch's own buffer: valid until the next run()."""
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t(0)
        _native.check(_native.lib().sushi_hip_batch_workspace_view(self._handle, int(which), ctypes.byref(ptr), ctypes.byref(nbytes)),
                      "sushi_hip_batch_workspace_view")
        if not ptr.value:
            return None
        off = ptr.value - self._mem.data_ptr()
        return self._mem[off:off + nbytes.value].view(torch.float16)

    def ranking_errors(self):
        """FFT path: |f32 FFT score - exact score| at every search's result position in the last run()
        (0 for searches the tile kernel finished)."""
        return self.diagnostics(per_search=True)["ranking_err"]

    def fallback_count(self):
        """FFT path: how many searches of the last run() needed the collection pass + exact tiles."""
        return int(self.diagnostics()["flagged"]) if self.path == "fft" else 0

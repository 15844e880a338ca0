"""WavStream.__init__ / DownmixedWavFile.readframes against tests/golden/wav_init.json, which was produced by
executing the REFERENCE's own bytecode (wav.py:64-91, 108-162; tests/golden/gen_wav_init_golden.py) on the seeded
WAV files of tests/wav_cases.py.  Bit-for-bit: sha256 of the stream's bytes.

CPU: the NumPy pipeline (`SUSHI_HIP_LOAD=host`) and the oracle's restatement.  GPU: the device pipeline
(decode + downmix + decimation + padding + medians + normalisation in libsushi_hip.so)."""
import hashlib
import json
import os

import numpy as np
import pytest

import wav_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

with open(os.path.join(ROOT, "tests", "golden", "wav_init.json")) as _f:
    GOLDEN = json.load(_f)["cases"]
IDS = [g["case"]["name"] for g in GOLDEN]


def _write(tmp_path, g):
    blob, _ = wav_cases.wav_bytes(g["case"])
    assert hashlib.sha256(blob).hexdigest() == g["wav_sha256"], "seeded WAV bytes differ from the generator's"
    path = os.path.join(str(tmp_path), g["case"]["name"] + ".wav")
    with open(path, "wb") as f:
        f.write(blob)
    return path


def _check(stream_data, sample_count, padding_size, g):
    assert float(sample_count) == g["sample_count"]
    assert int(padding_size) == g["padding_size"]
    assert list(stream_data.shape) == g["shape"] and str(stream_data.dtype) == g["dtype"]
    got = hashlib.sha256(np.ascontiguousarray(stream_data).tobytes()).hexdigest()
    if got != g["data_sha256"]:
        probe = [float(stream_data[0, p]) for p in g["probe_index"]]
        bad = [(p, a, b) for p, a, b in zip(g["probe_index"], probe, g["probe_value"]) if a != b]
        other = " (it IS the NumPy-2-promotion variant)" if got == g["data_sha256_nep50"] else ""
        raise AssertionError("stream differs from the reference's%s; probes that differ: %r; sum %r vs %r"
                             % (other, bad[:6], float(stream_data.astype(np.float64).sum()), g["sum"]))


@pytest.mark.parametrize("g", GOLDEN, ids=IDS)
def test_host_pipeline_equals_reference_bytecode(g, tmp_path, monkeypatch):
    monkeypatch.setenv("SUSHI_HIP_LOAD", "host")
    from sushi_amd.wav import WavStream
    c = g["case"]
    s = WavStream(_write(tmp_path, g), sample_rate=c["sample_rate"], sample_type=c["sample_type"])
    _check(s.data, s.sample_count, s.padding_size, g)
    assert s.duration_seconds == g["duration_seconds"] and s.sample_rate == g["sample_rate"]


@pytest.mark.parametrize("g", GOLDEN, ids=IDS)
def test_oracle_load_equals_reference_bytecode(g, tmp_path, oracle):
    c = g["case"]
    s = oracle.load_wav_stream(_write(tmp_path, g), sample_rate=c["sample_rate"], sample_type=c["sample_type"])
    _check(s.data, s.sample_count, s.padding_size, g)


@pytest.mark.gpu
@pytest.mark.parametrize("g", GOLDEN, ids=IDS)
def test_device_pipeline_equals_reference_bytecode(g, tmp_path, monkeypatch):
    monkeypatch.setenv("SUSHI_HIP_LOAD", "auto")
    from sushi_amd.wav import WavStream
    c = g["case"]
    s = WavStream(_write(tmp_path, g), sample_rate=c["sample_rate"], sample_type=c["sample_type"])
    assert s._dev_row is not None or s._dev is not None, "the device pipeline did not run"
    _check(s.data, s.sample_count, s.padding_size, g)
    # and the HBM-resident row the matching reads is that same stream
    row = s.device_stream().raw.cpu().numpy()
    assert hashlib.sha256(row.tobytes()).hexdigest() == g["data_sha256"]

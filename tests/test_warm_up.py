"""sushi_amd.device.warm_up: the process's one-time GPU start-up (context, code object), paid when the caller chooses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_warm_up_is_idempotent_and_a_stream_waits_for_a_background_one():
    from sushi_amd import device
    first = device.warm_up(background=True)          # None while it runs; a number if an earlier test's call has finished
    assert first is None or first > 0
    s = device.DeviceStream((np.arange(50000, dtype=np.float32) % 977) / 977.0)      # waits for the warm-up, then works as ever
    assert s.n == 50000 and s.searchable()
    ms = device.warm_up()
    assert ms is not None and ms > 0
    assert device.warm_up() == ms and device.warm_up(background=True) == ms

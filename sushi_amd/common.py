"""Names the hot path shares with the reference's ``common.py``."""
import math


class SushiError(Exception):
    """The reference's only domain error (common.py:4); sushi.py:841-843 turns it into exit code 2."""
    pass


def clip(value, minimum, maximum):
    """common.py:41-42"""
    return max(min(value, maximum), minimum)


def py2_round(x):
    """Python 2's round(): half away from zero (the reference is Python 2: common.py:32, wav.py:127)."""
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def format_time(seconds):
    """common.py:31-38 (h:mm:ss.cc, as in the reference's log lines)"""
    cs = py2_round(seconds * 100)
    return u'{0}:{1:02d}:{2:02d}.{3:02d}'.format(int(cs // 360000), int((cs // 6000) % 60),
                                                 int((cs // 100) % 60), int(cs % 100))

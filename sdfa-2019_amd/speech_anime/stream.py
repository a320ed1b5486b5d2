"""saber.stream.seek (saber/data/stream/stream.py:20-46): timestamp lookup with linear interpolation, used by
evaluate() to resample the irregular-millisecond animation track to the video frame rate."""
import numpy as np


def seek(ts, timestamps, sequence):
    assert len(timestamps) == len(sequence)
    left, right = 0, len(timestamps)
    m = (left + right) // 2
    while left < right:
        m = (left + right) // 2
        tm = timestamps[m]
        tn = timestamps[m + 1] if m + 1 < len(timestamps) else ts + 1
        if tm <= ts < tn:
            break
        elif tm > ts:
            right = m
        else:
            left = m + 1
    if ts < timestamps[m] or ts > timestamps[-1]:
        return np.copy(sequence[m])
    if m + 1 >= len(timestamps):
        return np.copy(sequence[m])
    n = m + 1
    a = (timestamps[n] - ts) / (timestamps[n] - timestamps[m])
    return a * sequence[m] + (1 - a) * sequence[n]

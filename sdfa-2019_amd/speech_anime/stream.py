"""Timestamp lookup with linear interpolation -- the behaviour of saber.stream.seek (saber/data/stream/stream.py:20-46),
used by evaluate() to resample the irregular-millisecond animation track to the video frame rate.

Semantics kept (pinned by tests/golden/host_rows.npz): timestamps ascending; a query before the first or after the
last timestamp returns a copy of the first / last row; otherwise rows m and m+1 with ts[m] <= q < ts[m+1] are blended
with weight (ts[m+1] - q) / (ts[m+1] - ts[m]) on row m.
"""
import bisect

import numpy as np


def seek(ts, timestamps, sequence):
    n = len(timestamps)
    assert n == len(sequence)
    if ts < timestamps[0]:
        return np.copy(sequence[0])
    if ts > timestamps[-1]:
        return np.copy(sequence[-1])
    m = bisect.bisect_right(timestamps, ts) - 1          # last index with timestamps[m] <= ts
    if m + 1 >= n:
        return np.copy(sequence[m])
    w = (timestamps[m + 1] - ts) / (timestamps[m + 1] - timestamps[m])
    return w * sequence[m] + (1 - w) * sequence[m + 1]

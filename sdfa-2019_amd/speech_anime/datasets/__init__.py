"""Inference-side dataset helpers: only the sliding-window feature fetcher of the reference is on the hot path."""
from . import sliding_window as _sw

DatasetSlidingWindow = _sw.DatasetSlidingWindow
__all__ = ["DatasetSlidingWindow"]

from .sliding_window import DatasetSlidingWindow  # noqa: F401

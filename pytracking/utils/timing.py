"""Wall-clock and device-event stopwatches (the reference's utils/timing.py:7-50 surface)."""
from timeit import default_timer as timer

import torch


class time_measurer:
    def __init__(self, units="ms", desc=None):
        self.start_time, self.units, self.desc = timer(), units, desc

    def __call__(self):
        return self.elapsed()

    def elapsed(self):
        value = float(timer() - self.start_time)
        return round(1000 * value, 1) if self.units == "ms" else value


class cuda_time_measurer:
    """HIP-event stopwatch on the current stream."""

    def __init__(self, units="ms"):
        assert units == "ms"
        self.start_event = torch.cuda.Event(enable_timing=True)
        self.end_event = torch.cuda.Event(enable_timing=True)
        self.start_event.record()

    def __call__(self):
        self.end_event.record()
        torch.cuda.synchronize()
        return self.start_event.elapsed_time(self.end_event)

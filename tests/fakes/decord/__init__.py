"""Stand-in for the `decord` package (not installable here: no network) serving tests/fakes/fake_clip.py's synthetic clips through the part
of decord's API the reference uses (vid_utils.py:10-21): VideoReader(uri, ctx=cpu(0), num_threads=n), len(), get_avg_fps(), get_batch(idx).asnumpy()."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fake_clip  # noqa: E402


class _Ctx:
    def __init__(self, dev_id=0):
        self.device_id = dev_id


def cpu(dev_id=0):
    return _Ctx(dev_id)


class _NDArray:
    def __init__(self, a):
        self._a = a

    def asnumpy(self):
        return self._a


class VideoReader:
    def __init__(self, uri, ctx=None, width=-1, height=-1, num_threads=0, fault_tol=-1):
        self._meta = fake_clip.read_clip(uri)
        self.requests = []

    def __len__(self):
        return self._meta["frames"]

    def get_avg_fps(self):
        return float(self._meta["fps"])

    def get_batch(self, indices):
        idx = [int(i) for i in indices]
        self.requests.append(idx)
        if not idx:
            return _NDArray(np.zeros((0, self._meta["height"], self._meta["width"], 3), dtype=np.uint8))
        return _NDArray(np.stack([fake_clip.frame(self._meta, i) for i in idx]))

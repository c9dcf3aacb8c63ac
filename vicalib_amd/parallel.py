"""Frame sharding across processes (one process per GPU) -- SURVEY.md 8(e).

The reference is single-process; the MI355X build shards the frames of one calibration problem over
ranks.  Every rank runs the same LM loop on its own frames; per iteration the library calls back twice:
once with the packed reduced system [S | g_red | diag(H_ss) | g_s | cost] (sum) and once with the step
scalars (sum / max).  This module supplies that callback on top of torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests) and the contiguous frame partition.
"""
from __future__ import annotations

import ctypes
import numpy as np


def frame_shard(n_frames: int, rank: int, world: int):
    """Contiguous range [lo, hi) of rank `rank` (rank r owns frames [r*N/P, (r+1)*N/P))."""
    lo = (n_frames * rank) // world
    hi = (n_frames * (rank + 1)) // world
    return lo, hi


class _DevArray:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 3}


class FrameShardComm:
    """Builds the vc_allreduce_fn callback.  device = "cuda:<i>" wraps the library's device buffers
    zero-copy and reduces on the calibrator's own HIP stream; device = "cpu" wraps host memory."""

    def __init__(self, group=None, device="cpu", stream_ptr=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device(device)
        self.stream = None
        if self.device.type == "cuda" and stream_ptr:
            self.stream = torch.cuda.ExternalStream(int(stream_ptr), device=self.device)
        self._views = {}
        self.calls = 0

    def _view(self, ptr, count):
        key = (int(ptr), int(count))
        t = self._views.get(key)
        if t is None:
            if self.device.type == "cuda":
                t = self.torch.as_tensor(_DevArray(ptr, count), device=self.device)
            else:
                arr = np.ctypeslib.as_array((ctypes.c_double * count).from_address(int(ptr)))
                t = self.torch.from_numpy(arr)
            self._views[key] = t
        return t

    def __call__(self, ctx, ptr, count, op):
        try:
            t = self._view(ptr, count)
            rop = self.dist.ReduceOp.SUM if op == 0 else self.dist.ReduceOp.MAX
            if self.stream is not None:
                with self.torch.cuda.stream(self.stream):
                    self.dist.all_reduce(t, op=rop, group=self.group)
            else:
                self.dist.all_reduce(t, op=rop, group=self.group)
            self.calls += 1
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("FrameShardComm: all_reduce failed:", e, flush=True)
            return -1

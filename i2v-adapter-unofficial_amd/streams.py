"""Fork / join of independent kernel chains onto side HIP streams.

At the 16 x 16 and 8 x 8 levels of the UNet a single GEMM / attention / convolution launch has fewer workgroups than the
chip has CUs (a 2048 x 1280 output is 32 tiles of 256 x 320 on 256 CUs), and the reference's op graph holds chains that
do not depend on each other: attn1 and the cross-frame adapter attention of a spatial block (i2v:468-473 vs 483-492), the
q|k and V^T projections, the K0 / V0^T projections of the frame-0 tokens, a resnet's 1x1 shortcut and its first
convolution.  `fork()` runs such a chain on a side stream: the side stream first waits for everything already queued on
the main stream, the chain's kernels are launched on it (the ctypes wrappers launch on torch's CURRENT stream, and
torch's allocator tags the chain's tensors with that stream), and leaving the block makes the main stream wait for it.
Inside `torch.cuda.graph` capture the same calls become fork / join dependencies of the captured hipGraph, so a replay
runs the branches concurrently with no host involvement; eager launches and replays stay bit-identical (the kernels and
their launch arguments do not change, only what may run beside them).

Lifetime rule that makes the allocator safe without record_stream: every use of a side stream begins by waiting for the
main stream, and every tensor a side chain reads or writes is referenced until after the join.
"""
import contextlib
import os

import torch

# MEASURED (round 3, same box, tools/stream_probe.py and bench.py): a fork / join pair costs ~7 us in a
# replayed hipGraph on ROCm 7.2.  Two 2048 x 1280 x 1280 GEMMs (32 workgroups each) take 46.8 us back to back and 39.9 us
# forked; two 8192-row ones 60.5 -> 67.1 us (each fills the chip: nothing to overlap, only the join to pay); eager launches
# lose everywhere (host-side event traffic).  Whole step: forks at the 16 x 16 and 8 x 8 levels 57.46 -> 57.48 ms, at the
# 8 x 8 level only 58.24 -> 58.48 ms: the chains that can overlap are too short to pay for their joins.  OFF by default;
# I2V_STREAMS=1 turns the forks on (the tests run both ways: replay == eager bit for bit either way).
ENABLED = os.environ.get("I2V_STREAMS", "0") == "1"
# fork only where one launch cannot fill the chip: token rows of the level (B*F*H*W) at or below this
MAX_ROWS = int(os.environ.get("I2V_STREAMS_MAX_ROWS", "2048"))

_side = {}
_force_off = [False]


def _side_stream(device, index):
    key = (device.index if device.index is not None else torch.cuda.current_device(), index)
    s = _side.get(key)
    if s is None:
        s = _side[key] = torch.cuda.Stream(device=device)
    return s


@contextlib.contextmanager
def disabled():
    """single-stream launches inside this block (per-kernel event timing needs back-to-back kernels on one stream)."""
    prev = _force_off[0]
    _force_off[0] = True
    try:
        yield
    finally:
        _force_off[0] = prev


class fork:
    """with fork(on, device) as fk:
           with fk.side():      # chain A: side stream (inline when `on` is false)
               ...
           ...                  # chain B: main stream
       # joined here"""

    def __init__(self, on: bool, device, n_sides: int = 1):
        self.on = bool(on) and ENABLED and not _force_off[0]
        self.device = device
        self.n = n_sides
        self.sides = []

    def __enter__(self):
        if self.on:
            self.main = torch.cuda.current_stream(self.device)
            self.sides = [_side_stream(self.device, i) for i in range(self.n)]
            for s in self.sides:
                s.wait_stream(self.main)
        return self

    def side(self, index: int = 0):
        if not self.on:
            return contextlib.nullcontext()
        return torch.cuda.stream(self.sides[index])

    def __exit__(self, *exc):
        if self.on:
            for s in self.sides:
                self.main.wait_stream(s)
        return False

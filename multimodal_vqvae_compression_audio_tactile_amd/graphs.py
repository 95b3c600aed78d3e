"""hipGraph capture of the launch sequence (the MI355X answer to per-call launch overhead: no tracing compiler).

The C ABI never allocates, frees or synchronises, outputs come from torch's caching allocator (which owns a private
pool during capture) and every kernel goes to the current stream, so a whole `encode_latents` / `T_DEC` call -- a few
hundred launches at B = 1 -- is capturable as ONE graph and replayed with one host call.
"""
from __future__ import annotations

import torch


class GraphedCall:
    """Capture `fn(*static_inputs)` once; `__call__(*inputs)` copies into the static inputs and replays."""

    def __init__(self, fn, *example_inputs, warmup: int = 2):
        self.static_in = [x.clone() for x in example_inputs]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                       # warm-up on a side stream: builds packed weights, caches
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.static_out

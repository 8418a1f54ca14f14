"""Host <-> HBM staging for tile streams: the step either side of the hot path.

Real slides live on the host; PCIe (Gen5 x16, ~63 GB/s spec per direction), not HBM, bounds them.
``TilePipeline`` keeps the GPU fed: batches of uint8 tiles are copied host->device, normalised and
copied device->host on three HIP streams with double-buffered pinned staging, so the two copies of
batch k+1 / k-1 overlap the kernels of batch k.  Reference counterpart: the caller's Python loop
over tiles (notebook cell 11); nothing inside stainlib stages data.
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator

import numpy as np
import torch


class TilePipeline:
    """Double-buffered H2D -> fn -> D2H.

    fn(tiles_cuda_uint8_NHWC, out_cuda) must enqueue its work on the CURRENT torch stream and write ``out``."""

    def __init__(self, fn: Callable[[torch.Tensor, torch.Tensor], None], batch_shape, depth: int = 2, device="cuda"):
        self.fn = fn
        self.shape = tuple(batch_shape)                      # (B, H, W, 3)
        self.depth = depth
        self.dev = torch.device(device)
        self.h_in = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.h_out = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.d_in = [torch.empty(self.shape, dtype=torch.uint8, device=self.dev) for _ in range(depth)]
        self.d_out = [torch.empty(self.shape, dtype=torch.uint8, device=self.dev) for _ in range(depth)]
        self.s_h2d, self.s_run, self.s_d2h = (torch.cuda.Stream(self.dev) for _ in range(3))
        self.e_in = [torch.cuda.Event() for _ in range(depth)]    # H2D of slot done
        self.e_run = [torch.cuda.Event() for _ in range(depth)]   # kernels of slot done
        self.e_out = [torch.cuda.Event() for _ in range(depth)]   # D2H of slot done
        self.copy_threads = 16                                # host threads staging pageable input into pinned memory
        self._pool = None

    def _host_copy(self, dst: torch.Tensor, b) -> None:
        """pageable batch -> pinned slot.  A single-threaded memcpy moves ~8 GB/s, a sixth of what the link takes:
        the batch is copied in slices by a few threads (numpy releases the GIL for large copies)."""
        src = np.ascontiguousarray(b) if isinstance(b, np.ndarray) else b.contiguous().numpy()
        out = dst.numpy()
        n = src.shape[0]
        k = min(self.copy_threads, n)
        if k <= 1 or src.nbytes < (1 << 24):
            np.copyto(out, src)
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.copy_threads)
        bounds = [n * i // k for i in range(k + 1)]
        list(self._pool.map(lambda i: np.copyto(out[bounds[i]:bounds[i + 1]], src[bounds[i]:bounds[i + 1]]), range(k)))

    def run(self, batches: Iterable[np.ndarray]) -> Iterator[np.ndarray]:
        """Yields one normalised uint8 array per input batch, in order.  A short last batch is allowed; a batch
        may be a numpy array (staged through pinned memory) or an already pinned torch uint8 tensor (zero-copy).
        The yielded array is a view of a pinned buffer that is reused `depth` batches later: copy it if kept."""
        pending = []                                          # (slot, n) in flight
        for k, b in enumerate(batches):
            slot = k % self.depth
            if len(pending) == self.depth:                    # the slot we are about to reuse must be drained
                s0, n0 = pending.pop(0)
                self.e_out[s0].synchronize()
                yield self.h_out[s0][:n0].numpy()
            n = b.shape[0]
            if isinstance(b, torch.Tensor) and b.is_pinned():
                src = b                                       # producer already wrote pinned memory: no staging copy
            else:                                             # pageable input: one host copy into the pinned slot
                self._host_copy(self.h_in[slot][:n], b)
                src = self.h_in[slot][:n]
            with torch.cuda.stream(self.s_h2d):
                self.s_h2d.wait_event(self.e_run[slot])       # previous kernels reading d_in[slot] are done
                self.d_in[slot][:n].copy_(src, non_blocking=True)
                self.e_in[slot].record()
            with torch.cuda.stream(self.s_run):
                self.s_run.wait_event(self.e_in[slot])
                self.s_run.wait_event(self.e_out[slot])       # previous D2H of d_out[slot] is done
                self.fn(self.d_in[slot][:n], self.d_out[slot][:n])
                self.e_run[slot].record()
            with torch.cuda.stream(self.s_d2h):
                self.s_d2h.wait_event(self.e_run[slot])
                self.h_out[slot][:n].copy_(self.d_out[slot][:n], non_blocking=True)
                self.e_out[slot].record()
            pending.append((slot, n))
        for s0, n0 in pending:
            self.e_out[s0].synchronize()
            yield self.h_out[s0][:n0].numpy()


def normalizer_pipeline(normalizer, batch_shape, depth: int = 2) -> TilePipeline:
    """Pipeline around a fitted stainlib_amd ExtractiveStainNormalizer (per-tile transform)."""
    from . import engine
    ws = engine.Workspace()

    transform = engine.macenko_transform if normalizer.method == "macenko" else engine.vahadane_transform
    # the 8 target doubles go to the device ONCE: handed over as numpy they would be a small pageable copy per batch,
    # which queues behind the next batch's 400 MB upload on the copy engine and stalls the kernels for its whole
    # duration (measured: 14.9 instead of 8.4 ms per 128-tile batch)
    M_t = torch.as_tensor(np.asarray(normalizer.stain_matrix_target, dtype=np.float64), device="cuda").reshape(2, 3).contiguous()
    c_t = torch.as_tensor(np.asarray(normalizer.maxC_target, dtype=np.float64), device="cuda").reshape(2).contiguous()

    def fn(tiles, out):
        transform(tiles, M_t, c_t, out=out, ws=ws)
    return TilePipeline(fn, batch_shape, depth)

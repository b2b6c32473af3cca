"""Input side of the train step: what `DataGenerator.__getitem__` (datageneratorv2.py:64-102) hands to the model, brought
onto the device without stalling the step.

The reference generator returns per batch `(spectrogram_in f32 [B,H,W,2], embedding i32 [B,2,16], spectrogram_out f32
[B,H,W,2])` built in Python on the host.  `DeviceBatchPipeline` wraps any iterable yielding such triples (NumPy arrays or
CPU tensors, NHWC as the reference or already NCHW): a background thread stages the next batches in pinned host memory and
copies them on a dedicated HIP stream while the current step runs; the consumer only waits on an event.  The engine's
boundary is contiguous NCHW fp32 `[B,2,H,W]` (+ the index tensor), so NHWC batches are permuted on the copy stream.
`synthetic_batches` is the device-side generator SURVEY.md §8(d) specifies for benchmarks.
"""
import queue
import threading

import numpy as np
import torch

from . import ops


class DeviceBatchPipeline:
    """for spec_in, emb, spec_out in DeviceBatchPipeline(generator, device): trainer.step(spec_in, emb, spec_out)

    depth = batches staged ahead.  The copy stream carries host -> device copies only (the NHWC -> NCHW permutation runs on the
    consumer's stream): a kernel on it can land on a hardware queue behind the step's compute kernels and hold every copy behind
    it up (round 1: 15.7-20 ms host-fed against 13.8 device-resident, 29.9 with two batches ahead).  Staging a batch is one
    foreign call without the interpreter lock.  Measured at cfg 2 (scripts/pcie_inclusive.py): 13.4 ms host-fed against 13.0 ms
    device-resident.  Do not pass a stream that also runs kernels
    (the optimizer stream: 30.7 ms)."""

    def __init__(self, source, device, depth=6, nhwc=True, stream=None):
        self.source, self.device, self.depth, self.nhwc = source, torch.device(device), max(1, depth), nhwc
        self.stream = stream        # copy stream (default: a stream of its own); it must not run kernels

    def _pinned_like(self, a, slot, k):
        shape = tuple(a.shape)
        dtype = torch.from_numpy(a[:0] if a.ndim else a).dtype if isinstance(a, np.ndarray) else a.dtype
        buf = slot.get(k)
        if buf is None or tuple(buf.shape) != shape or buf.dtype != dtype:
            buf = torch.empty(shape, dtype=dtype).pin_memory()
            slot[k] = buf
        return buf

    def __iter__(self):
        stream = self.stream if self.stream is not None else torch.cuda.Stream(device=self.device)
        q = queue.Queue(maxsize=self.depth)
        slots = [dict() for _ in range(self.depth + 2)]        # pinned staging buffers, reused round-robin
        slot_ev = [None] * len(slots)                          # event after the last host->device copies out of a slot
        stop = object()

        def producer():
            try:
                for i, (spec_in, emb, spec_out) in enumerate(self.source):
                    s = i % len(slots)
                    if slot_ev[s] is not None:                 # the copies that last read this slot have finished (waited for
                        slot_ev[s].synchronize()               # HERE, in the producer: the training thread never blocks on a copy)
                    # the copy stream carries NOTHING but host -> device copies (DMA engines): a kernel on it can land on a hardware
                    # queue behind the step's compute kernels and hold the copies behind it up.  Staging + copies of the three arrays
                    # are ONE foreign call (ops.stage_h2d: memcpy into the pinned slot, hipMemcpyAsync), which runs without the
                    # interpreter lock - the training thread needs that lock ~300 times per step.
                    arrs = [np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a.contiguous() for a in (spec_in, emb, spec_out)]
                    pinned = [self._pinned_like(a, slots[s], k) for a, k in zip(arrs, ("in", "emb", "out"))]
                    with torch.cuda.stream(stream):
                        dev = [torch.empty(p.shape, dtype=p.dtype, device=self.device) for p in pinned]
                        ops.stage_h2d([a.ctypes.data if isinstance(a, np.ndarray) else a.data_ptr() for a in arrs], pinned, dev, stream)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        slot_ev[s] = ev
                    q.put((dev, ev))
                q.put(stop)
            except BaseException as e:                         # surface generator errors in the consumer
                q.put(e)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is stop:
                break
            if isinstance(item, BaseException):
                raise item
            dev, ev = item
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                                 # a wait in the stream; the host keeps running ahead
            for t in dev:
                t.record_stream(cur)
            if self.nhwc:                                      # [B,H,W,2] -> contiguous [B,2,H,W], on the consumer's stream (two 17 MB
                dev[0] = dev[0].permute(0, 3, 1, 2).contiguous()     # passes, ~10 us each at cfg 2)
                dev[2] = dev[2].permute(0, 3, 1, 2).contiguous()
            yield dev[0].float(), dev[1], dev[2].float()
        th.join()


def synthetic_batches(n, B, H, W, device, seed=1234, rank=0):
    """amplitude and phase ~ U[0,1) with the zero padding of the 129/144 x 151/160 STFT core (rows >=
    ceil(0.896 H), columns >= ceil(0.944 W) exactly 0), indices ~ randint[26, 1282); generated on the device, NCHW."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + rank)
    r0, c0 = int(np.ceil(0.896 * H)), int(np.ceil(0.944 * W))
    for _ in range(n):
        spec_in = torch.rand((B, 2, H, W), device=device, generator=g)
        spec_out = torch.rand((B, 2, H, W), device=device, generator=g)
        for s in (spec_in, spec_out):
            s[:, :, r0:, :] = 0.0
            s[:, :, :, c0:] = 0.0
        emb = torch.randint(26, 1282, (B, 2, 16), device=device, generator=g, dtype=torch.int64)
        yield spec_in, emb, spec_out

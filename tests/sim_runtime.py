"""Test infrastructure: a simulated stream runtime for the product's scheduling code.

The engines and the trainer issue every stream / event / collective call through a runtime object (unet_rir_amd/device.py).
``SimRuntime`` is the CPU stand-in used by the tests: launches execute eagerly in enqueue order on CPU tensors (the kernels
behind ``ops`` are replaced by tests/cpu_ops.py), while every stream carries a vector clock, every event is a snapshot of
one, every wait merges one - and every tensor access the ops report is checked against the earlier accesses it conflicts
with: a read after a write (or a write after a read / write) issued from another stream with NO happens-before edge between
them is a data race on the real device, and raises ``RaceError`` here.  So the tests cover both what a schedule computes
(enqueue order) and whether its cross-stream dependencies are all expressed (events), including the gradient-bucket
hand-over to the collective and the bucket-wise optimizer on its own stream.
"""
import contextlib
import time

import torch
import torch.distributed as dist


class RaceError(AssertionError):
    pass


class SimStream:
    def __init__(self, name):
        self.name = name
        self.clock = {name: 0}

    def tick(self):
        self.clock[self.name] += 1
        return self.clock[self.name]

    def merge(self, clock):
        for k, v in clock.items():
            if self.clock.get(k, 0) < v:
                self.clock[k] = v

    def __repr__(self):
        return f"<stream {self.name}>"


class SimEvent:
    def __init__(self, clock):
        self.clock = dict(clock)
        self.t = time.perf_counter()       # launches execute eagerly: host time of the record stands in for the device time stamp


class SimWork:
    def __init__(self, clock):
        self.clock = dict(clock)

    def wait(self):          # GradBucketer without a runtime would call this; the runtime path uses wait_work
        raise AssertionError("collective handles must be waited for through the runtime")


class _Region:
    """Bytes an op touches: `rows` rows of `width` bytes, `ld` bytes apart, starting at absolute address `start`."""
    __slots__ = ("base", "ld", "lo", "hi", "a_lo", "a_hi")

    def __init__(self, obj):
        if hasattr(obj, "base") and hasattr(obj, "ld"):              # ops.Act: a channel slice of an NHWC buffer
            es = obj.base.element_size()
            self.base, self.ld = obj.base.data_ptr(), obj.ld * es
            self.lo = obj.ptr - self.base
            self.hi = self.lo + obj.C * es
            self.a_lo = obj.ptr
            self.a_hi = obj.ptr + (obj.P - 1) * self.ld + obj.C * es
        elif hasattr(obj, "buf"):                                     # ops.Workspace
            t = obj.buf
            self.base, self.ld, self.lo, self.hi = t.data_ptr(), 0, 0, t.numel()
            self.a_lo, self.a_hi = t.data_ptr(), t.data_ptr() + t.numel()
        else:
            t = obj
            if not t.is_contiguous():
                raise AssertionError("ops take contiguous tensors or Act views")
            n = t.numel() * t.element_size()
            self.base, self.ld, self.lo, self.hi = t.data_ptr(), 0, 0, n
            self.a_lo, self.a_hi = t.data_ptr(), t.data_ptr() + n

    def overlaps(self, o):
        if self.a_hi <= o.a_lo or o.a_hi <= self.a_lo:
            return False
        if self.base == o.base and self.ld == o.ld and self.ld > 0:   # two channel slices of one buffer
            return self.lo < o.hi and o.lo < self.hi
        return True

    def covers(self, o):
        if self.base == o.base and self.ld == o.ld:
            return self.lo <= o.lo and o.hi <= self.hi and self.a_hi >= o.a_hi
        return self.ld == 0 and o.ld == 0 and self.a_lo <= o.a_lo and o.a_hi <= self.a_hi


class SimRuntime:
    def __init__(self, name="r"):
        self.device = torch.device("cpu")
        self.main = SimStream("main")
        self._stack = [self.main]
        self._n = 0
        self.comm = SimStream("comm")          # the collective library's own stream
        self.accesses = []                     # (region, is_write, stream name, tick, what)
        self.n_checked = 0
        self.n_cross_stream = 0                # ordered conflicts between different streams (the edges that mattered)
        self.collectives = []                  # (lo address, bytes) per all-reduce, in issue order

    # ---- the runtime interface of unet_rir_amd/device.py
    def current_stream(self):
        return self._stack[-1]

    def record(self, stream=None, timing=False):
        s = stream if stream is not None else self.current_stream()
        return SimEvent(s.clock)

    def elapsed_ms(self, a, b):
        return (b.t - a.t) * 1e3

    def wait(self, stream, ev):
        stream.merge(ev.clock)

    @contextlib.contextmanager
    def on(self, stream):
        self._stack.append(stream)
        try:
            yield
        finally:
            self._stack.pop()

    def synchronize(self):
        for s in [self.main, self.comm] + getattr(self, "_streams", []):
            for o in [self.main, self.comm] + getattr(self, "_streams", []):
                s.merge(o.clock)

    def concurrent_streams(self, n):
        out = []
        for _ in range(n):
            self._n += 1
            out.append(SimStream(f"s{self._n}"))
        self._streams = getattr(self, "_streams", []) + out
        return out

    def all_reduce_sum(self, tensor, group=None):
        # RCCL semantics: the collective runs on the library's stream after the work queued on the CURRENT stream so far
        self.comm.merge(self.current_stream().clock)
        self.touch(reads=[tensor], writes=[tensor], what="all_reduce", stream=self.comm)
        self.collectives.append((tensor.data_ptr(), tensor.numel() * tensor.element_size()))
        if dist.is_initialized():
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
        return SimWork(self.comm.clock)

    def wait_work(self, work):
        self.current_stream().merge(work.clock)

    def broadcast(self, tensor, src=0, group=None):
        self.touch(reads=[tensor], writes=[tensor], what="broadcast")
        if dist.is_initialized():
            dist.broadcast(tensor, src=src, group=group)

    # ---- the race check
    def touch(self, reads=(), writes=(), what="", stream=None):
        s = stream if stream is not None else self.current_stream()
        tick = s.tick()
        new = [(_Region(o), False) for o in reads if o is not None] + [(_Region(o), True) for o in writes if o is not None]
        for reg, is_w in new:
            keep = []
            for (r0, w0, s0, t0, what0) in self.accesses:
                if (is_w or w0) and reg.overlaps(r0):
                    self.n_checked += 1
                    if s0 != s.name:
                        if s.clock.get(s0, 0) < t0:
                            raise RaceError(f"{what} on {s.name} {'writes' if is_w else 'reads'} memory that {what0} on {s0} "
                                            f"{'wrote' if w0 else 'read'} with no happens-before edge between them")
                        self.n_cross_stream += 1
                    if is_w and reg.covers(r0):
                        continue                 # ordered before a covering write: can never conflict again unobserved
                elif not is_w and not w0 and s0 == s.name and reg.covers(r0):
                    continue                     # an older read of the same bytes from the same stream is implied by this one
                keep.append((r0, w0, s0, t0, what0))
            self.accesses = keep
        for reg, is_w in new:
            self.accesses.append((reg, is_w, s.name, tick, what))

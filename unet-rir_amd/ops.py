"""Host-side operator wrappers: torch tensors (device memory, streams) -> the C ABI.

Each function mirrors one Keras call site of the reference graph and forwards to the HIP
library.  PyTorch is used for device memory and streams only; no arithmetic happens here.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvGeom, check


class Act:
    """An NHWC activation view (fp32 or bf16 storage): channels [c0, c0+C) of a [B,H,W,ld] device buffer.
    A channel-concat (dl_models/u_net.py:308) is two Acts over one buffer."""
    __slots__ = ("base", "B", "H", "W", "C", "ld", "c0", "ptr", "sfx")

    def __init__(self, base: torch.Tensor, c0: int = 0, C_: int = None):
        if base.dim() != 4 or not base.is_contiguous() or base.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("Act needs a contiguous fp32 or bf16 [B,H,W,ld] tensor")
        self.base = base
        self.B, self.H, self.W, self.ld = base.shape
        self.c0 = c0
        self.C = self.ld - c0 if C_ is None else C_
        self.sfx = "f32" if base.dtype == torch.float32 else "bf16"
        al = 16 // base.element_size()                    # rows and slices start on 16-byte boundaries
        if self.c0 % al or self.ld % al or self.c0 + self.C > self.ld:
            raise ValueError(f"channel offset and pixel stride must be multiples of {al}")
        self.ptr = base.data_ptr() + base.element_size() * c0

    @property
    def P(self):
        return self.B * self.H * self.W

    def slice(self, c0, C_):
        return Act(self.base, self.c0 + c0, C_)

    def dense(self):
        """A contiguous [B,H,W,C] copy (tests / debugging)."""
        return self.base[..., self.c0:self.c0 + self.C].contiguous()


def new_act(B, H, W, C_, device, ld=None, dtype=torch.float32):
    return Act(torch.empty((B, H, W, C_ if ld is None else ld), dtype=dtype, device=device), 0, C_)


def _fn(name, sfx):
    """C entry point `unetrir_<name>_<f32|bf16>`."""
    return getattr(_lib.lib(), f"unetrir_{name}_{sfx}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if isinstance(t, Act):
        return C.c_void_p(t.ptr)
    return C.c_void_p(t.data_ptr())


def geom(B, H, W, Cin, Cout, k, stride):
    return ConvGeom(B, H, W, Cin, Cout, k, stride)


class Workspace:
    """A reusable scratch buffer (the C ABI never allocates)."""

    def __init__(self, device, nbytes=0):
        self.device = device
        self.buf = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)

    def reserve(self, nbytes):
        if nbytes > self.buf.numel():
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr())

    @property
    def nbytes(self):
        return self.buf.numel()


# ---- Conv2D / Conv2DTranspose --------------------------------------------------------------

def conv3x3s2_packed_elems(N, C_):
    """Elements of the packed copy the stride-2 3x3 forward kernel reads (0: none defined for these channel counts)."""
    return int(_lib.lib().unetrir_conv3x3s2_packed_elems(int(N), int(C_)))


def conv2d_fwd(g, x: Act, w, bias, y: Act, addend: Act = None, w_packed=None):
    """Conv2D(padding='same') forward (dl_models/u_net.py:269-276, :366, :248, :262).  w_packed: the packed copy of a 3x3
    stride-2 kernel (bf16 storage; made by cast_weights_batched), passed beside the plain one."""
    if w_packed is not None:
        check(_lib.lib().unetrir_conv2d_fwd_packed_bf16(C.byref(g), _p(x), x.ld, _p(w), _p(w_packed), _p(bias), _p(addend),
                                                        addend.ld if addend is not None else 0, _p(y), y.ld, _stream()),
              "conv2d_fwd_packed")
        return
    check(_fn("conv2d_fwd", x.sfx)(C.byref(g), _p(x), x.ld, _p(w), _p(bias), _p(addend),
                                            addend.ld if addend is not None else 0, _p(y), y.ld, _stream()),
          "conv2d_fwd")


def conv2d_dgrad(g, dy: Act, wt, dx: Act, addend: Act = None):
    check(_fn("conv2d_dgrad", dy.sfx)(C.byref(g), _p(dy), dy.ld, _p(wt), _p(addend),
                                              addend.ld if addend is not None else 0, _p(dx), dx.ld, _stream()),
          "conv2d_dgrad")


def conv2d_colstat_rows(g, dgrad, x: Act):
    """Rows of fused column statistics the conv kernel serving this layer emits (0: none; bf16 storage only)."""
    if x.sfx != "bf16":
        return 0
    return int(_lib.lib().unetrir_conv2d_colstat_rows_bf16(C.byref(g), int(bool(dgrad)), x.ld))


K3_NAMES = ("tap-table", "conv3x3r", "conv3x3g", "conv3x3g pair", "conv3x3h", "conv3x3s", "conv3x3p", "stem")


def conv3x3_kernel(g, dgrad, x: Act):
    """Name of the kernel that serves this 3x3 stride-1 layer (bf16 storage) under the switches in effect."""
    return K3_NAMES[int(_lib.lib().unetrir_conv3x3_kernel_id_bf16(C.byref(g), int(bool(dgrad)), x.ld))]


def conv2d_fwd_colstat(g, x: Act, w, bias, y: Act, colstat, addend: Act = None):
    """conv2d_fwd that also writes per-tile (sum, sum of squares) of the stored output: colstat [rows][Cout][2] fp32."""
    check(_lib.lib().unetrir_conv2d_fwd_colstat_bf16(C.byref(g), _p(x), x.ld, _p(w), _p(bias), _p(addend),
                                                     addend.ld if addend is not None else 0, _p(y), y.ld, _p(colstat), _stream()),
          "conv2d_fwd_colstat")


def conv2d_dgrad_colstat(g, dy: Act, wt, dx: Act, colstat, addend: Act = None):
    check(_lib.lib().unetrir_conv2d_dgrad_colstat_bf16(C.byref(g), _p(dy), dy.ld, _p(wt), _p(addend),
                                                       addend.ld if addend is not None else 0, _p(dx), dx.ld, _p(colstat), _stream()),
          "conv2d_dgrad_colstat")


def conv2d_transpose_colstat_rows(g, x: Act):
    """Rows of fused column statistics of a Conv2DTranspose forward launch (0: none; bf16 storage only)."""
    if x.sfx != "bf16":
        return 0
    return int(_lib.lib().unetrir_conv2d_transpose_colstat_rows_bf16(C.byref(g), x.ld))


def conv2d_transpose_fwd_colstat(g, x: Act, wt, bias, y: Act, colstat):
    """conv2d_transpose_fwd that also writes per-tile (sum, sum of squares) of the stored output: colstat [rows][Cout][2] fp32."""
    check(_lib.lib().unetrir_conv2d_transpose_fwd_colstat_bf16(C.byref(g), _p(x), x.ld, _p(wt), _p(bias), _p(y), y.ld, _p(colstat),
                                                               _stream()), "conv2d_transpose_fwd_colstat")


def bn_stats_colstat(colstat, rows, P, C_, gamma, beta, affine, saved, moving_mean=None, moving_var=None, eps=1e-3, momentum=0.99):
    """BatchNormalization() batch statistics from fused conv-epilogue partials."""
    check(_lib.lib().unetrir_bn_stats_colstat(_p(colstat), rows, P, C_, _p(gamma), _p(beta), eps, momentum, _p(moving_mean),
                                              _p(moving_var), _p(affine), _p(saved), _stream()), "bn_stats_colstat")


def bn_colstat_act_add(colstat, rows, x: Act, gamma, beta, affine, saved, y: Act, act=2, addend: Act = None, moving_mean=None,
                       moving_var=None, eps=1e-3, momentum=0.99):
    """BatchNormalization (training) from fused conv-epilogue partials + Add + activation in one call (tensors <= 64 MB: one launch)."""
    check(_fn("bn_colstat_act_add", x.sfx)(_p(colstat), rows, _p(x), x.ld, x.P, x.C, _p(gamma), _p(beta), eps, momentum, _p(moving_mean),
                                            _p(moving_var), _p(affine), _p(saved), int(act), _p(addend),
                                            addend.ld if addend is not None else 0, _p(y), y.ld, _stream()), "bn_colstat_act_add")


def colsum_colstat(colstat, rows, ldc, c0, C_, out):
    """Bias gradient from fused conv-epilogue partials: out[c] = sum over rows of colstat[row][c0 + c][0]."""
    check(_lib.lib().unetrir_colsum_colstat(_p(colstat), rows, ldc, c0, C_, _p(out), _stream()), "colsum_colstat")


def conv2d_wgrad_ws_bytes(g):
    return _lib.lib().unetrir_conv2d_wgrad_ws_bytes(C.byref(g))


def wgrad_defer_supported(storage):
    """Whether the weight gradients of a trunk with this storage type ("f32" | "bf16") can leave their split-K reduction to a
    ReduceBatch (the *_partials_* entry points exist for bf16 storage)."""
    return storage == "bf16"


class ReduceBatch:
    """Deferred split-K reductions of weight gradients (bf16 storage): `conv2d_wgrad(..., defer=batch)` runs the weight-gradient kernel
    only, leaving its fp32 partial slabs in this object's arena; `flush()` reduces everything gathered so far in one launch per 16
    reductions (unetrir_splitk_reduce_batched) - same sums in the same order as the immediate form, bit-identical.  A step has 23
    (configs[1]) to 57 (configs[4]) such reductions of 5-20 us, most of it launch latency.  The owner flushes before anything reads
    the gradients (bucket hand-over, optimizer) and on the stream the weight gradients ran on."""

    def __init__(self, device, arena_bytes, capacity=64, park_max_bytes=None):
        """park_max_bytes: weight gradients whose slabs are larger reduce at once instead (None: park everything).  Measured on
        MI355X (scripts/ab_switch.py attr:park_reduces): the 37.7 MB slab sets of configs[1] are read back from the Infinity Cache when
        reduced at once and from HBM when parked - parking them costs 0.06-0.09 ms per step; the few-MB slab sets of configs[4] gain
        0.1 (side-stream schedule) to 0.57 ms (single stream) from sharing a launch."""
        self.arena = Workspace(device, int(arena_bytes))
        self.descs = (_lib.ReduceDesc * capacity)()
        self.n, self.used, self.capacity = 0, 0, capacity
        self.park_max_bytes = park_max_bytes

    def __len__(self):
        return self.n

    def take(self, nbytes):
        """(pointer, bytes) of a fresh 256-byte-aligned arena region, or None when the arena or the descriptor table is full (the
        caller flushes first, or reduces immediately)."""
        nbytes = -(-int(nbytes) // 256) * 256
        if self.n == 0 and nbytes > self.arena.nbytes:
            self.arena.reserve(4 * nbytes)          # empty: nothing parked points into the old buffer
        if self.n == self.capacity or self.used + nbytes > self.arena.nbytes:
            return None
        ptr = self.arena.buf.data_ptr() + self.used
        self.used += nbytes
        return C.c_void_p(ptr), nbytes

    def flush(self):
        if self.n:
            check(_lib.lib().unetrir_splitk_reduce_batched(self.descs, self.n, _stream()), "splitk_reduce_batched")
        self.n, self.used = 0, 0


def _wgrad(name, g, x: Act, dy: Act, dw, ws: Workspace, reg, w, defer, need):
    if defer is not None and x.sfx == "bf16" and (defer.park_max_bytes is None or need <= defer.park_max_bytes):
        got = defer.take(need)
        if got is None and len(defer):
            defer.flush()                  # arena or table full: reduce what is there (same stream), then defer this one
            got = defer.take(need)
        if got is not None:
            ptr, nbytes = got
            check(getattr(_lib.lib(), f"unetrir_{name}_partials_bf16")(C.byref(g), _p(x), x.ld, _p(dy), dy.ld, _p(dw), float(reg), _p(w), ptr, nbytes,
                                                                       C.byref(defer.descs[defer.n]), _stream()), name + "_partials")
            if defer.descs[defer.n].nsplit:
                defer.n += 1
            else:
                defer.used -= nbytes       # written straight into dw: the region is free again
            return
    ws.reserve(need)
    check(_fn(name, x.sfx)(C.byref(g), _p(x), x.ld, _p(dy), dy.ld, _p(dw), float(reg), _p(w), ws.ptr, ws.nbytes, _stream()), name)


def conv2d_wgrad(g, x: Act, dy: Act, dw, ws: Workspace, reg=0.0, w=None, defer: ReduceBatch = None):
    """Conv2DBackpropFilter.  defer: leave the split-K reduction to `defer.flush()` (bf16 storage; see ReduceBatch)."""
    _wgrad("conv2d_wgrad", g, x, dy, dw, ws, reg, w, defer, conv2d_wgrad_ws_bytes(g))


def conv2d_transpose_fwd(g, x: Act, wt, bias, y: Act):
    """Conv2DTranspose(strides=2, padding='same') forward (dl_models/u_net.py:297-304)."""
    check(_fn("conv2d_transpose_fwd", x.sfx)(C.byref(g), _p(x), x.ld, _p(wt), _p(bias), _p(y), y.ld,
                                                      _stream()), "conv2d_transpose_fwd")


def conv2d_transpose_dgrad(g, dy: Act, w, dx: Act, addend: Act = None, w_packed=None):
    if w_packed is not None:
        check(_lib.lib().unetrir_conv2d_transpose_dgrad_packed_bf16(C.byref(g), _p(dy), dy.ld, _p(w), _p(w_packed), _p(addend),
                                                                    addend.ld if addend is not None else 0, _p(dx), dx.ld,
                                                                    _stream()), "conv2d_transpose_dgrad_packed")
        return
    check(_fn("conv2d_transpose_dgrad", dy.sfx)(C.byref(g), _p(dy), dy.ld, _p(w), _p(addend),
                                                        addend.ld if addend is not None else 0, _p(dx), dx.ld,
                                                        _stream()), "conv2d_transpose_dgrad")


def conv2d_transpose_wgrad_ws_bytes(g):
    return _lib.lib().unetrir_conv2d_transpose_wgrad_ws_bytes(C.byref(g))


def conv2d_transpose_wgrad(g, x: Act, dy: Act, dw, ws: Workspace, reg=0.0, w=None, defer: ReduceBatch = None):
    _wgrad("conv2d_transpose_wgrad", g, x, dy, dw, ws, reg, w, defer, conv2d_transpose_wgrad_ws_bytes(g))


def dense_fwd(x: Act, w, bias, y: Act, ws: Workspace):
    """Dense on a small batch (dl_models/u_net.py:259): x [B,1,1,K] . w[N][K]^T + bias -> y [B,1,1,N]."""
    B, K, N = x.B, x.C, y.C
    ws.reserve(_lib.lib().unetrir_dense_fwd_ws_bytes(B, K, N))
    check(_lib.lib().unetrir_dense_fwd_f32(_p(x), x.ld, _p(w), _p(bias), _p(y), y.ld, B, K, N, ws.ptr, ws.nbytes, _stream()),
          "dense_fwd")


def dense_dgrad_supported(B, K, N):
    return bool(_lib.lib().unetrir_dense_dgrad_supported(B, K, N))


def dense_dgrad(dy: Act, w, dx: Act, ws: Workspace):
    """dL/dx of Dense(N) (dl_models/u_net.py:259) from the [N][K] kernel as stored: dx[b][k] = sum_n dy[b][n] w[n][k]."""
    B, N, K = dy.P, dy.C, dx.C
    ws.reserve(_lib.lib().unetrir_dense_dgrad_ws_bytes(B, K, N))
    check(_lib.lib().unetrir_dense_dgrad_f32(_p(dy), dy.ld, _p(w), _p(dx), dx.ld, B, K, N, ws.ptr, ws.nbytes, _stream()),
          "dense_dgrad")


def transpose_weight(w, wt, N, T, C_):
    """[N][T][C] -> [C][T][N]."""
    check(_lib.lib().unetrir_transpose_weight_f32(_p(w), _p(wt), N, T, C_, _stream()), "transpose_weight")


def cast_weight_bf16(w, o, N, T, C_, Cp):
    """fp32 master [N][T][C] -> bf16 [N][T][Cp] (zero-padded channels)."""
    check(_lib.lib().unetrir_cast_weight_bf16(_p(w), _p(o), N, T, C_, Cp, _stream()), "cast_weight_bf16")


def transpose_cast_weight_bf16(w, wt, N, T, C_, Np):
    """fp32 master [N][T][C] -> bf16 [C][T][Np] (zero-padded rows)."""
    check(_lib.lib().unetrir_transpose_cast_weight_bf16(_p(w), _p(wt), N, T, C_, Np, _stream()),
          "transpose_cast_weight_bf16")


def make_cast_table(entries, device):
    """Device-resident unetrir_cast_desc array for cast_weights_batched: entries = [(w fp32 [N][T][C], same bf16 or None,
    transposed bf16 or None, N, T, C, Cp, Np[, packed bf16 or None])].  The returned tensor keeps the table alive; the weight
    tensors must too."""
    arr = (_lib.CastDesc * len(entries))()
    for i, ent in enumerate(entries):
        w, same, tr, N, T, C_, Cp, Np = ent[:8]
        pk = ent[8] if len(ent) > 8 else None
        arr[i] = _lib.CastDesc(w.data_ptr(), same.data_ptr() if same is not None else None,
                               tr.data_ptr() if tr is not None else None, N, T, C_, Cp, Np, 0,
                               pk.data_ptr() if pk is not None else None)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return raw.to(device), len(entries)


def cast_weights_batched(table):
    """fp32 master kernels -> bf16 work copies (both orientations) for every layer of the table in two launches."""
    dev, n = table
    check(_lib.lib().unetrir_cast_weights_batched_bf16(dev.data_ptr(), n, _stream()), "cast_weights_batched")


def stage_h2d(src_ptrs, pinned, dev, stream):
    """Host arrays (addresses src_ptrs) -> pinned staging tensors -> device tensors on `stream`, in one foreign call (no interpreter
    lock while it copies).  src_ptrs[k] must stay alive until the call returns; pinned[k] until the copies have completed."""
    n = len(src_ptrs)
    vp, sz = C.c_void_p * n, C.c_size_t * n
    nbytes = [p.numel() * p.element_size() for p in pinned]
    check(_lib.lib().unetrir_stage_h2d(n, vp(*src_ptrs), vp(*[p.data_ptr() for p in pinned]), vp(*[d.data_ptr() for d in dev]),
                                       sz(*nbytes), C.c_void_p(stream.cuda_stream)), "stage_h2d")


def add_f32_to_bf16(a: Act, b: Act, y: Act):
    """y = a + b with a, y bf16 and b fp32, dense buffers (Add() of dl_models/u_net.py:229 in the bf16 trunk)."""
    check(_lib.lib().unetrir_add_f32_to_bf16(_p(a), _p(b), _p(y), a.base.numel(), _stream()), "add_f32_to_bf16")


def cast_f32_to_bf16(a: Act, y: Act):
    check(_lib.lib().unetrir_cast_f32_to_bf16(_p(a), _p(y), a.base.numel(), _stream()), "cast_f32_to_bf16")


def cast_bf16_to_f32(a: Act, y: Act):
    check(_lib.lib().unetrir_cast_bf16_to_f32(_p(a), _p(y), a.base.numel(), _stream()), "cast_bf16_to_f32")


# ---- BatchNormalization + ReLU ---------------------------------------------------------------

def bn_ws_bytes(P, C_):
    return _lib.lib().unetrir_bn_ws_bytes(P, C_)


def bn_stats(x: Act, gamma, beta, affine, saved, ws: Workspace, moving_mean=None, moving_var=None,
             eps=1e-3, momentum=0.99):
    """BatchNormalization() batch statistics (dl_models/u_net.py:368)."""
    ws.reserve(bn_ws_bytes(x.P, x.C))
    check(_fn("bn_stats", x.sfx)(_p(x), x.ld, x.P, x.C, _p(gamma), _p(beta), eps, momentum,
                                          _p(moving_mean), _p(moving_var), _p(affine), _p(saved), ws.ptr, ws.nbytes,
                                          _stream()), "bn_stats")


def bn_apply(x: Act, affine, y: Act, relu=True):
    check(_fn("bn_apply", x.sfx)(_p(x), x.ld, x.P, x.C, _p(affine), int(relu), _p(y), y.ld, _stream()),
          "bn_apply")


def bn_bwd(da: Act, x: Act, gamma, affine, saved, dx: Act, dgamma, dbeta, ws: Workspace, relu=True):
    ws.reserve(bn_ws_bytes(x.P, x.C))
    if x.sfx == "f32":
        check(_lib.lib().unetrir_bn_bwd_f32(_p(da), da.ld, _p(x), x.ld, x.P, x.C, _p(gamma), _p(affine), _p(saved),
                                            int(relu), _p(dx), dx.ld, _p(dgamma), _p(dbeta), ws.ptr, ws.nbytes,
                                            _stream()), "bn_bwd")
    else:
        check(_lib.lib().unetrir_bn_bwd_bf16(_p(da), da.ld, _p(x), x.ld, x.P, x.C, _p(affine), _p(saved), int(relu),
                                             _p(dx), dx.ld, _p(dgamma), _p(dbeta), ws.ptr, ws.nbytes, _stream()),
              "bn_bwd")


def bn_inference_affine(gamma, beta, moving_mean, moving_var, eps, affine):
    """training=False: the BatchNormalization affine from the moving statistics (rir_generation.py:165)."""
    check(_lib.lib().unetrir_bn_inference_affine_f32(_p(gamma), _p(beta), _p(moving_mean), _p(moving_var), eps, moving_mean.numel(),
                                                     _p(affine), _stream()), "bn_inference_affine")


def bn_act_add(x: Act, affine, y: Act, act=2, addend: Act = None):
    """BatchNormalization -> Add -> activation (dl_models/res_ae.py:331-336); act 0 none, 1 ReLU, 2 LeakyReLU(0.3)."""
    check(_fn("bn_act_add", x.sfx)(_p(x), x.ld, x.P, x.C, _p(affine), int(act), _p(addend),
                                   addend.ld if addend is not None else 0, _p(y), y.ld, _stream()), "bn_act_add")


def bn_bwd_junction(da: Act, x: Act, out: Act, affine, saved, dx: Act, dgamma, dbeta, ws: Workspace, act=2, gskip: Act = None,
                    gskip_add: Act = None):
    """Backward of out = act(BatchNormalization(x) + skip) (dl_models/res_ae.py:334-336, :478-480): dx, dgamma, dbeta and the
    gradient of `skip` (gskip = g (+ gskip_add)) from one reduce / finalize / apply sequence."""
    ws.reserve(bn_ws_bytes(x.P, x.C))
    check(_fn("bn_bwd_junction", x.sfx)(_p(da), da.ld, _p(x), x.ld, _p(out), out.ld, x.P, x.C, _p(affine), _p(saved), int(act), _p(dx), dx.ld,
                                        _p(gskip), gskip.ld if gskip is not None else 0, _p(gskip_add),
                                        gskip_add.ld if gskip_add is not None else 0, _p(dgamma), _p(dbeta), ws.ptr, ws.nbytes, _stream()),
          "bn_bwd_junction")


def act_bwd(da: Act, out: Act, g: Act, act=2):
    check(_fn("act_bwd", out.sfx)(_p(da), da.ld, _p(out), out.ld, out.P, out.C, int(act), _p(g), g.ld, _stream()), "act_bwd")


def add(a, b, y):
    check(_lib.lib().unetrir_add_f32(_p(a), _p(b), _p(y), a.numel(), _stream()), "add")


def colsum(x: Act, out, ws: Workspace):
    """Bias gradient: out[c] = sum over pixels."""
    ws.reserve(bn_ws_bytes(x.P, x.C))
    check(_fn("colsum", x.sfx)(_p(x), x.ld, x.P, x.C, _p(out), ws.ptr, ws.nbytes, _stream()), "colsum")


def relu_fwd(x: Act, y: Act):
    if x.sfx == "f32":
        check(_lib.lib().unetrir_relu_fwd_f32(_p(x), x.ld, x.P, x.C, _p(y), y.ld, _stream()), "relu_fwd")
    else:
        check(_lib.lib().unetrir_bn_apply_bf16(_p(x), x.ld, x.P, x.C, None, 1, _p(y), y.ld, _stream()), "relu_fwd")


def relu_bwd(da: Act, x: Act, dx: Act):
    check(_fn("relu_bwd", x.sfx)(_p(da), da.ld, _p(x), x.ld, x.P, x.C, _p(dx), dx.ld, _stream()),
          "relu_bwd")


# ---- boundary / head / loss ------------------------------------------------------------------

def nchw_to_nhwc_pad(x, y: Act):
    B, C_, H, W = x.shape
    check(_fn("nchw_to_nhwc_pad", y.sfx)(_p(x), B, C_, H, W, _p(y), y.ld, _stream()), "nchw_to_nhwc_pad")


def head6x6_supported(C_):
    return bool(_lib.lib().unetrir_head6x6_supported(C_))


def head6x6_fwd(x: Act, w, bias, y: Act):
    """Conv2D(2, (6,6), padding='same') (dl_models/u_net.py:248), direct kernel."""
    check(_fn("head6x6_fwd", x.sfx)(_p(x), x.ld, x.B, x.H, x.W, x.C, _p(w), _p(bias), _p(y), y.ld, _stream()),
          "head6x6_fwd")


def head6x6_wgrad(x: Act, dy: Act, dw, ws: Workspace):
    ws.reserve(_lib.lib().unetrir_head6x6_wgrad_ws_bytes(x.C))
    check(_fn("head6x6_wgrad", x.sfx)(_p(x), x.ld, x.B, x.H, x.W, x.C, _p(dy), dy.ld, _p(dw), ws.ptr, ws.nbytes,
                                               _stream()), "head6x6_wgrad")


def head6x6_dgrad_supported(W_, C_):
    return bool(_lib.lib().unetrir_head6x6_dgrad_supported(W_, C_))


def head6x6_dgrad(dy: Act, w, dx: Act):
    """Adjoint of head6x6_fwd on bf16 activations (matrix-core kernel): dx = conv-transpose of dy[..., :2] with the fp32 kernel w."""
    if dy.sfx != "bf16" or dx.sfx != "bf16":
        raise ValueError("head6x6_dgrad is a bf16-storage kernel")
    check(_lib.lib().unetrir_head6x6_dgrad_bf16(_p(dy), dy.ld, dy.B, dy.H, dy.W, _p(w), dx.C, _p(dx), dx.ld, _stream()),
          "head6x6_dgrad")


# ---- waveform <-> feature transforms ----------------------------------------------------------------

PAD_MODES = {"reflect": 0, "constant": 1}


def stft_frames(T, hop_length):
    return _lib.lib().unetrir_stft_frames(int(T), int(hop_length))


def stft_features(wav, out, n_fft=256, win_length=128, hop_length=64, pad_mode="reflect", remove_mean=True, normalize=True):
    """wav fp32 [B, T] -> out fp32 [B, 2, H, W] (amplitude / phase planes, zero padded): Loader mean removal +
    FeatureExtractor.extract + Normalizer.normalize + TensorPadder.pad_amp_phase (preprocess.py:13-18, :26-32, :56, :65-70)."""
    if wav.dtype != torch.float32 or out.dtype != torch.float32 or not wav.is_contiguous() or not out.is_contiguous():
        raise ValueError("stft_features takes contiguous fp32 tensors")
    if wav.dim() != 2 or out.dim() != 4 or out.shape[0] != wav.shape[0] or out.shape[1] != 2:
        raise ValueError("stft_features: wav [B, T], out [B, 2, H, W]")
    if pad_mode not in PAD_MODES:
        raise ValueError(f"pad_mode must be one of {sorted(PAD_MODES)}")
    B, T = wav.shape
    check(_lib.lib().unetrir_stft_features_f32(_p(wav), B, T, n_fft, win_length, hop_length, PAD_MODES[pad_mode], int(remove_mean),
                                               int(normalize), _p(out), out.shape[2], out.shape[3], _stream()), "stft_features")


def istft_features(feat, wav, n_bins, n_frames, n_fft=256, win_length=128, hop_length=64, denormalize=True):
    """feat fp32 [B, 2, H, W] -> wav fp32 [B, hop (n_frames - 1)]: un_pad + Normalizer.denormalize + librosa.istft
    (postprocess.py:68-71, :127-134; preprocess.py:34-41, :107-113)."""
    if feat.dtype != torch.float32 or wav.dtype != torch.float32 or not feat.is_contiguous() or not wav.is_contiguous():
        raise ValueError("istft_features takes contiguous fp32 tensors")
    if feat.dim() != 4 or feat.shape[1] != 2 or wav.dim() != 2 or wav.shape[0] != feat.shape[0] or \
            wav.shape[1] != hop_length * (n_frames - 1):
        raise ValueError("istft_features: feat [B, 2, H, W], wav [B, hop_length * (n_frames - 1)]")
    B, _, H, W = feat.shape
    check(_lib.lib().unetrir_istft_features_f32(_p(feat), B, H, W, n_bins, n_frames, n_fft, win_length, hop_length,
                                                int(denormalize), _p(wav), _stream()), "istft_features")


def sigmoid_loss(logits: Act, target, alpha, inv_norm, pred, dlogits: Act, loss_out, ws: Workspace, phase_ref=None, phase_weight=None):
    """sigmoid head (dl_models/u_net.py:249) + compute_loss (main_training.py:203-231) + dL/dlogits.
    phase_ref: the network input [B,2,H,W] (the `diff_loss` switch, main_training.py:214-217); phase_weight: fp32 [W] column
    weights of the phase term (the `sigmoid_loss` switch, :221-222)."""
    B, _, H, W = target.shape
    ws.reserve(_lib.lib().unetrir_loss_ws_bytes(B * H * W))
    if phase_ref is None and phase_weight is None:
        check(_fn("sigmoid_loss", dlogits.sfx)(_p(logits), logits.ld, _p(target), B, H, W, alpha, inv_norm, _p(pred),
                                               _p(dlogits), _p(loss_out), ws.ptr, ws.nbytes, _stream()), "sigmoid_loss")
        return
    if phase_ref is not None and (tuple(phase_ref.shape) != tuple(target.shape) or phase_ref.dtype != torch.float32 or
                                  not phase_ref.is_contiguous() or phase_ref.device != target.device):
        raise ValueError("phase_ref must be a contiguous float32 tensor of the target's shape on the same device")
    if phase_weight is not None and (tuple(phase_weight.shape) != (W,) or phase_weight.dtype != torch.float32 or
                                     phase_weight.device != target.device):
        raise ValueError(f"phase_weight must be float32 [{W}] on the target's device")
    check(_fn("sigmoid_loss_ex", dlogits.sfx)(_p(logits), logits.ld, _p(target), _p(phase_ref), _p(phase_weight), B, H, W, alpha,
                                              inv_norm, _p(pred), _p(dlogits), _p(loss_out), ws.ptr, ws.nbytes, _stream()),
          "sigmoid_loss_ex")


def sigmoid_nchw(logits: Act, pred):
    B, _, H, W = pred.shape
    check(_lib.lib().unetrir_sigmoid_nchw_f32(_p(logits), logits.ld, B, H, W, _p(pred), _stream()), "sigmoid_nchw")


def sigmoid_bwd(pred, dpred, dlogits: Act):
    B, _, H, W = pred.shape
    check(_fn("sigmoid_bwd", dlogits.sfx)(_p(pred), _p(dpred), B, H, W, _p(dlogits), _stream()), "sigmoid_bwd")


# ---- information vector ----------------------------------------------------------------------

def embedding_fwd(idx, table, out):
    check(_lib.lib().unetrir_embedding_fwd_f32(_p(idx), idx.numel(), _p(table), table.shape[0], table.shape[1],
                                               _p(out), _stream()), "embedding_fwd")


def embedding_bwd(idx, dout, dtable):
    check(_lib.lib().unetrir_embedding_bwd_f32(_p(idx), idx.numel(), _p(dout), dtable.shape[0], dtable.shape[1],
                                               _p(dtable), _stream()), "embedding_bwd")


def dropout_mask(mask, p, seed, step):
    """Keep mask of Dropout(p) scaled by 1/(1-p) (dl_models/u_net.py:260), draw number `step` of stream `seed`."""
    check(_lib.lib().unetrir_dropout_mask_f32(_p(mask), mask.numel(), float(p), int(seed), int(step), _stream()), "dropout_mask")


def index_to_i32(idx, out):
    """int32 / int64 index tensor -> the int32 array the embedding kernels read."""
    if idx.dtype not in (torch.int32, torch.int64) or not idx.is_contiguous() or idx.numel() != out.numel():
        raise ValueError("indices must be a contiguous int32 or int64 tensor of the expected size")
    if idx.device != out.device:      # a host pointer handed to the kernel is a GPU memory fault, not an exception
        raise ValueError(f"indices live on {idx.device}, the engine on {out.device}")
    check(_lib.lib().unetrir_index_to_i32(_p(idx), idx.element_size(), idx.numel(), _p(out), _stream()), "index_to_i32")


def mul(x, m, y):
    check(_lib.lib().unetrir_mul_f32(_p(x), _p(m), _p(y), x.numel(), _stream()), "mul")


def sumsq(x, coef, out, accumulate, ws: Workspace):
    ws.reserve(512 * 8)
    check(_lib.lib().unetrir_sumsq_f32(_p(x), x.numel(), float(coef), _p(out), int(accumulate), ws.ptr, ws.nbytes,
                                       _stream()), "sumsq")


def adam(theta, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    """tf.keras.optimizers.Adam step over a flat buffer (main_training.py:168-169, :268)."""
    check(_lib.lib().unetrir_adam_f32(_p(theta), _p(g), _p(m), _p(v), theta.numel(), float(lr_t), beta1, beta2, eps,
                                      float(grad_scale), _stream()), "adam")


def sgd(theta, g, lr, grad_scale=1.0):
    """tf.keras.optimizers.SGD(learning_rate) step (main_training.py:166-167)."""
    check(_lib.lib().unetrir_sgd_f32(_p(theta), _p(g), theta.numel(), float(lr), float(grad_scale), _stream()), "sgd")


def nadam(theta, g, m, v, lr, beta1, beta2, eps, c_g, c_m, c_v, grad_scale=1.0):
    """tf.keras.optimizers.Nadam step (main_training.py:164-165) with the step's schedule coefficients (engine.DeviceCounters)."""
    check(_lib.lib().unetrir_nadam_f32(_p(theta), _p(g), _p(m), _p(v), theta.numel(), float(lr), beta1, beta2, eps, float(c_g), float(c_m),
                                       float(c_v), float(grad_scale), _stream()), "nadam")


def adam_dev(theta, g, m, v, hyper):
    """adam() with its five scalars read from device memory (hyper: fp32 [>=5], written by step_advance): HIP-graph replay."""
    check(_lib.lib().unetrir_adam_dev_f32(_p(theta), _p(g), _p(m), _p(v), theta.numel(), _p(hyper), _stream()), "adam_dev")


def dropout_mask_dev(mask, p, seed, state, offset):
    """dropout_mask() with draw number state[2] + offset read from device memory (state: the uint64 [3] of step_advance)."""
    check(_lib.lib().unetrir_dropout_mask_dev_f32(_p(mask), mask.numel(), float(p), int(seed), C.c_void_p(state.data_ptr() + 16),
                                                  int(offset), _stream()), "dropout_mask_dev")


def step_advance(state, cfg, hyper, n_draws, advance_t=True):
    """Start of a step whose counters live in device memory: t += 1, hyper = (lr_t, beta1, beta2, eps, grad_scale), draw base.
    advance_t=False: a forward-only pass (the validation loop draws dropout masks, main_training.py:297-300)."""
    check(_lib.lib().unetrir_step_advance(_p(state), _p(cfg), _p(hyper), int(n_draws), int(bool(advance_t)), _stream()), "step_advance")


def reset_tile_tickets():
    check(_lib.lib().unetrir_reset_tile_tickets(), "reset_tile_tickets")


# ---- kernel-selection switches --------------------------------------------------------------------

def get_config():
    """The kernel-selection switches in effect (unetrir_config) as a dict."""
    c = _lib.Config()
    check(_lib.lib().unetrir_get_config(C.byref(c)), "get_config")
    return {n: getattr(c, n) for n, _ in _lib.Config._fields_}


_CONFIG_GEN = [0]


def config_generation():
    """Counts the set_config calls that changed a switch.  Anything derived from the dispatch (which kernel serves a layer, how many
    rows of column statistics it writes) is cached against this number by the engines and recomputed when it moves."""
    return _CONFIG_GEN[0]


def set_config(**switches):
    """Replace kernel-selection switches (tests, A/B scripts), e.g. set_config(conv3x3s=0); returns the previous values."""
    old = get_config()
    new = dict(old)
    for k, v in switches.items():
        if k not in new:
            raise KeyError(f"unknown switch {k}")
        new[k] = int(v)
    c = _lib.Config(*[new[n] for n, _ in _lib.Config._fields_])
    check(_lib.lib().unetrir_set_config(C.byref(c)), "set_config")
    if new != old:
        _CONFIG_GEN[0] += 1
    return old


# ---- profiling hooks -------------------------------------------------------------------------

def prof_enable(on):
    _lib.lib().unetrir_prof_enable(int(on))


def prof_collect():
    n = _lib.PROF_FAMILIES
    counts = (C.c_int * n)()
    ms = (C.c_double * n)()
    fl = (C.c_double * n)()
    _lib.lib().unetrir_prof_collect(counts, ms, fl)
    return list(counts), list(ms), list(fl)

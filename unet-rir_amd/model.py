"""Drop-in boundary: the model classes ``main_training.py`` constructs (main_training.py:130-161), as ``torch.nn.Module``s
whose forward and backward run on the HIP engines.

  ``UNet``   dl_models/u_net.py:34-64  - constructor names / order / defaults of the reference, every ``mode`` (0: the
             hand-scheduled ``UNetEngine``, fp32 or bf16 storage, optional side-stream schedule; 1-3: ``UNetGraphEngine``)
  ``ResAE``  dl_models/res_ae.py:35-70 - the residual autoencoder, with ``.encoder`` / ``.decoder`` / ``.model``
  ``Autoencoder``  dl_models/autoencoder.py:34-62 - the plain conv / conv-transpose autoencoder, same surface

Both keep the reference's call shape ``model.model([spec_in, emb], training=...)`` (NHWC, main_training.py:261),
``model.model.trainable_variables`` / ``.losses``, ``summary()``, ``save()`` / ``load()`` / ``load_weights()``,
``predict_stft()``.  There is no CPU or eager fallback: without the HIP library / a GPU construction raises.

How a module maps onto the engines
  * Parameters exist from construction on (the Keras classes call ``_build()`` in ``__init__``): an engine for
    ``batch_size`` (default 1) is built at once, and every ``nn.Parameter`` is a VIEW of that engine's flat parameter
    buffer - created once, never replaced.  A forward pass with another batch size builds another engine that ALIASES
    the same parameters, gradients, Adam moments, work copies and BatchNorm moving statistics (only activation buffers
    depend on the batch size), so an optimizer created from ``model.parameters()`` keeps training the live model.
  * ``loss.backward()`` runs the engine's backward pass; each parameter's ``.grad`` is then a view of the engine's flat
    gradient buffer (no copies).  Gradients are WRITTEN, not accumulated, by a backward pass - the semantics of
    ``tape.gradient`` in main_training.py:267.  The l2(0.001) terms of the strided / transposed kernels are folded into
    the weight-gradient kernels (``fold_l2=True``, as ``Trainer`` does), and ``model.model.losses`` then carries their
    VALUES; with ``fold_l2=False`` the terms are ordinary differentiable torch expressions and gradients are cloned.
  * Parameters changed from outside (``torch.optim`` step, ``load_state_dict``, ``load_weights``) are noticed through
    their version counters and the engine's transposed / bf16 work copies are refreshed before the next forward.
  * The returned prediction is the engine's own output buffer (the next forward pass overwrites it); ``predict_stft``
    returns a copy.
"""
import os
import pickle

import numpy as np
import torch
from torch import nn

from . import ops
from .device import HipRuntime
from .ae import AutoencoderEngine
from .engine import L2_COEF, UNetEngine
from .resae import ResAEEngine
from .unet_graph import UNetGraphEngine


class _ModelFunction(torch.autograd.Function):
    """Whole-network autograd node: forward = engine forward, backward = engine backward.  Lets a main_training.py-style
    loop (any torch loss on the prediction, any torch optimizer) drive the HIP path."""

    @staticmethod
    def forward(ctx, module, eng, anchor, spec, emb, dropout_mask, *params):
        ctx.module, ctx.eng = module, eng
        ctx.set_materialize_grads(False)
        # an alias of the engine's output buffer: autograd marks the returned OBJECT as a graph output, the engine's own
        # tensor object stays a plain buffer
        return eng.forward(spec, emb, dropout_mask=dropout_mask).detach()

    @staticmethod
    def backward(ctx, dpred):
        module, eng = ctx.module, ctx.eng
        n_in = 6 + len(module._params) if not module.fold_l2 else 6
        if dpred is None:
            return (None,) * n_in
        eng.backward(dpred=dpred.contiguous(), include_reg=module.fold_l2)
        if module.fold_l2:
            module._point_grads_at_engine()
            return (None,) * 6
        return (None,) * 6 + tuple(eng.g[n].clone() for n in module._param_names)


class _KerasModelAdapter:
    """``model.model`` of the reference: callable as ``model.model([spec_in, emb], training=bool)`` with NHWC
    spectrograms (main_training.py:261), plus ``trainable_variables`` and ``losses`` (main_training.py:263-268)."""

    def __init__(self, module):
        self._m = module

    def __call__(self, inputs, training=False):
        spec, emb = inputs
        self._m.train(training)
        out = self._m(spec.permute(0, 3, 1, 2), emb)
        return out.permute(0, 2, 3, 1)

    @property
    def trainable_variables(self):
        return list(self._m.parameters())

    @property
    def losses(self):
        return self._m.regularization_losses()

    def predict(self, inputs):
        with torch.no_grad():
            return self(inputs, training=False).clone()


class _EngineModule(nn.Module):
    """Shared machinery of the boundary classes: engines per batch size over one parameter set."""

    def __init__(self, device, batch_size, n_replicas, dropout, fold_l2, runtime):
        super().__init__()
        self._device = torch.device(device)
        self._rt = runtime if runtime is not None else HipRuntime(self._device)      # raises without a GPU: no CPU fallback
        self.n_replicas = n_replicas
        self.dropout = dropout
        self.fold_l2 = fold_l2
        self._engines = {}
        self._main_B = batch_size if batch_size is not None else 1
        self.engine = None

    # subclasses: _new_engine(B, share) -> engine
    def _finish_init(self):
        eng = self._new_engine(self._main_B, None)
        eng.reset_parameters()
        self._engines[self._main_B] = eng
        self.engine = eng
        self._param_names = list(eng.specs)
        self._params = nn.ParameterList([nn.Parameter(eng.p[n]) for n in self._param_names])
        for n, b in eng.moving.items():
            self.register_buffer(n.replace(".", "_"), b, persistent=True)
        self._anchor = nn.Parameter(torch.zeros((), device=self._device), requires_grad=True)
        self._seen_version = self._version()
        self._reg_buf = torch.zeros(max(len(eng.l2_names), 1), dtype=torch.float32, device=self._device)

    def parameters(self, recurse=True):
        """trainable_variables: the anchor (an implementation detail of the autograd bridge) is not one of them."""
        return iter(self._params)

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        for n, p in zip(self._param_names, self._params):
            yield (prefix + ("." if prefix else "") + n, p)

    def named_engine_parameters(self):
        return dict(zip(self._param_names, self._params))

    def state_dict(self, *args, **kwargs):
        sd = {n: p.detach() for n, p in zip(self._param_names, self._params)}
        sd.update({n: b for n, b in self.engine.moving.items()})
        return sd

    def load_state_dict(self, state_dict, strict=True):
        eng = self.engine
        want = set(self._param_names) | set(eng.moving)
        if strict and set(state_dict) != want:
            raise KeyError(f"state_dict keys differ: missing {sorted(want - set(state_dict))}, unexpected {sorted(set(state_dict) - want)}")
        with torch.no_grad():
            for n, p in zip(self._param_names, self._params):
                if n in state_dict:
                    p.copy_(state_dict[n])
            for n, b in eng.moving.items():
                if n in state_dict:
                    b.copy_(state_dict[n])
        eng.t_dirty = True

    def _version(self):
        return sum(p._version for p in self._params)

    def _engine_for(self, B):
        eng = self._engines.get(B)
        if eng is None:
            eng = self._new_engine(B, self.engine)          # aliases parameters / moments / moving statistics of the first engine
            self._engines[B] = eng
        return eng

    def _point_grads_at_engine(self):
        eng = self.engine
        for n, p in zip(self._param_names, self._params):
            p.grad = eng.g[n]

    @property
    def model(self):
        return _KerasModelAdapter(self)

    def summary(self):
        eng = self.engine
        print(f'Model: "{self.name}"  input [B,2,{self.H},{self.W}] + [B,{self.inf_vector_shape}] -> [B,2,{self.H},{self.W}]')
        for n, s_ in eng.specs.items():
            print(f"  {n:32s} {str(s_.keras_shape):24s}")
        print(f"Total params: {eng.n_params():,}")

    def regularization_losses(self):
        """model.model.losses: one l2(0.001) term per regularised kernel (strided Conv2D / Conv2DTranspose kernels,
        dl_models/u_net.py:274, :302).  fold_l2=True: values from the device reduction (their gradient comes out of the
        weight-gradient kernels); fold_l2=False: differentiable torch expressions."""
        eng = self.engine
        if not self.fold_l2:
            named = self.named_engine_parameters()
            return [L2_COEF * (named[n] ** 2).sum() for n in eng.l2_names]
        for i, n in enumerate(eng.l2_names):
            s_ = eng.specs[n]
            ops.sumsq(eng.theta[s_.offset:s_.offset + s_.numel], L2_COEF, self._reg_buf[i:i + 1], False, eng.ws)
        return [self._reg_buf[i] for i in range(len(eng.l2_names))]

    def forward(self, spec, emb, dropout_mask=None):
        if spec.dim() != 4 or spec.shape[1] != 2:
            raise ValueError("spec must be NCHW [B,2,H,W]")
        eng = self._engine_for(spec.shape[0])
        eng.training = self.training
        v = self._version()
        if v != self._seen_version:          # an optimizer step / load_state_dict wrote the parameters in place
            eng.t_dirty = True
            self._seen_version = v
        eng.use_device_counters(False)       # this path passes its per-step scalars as launch arguments (trainer.Trainer(graph=True) does not)
        if dropout_mask is None and self.training and self.dropout:
            dropout_mask = eng.make_dropout_mask()
        spec = spec.to(self._device, torch.float32).contiguous()
        emb = emb.to(self._device)
        if torch.is_grad_enabled():
            extra = () if self.fold_l2 else tuple(self._params)
            return _ModelFunction.apply(self, eng, self._anchor, spec, emb, dropout_mask, *extra)
        return eng.forward(spec, emb, dropout_mask=dropout_mask)

    # ---- dl_models/u_net.py:72-118 (the same two methods in res_ae.py:78-127 and autoencoder.py:72-121)
    def get_callbacks(self):
        """CSVLogger(f'{name}.log') + EarlyStopping(monitor='val_loss', patience=20), dl_models/u_net.py:72-81."""
        from . import callbacks
        return [callbacks.CSVLogger(f"{self.name}.log", separator=",", append=False),
                callbacks.EarlyStopping(monitor="val_loss", patience=20)]

    def _compile_and_fit(self, x_train1, x_train2, y_train, x_val1, x_val2, y_val, batch_size, num_epochs, steps_per_epoch, learning_rate):
        """dl_models/u_net.py:83-118 (UNet: the rate is the constructor's; res_ae.py:89-127, autoencoder.py:83-121: an argument,
        default 1e-5): `model.compile(Adam(InverseTimeDecay(learning_rate, decay_steps=steps_per_epoch * 100, decay_rate=1)),
        loss=MeanSquaredError())` + `model.fit(x=[x_train1, x_train2], y=y_train, validation_data=..., batch_size, epochs, shuffle=False,
        callbacks=get_callbacks())`; returns `History.history` ({'loss': [...], 'val_loss': [...]}).

        Spectrograms are the reference's NHWC arrays [N, H, W, 2] (numpy or torch), information vectors [N, 2, 16] integers.  As in Keras
        the reported losses are the batch-size weighted means over an epoch of MSE + the l2 terms of `model.losses`, the validation pass
        runs on the moving BatchNorm statistics without Dropout, and the optimizer is the engine's Keras-Adam kernel (epsilon 1e-7
        outside the root) at the rate the schedule gives for the number of steps already taken."""
        lr0, decay_steps = float(learning_rate), float(steps_per_epoch * 100)
        self.summary()
        cbs = self.get_callbacks()
        self.stop_training = False
        for cb in cbs:
            cb.set_model(self)
            cb.on_train_begin()
        history = {"loss": [], "val_loss": []}

        def batches(x1, x2, y):
            n = len(x1)
            for i in range(0, n, batch_size):
                sl = slice(i, min(i + batch_size, n))
                spec = torch.as_tensor(x1[sl]).to(self._device, torch.float32).permute(0, 3, 1, 2).contiguous()
                tgt = torch.as_tensor(y[sl]).to(self._device, torch.float32).permute(0, 3, 1, 2).contiguous()
                yield spec, torch.as_tensor(x2[sl]).to(self._device), tgt

        def total_loss(pred, tgt):
            return ((pred - tgt) ** 2).mean() + sum(self.regularization_losses())

        iterations = 0
        for epoch in range(num_epochs):
            self.train()
            tot, cnt = 0.0, 0
            for spec, emb, tgt in batches(x_train1, x_train2, y_train):
                for p in self._params:
                    p.grad = None
                loss = total_loss(self(spec, emb), tgt)
                loss.backward()
                eng = self._engine_for(spec.shape[0])
                if not self.fold_l2:                      # gradients arrived as autograd tensors: hand them to the engine's buffer
                    for n, p in zip(self._param_names, self._params):
                        eng.g[n].copy_(p.grad if p.grad is not None else torch.zeros_like(p))
                eng.adam_step(lr0 / (1.0 + iterations / decay_steps))
                eng.t_dirty = True
                iterations += 1
                tot += float(loss.detach()) * spec.shape[0]
                cnt += spec.shape[0]
            logs = {"loss": tot / max(cnt, 1)}
            self.eval()
            tot, cnt = 0.0, 0
            with torch.no_grad():
                for spec, emb, tgt in batches(x_val1, x_val2, y_val):
                    tot += float(total_loss(self(spec, emb), tgt)) * spec.shape[0]
                    cnt += spec.shape[0]
            if cnt:
                logs["val_loss"] = tot / cnt
            for k, v in logs.items():
                history[k].append(v)
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end()
        if not history["val_loss"]:
            del history["val_loss"]
        return history

    def predict_stft(self, inputs):
        """dl_models/u_net.py:138-146: model.predict([spectrograms NHWC, vectors]) -> generated spectrograms NHWC."""
        return self.model.predict(inputs)

    # ---- persistence, dl_models/u_net.py:120-199 / dl_models/res_ae.py:128-210.  `weights.npz` replaces `weights.h5` (h5py is
    # not part of this image) and holds every variable in its Keras layout (HWIO Conv2D, HWOI Conv2DTranspose, [in, out]
    # Dense) plus the BatchNorm moving statistics.
    def save(self, save_folder="."):
        os.makedirs(save_folder, exist_ok=True)
        with open(os.path.join(save_folder, "parameters.pkl"), "wb") as f:
            pickle.dump(self._ctor_parameters(), f)
        arrays = {n: np.asarray(v) for n, v in self.engine.export_keras_params().items()}
        arrays.update({"moving/" + n: b.detach().cpu().numpy() for n, b in self.engine.moving.items()})
        np.savez(os.path.join(save_folder, "weights.npz"), **arrays)

    def load_weights(self, weights_path):
        with np.load(weights_path) as z:
            self.engine.load_keras_params({n: z[n] for n in self.engine.specs})
            for n, b in self.engine.moving.items():
                b.copy_(torch.from_numpy(z["moving/" + n]).to(b.device))
        self.engine.t_dirty = True


class UNet(_EngineModule):
    """U-Net generator of the reference (dl_models/u_net.py:34-64).

    Constructor arguments keep the reference's names, order and defaults.  Additions (keyword only in practice): ``depth``
    (number of stride-2 levels, hard-coded to 4 in the reference), ``batch_size`` (the batch size whose engine is built at
    construction and driven by ``Trainer``; default 1), ``device``, ``n_replicas``, ``dropout``, ``dtype`` ("f32": the
    reference's arithmetic, the mode the fp32-tolerance parity tests run in; "bf16": BASELINE.json configs[1] - bf16
    activations / gradients / weight work copies, fp32 accumulation, statistics and master weights; mode 0 only),
    ``overlap`` (weight gradients and the information-vector branch on a side HIP stream, bucket-wise Adam on a third),
    ``fold_l2``.  ``input_shape`` may be the reference's (H, W, 2) or (2, H, W).
    ``forward(spec[B,2,H,W] float32 NCHW, emb[B,2,16] int) -> [B,2,H,W]``.
    """

    def __init__(self, input_shape, inf_vector_shape, learning_rate=1e-5, mode=0, number_filters_0=32, kernels=6,
                 BatchNorm=True, resize_factor_0=None, res_factor=None, name="U-Net", depth=4, batch_size=None,
                 device="cuda:0", n_replicas=1, dropout=True, dtype="f32", overlap=False, fold_l2=True, runtime=None):
        super().__init__(device, batch_size, n_replicas, dropout, fold_l2, runtime)
        # unlike the reference (dl_models/u_net.py:46-49) explicit factors are honoured, not dropped
        self.res_factor = [2, 2] if res_factor is None else list(res_factor)
        self.resize_factor_0 = [1, 1] if resize_factor_0 is None else list(resize_factor_0)
        if mode not in (0, 1, 2, 3):
            raise ValueError("mode must be 0..3 (dl_models/u_net.py:280-287)")
        if self.res_factor != [2, 2] or self.resize_factor_0 != [1, 1]:
            raise NotImplementedError("only res_factor=[2,2], resize_factor_0=[1,1] are implemented")
        shp = tuple(input_shape)
        if len(shp) != 3 or 2 not in (shp[0], shp[2]):
            raise ValueError("input_shape must be (H, W, 2) or (2, H, W)")
        self.H, self.W = (shp[0], shp[1]) if shp[2] == 2 else (shp[1], shp[2])
        self.input_shape = shp
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.learning_rate = learning_rate
        self.mode = mode
        self.number_filters_0 = number_filters_0
        self.kernels = kernels
        self.BatchNorm = BatchNorm
        self.name = name
        self.depth = depth
        self.dtype_name, self.overlap = dtype, overlap
        self._finish_init()

    def compile_and_fit(self, x_train1, x_train2, y_train, x_val1, x_val2, y_val, batch_size, num_epochs, steps_per_epoch):
        """dl_models/u_net.py:83-118 (signature as there: the rate is the constructor's `learning_rate`)."""
        return self._compile_and_fit(x_train1, x_train2, y_train, x_val1, x_val2, y_val, batch_size, num_epochs, steps_per_epoch,
                                     self.learning_rate)

    def _new_engine(self, B, share):
        if self.mode == 0:
            return UNetEngine(self.H, self.W, B, F0=self.number_filters_0, k=self.kernels, depth=self.depth,
                              batchnorm=self.BatchNorm, inf_vector_shape=self.inf_vector_shape, device=self._device,
                              n_replicas=self.n_replicas, dtype=self.dtype_name, overlap_wgrad=self.overlap, runtime=self._rt,
                              share=share)
        return UNetGraphEngine(self.H, self.W, B, F0=self.number_filters_0, k=self.kernels, depth=self.depth, mode=self.mode,
                               batchnorm=self.BatchNorm, inf_vector_shape=self.inf_vector_shape, device=self._device,
                               n_replicas=self.n_replicas, runtime=self._rt, share=share, dtype=self.dtype_name,
                               overlap_wgrad=self.overlap)

    # `parameters.pkl` holds the reference's list (u_net.py:180-187) followed by the arguments it forgets (kernels, depth, and
    # this build's dtype) - `UNet.load` of the reference feeds BatchNorm into the `kernels` slot because of that omission.
    def _ctor_parameters(self):
        return [self.input_shape, self.inf_vector_shape, self.learning_rate, self.mode, self.number_filters_0,
                self.BatchNorm, self.kernels, self.depth, self.dtype_name]

    @classmethod
    def load(cls, save_folder=".", batch_size=1, device="cuda:0", **kw):
        with open(os.path.join(save_folder, "parameters.pkl"), "rb") as f:
            prm = pickle.load(f)
        input_shape, inf_vector_shape, lr, mode, f0, bn, kernels, depth = prm[:8]
        dtype = prm[8] if len(prm) > 8 else "f32"
        ue = cls(input_shape, inf_vector_shape, lr, mode, f0, kernels, bn, depth=depth, batch_size=batch_size, device=device,
                 dtype=dtype, **kw)
        ue.load_weights(os.path.join(save_folder, "weights.npz"))
        return ue


class _SubModel:
    """``model.encoder`` / ``model.decoder`` of the autoencoder family (dl_models/res_ae.py:62-64, main_training.py:258-259)."""

    def __init__(self, module, which):
        self._m, self._which = module, which

    def __call__(self, inputs, training=False):
        m = self._m
        m.train(training)
        with torch.no_grad():
            if self._which == "encoder":
                spec, emb = inputs
                eng = m._engine_for(spec.shape[0])
                eng.training = training
                eng.use_device_counters(False)
                mask = eng.make_dropout_mask() if (training and m.dropout) else None
                return eng.encode(spec.permute(0, 3, 1, 2).to(m._device, torch.float32).contiguous(), emb.to(m._device), mask)
            z = inputs
            eng = m._engine_for(z.shape[0])
            eng.training = training
            eng.use_device_counters(False)
            mask = eng.make_dropout_mask() if (training and m.dropout) else None
            return eng.decode(z.to(m._device, torch.float32).contiguous(), mask).permute(0, 2, 3, 1)

    predict = __call__


class _AEFamily(_EngineModule):
    """Constructor surface shared by the reference's autoencoder classes (dl_models/res_ae.py:41-50, dl_models/autoencoder.py:41-46):
    (input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons, name)."""
    ENGINE = None

    def __init__(self, input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons,
                 name, batch_size=None, device="cuda:0", n_replicas=1, dropout=True, fold_l2=True, runtime=None, dtype="f32",
                 overlap=False):
        super().__init__(device, batch_size, n_replicas, dropout, fold_l2, runtime)
        self.dtype_name, self.overlap = dtype, overlap
        shp = tuple(input_shape)
        if len(shp) != 3 or 2 not in (shp[0], shp[2]):
            raise ValueError("input_shape must be (H, W, 2) or (2, H, W)")
        self.H, self.W = (shp[0], shp[1]) if shp[2] == 2 else (shp[1], shp[2])
        self.input_shape = shp
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.conv_filters, self.conv_kernels, self.conv_strides = tuple(conv_filters), tuple(conv_kernels), tuple(conv_strides)
        self.latent_space_dim, self.n_neurons = latent_space_dim, n_neurons
        self.name = name
        self._finish_init()
        self._shape_before_bottleneck = self.engine.shape_before_bottleneck

    def _new_engine(self, B, share):
        return self.ENGINE(self.H, self.W, B, self.conv_filters, self.conv_kernels, self.conv_strides, self.latent_space_dim,
                           self.n_neurons, self.inf_vector_shape, device=self._device, n_replicas=self.n_replicas,
                           runtime=self._rt, share=share, dtype=self.dtype_name, overlap_wgrad=self.overlap)

    def compile_and_fit(self, x_train1, x_train2, y_train, x_val1, x_val2, y_val, batch_size, num_epochs, steps_per_epoch,
                        learning_rate=0.00001):
        """dl_models/res_ae.py:89-127, dl_models/autoencoder.py:83-121 (signature as there)."""
        return self._compile_and_fit(x_train1, x_train2, y_train, x_val1, x_val2, y_val, batch_size, num_epochs, steps_per_epoch,
                                     learning_rate)

    @property
    def encoder(self):
        return _SubModel(self, "encoder")

    @property
    def decoder(self):
        return _SubModel(self, "decoder")

    def _ctor_parameters(self):            # dl_models/res_ae.py:193-201, dl_models/autoencoder.py:177-186
        return [self.input_shape, self.inf_vector_shape, self.conv_filters, self.conv_kernels, self.conv_strides,
                self.latent_space_dim, self.n_neurons]

    @classmethod
    def load(cls, save_folder=".", batch_size=1, device="cuda:0", **kw):
        with open(os.path.join(save_folder, "parameters.pkl"), "rb") as f:
            prm = pickle.load(f)
        ae = cls(*prm, batch_size=batch_size, device=device, **kw)
        ae.load_weights(os.path.join(save_folder, "weights.npz"))
        return ae


class Autoencoder(_AEFamily):
    """Convolutional autoencoder of the reference (dl_models/autoencoder.py:34-62; main_training.py:118-129 builds it with
    filters (64,128,256,512), kernels 3, strides 2, latent 64, n_neurons 2048).  Constructor arguments keep the reference's
    names and order; ``batch_size``, ``device``, ``n_replicas``, ``dropout``, ``fold_l2``, ``dtype``, ``overlap`` are additions."""
    ENGINE = AutoencoderEngine

    def __init__(self, input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons,
                 name="Autoencoder", **kw):
        super().__init__(input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons,
                         name, **kw)


class ResAE(_AEFamily):
    """Residual autoencoder of the reference (dl_models/res_ae.py:35-70; main_training.py:130-140 builds it with filters
    (32,64,128,256), kernels 3, strides 2, latent 32, n_neurons 1024 - BASELINE.json configs[4]).  Constructor arguments keep
    the reference's names and order; ``batch_size``, ``device``, ``n_replicas``, ``dropout``, ``fold_l2``, ``dtype`` ("f32" /
    "bf16" storage of the convolutional trunk), ``overlap`` (weight gradients on a side stream, bucket-wise Adam on a third) are additions."""
    ENGINE = ResAEEngine

    def __init__(self, input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons,
                 name="ResAE", **kw):
        super().__init__(input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons,
                         name, **kw)

"""Drop-in boundary: ``UNet`` keeps the constructor signature of the reference class
(dl_models/u_net.py:40-45) and the ``model.model([spec_in, emb], training=...)`` call shape used by
main_training.py:261 / trainer.py:137, as a ``torch.nn.Module`` whose forward and backward run on the
HIP engine.  There is no CPU or eager fallback: without the HIP library / a GPU this raises."""
import os
import pickle

import numpy as np
import torch
from torch import nn

from .engine import UNetEngine


class _UNetFunction(torch.autograd.Function):
    """Whole-network autograd node: forward = engine.forward, backward = engine.backward.  Lets a
    main_training.py-style loop (loss on the prediction, any torch optimizer) drive the HIP path."""

    @staticmethod
    def forward(ctx, module, spec, emb, dropout_mask, *params):
        eng = module.engine
        pred = eng.forward(spec, emb, dropout_mask=dropout_mask)
        ctx.module = module
        return pred.clone()

    @staticmethod
    def backward(ctx, dpred):
        module = ctx.module
        eng = module.engine
        eng.backward(dpred=dpred.contiguous(), include_reg=False)   # the l2 terms reach autograd via regularization_losses()
        grads = tuple(eng.g[n].clone() for n in module._param_names)
        return (None, None, None, None) + grads


class _KerasModelAdapter:
    """``model.model`` of the reference: callable as ``model.model([spec_in, emb], training=bool)`` with NHWC
    spectrograms (main_training.py:261), plus ``trainable_variables`` and ``losses`` (main_training.py:263-268)."""

    def __init__(self, module):
        self._m = module

    def __call__(self, inputs, training=False):
        spec, emb = inputs
        self._m.train(training)
        out = self._m(spec.permute(0, 3, 1, 2).contiguous(), emb)
        return out.permute(0, 2, 3, 1)

    @property
    def trainable_variables(self):
        return list(self._m.parameters())

    @property
    def losses(self):
        return self._m.regularization_losses()


class UNet(nn.Module):
    """U-Net generator of the reference (dl_models/u_net.py:34-64).

    Constructor arguments keep the reference's names, order and defaults; ``depth`` (number of stride-2
    levels, hard-coded to 4 in the reference) and ``batch_size`` / ``device`` are additions.  ``input_shape`` may be
    the reference's (H, W, 2) or (2, H, W).  ``forward(spec[B,2,H,W] float32 NCHW, emb[B,2,16] int) -> [B,2,H,W]``.

    Only the configurations the reference's live driver uses run on HIP kernels: mode=0,
    resize_factor_0=[1,1], res_factor=[2,2]; anything else raises NotImplementedError.
    """

    def __init__(self, input_shape, inf_vector_shape, learning_rate=1e-5, mode=0, number_filters_0=32, kernels=6,
                 BatchNorm=True, resize_factor_0=None, res_factor=None, name="U-Net", depth=4, batch_size=None,
                 device="cuda:0", n_replicas=1, dropout=True):
        super().__init__()
        # unlike the reference (dl_models/u_net.py:46-49) explicit factors are honoured, not dropped
        self.res_factor = [2, 2] if res_factor is None else list(res_factor)
        self.resize_factor_0 = [1, 1] if resize_factor_0 is None else list(resize_factor_0)
        if mode != 0:
            raise NotImplementedError("the nn.Module wrapper drives the hand-scheduled mode-0 engine; modes 1-3 "
                                      "(dl_models/u_net.py:324-386) run on unet_rir_amd.UNetGraphEngine")
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
        self.dropout = dropout
        self.n_replicas = n_replicas
        self._device = torch.device(device)
        if self._device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("unet-rir_amd needs an AMD GPU (HIP); there is no CPU fallback")
        self.engine = None
        self._param_names = []
        self._params = nn.ParameterList()
        if batch_size is not None:
            self._build(batch_size)

    # the engine is built for a fixed per-replica batch size (buffers are allocated once)
    def _build(self, batch_size):
        old = self.engine
        eng = UNetEngine(self.H, self.W, batch_size, F0=self.number_filters_0, k=self.kernels, depth=self.depth,
                         batchnorm=self.BatchNorm, inf_vector_shape=self.inf_vector_shape, device=self._device,
                         n_replicas=self.n_replicas)
        if old is None:
            eng.reset_parameters()
        else:
            eng.theta.copy_(old.theta); eng.adam_m.copy_(old.adam_m); eng.adam_v.copy_(old.adam_v)
            eng.adam_t = old.adam_t
            for n in eng.moving:
                eng.moving[n].copy_(old.moving[n])
        self.engine = eng
        self._param_names = list(eng.specs)
        self._params = nn.ParameterList([nn.Parameter(eng.p[n]) for n in self._param_names])
        for prm, n in zip(self._params, self._param_names):
            prm.grad = None
        for n, b in eng.moving.items():
            self.register_buffer(n.replace(".", "_"), b, persistent=True)

    def named_engine_parameters(self):
        return dict(zip(self._param_names, self._params))

    @property
    def model(self):
        return _KerasModelAdapter(self)

    def summary(self):
        eng = self.engine
        print(f'Model: "{self.name}"  input [B,2,{self.H},{self.W}] + [B,{self.inf_vector_shape}] -> [B,2,{self.H},{self.W}]')
        if eng is not None:
            for n, s_ in eng.specs.items():
                print(f"  {n:28s} {str(s_.keras_shape):24s}")
            print(f"Total params: {eng.n_params():,}")

    # ---- persistence, dl_models/u_net.py:120-199.  `parameters.pkl` holds the reference's list (u_net.py:180-187) followed by
    # the arguments it forgets (kernels, depth) - `UNet.load` of the reference feeds BatchNorm into the `kernels` slot because
    # of that omission; `weights.npz` replaces `weights.h5` (h5py is not part of this image) and holds every variable in its
    # Keras layout (HWIO Conv2D, HWOI Conv2DTranspose, [in, out] Dense) plus the BatchNorm moving statistics.
    def save(self, save_folder="."):
        if self.engine is None:
            raise RuntimeError("the model has no variables yet: build it with batch_size= or run one forward pass")
        os.makedirs(save_folder, exist_ok=True)
        parameters = [self.input_shape, self.inf_vector_shape, self.learning_rate, self.mode, self.number_filters_0,
                      self.BatchNorm, self.kernels, self.depth]
        with open(os.path.join(save_folder, "parameters.pkl"), "wb") as f:
            pickle.dump(parameters, f)
        arrays = {n: np.asarray(v) for n, v in self.engine.export_keras_params().items()}
        arrays.update({"moving/" + n: b.detach().cpu().numpy() for n, b in self.engine.moving.items()})
        np.savez(os.path.join(save_folder, "weights.npz"), **arrays)

    def load_weights(self, weights_path):
        if self.engine is None:
            raise RuntimeError("build the model (batch_size=) before loading weights")
        with np.load(weights_path) as z:
            self.engine.load_keras_params({n: z[n] for n in self.engine.specs})
            for n, b in self.engine.moving.items():
                b.copy_(torch.from_numpy(z["moving/" + n]).to(b.device))

    @classmethod
    def load(cls, save_folder=".", batch_size=1, device="cuda:0"):
        with open(os.path.join(save_folder, "parameters.pkl"), "rb") as f:
            input_shape, inf_vector_shape, lr, mode, f0, bn, kernels, depth = pickle.load(f)
        ue = cls(input_shape, inf_vector_shape, lr, mode, f0, kernels, bn, depth=depth, batch_size=batch_size, device=device)
        ue.load_weights(os.path.join(save_folder, "weights.npz"))
        return ue

    def predict_stft(self, inputs):
        """dl_models/u_net.py:138-146: model.predict([spectrograms NHWC, vectors]) -> generated spectrograms NHWC."""
        with torch.no_grad():
            return self.model(inputs, training=False)

    def regularization_losses(self):
        """model.model.losses: one l2(0.001) term per strided Conv2D / Conv2DTranspose kernel."""
        named = self.named_engine_parameters()
        return [1e-3 * (named[n] ** 2).sum() for n in self.engine.l2_names]

    def forward(self, spec, emb, dropout_mask=None):
        if spec.dim() != 4 or spec.shape[1] != 2:
            raise ValueError("spec must be NCHW [B,2,H,W]")
        if self.engine is None or self.engine.B != spec.shape[0]:
            self._build(spec.shape[0])
        eng = self.engine
        eng.training = self.training
        if dropout_mask is None and self.training and self.dropout:
            dropout_mask = eng.make_dropout_mask()
        spec = spec.to(self._device, torch.float32).contiguous()
        emb = emb.to(self._device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._params):
            return _UNetFunction.apply(self, spec, emb, dropout_mask, *self._params)
        return eng.forward(spec, emb, dropout_mask=dropout_mask).clone()

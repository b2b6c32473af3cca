"""Generates tests/golden/*.npz from the CPU oracle (oracle/torch_ref.py, fp64).

PARITY UNPINNED: the TensorFlow/Keras reference cannot run here and ships no fixtures, so these vectors pin the
*oracle* (and through it the HIP path) against regressions; they are not outputs of the reference itself.
Inputs and parameters are regenerated from oracle/detrand.py (platform independent), so only outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import detrand, torch_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (H, W, F0, B, dropout, global_batch, n_replicas)
    "tiny_32x48_f4": (32, 48, 4, 2, False, 2, 1),
    "cfg1_shaped_64x64_f16": (64, 64, 16, 2, True, 4, 2),     # BASELINE configs[0] model at reduced size, 2-replica loss scaling
}
LR = 1e-3


def case_inputs(name):
    H, W, F0, B, dropout, gb, nrep = CASES[name]
    cfg = R.Config(H, W, F0, 3)
    params = R.init_params(cfg, seed_name=f"golden/{name}/p", randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B, seed_name=f"golden/{name}/d")
    mask = None
    if dropout:
        h5, w5 = cfg.bottleneck_hw()
        keep = detrand.uniform(f"golden/{name}/mask", (B, h5 * w5 * 16)) >= 0.3
        mask = keep.astype(np.float64) / 0.7
    return cfg, params, spec_in, emb, spec_out, mask, gb, nrep


def compute(name, dtype=torch.float64):
    cfg, params, spec_in, emb, spec_out, mask, gb, nrep = case_inputs(name)
    loss, dl, pred, grads = R.loss_and_grads(params, spec_in, emb, spec_out, cfg, 0.9, gb, nrep, mask, dtype)
    out = {"loss": np.float64(loss), "data_loss": np.float64(dl), "pred": pred.numpy().astype(np.float32)}
    for n, g in grads.items():
        g = g.double()
        out[f"gnorm/{n}"] = np.float64(g.norm())
        out[f"ghead/{n}"] = g.flatten()[:8].numpy()
        th = torch.tensor(params[n]).double()
        new, _, _ = R.adam_update(th, g, torch.zeros_like(th), torch.zeros_like(th), 1, LR)
        out[f"psum/{n}"] = np.float64(new.sum())
    return out


if __name__ == "__main__":
    for name in CASES:
        out = compute(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes; loss", out["loss"])

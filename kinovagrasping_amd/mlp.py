"""Fused 3-layer MLP forward (kr_mlp3_forward, csrc/ks_mlp.hip): the reference's Actor / Critic forward passes
(DDPGfD.py:29-32, 47-50) as one fp32-MFMA launch each instead of three library GEMMs + bias / activation kernels."""
from __future__ import annotations

import ctypes

import torch

from . import sim as _sim

ACT_NONE, ACT_SIGMOID = 0, 1
SUPPORTED_TILES = {(16, 16), (25, 19), (8, 8), (4, 4)}      # ceil(hidden / 16) of the two hidden layers (ks_mlp.hip)
SHADOW_TILES = {(16, 16), (8, 8), (4, 4)}                   # ... of the LDS-free variant (kr_mlp3_forward_shadow)


def layers_of(module):
    """[(weight, bias)] x 3 of an Actor / Critic module (l1, l2, l3), as the tensors the kernel reads"""
    return [(getattr(module, k).weight.data, getattr(module, k).bias.data) for k in ("l1", "l2", "l3")]


def supported(layers, in_dim: int, shadow: bool = False) -> bool:
    (w1, _), (w2, _), (w3, _) = layers
    tiles = ((w1.shape[0] + 15) // 16, (w2.shape[0] + 15) // 16)
    return (w1.is_cuda and w1.dtype == torch.float32 and tiles in (SHADOW_TILES if shadow else SUPPORTED_TILES) and in_dim <= 96 and w3.shape[0] <= 4
            and all(w.is_contiguous() and b.is_contiguous() for w, b in layers))


def mlp3_forward(layers, xa: torch.Tensor, xb: torch.Tensor | None = None, act: int = ACT_NONE, scale: float = 1.0,
                 out: torch.Tensor | None = None, h1_out: torch.Tensor | None = None, h2_out: torch.Tensor | None = None,
                 shadow: bool = False) -> torch.Tensor:
    """out[n, out_dim] = f(W3 relu(W2 relu(W1 [xa | xb] + b1) + b2) + b3) on the current stream.  xa / xb: fp32 [n, *]
    with unit column stride (row stride free: slices of wider tensors are fine).  h1_out [n, h1] / h2_out [n, h2]
    (contiguous, optional) receive the hidden activations for a backward pass.  shadow=True: the LDS-free kernel whose
    waves fit beside the resident simulator kernel (include/kinova_rollout.h: kr_mlp3_forward_shadow)."""
    (w1, b1), (w2, b2), (w3, b3) = layers
    n, in_a = xa.shape
    in_b = 0 if xb is None else xb.shape[1]
    assert w1.shape[1] == in_a + in_b and w2.shape[1] == w1.shape[0] and w3.shape[1] == w2.shape[0]
    assert xa.dtype == torch.float32 and xa.stride(1) == 1 and (xb is None or (xb.dtype == torch.float32 and xb.stride(1) == 1 and xb.shape[0] == n))
    if out is None:
        out = torch.empty(n, w3.shape[0], device=xa.device, dtype=torch.float32)
    assert out.is_contiguous() and tuple(out.shape) == (n, w3.shape[0])
    for h, w in ((h1_out, w1), (h2_out, w2)):
        assert h is None or (h.is_contiguous() and tuple(h.shape) == (n, w.shape[0]) and h.dtype == torch.float32)
    lib, P = _sim.load_library(), _sim._ptr
    rc = (lib.kr_mlp3_forward_shadow if shadow else lib.kr_mlp3_forward)(n, in_a, in_b, w1.shape[0], w2.shape[0], w3.shape[0], P(xa), xa.stride(0), P(xb) if xb is not None else None,
                             xb.stride(0) if xb is not None else 0, P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), act, float(scale), P(out), P(h1_out), P(h2_out),
                             ctypes.c_void_p(torch.cuda.current_stream(xa.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"kr_mlp3_forward failed ({rc}): unsupported layer widths or bad arguments")
    return out

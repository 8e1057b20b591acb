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


def _split_waves(n: int) -> int:
    """waves per workgroup of the split LDS-free forward: 4 while the launch still fits the GPU in one round beside the
    stepping kernel (one learner wave per SIMD: 1024 of them), 2 for larger batches; KS_MLP_SPLIT=0 / 2 / 4 forces."""
    import os
    forced = os.environ.get("KS_MLP_SPLIT")
    if forced is not None:
        return int(forced)
    return 4 if (n + 15) // 16 * 4 <= 1024 else 2


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
    waves = _split_waves(n) if shadow and w1.shape[0] % 16 == 0 and w2.shape[0] % 16 == 0 and w1.shape[0] // 16 in (4, 8, 16) and w1.shape[0] == w2.shape[0] else 0
    if waves:
        # the LDS-free launch with each layer's tiles split over 2 / 4 waves of a workgroup (kr_mlp3_forward_split)
        blocks = (n + 15) // 16
        need = (0 if h1_out is not None else blocks * 16 * w1.shape[0]) + blocks * waves * 64
        scratch = torch.empty(need, device=xa.device, dtype=torch.float32)
        rc = lib.kr_mlp3_forward_split(n, in_a, in_b, w1.shape[0], w2.shape[0], w3.shape[0], P(xa), xa.stride(0), P(xb) if xb is not None else None,
                                       xb.stride(0) if xb is not None else 0, P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), act, float(scale), P(out),
                                       P(h1_out), P(h2_out), P(scratch), need, waves, ctypes.c_void_p(torch.cuda.current_stream(xa.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"kr_mlp3_forward_split failed ({rc})")
        return out
    rc = (lib.kr_mlp3_forward_shadow if shadow else lib.kr_mlp3_forward)(n, in_a, in_b, w1.shape[0], w2.shape[0], w3.shape[0], P(xa), xa.stride(0), P(xb) if xb is not None else None,
                             xb.stride(0) if xb is not None else 0, P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), act, float(scale), P(out), P(h1_out), P(h2_out),
                             ctypes.c_void_p(torch.cuda.current_stream(xa.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"kr_mlp3_forward failed ({rc}): unsupported layer widths or bad arguments")
    return out


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def mlp3_backward(layers, dz3: torch.Tensor, h1: torch.Tensor, h2: torch.Tensor, want_dz: bool = True, dx_cols: tuple[int, int] | None = None,
                  act_out: torch.Tensor | None = None, scale: float = 1.0):
    """Data gradients of the 3-layer MLP without LDS (kr_mlp3_backward_shadow).  dz3 [n, out_dim] = dLoss/d(last
    pre-activation); h1 / h2 the activations of the forward pass.  Returns (dz2, dz1, dx): dz2 / dz1 None unless want_dz;
    dx [n, ncol] = dLoss/d(input columns col0..col0+ncol) for dx_cols = (col0, ncol), None otherwise - multiplied by the
    derivative of scale * sigmoid when act_out (that sigmoid's output) is given."""
    (w1, _), (w2, _), (w3, _) = layers
    n = dz3.shape[0]
    assert dz3.is_contiguous() and h1.is_contiguous() and h2.is_contiguous() and tuple(h1.shape) == (n, w1.shape[0]) and tuple(h2.shape) == (n, w2.shape[0])
    dz2 = torch.empty_like(h2) if want_dz else None
    dz1 = torch.empty_like(h1) if want_dz else None
    dx, col0, ncol = None, 0, 0
    if dx_cols is not None:
        col0, ncol = dx_cols
        dx = torch.empty(n, ncol, device=dz3.device, dtype=torch.float32)
        assert act_out is None or (act_out.is_contiguous() and tuple(act_out.shape) == (n, ncol))
    lib, P = _sim.load_library(), _sim._ptr
    waves = _split_waves(n) if w1.shape[0] == w2.shape[0] and w1.shape[0] % 16 == 0 and w1.shape[0] // 16 in (4, 8, 16) else 0
    if waves:
        need = (n + 15) // 16 * waves * 64 if dx is not None else 0
        scratch = torch.empty(max(need, 4), device=dz3.device, dtype=torch.float32)
        rc = lib.kr_mlp3_backward_split(n, w1.shape[1], w1.shape[0], w2.shape[0], w3.shape[0], P(dz3), P(w3), P(h2), P(w2), P(h1), P(dz2), P(dz1), P(w1),
                                        col0, ncol, P(act_out), float(scale), P(dx), P(scratch), need, waves, _stream(dz3))
    else:
        rc = lib.kr_mlp3_backward_shadow(n, w1.shape[1], w1.shape[0], w2.shape[0], w3.shape[0], P(dz3), P(w3), P(h2), P(w2), P(h1), P(dz2), P(dz1), P(w1),
                                         col0, ncol, P(act_out), float(scale), P(dx), _stream(dz3))
    if rc != 0:
        raise RuntimeError(f"kr_mlp3_backward failed ({rc})")
    return dz2, dz1, dx


def weight_grad(dz: torch.Tensor, ha: torch.Tensor, hb: torch.Tensor | None, dW: torch.Tensor, db: torch.Tensor, rows_per_chunk: int = 400):
    """dW [M, N] = dz^T [ha | hb], db [M] = dz.sum(0) without LDS (kr_weight_grad_shadow), written in place."""
    n, M = dz.shape
    Na, Nb = ha.shape[1], (0 if hb is None else hb.shape[1])
    assert dz.is_contiguous() and ha.stride(1) == 1 and (hb is None or hb.stride(1) == 1) and dW.is_contiguous() and tuple(dW.shape) == (M, Na + Nb)
    assert db.is_contiguous() and db.numel() == M
    chunks = max(1, (n + rows_per_chunk - 1) // rows_per_chunk)
    ws = torch.empty(chunks * (M * (Na + Nb) + M), device=dz.device, dtype=torch.float32)
    lib, P = _sim.load_library(), _sim._ptr
    rc = lib.kr_weight_grad_shadow(n, M, Na, Nb, P(dz), P(ha), ha.stride(0), P(hb), 0 if hb is None else hb.stride(0), chunks, P(ws), P(dW), P(db), _stream(dz))
    if rc != 0:
        raise RuntimeError(f"kr_weight_grad_shadow failed ({rc})")

"""CPU emulation of the `f16f8` operand split of the local-MI backward (csrc/mi_local_bwd_f8.hip): every product g * s as
f16(g) f16(s) + e4m3(2^7 gh) e4m3(2^20 sl) 2^-27 + e4m3(2^19 gl) e4m3(2^8 sh) 2^-27 with exact accumulation, against the fp64 oracle
(ref contrastyou/losses/iic_loss.py:107-149 and its gradient).  This is the arithmetic argument for the kernel, checked before it
was written: the error the split itself introduces, apart from any kernel, stays a factor 3 inside the gradient bound the GPU
tests apply (1e-4 of the gradient's scale) and inside the 1e-5-relative loss bound on peaked inputs."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import synth  # noqa: E402
from oracle import iic as OI  # noqa: E402

F8 = getattr(torch, "float8_e4m3fn", None)
pytestmark = pytest.mark.skipif(F8 is None, reason="torch without float8_e4m3fn")


def q_f16(v):
    return v.float().to(torch.float16).double()


def q_f8(v, scale):
    s = (v.double() * scale).float()
    assert float(s.abs().max()) <= 448.0           # e4m3 saturates into NaN above: the kernel's operand scaling keeps below 256
    return s.to(F8).float().double() / scale


def bilinear_grads(terms, x, y, p):
    """sum over (G, y_for_gx, x_for_gy) of d/dx, d/dy of <G, joint(x, y)> with the OTHER operand quantised as given."""
    gx, gy = torch.zeros_like(x), torch.zeros_like(y)
    for G, yq, xq in terms:
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        gx += torch.autograd.grad((G * OI.local_joint_raw(xr, yq, p)).sum(), xr)[0]
        gy += torch.autograd.grad((G * OI.local_joint_raw(xq, yr, p)).sum(), yr)[0]
    return gx, gy


def split_backward(G, x, y, p):
    s = 2.0 ** -np.floor(np.log2(float(G.abs().max())))      # the kernel's per-(sub-head, window) power-of-two pre-scale of G
    Gs = G * s
    Gh = q_f16(Gs)
    Gl = Gs - Gh
    xh, yh = q_f16(x), q_f16(y)
    xl, yl = x - xh, y - yh
    gx, gy = bilinear_grads([(Gh, yh, xh), (q_f8(Gh, 128.0), q_f8(yl, 2.0 ** 20), q_f8(xl, 2.0 ** 20)),
                             (q_f8(Gl, 2.0 ** 19), q_f8(y, 256.0), q_f8(x, 256.0))], x, y, p)
    return gx / s, gy / s


CASES = [("softmax", (4, 20, 32, 32), 3), ("softmax", (2, 20, 37, 45), 3), ("peaked", (3, 5, 12, 10), 2), ("peaked", (4, 20, 32, 32), 3)]


@pytest.mark.parametrize("kind,shape,p", CASES)
def test_f16_hi_plus_fp8_cross_terms_hold_the_gradient_bound(kind, shape, p):
    n, k, h, w = shape
    if kind == "peaked":
        xs, ys = synth.peaked_pair(f"lpeak_f32_n{n}_k{k}_h{h}_w{w}_p{p}", shape)
        x32, y32 = torch.from_numpy(xs), torch.from_numpy(ys)
    else:
        gen = torch.Generator().manual_seed(n * 100 + h + p)
        x32, y32 = torch.randn(shape, generator=gen).softmax(1), torch.randn(shape, generator=gen).softmax(1)
    x64, y64 = x32.double().requires_grad_(True), y32.double().requires_grad_(True)
    raw = OI.local_joint_raw(x64, y64, p)
    truth = OI.local_mi_from_raw(raw)
    gx64, gy64 = torch.autograd.grad(truth, [x64, y64])
    G = OI.local_mi_grad_wrt_raw(raw.detach())
    gx, gy = split_backward(G, x32.double(), y32.double(), p)
    scale = float(gx64.abs().max())
    err = max(float((gx - gx64).abs().max()), float((gy - gy64).abs().max())) / scale
    assert err < 3.5e-5, err                        # measured 5e-6 .. 3.1e-5; the GPU tests' bound is 1e-4 of the scale

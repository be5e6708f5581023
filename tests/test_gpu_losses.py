"""GPU parity: supervised KL, UDA MSE (with fused flip), flip, argmax/Dice counts, Adam -- HIP vs oracle/golden."""
import numpy as np
import pytest
import torch

import synth
from oracle import losses as OL

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda"


def ops():
    from miseg_amd import ops as _ops
    return _ops


def test_softmax_kl_vs_golden(golden):
    g = golden("losses")
    logits = T(synth.normal("kl/logits", (3, 4, 16, 16))).to(DEV).requires_grad_(True)
    target = T(synth.integers("kl/target", (3, 16, 16), 4)).to(DEV)
    loss = ops().softmax_kl(logits, target)
    np.testing.assert_allclose(float(loss), float(g["kl/loss"]), rtol=1e-5)      # north-star tolerance
    loss.backward()
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g["kl/glogits"], rtol=1e-4, atol=1e-9)


def test_softmax_mse_vs_golden_and_flip(golden):
    g = golden("losses")
    a = T(synth.normal("mse/a", (3, 4, 16, 16))).to(DEV).requires_grad_(True)
    b = T(synth.normal("mse/b", (3, 4, 16, 16))).to(DEV)
    loss = ops().softmax_mse(a, b)
    np.testing.assert_allclose(float(loss), float(g["mse/loss"]), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g["mse/ga"], rtol=1e-4, atol=1e-10)
    # fused flip of the detached branch == materialised flip (oracle)
    dec = OL.flip_decisions(123, 3)
    a2 = a.detach().clone().requires_grad_(True)
    fused = ops().softmax_mse(a2, b, ops().flips_to_tensor(dec, DEV))
    a_ref = a.detach().cpu().clone().requires_grad_(True)
    ref = OL.softmax_mse(a_ref, OL.apply_flips(b.cpu(), dec))
    np.testing.assert_allclose(float(fused), float(ref), rtol=1e-5)
    fused.backward(), ref.backward()
    np.testing.assert_allclose(a2.grad.cpu().numpy(), a_ref.grad.numpy(), rtol=1e-4, atol=1e-10)


@pytest.mark.parametrize("seed", [0, 123, 9999999, 4242])
def test_flip_bit_exact(golden, seed):
    g = golden("losses")
    x = torch.arange(4 * 2 * 3 * 5, dtype=torch.float32).view(4, 2, 3, 5)
    dec = OL.flip_decisions(seed, 4)
    np.testing.assert_array_equal(np.asarray(dec), g[f"flip/seed{seed}/decisions"])
    fl = ops().flips_to_tensor(dec, DEV)
    out = ops().flip(x.to(DEV), fl)
    np.testing.assert_array_equal(out.cpu().numpy(), g[f"flip/seed{seed}/out"])
    # channels_last strides, int64 and bf16 payloads
    xi = (x * 3).long().to(DEV)
    np.testing.assert_array_equal(ops().flip(xi, fl).cpu().numpy(), OL.apply_flips(xi.cpu(), dec).numpy())
    xb = x.to(DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    np.testing.assert_array_equal(ops().flip(xb, fl).float().cpu().numpy(), OL.apply_flips(xb.float().cpu(), dec).numpy())


def test_argmax_dice_bit_exact():
    logits = T(synth.normal("dice/logits", (5, 4, 24, 20))).to(DEV)
    labels = T(synth.integers("dice/labels", (5, 24, 20), 4)).to(DEV)
    pred, inter, uni = ops().argmax_dice(logits, labels)
    ref_pred = logits.cpu().max(1)[1]
    np.testing.assert_array_equal(pred.cpu().numpy(), ref_pred.numpy())
    ri, ru = OL.dice_counts(ref_pred, labels.cpu(), 4)
    np.testing.assert_array_equal(inter.cpu().numpy(), ri.numpy())
    np.testing.assert_array_equal(uni.cpu().numpy(), ru.numpy())


def test_adam_matches_oracle():
    from miseg_amd import unet_ops
    import math
    p0 = T(synth.normal("adam/p", (1000,)))
    ps, ms, vs = [p0.clone()], [torch.zeros(1000)], [torch.zeros(1000)]
    pd, md, vd = p0.to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    lr, wd, b1, b2, eps = 1e-3, 1e-5, 0.9, 0.999, 1e-8
    for t in range(1, 4):
        gr = T(synth.normal(f"adam/g{t}", (1000,)))
        OL.adam_step(ps, [gr], ms, vs, t, lr, weight_decay=wd)
        hyper = torch.tensor([lr / (1 - b1 ** t), 1 / math.sqrt(1 - b2 ** t), eps, wd], dtype=torch.float32, device=DEV)
        unet_ops.adam_step(pd, gr.to(DEV), md, vd, hyper, b1, b2)
    np.testing.assert_allclose(pd.cpu().numpy(), ps[0].numpy(), rtol=1e-6, atol=1e-7)


def test_softmax_kl_consistency_vs_golden_and_flip(golden):
    """`UDARegCriterion.name: kl` (ref semi_seg/trainer.py:137,194): KL_div()(softmax(a), softmax(flip(b)).detach()) fused, against
    the reference's own value / gradient (losses.npz klc/*) at 1e-5 relative, and with the flip replay against the oracle."""
    g = golden("losses")
    a = T(synth.normal("mse/a", (3, 4, 16, 16))).to(DEV).requires_grad_(True)
    b = T(synth.normal("mse/b", (3, 4, 16, 16))).to(DEV)
    loss = ops().softmax_kl_consistency(a, b, None)
    np.testing.assert_allclose(float(loss), float(g["klc/loss"]), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g["klc/ga"], rtol=1e-4, atol=1e-9)
    dec = [[True, False], [False, True], [True, True]]
    a2 = a.detach().clone().requires_grad_(True)
    fused = ops().softmax_kl_consistency(a2, b, ops().flips_to_tensor(dec, DEV))
    a_ref = a.detach().cpu().clone().requires_grad_(True)
    ref = OL.kl_div(a_ref.softmax(1), OL.apply_flips(b.cpu(), dec).softmax(1))
    np.testing.assert_allclose(float(fused), float(ref), rtol=1e-5)
    fused.backward(), ref.backward()
    np.testing.assert_allclose(a2.grad.cpu().numpy(), a_ref.grad.numpy(), rtol=1e-4, atol=1e-9)

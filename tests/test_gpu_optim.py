"""Fused multi-tensor Adam against torch.optim.Adam on the CPU -- the reference's optimizer
(GLfusion/main.py:162-165) -- on the same parameters and gradients."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(7,), (64, 3, 7, 7), (256,), (1,), (300, 1000), (2048, 256, 1, 1), (5, 13)]     # unaligned, tiny, > one chunk


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) * 0.3 for s in SHAPES]


@pytest.mark.parametrize("wd", [0.0, 1e-5, 0.1])
def test_adam_matches_torch_adam(wd):
    from glfusion_amd.optim import Adam
    cpu = [torch.nn.Parameter(p.clone()) for p in _params(0)]
    gpu = [torch.nn.Parameter(p.clone().to(DEV)) for p in _params(0)]
    ref = torch.optim.Adam(cpu, lr=3e-4, weight_decay=wd)           # main.py:162-165 (default betas / eps)
    opt = Adam(gpu, lr=3e-4, weight_decay=wd)
    sched_r = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=5)        # main.py:168
    sched_o = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=5)
    for step in range(4):
        grads = _params(100 + step)
        for i, (a, b, g) in enumerate(zip(cpu, gpu, grads)):
            if i == 3 and step < 2:                 # a parameter that gets its first gradient late: own step count
                a.grad = None
                b.grad = None
                continue
            a.grad = g.clone()
            b.grad = g.clone().to(DEV)
        ref.step()
        opt.step()
        sched_r.step()
        sched_o.step()
        for i, (a, b) in enumerate(zip(cpu, gpu)):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-8, err_msg=f"param {i} step {step}")
    for a, b in zip(cpu, gpu):
        sa, sb = ref.state[a], opt.state[b]
        assert int(sa["step"]) == int(sb["step"])
        # ATen's CPU kernels fuse a*b+c (vec::fmadd), the HIP kernel rounds every operation: differences of one ulp of
        # the LARGER operand where the two terms of the moving average cancel => absolute tolerance at operand scale
        np.testing.assert_allclose(sb["exp_avg"].cpu().numpy(), sa["exp_avg"].numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(sb["exp_avg_sq"].cpu().numpy(), sa["exp_avg_sq"].numpy(), rtol=2e-6, atol=1e-9)


def test_adam_state_dict_is_torch_compatible():
    from glfusion_amd.optim import Adam
    gpu = [torch.nn.Parameter(p.clone().to(DEV)) for p in _params(1)]
    opt = Adam(gpu, lr=1e-3, weight_decay=1e-5)
    for b, g in zip(gpu, _params(7)):
        b.grad = g.to(DEV)
    opt.step()
    sd = copy.deepcopy(opt.state_dict())
    twin = [torch.nn.Parameter(b.detach().cpu().clone()) for b in gpu]          # torch's Adam on the CPU
    t_opt = torch.optim.Adam(twin, lr=1e-3, weight_decay=1e-5)
    t_opt.load_state_dict(sd)                        # torch's own Adam accepts the state ...
    back = Adam(gpu, lr=1e-3, weight_decay=1e-5)
    back.load_state_dict(t_opt.state_dict())         # ... and ours accepts torch's
    for b, t, g in zip(gpu, twin, _params(8)):
        b.grad = g.to(DEV)
        t.grad = g.clone()
    back.step()
    t_opt.step()
    for b, t in zip(gpu, twin):
        np.testing.assert_allclose(b.detach().cpu().numpy(), t.detach().cpu().numpy(), rtol=2e-6, atol=1e-8)


def test_adam_refuses_what_it_does_not_build():
    from glfusion_amd.optim import Adam
    p = [torch.nn.Parameter(torch.zeros(4, device=DEV))]
    with pytest.raises(NotImplementedError):
        Adam(p, amsgrad=True)
    with pytest.raises(ValueError):
        Adam(p, lr=-1.0)
    cpu_p = [torch.nn.Parameter(torch.zeros(4))]
    o = Adam(cpu_p)
    cpu_p[0].grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        o.step()

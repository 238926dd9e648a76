"""Training mode of the attentive pools (BatchNorm with batch statistics,
backward through the CSR kernels) against one training step of the reference's
own modules: output, input gradients, parameter gradients and running
statistics.  float32 on both sides with different summation orders: compared
to 2e-4 of the largest magnitude of each tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def close(got, want, tol=2e-4):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    scale = max(np.abs(want).max(), 1e-6)
    return np.abs(got - want).max() <= tol * scale


@pytest.mark.parametrize("name", ["train_qkv", "train_qkv_groups4_modk", "train_group"])
def test_training_step_matches_reference(golden, name):
    from mvfusion import pooling
    g = golden(name)
    kw = {k[3:]: v.item() for k, v in g.items() if k.startswith("kw.")}
    cls = str(g["cls"])
    F_map, C = g["x_map"].shape[1], g["x_mod"].shape[1]
    mod = getattr(pooling, cls)(in_map=F_map, in_mod=C, out_mod=C, **kw)
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd.")}
    mod.load_state_dict(sd)
    mod = mod.cuda().train()
    xs = {k: T(g[k]).cuda().requires_grad_(True) for k in ("x_main", "x_mod", "x_map")}
    out = mod(xs["x_main"] if cls.startswith("QKV") else None, xs["x_mod"], xs["x_map"],
              T(g["ptr"]).cuda())
    (out * T(g["w"]).cuda()).sum().backward()
    assert close(out.detach().cpu().numpy(), g["x_pool"])
    for k in ("x_main", "x_mod", "x_map"):
        if "g_" + k in g:
            assert close(xs[k].grad.cpu().numpy(), g["g_" + k]), k
    n_checked = 0
    for k, p in mod.named_parameters():
        assert close(p.grad.cpu().numpy(), g["grad." + k]), k
        n_checked += 1
    assert n_checked >= 20
    for k, v in mod.state_dict().items():
        if "running_" in k:
            assert close(v.cpu().numpy(), g["after." + k], 1e-5), k
    # back in eval mode the fused inference kernels take over and agree with the
    # composed path on the same (updated) statistics
    mod.eval()
    with torch.no_grad():
        fused = mod(xs["x_main"] if cls.startswith("QKV") else None, xs["x_mod"], xs["x_map"],
                    T(g["ptr"]).cuda())
        from mvfusion import training
        fn = training.qkv_pool if cls.startswith("QKV") else training.group_pool
        composed, _ = fn(mod, xs["x_main"], xs["x_mod"], xs["x_map"], T(g["ptr"]).cuda())
    assert close(fused.cpu().numpy(), composed.cpu().numpy(), 1e-5)


def test_differentiable_csr_primitives():
    """gather_csr and segment_softmax_csr gradients against torch autograd of
    plain torch restatements on the CPU."""
    from mvfusion import pooling as P
    rng = np.random.default_rng(61)
    sizes = rng.poisson(4, 400)
    sizes[rng.random(400) < 0.2] = 0
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows, G = int(ptr[-1]), 4
    seg = np.repeat(np.arange(400), sizes)
    for scaling in (False, True):
        x = rng.normal(size=(rows, G)).astype(np.float32)
        w = rng.normal(size=(rows, G)).astype(np.float32)
        xg = T(x).cuda().requires_grad_(True)
        y = P.segment_softmax_csr(xg, T(ptr).cuda(), scaling=scaling)
        (y * T(w).cuda()).sum().backward()
        xc = T(x).double().requires_grad_(True)
        yc = torch.zeros_like(xc)
        for s in range(400):
            a, b = int(ptr[s]), int(ptr[s + 1])
            if b > a:
                z = xc[a:b] - xc[a:b].max(0).values
                if scaling:
                    z = z / np.sqrt(b - a)
                e = z.exp()
                yc[a:b] = e / (e.sum(0) + 1e-12)
        (yc * T(w).double()).sum().backward()
        assert close(y.detach().cpu().numpy(), yc.detach().numpy(), 1e-5)
        assert close(xg.grad.cpu().numpy(), xc.grad.numpy(), 1e-5)
    src = rng.normal(size=(400, 7)).astype(np.float32)
    w = rng.normal(size=(rows, 7)).astype(np.float32)
    sg = T(src).cuda().requires_grad_(True)
    (P.gather_csr(sg, T(ptr).cuda()) * T(w).cuda()).sum().backward()
    want = np.zeros_like(src, dtype=np.float64)
    np.add.at(want, seg, w.astype(np.float64))
    assert close(sg.grad.cpu().numpy(), want, 1e-5)

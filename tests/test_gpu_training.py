"""Training mode of the attentive pools (BatchNorm with batch statistics,
backward through the CSR kernels) against one training step of the reference's
own modules: output, input gradients, parameter gradients and running
statistics.  float32 on both sides with different summation orders: compared
to 1e-5 of the largest magnitude of each tensor (observed: at most 1.6e-6, on
weight gradients, which sum over every row)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def close(got, want, tol=1e-5):
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


def test_feature_map_gathers_are_differentiable():
    """Gradients of the feature-map gathers with respect to the image feature maps
    (what reaches the 2D encoder): x[feature_map_indexing], the fused gather +
    pixel -> view pool and the bilinear sparse_interpolation, against torch's own
    autograd through plain indexing."""
    from mvfusion import pooling
    from mvfusion.mapping import ImageMapping
    rng = np.random.default_rng(3)
    B, C, H, W, n_pts = 3, 7, 20, 24, 50
    # a few points, up to 3 views each, 1 to 4 pixels per view; pixels repeat across views
    point_ids, image_ids, pix = [], [], []
    for pt in range(n_pts):
        for im in rng.choice(B, size=rng.integers(0, 4), replace=False):
            for _ in range(rng.integers(1, 5)):
                point_ids.append(pt); image_ids.append(im)
                pix.append([rng.integers(0, 6), rng.integers(0, 5)])        # few pixels: many collisions
    m = ImageMapping.from_dense(
        T(np.asarray(point_ids)).cuda(), T(np.asarray(image_ids)).cuda(),
        T(np.asarray(pix, np.int16)).cuda(), T(rng.random((len(pix), 2)).astype(np.float32)).cuda(),
        num_points=n_pts)
    x0 = T(rng.normal(size=(B, C, H, W)).astype(np.float32)).cuda()
    idx = m.feature_map_indexing

    # plain gather
    w = T(rng.normal(size=(m.pixels.shape[0], C)).astype(np.float32)).cuda()
    xa = x0.clone().requires_grad_(True)
    (pooling.gather_features(xa, m) * w).sum().backward()
    xb = x0.clone().requires_grad_(True)
    (xb[idx] * w).sum().backward()
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)
    assert float(xa.grad.abs().max()) > 0

    # fused gather + atomic pool, every reduction
    vptr = m.atomic_csr_indexing
    wv = T(rng.normal(size=(vptr.shape[0] - 1, C)).astype(np.float32)).cuda()
    for mode in ("max", "min", "sum", "mean"):
        xa = x0.clone().requires_grad_(True)
        out = pooling.gather_atomic_pool(xa, m, mode)
        (out * wv).sum().backward()
        xb = x0.clone().requires_grad_(True)
        ref = pooling.segment_csr(xb[idx], vptr, reduce=mode)
        (ref * wv).sum().backward()
        torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5)

    # bilinear interpolation against a torch composition of the same four taps
    n = 300
    coords = T(rng.random((n, 2)).astype(np.float32)).cuda()
    batch = T(rng.integers(0, B, n)).cuda()
    wi = T(rng.normal(size=(n, C)).astype(np.float32)).cuda()
    xa = x0.clone().requires_grad_(True)
    got = pooling.sparse_interpolation(xa, coords, batch, padding_mode="border")
    (got * wi).sum().backward()
    xb = x0.clone().requires_grad_(True)
    xp = torch.nn.ReplicationPad2d(1)(xb)
    p0, p1 = coords[:, 0] * H + 0.5, coords[:, 1] * W + 0.5
    top, bottom, left, right = p0.floor(), (p0 + 1).floor(), p1.floor(), (p1 + 1).floor()
    taps = ((top, left, (p0 - bottom) * (p1 - right)), (top, right, (p0 - bottom) * (p1 - left)),
            (bottom, left, (p0 - top) * (p1 - right)), (bottom, right, (p0 - top) * (p1 - left)))
    ref = sum(xp[batch, :, r.long(), c.long()] * wt.abs().view(-1, 1) for r, c, wt in taps)
    (ref * wi).sum().backward()
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-4, atol=1e-5)


def test_heuristic_pool_and_eval_mode_calls_are_differentiable():
    """HeuristicBimodalCSRPool routes the gradient to the selected view rows; an
    attentive pool in eval mode with autograd on records a graph (BatchNorm on its
    running statistics) and agrees with the fused inference kernels."""
    from mvfusion import pooling
    rng = np.random.default_rng(4)
    counts = rng.integers(0, 5, 40)
    ptr = T(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).cuda()
    V, C = int(ptr[-1]), 6
    x_mod = T(rng.normal(size=(V, C)).astype(np.float32)).cuda().requires_grad_(True)
    x_map = T(rng.random((V, 8)).astype(np.float32)).cuda()
    pool = pooling.HeuristicBimodalCSRPool(mode="max", feat=2)
    out = pool(None, x_mod, x_map, ptr)
    w = T(rng.normal(size=out.shape).astype(np.float32)).cuda()
    (out * w).sum().backward()
    want = torch.zeros_like(x_mod)
    for s in range(40):
        a, b = int(ptr[s]), int(ptr[s + 1])
        if b > a:
            want[a + int(torch.argmax(x_map[a:b, 2]))] = w[s]
    assert torch.equal(x_mod.grad, want)

    torch.manual_seed(0)
    qkv = pooling.QKVBimodalCSRPool(in_main=5, in_map=8, in_mod=C, out_mod=C, nc_qk=4,
                                    gating=False).cuda().eval()
    x_main = T(rng.normal(size=(40, 5)).astype(np.float32)).cuda()
    xm = x_mod.detach().clone().requires_grad_(True)
    got = qkv(x_main, xm, x_map, ptr)
    assert got.requires_grad
    got.sum().backward()
    assert xm.grad is not None and float(xm.grad.abs().max()) > 0
    assert all(p.grad is not None for p in qkv.E_mod.parameters())
    with torch.no_grad():
        fused = qkv(x_main, xm, x_map, ptr)
    assert not fused.requires_grad
    torch.testing.assert_close(got.detach(), fused, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,C,train,affine,track,slope", [
    (5003, 64, True, True, True, 0.2), (5003, 64, False, True, True, 0.2), (777, 13, True, True, True, 0.2),
    (2, 8, True, True, True, 0.01), (4099, 1, True, True, True, 0.2), (300, 1040, True, True, True, 0.2),
    (301, 257, True, True, True, None), (1500, 32, True, False, True, 0.2), (1500, 32, True, True, False, 0.2),
    (1500, 32, False, True, False, 0.2), (100003, 16, True, True, True, 0.2)])
def test_batch_norm_act_kernels_match_the_modules(rows, C, train, affine, track, slope):
    """BatchNorm1d + LeakyReLU behind a dense layer's Linear on this library's kernels
    (training.batch_norm_act) against the torch modules evaluated in float64: output, the three
    gradients, running statistics and the batch counter; batch and running statistics, column
    counts off the 16-byte path and beyond one block's width, a mean far from zero (the variance
    is a difference of float64 sums; not in the cases of a few rows, where the float32 rounding of
    the saved mean alone exceeds the tolerance), no affine terms, no tracked statistics."""
    from mvfusion import training
    torch.manual_seed(rows + C)
    bn = torch.nn.BatchNorm1d(C, affine=affine, track_running_stats=track, momentum=0.1).cuda()
    if affine:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    if track:
        with torch.no_grad():
            bn.running_mean.uniform_(-1, 1)
            bn.running_var.uniform_(0.5, 2.0)
    ref = torch.nn.BatchNorm1d(C, affine=affine, track_running_stats=track, momentum=0.1).cuda().double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bn.train(train)
    ref.train(train)
    act = torch.nn.LeakyReLU(slope) if slope is not None else torch.nn.Identity()
    z = (torch.randn(rows, C, device="cuda") * torch.linspace(0.5, 3.0, C, device="cuda")
         + torch.linspace(-1.0, 1.0, C, device="cuda") * (40.0 if rows >= 500 else 1.0)).requires_grad_(True)
    wgt = torch.randn(rows, C, device="cuda")
    y = training.batch_norm_act(z, bn, act)
    (y * wgt).sum().backward()
    zd = z.detach().double().requires_grad_(True)
    yd = act(ref(zd))
    (yd * wgt.double()).sum().backward()
    assert y.dtype == torch.float32 and y.shape == (rows, C)
    assert close(y.detach().cpu().numpy(), yd.detach().cpu().numpy(), 2e-6)
    if rows >= 100:
        assert close(z.grad.cpu().numpy(), zd.grad.cpu().numpy(), 1e-5)
    else:
        # with a handful of rows the input gradient is what is left after the batch terms cancel
        # most of the incoming gradient: compared on the scale of the terms that cancel
        invstd = 1.0 / torch.sqrt(zd.detach().var(0, unbiased=False) + ref.eps)
        scale = float((wgt.abs().max(0).values.double() * invstd).max()) * (1.5 if affine else 1.0)
        assert float((z.grad.double() - zd.grad).abs().max()) <= 1e-5 * scale
    if affine:
        # (the float32 rounding of the saved mean shifts every normalised value of a column alike)
        assert close(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), 1e-5)
        assert close(bn.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), 2e-6)
    if track:
        assert close(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy(), 1e-6)
        assert close(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy(), 1e-6)
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == (1 if train else 0)
    # the same call again returns the same bits (fixed summation order)
    bn2 = torch.nn.BatchNorm1d(C, affine=affine, track_running_stats=track, momentum=0.1).cuda().train(train)
    bn2.load_state_dict(bn.state_dict())
    y1 = training.batch_norm_act(z.detach(), bn2, act)
    bn2.load_state_dict(bn.state_dict())
    y2 = training.batch_norm_act(z.detach(), bn2, act)
    assert torch.equal(y1, y2)


def test_batch_norm_act_single_row_in_train_mode_raises():
    from mvfusion import training
    bn = torch.nn.BatchNorm1d(8).cuda().train()
    with pytest.raises(ValueError):
        training.batch_norm_act(torch.zeros(1, 8, device="cuda"), bn, torch.nn.LeakyReLU(0.2))


@pytest.mark.parametrize("rows,A,B", [(100003, 64, 64), (5003, 32, 64), (777, 16, 8), (40, 4, 1), (1, 64, 64),
                                      (0, 8, 8), (30011, 64, 72), (9001, 130, 33), (2048, 1, 64), (15, 13, 7)])
def test_xty_weight_gradient_kernel(rows, A, B):
    """mvf_xty (x^T . y over the rows: the weight gradient of a Linear) against float64, for widths
    below, at and above one tile of 64 x 64 sums, row counts around the batch of 16 rows, one row,
    no row; the same bits on a second call; and as the backward of training.linear."""
    from mvfusion import training
    g = torch.Generator(device="cuda").manual_seed(rows + A)
    x = torch.randn(rows, A, device="cuda", generator=g)
    y = torch.randn(rows, B, device="cuda", generator=g) + 0.5
    got = training.xty(x, y)
    want = x.double().T @ y.double()
    mag = (x.double().abs().T @ y.double().abs()).clamp_min(1e-30)   # what the sums are made of
    assert got.shape == (A, B)
    assert float(((got.double() - want).abs() / mag).max()) < 2e-6 if rows else float(got.abs().max()) == 0.0
    assert torch.equal(got, training.xty(x, y))
    sums = training.xty(x)
    assert sums.shape == (A, 1)
    if rows:
        assert float(((sums.view(-1).double() - x.double().sum(0)).abs() / x.double().abs().sum(0).clamp_min(1e-30)).max()) < 2e-6
    if rows:
        lin = torch.nn.Linear(B, A).cuda()
        inp = y.clone().requires_grad_(True)
        training.linear(lin, inp).mul(x).sum().backward()
        ref = torch.nn.Linear(B, A).cuda().double()
        ref.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
        inp_d = y.double().requires_grad_(True)
        ref(inp_d).mul(x.double()).sum().backward()
        assert close(lin.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), 2e-6)
        assert close(lin.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), 1e-5)
        assert close(inp.grad.cpu().numpy(), inp_d.grad.cpu().numpy(), 1e-5)


@pytest.mark.parametrize("C,G", [(64, 1), (64, 4), (32, 8), (16, 1), (128, 2), (8, 2), (48, 3), (64, 64)])
def test_weighted_segment_sum_and_its_gradients(C, G):
    """training.weighted_segment_sum (the attention-weighted sum of the views of a point, one
    kernel forward and one backward) against the composition it replaces evaluated in float64;
    empty segments, one long segment, shapes that fall back to the composition (48 / 3: the
    group is no power-of-two multiple of 4 channels; one channel per group)."""
    from mvfusion import pooling, training
    rng = np.random.default_rng(C + G)
    counts = np.minimum(rng.poisson(3, 3000), 20)
    counts[17] = 300
    ptr = T(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).cuda()
    extra = 5                               # rows behind the last segment: no part in any sum
    rows = int(counts.sum()) + extra
    x = torch.randn(rows, C, device="cuda").requires_grad_(True)
    att = torch.rand(rows, G, device="cuda").requires_grad_(True)
    wgt = torch.randn(len(counts), C, device="cuda")
    out = training.weighted_segment_sum(x, att, ptr, G)
    (out * wgt).sum().backward()
    xd, ad = x.detach().double().requires_grad_(True), att.detach().double().requires_grad_(True)
    seg = torch.repeat_interleave(torch.arange(len(counts), device="cuda"), T(counts).cuda())
    full = xd * (ad.repeat_interleave(C // G, dim=1) if G < C else ad)
    ref = torch.zeros(len(counts), C, dtype=torch.float64, device="cuda").index_add_(0, seg, full[:rows - extra])
    (ref * wgt.double()).sum().backward()
    assert close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), 2e-6)
    assert close(x.grad.cpu().numpy(), xd.grad.cpu().numpy(), 2e-6)
    assert close(att.grad.cpu().numpy(), ad.grad.cpu().numpy(), 2e-6)
    assert float(x.grad[rows - extra:].abs().sum()) == 0.0 and float(att.grad[rows - extra:].abs().sum()) == 0.0
    assert float(out.detach()[T(counts == 0).cuda()].abs().sum()) == 0.0


@pytest.mark.parametrize("rows,cin,cout,bias", [(100003, 64, 64, False), (5003, 8, 32, True), (4099, 32, 16, True),
                                                (65536, 40, 36, False), (4096, 16, 4, True), (9999, 64, 8, False),
                                                (31, 24, 64, True), (1, 8, 8, True)])
def test_linear_rows_streaming_kernel(rows, cin, cout, bias):
    """mvf_linear_rows (y = x W^T + b in one streaming pass, the Linear outputs and input gradients
    of the differentiable path) against float64: every supported width class, a ragged last tile,
    fewer rows than one tile."""
    from mvfusion import _lib
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(rows + cin)
    x = torch.randn(rows, cin, device="cuda", generator=g)
    w = torch.randn(cout, cin, device="cuda", generator=g)
    b = torch.randn(cout, device="cuda", generator=g) if bias else None
    assert lib.mvf_linear_rows_supported(cin, cout) == 1
    assert lib.mvf_linear_rows_supported(cin + 1, cout) == 0 and lib.mvf_linear_rows_supported(cin, 65) == 0
    y = torch.full((rows + 1, cout), 7.0, device="cuda")         # a guard row behind the output
    _lib.check(lib.mvf_linear_rows(_lib.ptr(x), cin, _lib.ptr(w), _lib.ptr(b), cout, rows, _lib.ptr(y),
                                   _lib.stream_ptr(x.device)))
    want = x.double() @ w.double().T + (b.double() if bias else 0.0)
    mag = x.double().abs() @ w.double().abs().T + 1e-30
    assert float(((y[:rows].double() - want).abs() / mag).max()) < 1e-6
    assert float((y[rows] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("rows,cin,cout", [(100003, 64, 64), (5003, 8, 32), (4099, 40, 36), (65, 16, 4)])
def test_linear_rows_column_sums_feed_batch_norm(rows, cin, cout):
    """The Linear's streaming pass leaves the column sums of its output (mvf_linear_rows_sums); the
    BatchNorm forward that folds its batch statistics from them equals the one that reads the rows,
    and both equal float64 (the MLP block of the differentiable path, training.mlp)."""
    from mvfusion import training, pooling
    torch.manual_seed(rows)
    x = (torch.randn(rows, cin, device="cuda") + 0.3).requires_grad_(True)
    seq = pooling.MLP([cin, cout], bias=False).cuda().train()
    with torch.no_grad():
        seq[0][1].batch_norm.weight.uniform_(0.5, 1.5)
        seq[0][1].batch_norm.bias.uniform_(-0.3, 0.3)
    lin, bn = seq[0][0], seq[0][1].batch_norm
    z, sums = training.linear(lin, x, want_sums=True)
    if rows >= 4096:
        assert sums is not None and sums.shape[1:] == (2, cout)
        want = torch.stack([z.detach().double().sum(0), (z.detach().double() ** 2).sum(0)])
        got = sums.sum(0)
        assert float(((got - want).abs() / want.abs().clamp_min(1e-30)).max()) < 1e-6
    state = {k: v.clone() for k, v in bn.state_dict().items()}
    y_fused = training.mlp(seq, x)
    y_fused.sum().backward()
    assert x.grad is not None and lin.weight.grad is not None and bool(torch.isfinite(x.grad).all())
    after = {k: v.clone() for k, v in bn.state_dict().items()}
    bn.load_state_dict(state)
    y_plain = training.batch_norm_act(training.linear(lin, x), bn, seq[0][2])
    assert close(y_fused.detach().cpu().numpy(), y_plain.detach().cpu().numpy(), 2e-6)
    for k in ("running_mean", "running_var"):
        assert close(after[k].cpu().numpy(), bn.state_dict()[k].cpu().numpy(), 1e-6)


def test_batch_statistics_of_a_large_mean_small_spread_column():
    """A column whose mean is 1000 times its standard deviation: sum(y^2) / n - mean^2 cancels
    six digits, so the column sums must be float64 from the first addition on.  The streaming
    Linear's sums (rows >= 4096, the path that feeds BatchNorm) and the row-reading BatchNorm
    kernel both reproduce torch.nn.BatchNorm1d's batch variance and output."""
    from mvfusion import training, pooling
    torch.manual_seed(5)
    rows, cin, cout = 200_000, 16, 32
    x = torch.randn(rows, cin, device="cuda") * 0.01
    x[:, 0] = 1.0                                   # a constant input column ...
    seq = pooling.MLP([cin, cout], bias=False).cuda().train()
    with torch.no_grad():
        seq[0][0].weight[:, 0] = 10.0               # ... gives every output column the mean 10
    lin, bn = seq[0][0], seq[0][1].batch_norm
    z = torch.nn.functional.linear(x, lin.weight).detach()
    assert float((z.mean(0).abs() / z.std(0)).min()) > 300
    ref_bn = torch.nn.BatchNorm1d(cout).cuda().train()
    ref_bn.load_state_dict(bn.state_dict())
    want = torch.nn.functional.leaky_relu(ref_bn(z.double().float()), seq[0][2].negative_slope)
    var64 = z.double().var(0, unbiased=True)
    with torch.no_grad():
        bn.running_var.zero_()                      # the update then leaves momentum * batch variance
    state = {k: v.clone() for k, v in bn.state_dict().items()}
    for fused in (True, False):
        bn.load_state_dict(state)
        y = training.mlp(seq, x) if fused else training.batch_norm_act(training.linear(lin, x), bn, seq[0][2])
        got_var = bn.running_var / bn.momentum
        assert float(((got_var.double() - var64).abs() / var64).max()) < 1e-4, fused
        assert float((y.detach() - want).abs().max()) < 2e-3 * float(want.abs().max()), fused


@pytest.mark.parametrize("D,G", [(16, 1), (8, 4), (8, 1), (4, 8), (6, 2)])
def test_segment_dot_and_its_gradients(D, G):
    """training.segment_dot (a view's key against the query of its point, per group) against the
    composition gather_csr + product + sum in float64: values and both gradients; a shape the
    kernels do not take (6 channels per group) runs the composition."""
    from mvfusion import training
    rng = np.random.default_rng(D * 10 + G)
    counts = np.minimum(rng.poisson(3, 2500), 15)
    counts[3] = 200
    ptr = T(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).cuda()
    rows, C = int(counts.sum()), D * G
    keys = torch.randn(rows, C, device="cuda").requires_grad_(True)
    q = torch.randn(len(counts), C, device="cuda").requires_grad_(True)
    wgt = torch.randn(rows, G, device="cuda")
    comp = training.segment_dot(keys, q, ptr, G)
    (comp * wgt).sum().backward()
    kd, qd = keys.detach().double().requires_grad_(True), q.detach().double().requires_grad_(True)
    seg = torch.repeat_interleave(torch.arange(len(counts), device="cuda"), T(counts).cuda())
    ref = (kd.view(-1, G, D) * qd[seg].view(-1, G, D)).sum(2)
    (ref * wgt.double()).sum().backward()
    assert comp.shape == (rows, G)
    assert close(comp.detach().cpu().numpy(), ref.detach().cpu().numpy(), 2e-6)
    assert close(keys.grad.cpu().numpy(), kd.grad.cpu().numpy(), 2e-6)
    assert close(q.grad.cpu().numpy(), qd.grad.cpu().numpy(), 2e-6)

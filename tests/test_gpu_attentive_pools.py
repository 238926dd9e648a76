"""GPU parity of the attentive pools (QKVBimodalCSRPool, GroupBimodalCSRPool,
DeepSetFeat, HeuristicBimodalCSRPool) through the C ABI: against the golden
vectors captured from the reference modules, and against the CPU oracle on
larger seeded inputs and on the option combinations the fixtures do not cover.
Tolerance (``_close``): 1e-5 relative, element by element, for every element
above a tenth of the output's largest magnitude, and 4e-6 of that magnitude for
every element (fp32 GEMMs; the summation order of the reference's BLAS is not
part of its contract)."""
import numpy as np
import pytest
import torch

from mvfusion import pooling

import mvf_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference_calls():
    """The tests of this file exercise the fused inference kernels: eval mode
    with autograd off (with autograd on, eval-mode calls whose parameters require
    a gradient take the differentiable composed path instead)."""
    with torch.no_grad():
        yield
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _sd(g):
    return {k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}


def _close(got, want, rtol=1e-5):
    """Element-wise |got - want| <= rtol * |want| for every element above a tenth
    of the largest magnitude, and |got - want| <= 0.4 * rtol of the largest
    magnitude for every element.  The rounding error of a float32 sum scales with
    the sum of the magnitudes of its terms, not with its result: elements far
    below the largest are sums that cancel, and no summation order (the
    reference's BLAS has its own) keeps them to 1e-5 of their own value.
    Observed over this suite: 5e-6 relative on the large elements, 1e-6 (once
    3e-6) of the largest magnitude overall.  Returns both observed values."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64).reshape(got.shape)
    scale = max(float(np.abs(want).max()), 1e-30)
    err = np.abs(got - want)
    big = np.abs(want) > 0.1 * scale
    worst_rel = float((err[big] / np.abs(want[big])).max()) if big.any() else 0.0
    worst_abs = float(err.max()) / scale if err.size else 0.0
    assert worst_rel <= rtol and worst_abs <= 0.4 * rtol, \
        f"worst relative error {worst_rel:.2e} (limit {rtol:.0e}) on the {int(big.sum())} elements above " \
        f"a tenth of the largest magnitude; worst error overall {worst_abs:.2e} of that magnitude " \
        f"(limit {0.4 * rtol:.0e})"
    return worst_rel, worst_abs


def _fixture_rtol(g):
    """Tolerance against a fixture's float32 output: 1e-5, unless the fixture shows that the
    reference's own float32 result is farther than a third of that from the same module
    evaluated in float64 (x_pool64) -- then three times that distance: two float32
    evaluations that are each that far from the exact value cannot be expected to be closer
    to each other.  (qkv_default: a nearly closed gate leaves outputs of 1e-2 that are
    cancelling sums; the reference's float32 forward is 1.1e-5 from its float64 one.)"""
    if "x_pool64" not in g:
        return 1e-5
    want = np.asarray(g["x_pool64"], np.float64)
    big = np.abs(want) > 0.1 * np.abs(want).max()
    noise = float((np.abs(g["x_pool"] - want)[big] / np.abs(want[big])).max())
    return max(1e-5, 3.0 * noise)


@pytest.mark.parametrize("name", ["qkv_default", "qkv_groups4", "qkv_nogate"])
def test_qkv_pool_golden(golden, name, mlp_precision):
    g = golden(name)
    rtol = _fixture_rtol(g)
    assert rtol < 4e-5 and (name == "qkv_default" or rtol == 1e-5)
    mod = pooling.QKVBimodalCSRPool(
        in_main=g["x_main"].shape[1], in_map=g["x_map"].shape[1], in_mod=g["x_mod"].shape[1],
        out_mod=g["x_mod"].shape[1], num_groups=int(g["num_groups"]), nc_qk=int(g["nc_qk"]),
        gating=bool(g["gating"]), dim_scaling=bool(g["dim_scaling"]),
        group_scaling=bool(g["group_scaling"]), use_num=bool(g["use_num"]), save_last=True,
        map_encoder="DeepSetFeat")
    missing = mod.load_state_dict(_sd(g), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    mod = mod.to(DEV).eval()
    out = mod(T(g["x_main"]), T(g["x_mod"]), T(g["x_map"]), T(g["ptr"]))
    _close(out, g["x_pool"], rtol=rtol)
    _close(mod._last_C, g["C"], rtol=1e-4)
    _close(mod._last_A, g["A"], rtol=1e-4)
    if bool(g["gating"]):
        _close(mod._last_G, g["G"], rtol=1e-4)
    # the fused path (no save_last: K chained behind the map encoder) only
    # differs in the order the key projection sums its input channels
    mod.save_last = False
    out2 = mod(T(g["x_main"]), T(g["x_mod"]), T(g["x_map"]), T(g["ptr"]))
    _close(out2, g["x_pool"], rtol=rtol)
    # and both are as close to the float64 evaluation as the reference's float32 one is
    _close(out2, g["x_pool64"], rtol=rtol)


def test_group_pool_golden(golden, mlp_precision):
    g = golden("group_pool")
    mod = pooling.GroupBimodalCSRPool(
        in_map=g["x_map"].shape[1], in_mod=g["x_mod"].shape[1], out_mod=g["x_mod"].shape[1],
        num_groups=int(g["num_groups"]), gating=bool(g["gating"]),
        group_scaling=bool(g["group_scaling"]), use_num=bool(g["use_num"]),
        map_encoder="DeepSetFeat")
    mod.load_state_dict(_sd(g), strict=True)
    mod = mod.to(DEV).eval()
    out = mod(None, T(g["x_mod"]), T(g["x_map"]), T(g["ptr"]))
    _close(out, g["x_pool"])


@pytest.mark.parametrize("name", ["qkv_usemod", "qkv_minmax", "qkv_residual", "qkv_both",
                                  "group_usemod"])
def test_attention_variants_golden(golden, name, mlp_precision):
    """Options beyond the defaults: modality features mixed into keys / queries /
    scores, MinMaxDiffSetFeat / MLPSetFeat encoders, the other fusions and pools."""
    import ast
    g = golden(name)
    kw = ast.literal_eval(str(g["kwargs"]))
    mod = getattr(pooling, str(g["cls"]))(save_last=True, **kw)
    mod.load_state_dict(_sd(g), strict=True)
    mod = mod.to(DEV).eval()
    out = mod(T(g["x_main"]), T(g["x_mod"]), T(g["x_map"]), T(g["ptr"]))
    _close(out, g["x_pool"])
    _close(mod._last_C, g["C"], rtol=1e-4)
    _close(mod._last_A, g["A"], rtol=1e-4)
    mod.save_last = False
    _close(mod(T(g["x_main"]), T(g["x_mod"]), T(g["x_map"]), T(g["ptr"])), g["x_pool"])


def _random_module(mod, seed):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
        if getattr(mod, "G", None) is not None:
            mod.G.weight.copy_(torch.rand(mod.G.weight.shape, generator=gen) + 0.5)
            mod.G.bias.copy_(torch.randn(mod.G.bias.shape, generator=gen) * 0.1)
    return mod


def _np_sd(mod):
    return {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}


def _ptr(rng, n, mean):
    counts = np.minimum(rng.poisson(mean, n), 32)
    counts[rng.random(n) < 0.05] = 0
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)


@pytest.mark.parametrize("groups,pool,fusion,in_mod,out_mod", [
    (1, "max", "concatenation", 64, 64),
    (8, "max_mean", "residual", 64, 64),
    (13, "min", "both", 13, 13),          # odd widths, one group per channel
    (2, "max", "concatenation", 160, 96),  # layers wider than one launch
])
def test_qkv_pool_vs_oracle(groups, pool, fusion, in_mod, out_mod, mlp_precision):
    rng = np.random.default_rng(5)
    N = 20000
    ptr = _ptr(rng, N, 4)
    Vw = int(ptr[-1])
    x_main = rng.random((N, 24)).astype(np.float32)
    x_mod = rng.random((Vw, in_mod)).astype(np.float32)
    x_map = rng.random((Vw, 8)).astype(np.float32)
    torch.manual_seed(groups)
    mod = _random_module(pooling.QKVBimodalCSRPool(
        in_main=24, in_map=8, in_mod=in_mod, out_mod=out_mod, num_groups=groups, nc_qk=8,
        gating=True, dim_scaling=True, group_scaling=True, pool=pool, fusion=fusion,
        use_num=True), 3)
    want = orc.qkv_pool(x_main, x_mod, x_map, ptr, _np_sd(mod), num_groups=groups, nc_qk=8,
                        gating=True, dim_scaling=True, group_scaling=True, use_num=True,
                        pool=tuple(pool.split("_")), fusion=fusion)
    out = mod.to(DEV).eval()(T(x_main), T(x_mod), T(x_map), T(ptr))
    assert out.shape == (N, out_mod)
    _close(out, want)


def test_group_pool_vs_oracle_and_empty(mlp_precision):
    rng = np.random.default_rng(6)
    N = 5000
    ptr = _ptr(rng, N, 3)
    Vw = int(ptr[-1])
    x_mod = rng.random((Vw, 32)).astype(np.float32)
    x_map = rng.random((Vw, 8)).astype(np.float32)
    torch.manual_seed(1)
    mod = _random_module(pooling.GroupBimodalCSRPool(
        in_map=8, in_mod=32, num_groups=4, gating=True, group_scaling=True), 4)
    want = orc.group_pool(x_mod, x_map, ptr, _np_sd(mod), num_groups=4, gating=True,
                          group_scaling=True)
    mod = mod.to(DEV).eval()
    _close(mod(None, T(x_mod), T(x_map), T(ptr)), want)
    # all groups empty: zeros of the right shape
    z = mod(None, T(x_mod[:0]), T(x_map[:0]), T(np.zeros(8, np.int64)))
    assert z.shape == (7, 32) and float(z.abs().max()) == 0.0


def test_deep_set_feat_vs_oracle(mlp_precision):
    rng = np.random.default_rng(7)
    ptr = _ptr(rng, 3000, 5)
    x = rng.random((int(ptr[-1]), 8)).astype(np.float32)
    torch.manual_seed(2)
    mod = _random_module(pooling.DeepSetFeat(8, 32, pool="max", fusion="concatenation"), 5)
    sd = {"E." + k: v for k, v in _np_sd(mod).items()}
    want = orc.deep_set_feat(x, ptr, sd, "E")
    _close(mod.to(DEV).eval()(T(x), T(ptr)), want)


def test_heuristic_pool_and_modes():
    rng = np.random.default_rng(8)
    ptr = _ptr(rng, 2000, 3)
    Vw = int(ptr[-1])
    x_mod = rng.random((Vw, 16)).astype(np.float32)
    x_map = rng.random((Vw, 8)).astype(np.float32)
    for mode in ("max", "min"):
        out = pooling.HeuristicBimodalCSRPool(mode=mode, feat="planarity")(
            None, T(x_mod), T(x_map), T(ptr)).cpu().numpy()
        want = np.zeros((2000, 16), np.float32)
        for s in range(2000):
            a, b = ptr[s], ptr[s + 1]
            if b > a:
                f = x_map[a:b, 2]
                want[s] = x_mod[a + (np.argmax(f) if mode == "max" else np.argmin(f))]
        np.testing.assert_array_equal(out, want)


def test_training_mode_records_a_graph_and_cpu_inputs_raise():
    mod = pooling.QKVBimodalCSRPool(in_main=4, in_map=4, in_mod=4).to(DEV)
    x = torch.rand(6, 4, device=DEV)
    ptr = torch.tensor([0, 2, 2, 6], device=DEV)
    with torch.enable_grad():
        out = mod(x[:3], x, x, ptr)                 # modules start in training mode
        assert out.shape == (3, 4) and out.requires_grad
        # eval mode with autograd on and parameters that require a gradient: a graph too
        assert mod.eval()(x[:3], x, x, ptr).requires_grad
    assert not mod.eval()(x[:3], x, x, ptr).requires_grad
    with pytest.raises(RuntimeError):
        mod.eval()(x[:3].cpu(), x.cpu(), x.cpu(), ptr.cpu())


def test_mlp_chain_dot_epilogue():
    """run_layers(..., dot=...) == (run_layers(...) * queries[row2seg]).sum(1) / divisor
    for one, two and three fused layers, ragged row counts and 8 / 16 / 32 keys."""
    from mvfusion import dense
    torch.manual_seed(5)
    for rows, nq, chans, D in [(1000, 300, [8, 32, 32], 16), (4097, 1000, [64, 32, 32], 8),
                               (33, 7, [24], 32), (70000, 20000, [40, 64], 16)]:
        mlp = _random_module(pooling.MLP(chans, bias=False), 9).to(DEV).eval() if len(chans) > 1 else None
        lin = torch.nn.Linear(chans[-1], D).to(DEV)
        specs = (dense.mlp_layers(mlp) if mlp is not None else []) + [dense.linear_layer(lin)]
        x = torch.rand((rows, chans[0]), device=DEV) - 0.5
        q = torch.rand((nq, D), device=DEV) - 0.5
        seg = torch.randint(0, nq, (rows,), device=DEV, dtype=torch.int32)
        assert dense.dot_fusable(specs, q)
        keys = dense.run_layers(specs, x)
        for div in (0.0, 4.0):
            got = dense.run_layers(specs, x, dot=(q, seg, div))
            want = (keys.double() * q[seg.long()].double()).sum(1, keepdim=True)
            if div:
                want = want / div
            assert got.shape == (rows, 1)
            err = (got.double() - want).abs().max().item()
            assert err <= 1e-5 * max(want.abs().max().item(), 1e-3), (rows, chans, D, err)


@pytest.mark.parametrize("C,gating,G", [(64, True, 1), (32, True, 1), (48, False, 1), (8, True, 1),
                                        (64, True, 4), (64, False, 8), (48, True, 4), (32, True, 2)])
def test_mlp_chain_pool_epilogue(C, gating, G):
    """The value encoder with the attention-weighted segment sum as epilogue
    against encoder launch + attention_pool on the same compatibilities:
    identical bits (without a gate) for every segment inside one 32-row tile (wide chains: of at
    most two rows, their sums are trees), 1e-6 of the output
    scale for the segments that cross a tile boundary; empty segments, segments of
    more than 64 rows (three tiles) and a ragged last tile included; two runs give
    the same bits for every segment.  A gate is
    folded into the attention weights, so with one every segment is compared at
    1e-6.  Then the two modules that use it, with and
    without save_last (which keeps the unfused launches)."""
    from mvfusion import dense
    rng = np.random.default_rng(11)
    N = 30011
    counts = np.minimum(rng.poisson(4, N), 32)
    counts[rng.random(N) < 0.05] = 0
    counts[[17, 5000, N - 1]] = [70, 131, 45]
    counts[0] = 0
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    Vw = int(ptr[-1])
    x_mod = T((rng.random((Vw, C)) - 0.3).astype(np.float32))
    comp = T((rng.random((Vw, G)) * 2 - 1).astype(np.float32))
    c2g = None if G == 1 else torch.arange(G, dtype=torch.int32, device=DEV).repeat_interleave(C // G)
    dptr = T(ptr)
    torch.manual_seed(C)
    e_mod = _random_module(pooling.MLP([C, C, C], bias=False), 3).to(DEV).eval()
    specs = dense.mlp_layers(e_mod)
    assert dense.pool_fusable(specs, x_mod, G)
    kw = dict(comp=comp, group_scaling=True, gating=gating,
              gate_w=torch.linspace(1.0, 1.6, G, device=DEV)[None] if gating else None,
              gate_b=torch.linspace(-0.1, 0.1, G, device=DEV)[None] if gating else None)
    plain, _, att, gate = dense.attention_pool(dense.run_layers(specs, x_mod), dptr, G, c2g, **kw)
    row2seg = dense.csr_row_segments(dptr, Vw)

    def fused_pass():
        out, _, att2, gate2 = dense.attention_pool(C, dptr, G, c2g, **kw)
        assert gate is None or torch.equal(gate2, gate)
        if gate is None:
            assert torch.equal(att2, att)
        else:
            assert torch.equal(att2, att * gate[row2seg.long()])
        return dense.pooled_layers(specs, x_mod, att2, row2seg, dptr, out)

    fused, again = fused_pass(), fused_pass()
    assert fused.shape == plain.shape == (N, C) and float(plain.abs().max()) > 0.01
    inside = (ptr[:-1] // 32 == (ptr[1:] - 1) // 32) | (counts == 0)
    two_tiles = (ptr[:-1] // 32 + 1 == (ptr[1:] - 1) // 32)
    assert inside.sum() > 0.8 * N and (~inside).sum() > 100
    m = torch.from_numpy(inside).to(DEV)
    if not gating:
        # chains of layers wider than 32 run the scan epilogue (a balanced tree over a segment's rows
        # instead of the in-order fold): there only segments of at most two rows keep every bit
        same_order = inside if C <= 32 else inside & (counts <= 2)
        assert same_order.sum() > 1000
        ms = torch.from_numpy(same_order).to(DEV)
        assert torch.equal(fused[ms], plain[ms])
    _close(fused, plain.cpu().numpy(), rtol=1e-6)
    assert float(fused[T(counts == 0)].abs().max()) == 0.0
    # run to run: every segment keeps its bits (the partial sums of the tiles a segment
    # crosses are added in tile order by a second launch, not atomically)
    assert two_tiles.sum() > 100 and (~(inside | two_tiles)).sum() >= 2
    assert torch.equal(fused, again)

    # the modules: save_last keeps the unfused launches (and sums the key / score
    # layers' inputs in another order, so no bit equality here)
    x_main = T(rng.random((N, 16)).astype(np.float32))
    x_map = T(rng.random((Vw, 8)).astype(np.float32))
    qkv = _random_module(pooling.QKVBimodalCSRPool(
        in_main=16, in_map=8, in_mod=C, out_mod=C, num_groups=G, nc_qk=8, gating=False), 3).to(DEV).eval()
    grp = _random_module(pooling.GroupBimodalCSRPool(
        in_map=8, in_mod=C, num_groups=G, gating=False, group_scaling=True), 4).to(DEV).eval()
    for mod, args in ((qkv, (x_main, x_mod, x_map, dptr)), (grp, (None, x_mod, x_map, dptr))):
        a = mod(*args)
        mod.save_last = True
        b = mod(*args)
        assert float(b.abs().max()) > 0.01
        _close(a, b.cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("cin,C", [(8, 32), (16, 64), (8, 8)])
def test_mlp_chain_segment_max_epilogue(cin, C):
    """Layer chain that stores its rows and their per-segment maximum in one
    launch against run_layers + segment_csr: identical bits, empty segments,
    segments over several 32-row tiles and a ragged last tile included."""
    from mvfusion import dense
    from mvfusion.pooling import segment_csr
    rng = np.random.default_rng(13)
    N = 20011
    counts = np.minimum(rng.poisson(4, N), 32)
    counts[rng.random(N) < 0.05] = 0
    counts[[0, 17, 5000, N - 1]] = [0, 70, 131, 45]
    ptr = T(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64))
    Vw = int(ptr[-1])
    x = T((rng.random((Vw, cin)) - 0.5).astype(np.float32))
    torch.manual_seed(C)
    mlp = _random_module(pooling.MLP([cin, C, C], bias=False), 3).to(DEV).eval()
    specs = dense.mlp_layers(mlp)
    assert dense.pool_fusable(specs, x, 1)
    row2seg = dense.csr_row_segments(ptr, Vw)
    rows, smax = dense.layers_with_segment_max(specs, x, ptr, row2seg)
    want_rows = dense.run_layers(specs, x)
    want_max = segment_csr(want_rows, ptr, reduce="max")
    assert (want_max < 0).any() and (want_max > 0).any()
    assert torch.equal(rows, want_rows)
    assert torch.equal(smax, want_max)
    assert float(smax[T(counts == 0)].abs().max()) == 0.0


@pytest.mark.parametrize("cin,mid,cout,scale", [(64, 64, 64, 1.0), (16, 48, 64, 1.0), (40, 64, 36, 1.0),
                                                (8, 33, 64, 1e-6), (64, 64, 64, 1e6), (24, 64, 48, 1.0)])
def test_wide_chain_on_split_bf16_matches_float32_chain(cin, mid, cout, scale):
    """Chains of two wide layers run on the bf16 matrix cores with three-way split operands
    (mvf_set_mlp_precision 1, the default): against the same chain on the float32 matrix cores
    (mode 0) and against a float64 evaluation, over input widths that leave padded k-steps, very
    small and very large magnitudes (bf16 keeps float32's exponent range), negative slopes'
    branch, signed zeros and a ragged last tile.  Both float32 paths must be as close to float64
    as a float32 dot product is: 2e-6 of the row's magnitude."""
    from mvfusion import _lib, dense
    lib = _lib.load()
    rng = np.random.default_rng(cin * 1000 + cout)
    rows = 5003
    x = ((rng.random((rows, cin)) * 2 - 1) * scale).astype(np.float32)
    x[rng.random((rows, cin)) < 0.05] = 0.0
    x[7] = -0.0
    torch.manual_seed(cin + cout)
    mlp = _random_module(pooling.MLP([cin, mid, cout], bias=False), 5).to(DEV).eval()
    specs = dense.mlp_layers(mlp)
    assert lib.mvf_get_mlp_precision() == 1
    got = dense.run_layers(specs, T(x))
    try:
        assert lib.mvf_set_mlp_precision(0) == 0
        plain = dense.run_layers(specs, T(x))
    finally:
        lib.mvf_set_mlp_precision(1)
    # float64 evaluation of Linear (no bias) -> BatchNorm1d (eval) -> LeakyReLU(0.2), twice
    sd = {k: v.double().cpu().numpy() for k, v in mlp.state_dict().items()}
    want = x.astype(np.float64)
    for i in range(2):
        want = want @ sd[f"{i}.0.weight"].T
        bn = f"{i}.1.batch_norm."
        want = (want - sd[bn + "running_mean"]) / np.sqrt(sd[bn + "running_var"] + mlp[i][1].batch_norm.eps) \
            * sd[bn + "weight"] + sd[bn + "bias"]
        want = np.where(want > 0, want, 0.2 * want)
    mag = np.abs(want).max(axis=1, keepdims=True) + 1e-300
    e_split = float((np.abs(got.cpu().numpy() - want) / mag).max())
    e_plain = float((np.abs(plain.cpu().numpy() - want) / mag).max())
    assert e_plain < 2e-6 and e_split < 2e-6, (e_split, e_plain)
    assert np.isfinite(got.cpu().numpy()).all()
    with pytest.raises(RuntimeError):
        _lib.check(lib.mvf_set_mlp_precision(7))


@pytest.mark.parametrize("cin,w1,w2,w3,dot", [(64, 32, 32, 16, True), (64, 32, 32, 16, False), (32, 32, 32, 32, False),
                                              (24, 20, 32, 8, True), (8, 32, 16, 12, False), (48, 32, 32, 4, True),
                                              (64, 32, 32, 1, False), (16, 32, 32, 7, False)])
def test_narrow_chain_on_split_bf16_matches_float32_chain(cin, w1, w2, w3, dot):
    """Chains of three layers of at most 32 outputs (two encoder layers and a Linear: the key /
    query branches) on the bf16 matrix cores with split operands, against the float32 matrix
    cores and a float64 evaluation: rows from two concatenated sources, the second one gathered
    per row (gather_csr of a per-segment tensor), rows stored or met with their query row (dot
    epilogue), a ragged last tile."""
    from mvfusion import _lib, dense
    lib = _lib.load()
    rng = np.random.default_rng(cin * 100 + w3)
    rows, nseg = 7001, 1803
    ca = cin // 2 if cin >= 16 else cin
    xa = (rng.random((rows, ca)) * 2 - 1).astype(np.float32)
    xb = (rng.random((nseg, cin - ca)) * 2 - 1).astype(np.float32) if cin > ca else None
    seg = np.sort(rng.integers(0, nseg, rows)).astype(np.int32)
    torch.manual_seed(cin + w3)
    mlp = _random_module(pooling.MLP([cin, w1, w2], bias=False), 7).to(DEV).eval()
    lin = torch.nn.Linear(w2, w3).to(DEV)
    specs = dense.mlp_layers(mlp) + [dense.linear_layer(lin)]
    q = (rng.random((nseg, w3)) * 2 - 1).astype(np.float32)
    kw = dict(dot=(T(q), T(seg), 2.0)) if dot else {}

    def run():
        if xb is None:
            return dense.run_layers(specs, T(xa), **kw)
        return dense.run_layers(specs, T(xa), None, T(xb), T(seg), **kw)

    assert lib.mvf_get_mlp_precision() == 1
    got = run().cpu().numpy()
    try:
        assert lib.mvf_set_mlp_precision(0) == 0
        plain = run().cpu().numpy()
    finally:
        lib.mvf_set_mlp_precision(1)
    sd = {k: v.double().cpu().numpy() for k, v in mlp.state_dict().items()}
    want = xa.astype(np.float64) if xb is None else np.concatenate([xa, xb[seg]], 1).astype(np.float64)
    for i in range(2):
        bn = f"{i}.1.batch_norm."
        want = want @ sd[f"{i}.0.weight"].T
        want = (want - sd[bn + "running_mean"]) / np.sqrt(sd[bn + "running_var"] + mlp[i][1].batch_norm.eps) \
            * sd[bn + "weight"] + sd[bn + "bias"]
        want = np.where(want > 0, want, 0.2 * want)
    want = want @ lin.weight.double().cpu().numpy().T + lin.bias.double().cpu().numpy()
    if dot:
        want = (want * q[seg].astype(np.float64)).sum(1, keepdims=True) / 2.0
        mag = np.abs(want).max() + 1e-300
    else:
        # (a row of one or a few channels can cancel to almost nothing: no less than a tenth of
        # the largest magnitude overall)
        mag = np.maximum(np.abs(want).max(axis=1, keepdims=True), 0.1 * np.abs(want).max())
    assert got.shape == want.shape == plain.shape
    e_split = float((np.abs(got - want) / mag).max())
    e_plain = float((np.abs(plain - want) / mag).max())
    assert e_plain < 3e-6 and e_split < 3e-6, (e_split, e_plain)


@pytest.mark.parametrize("cin,widths,offset", [(6, (64, 64), 0), (2, (32, 32), 0), (13, (32, 32, 16), 0),
                                               (8, (64, 64), 1), (3, (8, 8), 0)])
def test_unaligned_input_rows_take_the_aligned_kernels(cin, widths, offset):
    """Input rows whose width is no multiple of 8, or whose storage is not 16-byte aligned (a view
    that starts one float into a buffer), are padded with zero columns (dense.aligned_input)
    and run through the same kernels as aligned rows: against a float64 evaluation, and the
    DeepSetFeat encoder on such rows against the unfused evaluation of its parts."""
    from mvfusion import dense
    rng = np.random.default_rng(cin * 10 + len(widths))
    rows = 4099
    base = torch.from_numpy((rng.random(rows * cin + offset) * 2 - 1).astype(np.float32)).to(DEV)
    x = base[offset:].view(rows, cin)
    mlp = _random_module(pooling.MLP([cin, *widths], bias=False), 11).to(DEV).eval()
    specs = dense.mlp_layers(mlp)
    padded_specs, padded_x = dense.aligned_input(specs, x)
    assert padded_x.shape[1] % 8 == 0 and padded_x.data_ptr() % 16 == 0
    assert padded_specs[0].cin == padded_x.shape[1] and specs[0].cin == cin
    assert torch.equal(padded_x[:, :cin], x) and float(padded_x[:, cin:].abs().sum()) == 0.0
    got = dense.run_layers(specs, x).cpu().numpy()
    sd = {k: v.double().cpu().numpy() for k, v in mlp.state_dict().items()}
    want = x.double().cpu().numpy()
    for i in range(len(widths)):
        bn = f"{i}.1.batch_norm."
        want = want @ sd[f"{i}.0.weight"].T
        want = (want - sd[bn + "running_mean"]) / np.sqrt(sd[bn + "running_var"] + mlp[i][1].batch_norm.eps) \
            * sd[bn + "weight"] + sd[bn + "bias"]
        want = np.where(want > 0, want, 0.2 * want)
    mag = np.maximum(np.abs(want).max(axis=1, keepdims=True), 0.1 * np.abs(want).max())
    assert got.shape == want.shape
    assert float((np.abs(got - want) / mag).max()) < 3e-6

    # the set encoder on rows of that width: the fused launches (rows + segment maximum, gather
    # + concatenation) against its parts run one by one
    d_out = 32
    enc = _random_module(pooling.DeepSetFeat(cin, d_out, use_num=True), 13).to(DEV).eval()
    counts = np.minimum(rng.poisson(2, 1500), 9)
    assert counts.sum() <= rows
    ptr_np = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(ptr_np[-1])
    ptr = torch.from_numpy(ptr_np).to(DEV)
    xs = x[:n]
    out = enc(xs, ptr)
    x1 = dense.run_layers(dense.mlp_layers(enc.mlp_elt_1), xs)
    from mvfusion.pooling import segment_csr, gather_csr, _set_num
    x_set = torch.cat([segment_csr(x1, ptr, reduce="max"), _set_num(ptr).view(-1, 1)], 1)
    x_set = dense.run_layers(dense.mlp_layers(enc.mlp_set), x_set)
    ref = dense.run_layers(dense.mlp_layers(enc.mlp_elt_2), torch.cat([x1, gather_csr(x_set, ptr)], 1))
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) < 1e-5 * scale


def test_non_finite_view_row_stays_in_its_point_or_tile():
    """One view row of NaN features.  The reference's segment sum keeps it to the point that
    owns the row.  So do the float32 chains (mvf_set_mlp_precision(0)) and the unfused
    launches (save_last); the split-bf16 value encoder sums a tile of 32 rows with a masked
    scan, where 0 x NaN reaches the other points of that tile -- and no point beyond it."""
    from mvfusion import _lib
    lib = _lib.load()
    rng = np.random.default_rng(21)
    N, C = 4000, 64
    counts = np.full(N, 3)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    Vw = int(ptr[-1])
    x_main = rng.random((N, 32)).astype(np.float32)
    x_mod = rng.random((Vw, C)).astype(np.float32)
    x_map = rng.random((Vw, 8)).astype(np.float32)
    bad_point = 1500
    bad_row = int(ptr[bad_point]) + 1
    x_mod[bad_row] = np.nan
    tile = bad_row // 32
    in_tile = np.nonzero((ptr[:-1] < (tile + 1) * 32) & (ptr[1:] > tile * 32))[0]
    assert bad_point in in_tile and 8 <= len(in_tile) <= 12
    torch.manual_seed(4)
    mod = _random_module(pooling.QKVBimodalCSRPool(
        in_main=32, in_map=8, in_mod=C, out_mod=C, num_groups=1, nc_qk=16, gating=False), 5).to(DEV).eval()
    args = (T(x_main), T(x_mod), T(x_map), T(ptr))
    before = lib.mvf_get_mlp_precision()
    try:
        for precision, save_last in ((0, False), (1, True), (1, False)):
            lib.mvf_set_mlp_precision(precision)
            mod.save_last = save_last
            out = mod(*args).cpu().numpy()
            hit = np.nonzero(~np.isfinite(out).all(axis=1))[0]
            assert bad_point in hit
            if precision == 0 or save_last:
                assert list(hit) == [bad_point], (precision, save_last, hit)
            else:
                assert set(hit) <= set(in_tile), (hit, in_tile)
    finally:
        lib.mvf_set_mlp_precision(before)
        mod.save_last = False


@pytest.mark.parametrize("d_in,d_out", [(8, 8), (2, 32), (6, 16)])
def test_narrow_chain_as_accurate_as_the_oracle(d_in, d_out, mlp_precision):
    """Layers that sum few terms: the chain kernel (channel scale folded into the weights,
    channel shift added to the finished sums) is as close to the float64 value of the layers as
    the oracle's float32 order (sum, subtract the mean, scale, shift).  Started from the shift
    instead, the accumulators of such a layer round every partial sum at the shift's magnitude:
    that order measured 1.5 to 2.8 times the oracle's error for the encoder on these shapes and
    1.5 for its first chain, against 0.9 to 1.0 and 0.9 to 1.3 now."""
    import mvf_exact64 as x64
    from mvfusion import dense
    rng = np.random.default_rng(d_in * 100 + d_out)
    ptr = _ptr(rng, 4000, 5)
    x = rng.random((int(ptr[-1]), d_in)).astype(np.float32)
    torch.manual_seed(d_in)
    enc = _random_module(pooling.DeepSetFeat(d_in, d_out, pool="max", fusion="concatenation"), 11)
    sd = {"E." + k: v for k, v in _np_sd(enc).items()}
    exact = x64.deep_set_feat(x.astype(np.float64), ptr, sd, "E", ("max",), "concatenation", False)
    want = orc.deep_set_feat(x, ptr, sd, "E", ("max",), "concatenation", False)
    got = enc.to(DEV).eval()(T(x), T(ptr)).cpu().numpy()
    first = dense.run_layers(dense.mlp_layers(enc.mlp_elt_1), T(x)).cpu().numpy()
    rms = lambda a, e: float(np.sqrt(((a - e) ** 2).mean()))
    e1 = x64.mlp(x.astype(np.float64), sd, "E.mlp_elt_1")
    r_all, r_first = rms(got, exact) / rms(want, exact), rms(first, e1) / rms(orc.mlp(x, sd, "E.mlp_elt_1"), e1)
    print(f"error rms relative to the oracle's: encoder {r_all:.2f}, first chain {r_first:.2f}")
    assert r_all <= 1.2 and r_first <= 1.4
    _close(got, want)

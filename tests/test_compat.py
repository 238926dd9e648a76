"""The drop-in registration resolves the reference's lookups by name."""
import importlib
import sys


def test_aliases_resolve_reference_lookups():
    from mvfusion import compat
    saved = {k: sys.modules.get(k) for k in compat.ALIASES}
    try:
        compat.install()
        pooling = importlib.import_module("torch_points3d.modules.multimodal.pooling")
        for name in ("BimodalCSRPool", "QKVBimodalCSRPool", "GroupBimodalCSRPool",
                     "HeuristicBimodalCSRPool", "DeepSetFeat", "Gating", "segment_softmax_csr",
                     "gather_csr", "segment_gather_csr"):
            assert hasattr(pooling, name)
        vis = importlib.import_module("torch_points3d.core.multimodal.visibility")
        for name in ("camera_projection", "visibility_from_splatting", "visibility_from_depth_map",
                     "visibility", "VisibilityModel", "SplattingVisibility", "DepthBasedVisibility"):
            assert hasattr(vis, name)
        csr = importlib.import_module("torch_points3d.core.multimodal.csr")
        assert hasattr(csr, "CSRData") and hasattr(csr, "CSRBatch")
        # the pool factory path: class by name + kwargs, extra kwargs swallowed
        mod = getattr(pooling, "BimodalCSRPool")(mode="max", index=3)
        assert mod._mode == "max"
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_state_dict_keys_match_reference_layout():
    from mvfusion import pooling
    m = pooling.QKVBimodalCSRPool(in_main=8, in_map=4, in_mod=16, nc_qk=4, num_groups=2)
    keys = set(m.state_dict().keys())
    for k in ("E_main.0.0.weight", "E_main.0.1.batch_norm.running_mean",
              "E_map.mlp_elt_1.0.0.weight", "E_map.mlp_set.1.1.batch_norm.weight",
              "E_map.mlp_elt_2.1.0.weight", "E_mod.1.0.weight", "Q.weight", "Q.bias",
              "K.weight", "K.bias", "G.weight", "G.bias"):
        assert k in keys, k
    assert not any("bias" in k and ".0.0." in k for k in keys)   # MLP(bias=False)

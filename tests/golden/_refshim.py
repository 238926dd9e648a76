"""Load the reference's hot-path source files by file path, in this container.

Only used by ``make_golden.py`` (never by tests, smoke or bench: the
reference does not exist on the GPU box).  The reference package cannot be
imported as a package because numba, torch_scatter, pykeops, torch_geometric
and torchvision are not installed; the hot-path files themselves only need

* ``numba.njit``            -> identity decorator (the bodies are plain numpy),
* ``torch_scatter``         -> ``segment_csr`` / ``scatter_min`` / ``scatter_max``
                               restated with torch ops (torch_scatter's published
                               semantics: empty segment -> 0, mean = sum / count),
* ``pykeops.torch``         -> ``LazyTensor = None`` (Biasutti visibility only),
* ``torch_geometric.data``  -> a 20-line attribute bag standing in for ``Data``,
* ``torchvision.transforms``-> empty module,
* the radius samplers of ``core/data_transform/transforms.py:301-405`` restated
  on sklearn's KDTree (that file drags in the whole training stack).

Nothing of the reference is copied: its files are executed where they lie.
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/DeepViewAgg-Wildscenes/torch_points3d")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


def segment_csr(src, indptr, out=None, reduce="sum"):
    """torch_scatter.segment_csr semantics for 1-D indptr on dim 0."""
    n = indptr.shape[0] - 1
    sizes = (indptr[1:] - indptr[:-1])
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    for i in range(n):
        a, b = int(indptr[i]), int(indptr[i + 1])
        if b <= a:
            continue
        seg = src[a:b]
        if reduce in ("sum", "add"):
            # sequential accumulation in row order, like the CPU kernel
            acc = seg[0].clone()
            for r in range(1, b - a):
                acc = acc + seg[r]
            res[i] = acc
        elif reduce == "mean":
            acc = seg[0].clone()
            for r in range(1, b - a):
                acc = acc + seg[r]
            res[i] = acc / (b - a)
        elif reduce == "max":
            res[i] = seg.max(dim=0).values
        elif reduce == "min":
            res[i] = seg.min(dim=0).values
        else:
            raise ValueError(reduce)
    return res


def scatter_mean(src, index, dim=0, dim_size=None):
    """torch_scatter.scatter_mean on dim 0: out[j] = mean of src[index == j]
    (0 where no row maps to j), rows added in input order."""
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    cnt = torch.zeros(n, dtype=src.dtype)
    out.index_add_(0, index, src)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return out / cnt.view((-1,) + (1,) * (src.dim() - 1))


def scatter_add(src, index, dim=0, dim_size=None):
    """torch_scatter.scatter_add on dim 0."""
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def _scatter_minmax(src, index, dim=0, dim_size=None, is_min=True):
    n = int(index.max()) + 1 if dim_size is None else dim_size
    if src.dim() == 1:
        if src.dtype.is_floating_point:
            fill = float("inf") if is_min else float("-inf")
        else:
            fill = torch.iinfo(src.dtype).max if is_min else torch.iinfo(src.dtype).min
        out = torch.full((n,), fill, dtype=src.dtype)
        arg = torch.full((n,), src.shape[0], dtype=torch.long)
        for i in range(src.shape[0]):
            j = int(index[i])
            if (is_min and src[i] < out[j]) or (not is_min and src[i] > out[j]):
                out[j] = src[i]
                arg[j] = i
        out[arg == src.shape[0]] = 0
        return out, arg
    outs = [_scatter_minmax(src[:, c], index, 0, n, is_min) for c in range(src.shape[1])]
    return torch.stack([o[0] for o in outs], 1), torch.stack([o[1] for o in outs], 1)


class Data:
    """Attribute bag with the handful of members MapImages touches."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __contains__(self, k):
        return k in self.__dict__

    @property
    def num_nodes(self):
        return self.pos.shape[0]

    def clone(self):
        return Data(**{k: (v.clone() if torch.is_tensor(v) else v)
                       for k, v in self.__dict__.items()})


KDTREE_KEY = "kd_tree"


class _RadiusSampling:
    """Restates core/data_transform/transforms.py:301-405 (SphereSampling /
    CylinderSampling, align_origin=False): KDTree(leaf_size=50) on pos (or
    pos[:, :2]), ``query_radius`` result order kept as returned."""
    KDTREE_KEY = KDTREE_KEY
    _dims = 3

    def __init__(self, radius, centre, align_origin=True):
        assert not align_origin
        centre = np.asarray(centre)
        if self._dims == 2 and centre.shape[0] == 3:
            centre = centre[:-1]
        self._radius = radius
        self._centre = centre[None] if centre.ndim == 1 else centre

    def __call__(self, data):
        from sklearn.neighbors import KDTree
        n = data.pos.shape[0]
        if not hasattr(data, KDTREE_KEY):
            setattr(data, KDTREE_KEY, KDTree(
                np.asarray(data.pos[:, :self._dims]), leaf_size=50))
        tree = getattr(data, KDTREE_KEY)
        ind = torch.LongTensor(tree.query_radius(self._centre, r=self._radius)[0])
        out = Data()
        for key in set(data.keys):
            if key == KDTREE_KEY:
                continue
            item = data[key]
            if torch.is_tensor(item) and item.shape[0] == n:
                item = item[ind]
            elif torch.is_tensor(item):
                item = item.clone()
            setattr(out, key, item)
        out._ind = ind
        return out


class SphereSampling(_RadiusSampling):
    _dims = 3


class CylinderSampling(_RadiusSampling):
    _dims = 2


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, REF / relpath)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


_loaded = {}


def load_reference():
    """Returns a dict of the reference modules on the hot path."""
    if _loaded:
        return _loaded
    _mod("numba", njit=_njit)
    _mod("torch_scatter", segment_csr=segment_csr, scatter_mean=scatter_mean, scatter_add=scatter_add,
         scatter_min=lambda s, i, dim=0, dim_size=None: _scatter_minmax(s, i, dim, dim_size, True),
         scatter_max=lambda s, i, dim=0, dim_size=None: _scatter_minmax(s, i, dim, dim_size, False))
    _mod("pykeops")
    _mod("pykeops.torch", LazyTensor=None)
    _mod("torch_geometric")
    _mod("torch_geometric.data", Data=Data)
    _mod("torchvision")
    _mod("torchvision.transforms")
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    torch.jit.script = lambda f=None, *a, **k: f
    if not torch.cuda.is_available():
        torch.cuda.synchronize = lambda *a, **k: None

    for pkg in ["torch_points3d", "torch_points3d.utils", "torch_points3d.core",
                "torch_points3d.core.multimodal", "torch_points3d.core.common_modules",
                "torch_points3d.core.spatial_ops", "torch_points3d.modules",
                "torch_points3d.modules.multimodal"]:
        _mod(pkg).__path__ = []
    _mod("torch_points3d.core.data_transform", SphereSampling=SphereSampling,
         CylinderSampling=CylinderSampling, GridSampling3D=None, SaveOriginalPosId=None)
    _mod("torch_points3d.core.spatial_ops.neighbour_finder", FAISSGPUKNNNeighbourFinder=None)

    mm_utils = _load("torch_points3d.utils.multimodal", "utils/multimodal.py")
    csr = _load("torch_points3d.core.multimodal.csr", "core/multimodal/csr.py")
    core_mm = sys.modules["torch_points3d.core.multimodal"]
    core_mm.CSRData, core_mm.CSRBatch = csr.CSRData, csr.CSRBatch
    vis = _load("torch_points3d.core.multimodal.visibility", "core/multimodal/visibility.py")
    image = _load("torch_points3d.core.multimodal.image", "core/multimodal/image.py")
    base = _load("torch_points3d.core.common_modules.base_modules",
                 "core/common_modules/base_modules.py")
    sys.modules["torch_points3d.core.common_modules"].MLP = base.MLP
    pooling = _load("torch_points3d.modules.multimodal.pooling", "modules/multimodal/pooling.py")
    tr_image = _load("torch_points3d.core.data_transform.multimodal_image",
                     "core/data_transform/multimodal/image.py")
    _loaded.update(utils=mm_utils, csr=csr, visibility=vis, image=image,
                   base_modules=base, pooling=pooling, transforms=tr_image,
                   Data=Data)
    return _loaded

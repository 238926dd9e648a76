"""Minimal ``SameSettingImageData``: the per-sample holder of camera poses,
intrinsics and image-state scalars that ``MapImages`` reads and annotates
(subset of ``torch_points3d/core/multimodal/image.py:177-1297`` needed by the
mapping path: pinhole / fisheye cameras with 4x4 extrinsics, equirectangular
cameras with omega-phi-kappa poses; no image loading).

``MapImages`` is duck-typed: any object exposing these attributes works,
including the reference's own class.
"""
import copy

import numpy as np
import torch

from .csr import tensor_idx
from .mapping import ImageMapping


class SameSettingImageData:

    def __init__(self, path=None, pos=torch.empty([0, 3]), extrinsic=None,
                 fx=None, fy=None, mx=None, my=None, ref_size=(512, 256),
                 proj_upscale=2, downscale=1, crop_size=None, crop_offsets=None,
                 x=None, mappings=None, mask=None, visibility=None, opk=None,
                 xi=None, k1=None, k2=None, gamma1=None, gamma2=None, u0=None, v0=None,
                 **kwargs):
        self.pos = pos.double()
        self.opk = opk.double() if opk is not None else None
        self.xi, self.k1, self.k2 = xi, k1, k2
        self.gamma1, self.gamma2, self.u0, self.v0 = gamma1, gamma2, u0, v0
        n = self.pos.shape[0]
        self.path = np.array([f"view_{i}" for i in range(n)], dtype='O') \
            if path is None else np.array(path)
        self.extrinsic = extrinsic
        self.fx, self.fy, self.mx, self.my = fx, fy, mx, my
        self.ref_size = tuple(ref_size)
        self.proj_upscale = proj_upscale
        self.downscale = downscale
        self.crop_size = tuple(crop_size) if crop_size is not None else self.ref_size
        self.crop_offsets = crop_offsets if crop_offsets is not None \
            else torch.zeros((n, 2), dtype=torch.int64)
        self.x = x
        self.mappings = mappings
        self.mask = mask
        self.visibility = visibility

    # the reference exposes these as computed properties
    @property
    def num_views(self):
        return self.pos.shape[0]

    @property
    def device(self):
        return self.pos.device

    _fisheye_keys = ('xi', 'k1', 'k2', 'gamma1', 'gamma2', 'u0', 'v0')

    @property
    def has_opk(self):
        return self.opk is not None

    @property
    def has_extrinsic(self):
        return self.extrinsic is not None

    @property
    def is_pinhole(self):
        return all(a is not None for a in (self.fx, self.fy, self.mx, self.my))

    @property
    def is_fisheye(self):
        return all(getattr(self, a) is not None for a in self._fisheye_keys)

    @property
    def intrinsic_fisheye(self):
        """[V, 7] xi, k1, k2, gamma1, gamma2, u0, v0 (image.py:441-454)."""
        return torch.stack([torch.as_tensor(getattr(self, a)) for a in self._fisheye_keys]).T

    @property
    def intrinsic_pinhole(self):
        """[V, 4, 4] intrinsics from fx, fy, mx, my (image.py:424-439)."""
        K = torch.eye(4).repeat(self.num_views, 1, 1)
        K[:, 0, 0], K[:, 1, 1] = self.fx, self.fy
        K[:, 0, 2], K[:, 1, 2] = self.mx, self.my
        return K

    @property
    def proj_size(self):
        return tuple(int(x * self.proj_upscale) for x in self.ref_size)

    @property
    def img_size(self):
        return tuple(int(x / self.downscale) for x in self.crop_size)

    @property
    def mapping_size(self):
        """Image size the mappings refer to (image.py:656-659)."""
        return self.crop_size

    @property
    def pixel_dtype(self):
        for dtype in [torch.int16, torch.int32, torch.int64]:
            if torch.iinfo(dtype).max >= max(self.ref_size[0], self.ref_size[1]):
                break
        return dtype

    @property
    def num_points(self):
        return self.mappings.num_groups if self.mappings is not None else 0

    def __len__(self):
        return self.num_views

    def __getitem__(self, idx):
        idx = tensor_idx(idx).to(self.device)
        assert idx.unique().numel() == idx.shape[0], "Index must not contain duplicates."
        assert self.mappings is None, \
            "view selection of an already-mapped SameSettingImageData is not supported here"
        sel = lambda t: t[idx] if t is not None else None
        return self.__class__(
            path=self.path[np.asarray(idx.cpu())], pos=self.pos[idx],
            extrinsic=sel(self.extrinsic), fx=sel(self.fx), fy=sel(self.fy),
            mx=sel(self.mx), my=sel(self.my), ref_size=copy.deepcopy(self.ref_size),
            proj_upscale=copy.deepcopy(self.proj_upscale),
            downscale=copy.deepcopy(self.downscale),
            crop_size=copy.deepcopy(self.crop_size), crop_offsets=self.crop_offsets[idx],
            x=sel(self.x), mask=self.mask.clone() if self.mask is not None else None,
            visibility=copy.deepcopy(self.visibility), opk=sel(self.opk),
            **{a: sel(getattr(self, a)) for a in self._fisheye_keys})

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def feature_map_indexing(self):
        return self.mappings.feature_map_indexing if self.mappings is not None else None

    @property
    def atomic_csr_indexing(self):
        return self.mappings.atomic_csr_indexing if self.mappings is not None else None

    @property
    def view_csr_indexing(self):
        return self.mappings.view_csr_indexing if self.mappings is not None else None

    @property
    def mapping_features(self):
        return self.mappings.features

    def get_mapped_features(self, interpolate=False):
        """Features of ``self.x`` [V, C, H, W] at the mapped pixels, one row per
        atomic mapping (core/multimodal/image.py:1272-1297).  Without
        interpolation the mappings are first brought to the resolution of the
        feature map (``1 / downscale``); with it the map is sampled bilinearly
        at the mapping-resolution coordinates."""
        from . import pooling
        scale = 1 / self.downscale
        mappings = self.mappings if interpolate else self.mappings.rescale_images(scale)
        if interpolate and scale != 1:
            resolution = torch.Tensor([self.mapping_size]).to(mappings.pixels.device)
            coords = mappings.pixels / (resolution - 1)
            coords = coords[:, [1, 0]]       # pixel mappings are (W, H)
            batch = mappings.feature_map_indexing[0]
            return pooling.sparse_interpolation(self.x, coords, batch)
        return pooling.gather_features(self.x, mappings)

    def __repr__(self):
        return f"{self.__class__.__name__}(num_views={self.num_views}, " \
               f"num_points={self.num_points}, device={self.device})"


class ImageData:
    """List of SameSettingImageData with different settings
    (core/multimodal/image.py:1419-1606), as far as transforms need it."""

    def __init__(self, image_list):
        self._list = list(image_list)

    def __len__(self):
        return len(self._list)

    def __getitem__(self, i):
        return self._list[i]

    def __iter__(self):
        return iter(self._list)

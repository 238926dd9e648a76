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
from .mapping import ImageMapping, ImageMappingBatch, lexunique


class SameSettingImageData:

    def __init__(self, path=None, pos=torch.empty([0, 3]), extrinsic=None,
                 fx=None, fy=None, mx=None, my=None, ref_size=(512, 256),
                 proj_upscale=2, downscale=1, crop_size=None, crop_offsets=None,
                 x=None, mappings=None, mask=None, visibility=None, opk=None,
                 xi=None, k1=None, k2=None, gamma1=None, gamma2=None, u0=None, v0=None,
                 rollings=None, **kwargs):
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
        self._downscale = downscale
        self.crop_size = tuple(crop_size) if crop_size is not None else self.ref_size
        self.crop_offsets = crop_offsets if crop_offsets is not None \
            else torch.zeros((n, 2), dtype=torch.int64, device=self.pos.device)
        # horizontal roll of every (circular) image with respect to ref_size, int64 [V]
        self.rollings = rollings.long() if rollings is not None \
            else torch.zeros(n, dtype=torch.int64, device=self.pos.device)
        self._x = None
        self.x = x
        self.mappings = mappings
        self.mask = mask
        self.visibility = visibility

    @property
    def downscale(self):
        """Scale of the current feature map with respect to ``crop_size``
        (core/multimodal/image.py:722-744)."""
        return self._downscale

    @downscale.setter
    def downscale(self, scale):
        assert (self._x is None and getattr(self, 'mappings', None) is None) \
            or self._downscale == scale, \
            "Can't directly edit 'downscale' if 'x' or 'mappings' are not both None. " \
            "Setting 'x' will automatically adjust the scale."
        assert scale >= 1, f"Expected scalar larger than 1 but got {scale} instead."
        self._downscale = scale

    @property
    def x(self):
        """Feature maps [V, C, H, W] (core/multimodal/image.py:746-787)."""
        return self._x

    @x.setter
    def x(self, x):
        """Setting feature maps of another resolution adjusts ``downscale`` (the
        larger of the two axis ratios), so that the mappings can follow."""
        if x is None:
            self._x = None
            return
        assert isinstance(x, torch.Tensor) and x.shape[0] == self.num_views, \
            f"Expected a tensor of shape ({self.num_views}, :, {self.img_size[1]}, " \
            f"{self.img_size[0]}) but got {tuple(x.shape)} instead."
        scale = max(self.img_size[0] / x.shape[3], self.img_size[1] / x.shape[2])
        self._downscale = self._downscale * scale
        self._x = x

    def update_cropping(self, crop_size, crop_offsets):
        """Crop the feature maps and the mappings with respect to the current
        ``img_size``; ``crop_size`` / ``crop_offsets`` are kept with respect to
        ``ref_size`` (core/multimodal/image.py:688-720).  In place."""
        crop_offsets = crop_offsets.long()
        self.crop_size = tuple(int(v * self.downscale) for v in crop_size)
        self.crop_offsets = (self.crop_offsets + crop_offsets.to(self.crop_offsets.device)
                             * self.downscale).long()
        if self._x is not None:
            self._x = torch.cat([
                im[:, o[1]:o[1] + crop_size[1], o[0]:o[0] + crop_size[0]].unsqueeze(0)
                for im, o in zip(self._x, crop_offsets.tolist())])
        if self.mappings is not None:
            self.mappings = self.mappings.crop(crop_size, crop_offsets)
        return self

    def update_rollings(self, rollings):
        """Roll the feature maps and the mappings along the width, for images
        with a circular representation (first and last column adjacent);
        ``rollings`` [V] int64 is kept with respect to ``ref_size``.  Not after a
        crop along the width or a resize (core/multimodal/image.py:578-627).
        In place."""
        who = "CenterRoll cannot operate if images and mappings underwent prior cropping or resizing."
        assert self.ref_size[0] == self.img_size[0], who
        assert self.crop_size is None or tuple(self.crop_size) == tuple(self.ref_size), who
        assert self.downscale is None or self.downscale == 1, who
        self.rollings = rollings.long().to(self.device)
        if self._x is not None:
            self._x = torch.stack([torch.roll(im, int(r), dims=-1) for im, r in zip(self._x, self.rollings)])
        if self.mappings is not None:
            m = self.mappings
            per_pixel = self.rollings.to(m.images.device)[m.images].repeat_interleave(
                m.values[1].pointers[1:] - m.values[1].pointers[:-1])
            w = (m.pixels[:, 0].long() + per_pixel) % self.ref_size[0]
            m.pixels[:, 0] = w.to(self.pixel_dtype)
        return self

    # the reference exposes these as computed properties
    @property
    def num_views(self):
        return self.pos.shape[0]

    @property
    def device(self):
        return self.pos.device

    _fisheye_keys = ('xi', 'k1', 'k2', 'gamma1', 'gamma2', 'u0', 'v0')
    _pinhole_keys = ('fx', 'fy', 'mx', 'my')
    # per-view attributes (concatenated by a batch) / attributes all views share
    _torch_keys = ('pos', 'opk', 'extrinsic', 'crop_offsets', 'rollings') + _pinhole_keys + _fisheye_keys
    _shared_keys = ('ref_size', 'proj_upscale', 'downscale', 'crop_size', 'mask', 'visibility')

    def to_dict(self):
        """Constructor arguments of self (core/multimodal/image.py:395-397)."""
        keys = ('path',) + self._torch_keys + self._shared_keys + ('mappings', 'x')
        return {k: getattr(self, k) for k in keys}

    @property
    def settings_hash(self):
        """Identifier of the settings every view shares, mask and visibility
        model aside: only images with equal hashes can be batched together
        (core/multimodal/image.py:1217-1229)."""
        keys = ('ref_size', 'proj_upscale', 'downscale', 'crop_size')
        return hash(tuple(getattr(self, k) for k in keys))

    @staticmethod
    def get_batch_type():
        return SameSettingImageBatch

    def clone(self):
        """Shallow copy, except for x and mappings (image.py:1174-1183)."""
        out = copy.copy(self)
        out._x = self._x.clone() if self._x is not None else None
        out.mappings = self.mappings.clone() if self.mappings is not None else None
        return out

    def to(self, device):
        out = copy.copy(self)
        for k in self._torch_keys + ('_x', 'mappings', 'mask'):
            v = getattr(self, k)
            setattr(out, k, v.to(device) if v is not None else None)
        return out

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
        sel = lambda t: t[idx.to(t.device)] if t is not None else None
        return self.__class__(
            path=self.path[np.asarray(idx.cpu())], pos=sel(self.pos),
            extrinsic=sel(self.extrinsic), fx=sel(self.fx), fy=sel(self.fy),
            mx=sel(self.mx), my=sel(self.my), ref_size=copy.deepcopy(self.ref_size),
            proj_upscale=copy.deepcopy(self.proj_upscale),
            downscale=copy.deepcopy(self.downscale),
            crop_size=copy.deepcopy(self.crop_size), crop_offsets=sel(self.crop_offsets),
            rollings=sel(self.rollings), x=sel(self.x),
            mappings=self.mappings.select_images(idx) if self.mappings is not None else None,
            mask=self.mask.clone() if self.mask is not None else None,
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

    def _without_mappings(self, idx=None):
        """self[idx] (or a clone) built while the mappings are set aside: the
        caller installs re-indexed mappings itself."""
        kept, self.mappings = self.mappings, None
        try:
            return self.clone() if idx is None else self[idx]
        finally:
            self.mappings = kept

    def select_points(self, idx, mode='pick'):
        """Follow a new 3D sampling of the points (core/multimodal/image.py:
        826-907): 'pick' keeps the points ``idx`` and drops the images no kept
        point maps to; 'merge' agglomerates point i into ``idx[i]`` and keeps
        every image.  Returns a new object."""
        idx = tensor_idx(idx).to(self.device)
        if self.mappings is None or idx is None or idx.shape[0] == 0:
            return self.clone()
        if len(self) == 0:
            return self.clone()
        if mode == 'pick':
            mappings = self.mappings.select_points(idx, mode=mode)
            seen_image_idx = lexunique(mappings.images) if mappings.num_items > 0 \
                else torch.empty(0, dtype=torch.int64)
            images = self._without_mappings(seen_image_idx)
            images.mappings = mappings.select_images(seen_image_idx)
        elif mode == 'merge':
            images = self.clone()
            if not idx.shape[0] == self.num_points > 0:
                return images
            if not torch.arange(int(idx.max()) + 1, device=idx.device).equal(idx.unique()):
                return images
            images.mappings = images.mappings.select_points(idx, mode=mode)
        else:
            raise ValueError(f"Unknown point selection mode '{mode}'.")
        return images

    def select_views(self, view_mask):
        """Keep the views of the boolean ``view_mask`` over the view-level rows
        of the mappings; images left without a view are dropped
        (core/multimodal/image.py:909-950).  Returns a new object."""
        if self.mappings is None or view_mask is None or torch.all(view_mask) or len(self) == 0:
            return self.clone()
        mappings, seen_image_idx = self.mappings.select_views(view_mask)
        images = self._without_mappings(seen_image_idx)
        images.mappings = mappings
        return images

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


class SameSettingImageBatch(SameSettingImageData):
    """Concatenation of SameSettingImageData items that refer to different
    point clouds: per-view attributes and feature maps are concatenated, the
    mappings become an ImageMappingBatch whose point ids follow the order of
    the items (core/multimodal/image.py:1300-1416)."""

    def __init__(self, **kwargs):
        super(SameSettingImageBatch, self).__init__(**kwargs)
        self.__sizes__ = None

    @property
    def batch_pointers(self):
        return np.cumsum(np.concatenate(([0], self.__sizes__))) \
            if self.__sizes__ is not None else None

    @property
    def batch_items_sizes(self):
        return self.__sizes__ if self.__sizes__ is not None else None

    @property
    def num_batch_items(self):
        return len(self.__sizes__) if self.__sizes__ is not None else None

    @staticmethod
    def from_data_list(image_data_list):
        assert isinstance(image_data_list, list) and len(image_data_list) > 0
        assert all(isinstance(x, SameSettingImageData) for x in image_data_list)
        first = image_data_list[0]
        hash_ref = first.settings_hash
        assert all(im.settings_hash == hash_ref for im in image_data_list), \
            f"All SameSettingImageData values for shared keys " \
            f"{SameSettingImageData._shared_keys} must be the same (except for the 'mask')."
        # shared attributes (and the mask) come from the first item
        batch_dict = {k: getattr(first, k) for k in SameSettingImageData._shared_keys}
        batch_dict['path'] = np.concatenate([im.path for im in image_data_list])
        for key in SameSettingImageData._torch_keys:
            vals = [getattr(im, key) for im in image_data_list]
            batch_dict[key] = None if any(v is None for v in vals) else torch.cat(vals)
        xs = [im.x for im in image_data_list]
        batch_dict['x'] = None if any(x is None for x in xs) else torch.cat(xs)
        maps = [im.mappings for im in image_data_list]
        batch_dict['mappings'] = None if any(m is None for m in maps) \
            else ImageMappingBatch.from_csr_list(maps)
        batch = SameSettingImageBatch(**batch_dict)
        batch.__sizes__ = np.array([im.num_views for im in image_data_list])
        return batch

    def to_data_list(self):
        if self.__sizes__ is None:
            raise RuntimeError(
                'Cannot reconstruct image data list from batch because the batch object was '
                'not created using `SameSettingImageBatch.from_data_list()`.')
        ptr = self.batch_pointers
        items = []
        for i in range(self.num_batch_items):
            idx = torch.arange(int(ptr[i]), int(ptr[i + 1]))
            item = SameSettingImageData.__getitem__(self, idx)
            items.append(SameSettingImageData(**item.to_dict()))
        return items


class ImageData:
    """List of SameSettingImageData with different settings
    (core/multimodal/image.py:1419-1606), as far as transforms need it."""

    def __init__(self, image_list):
        self._list = list(image_list)

    @property
    def num_points(self):
        """Number of points the mappings refer to (0 when there is no image)."""
        return self._list[0].num_points if len(self._list) > 0 else 0

    @property
    def num_views(self):
        return sum(im.num_views for im in self._list)

    @staticmethod
    def get_batch_type():
        return ImageBatch

    def select_points(self, idx, mode='pick'):
        """core/multimodal/image.py:1501-1503"""
        return self.__class__([im.select_points(idx, mode=mode) for im in self])

    def select_views(self, view_mask_list):
        """core/multimodal/image.py:1505-1511"""
        assert isinstance(view_mask_list, list) and len(view_mask_list) == len(self)
        return self.__class__([im.select_views(m) for im, m in zip(self, view_mask_list)])

    def __len__(self):
        return len(self._list)

    def __getitem__(self, i):
        return self._list[i]

    def __iter__(self):
        return iter(self._list)


class ImageBatch(ImageData):
    """Batch of ImageData: images with equal settings are concatenated into one
    SameSettingImageBatch each, and every mapping is re-based on the
    concatenated point cloud of all items (core/multimodal/image.py:1608-1735).
    Settings groups are kept in order of first appearance."""

    def __init__(self, image_list):
        super(ImageBatch, self).__init__(image_list)
        self.__il_sizes__ = None
        self.__hashes__ = None
        self.__il_idx_dict__ = None
        self.__im_idx_dict__ = None
        self.__cum_pts__ = None

    @staticmethod
    def from_data_list(image_data_list):
        assert isinstance(image_data_list, list) and len(image_data_list) > 0
        assert all(isinstance(x, ImageData) for x in image_data_list)
        hashes = []
        for il in image_data_list:
            for im in il:
                if im.settings_hash not in hashes:
                    hashes.append(im.settings_hash)
        n_pts = torch.LongTensor([il.num_points for il in image_data_list])
        cum_pts = torch.cumsum(torch.cat((torch.LongTensor([0]), n_pts)), dim=0)
        il_sizes = [len(il) for il in image_data_list]
        il_idx_dict = {h: [] for h in hashes}
        im_idx_dict = {h: [] for h in hashes}
        groups = {h: [] for h in hashes}
        for il_idx, il in enumerate(image_data_list):
            for im_idx, im in enumerate(il):
                h = im.settings_hash
                il_idx_dict[h].append(il_idx)
                im_idx_dict[h].append(im_idx)
                groups[h].append(im)
        batches = [SameSettingImageBatch.from_data_list(groups[h]) for h in hashes]
        # mappings of a settings group only know the points of the items that
        # hold such images: insert the other items' points as empty groups
        for h, im in zip(hashes, batches):
            if im.num_points > 0:
                global_idx = torch.cat(
                    [torch.arange(int(cum_pts[i]), int(cum_pts[i + 1])) for i in il_idx_dict[h]])
                im.mappings.insert_empty_groups(
                    global_idx.to(im.mappings.device), num_groups=int(cum_pts[-1]))
        out = ImageBatch(batches)
        out.__il_sizes__ = il_sizes
        out.__hashes__ = hashes
        out.__il_idx_dict__ = il_idx_dict
        out.__im_idx_dict__ = im_idx_dict
        out.__cum_pts__ = cum_pts
        return out

    def to_data_list(self):
        assert self.__il_sizes__ is not None, \
            "Cannot reconstruct the list of ImageData because the ImageBatch was not " \
            "created using 'ImageBatch.from_data_list'."
        out = [[None] * s for s in self.__il_sizes__]
        for h, ib in zip(self.__hashes__, self):
            for il_idx, im_idx, im in zip(
                    self.__il_idx_dict__[h], self.__im_idx_dict__[h], ib.to_data_list()):
                start, end = int(self.__cum_pts__[il_idx]), int(self.__cum_pts__[il_idx + 1])
                im.mappings = im.mappings[torch.arange(start, end)]
                out[il_idx][im_idx] = im
        return [ImageData(x) for x in out]

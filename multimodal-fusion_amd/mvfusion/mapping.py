"""``ImageMapping``: the point -> view -> pixel nested CSR produced by
``MapImages`` and consumed by the pools, with the interface of
``torch_points3d/core/multimodal/image.py:1717-1909``.

Layout (all on one device):
``pointers`` int64 [N+1] point -> views; ``values[0]`` = ``images`` int64 [Vw];
``values[1]`` = CSRData{``pointers`` int64 [Vw+1] view -> pixels, ``values[0]`` =
``pixels`` [P, 2] (x, y)}; ``values[2]`` = ``features`` float32 [Vw, F].
"""
import numpy as np
import torch

from .csr import CSRData, CSRBatch, tensor_idx
from . import pooling as _pooling


def composite_key(*cols):
    """Lexicographic composite of non-negative int64 columns
    (CompositeTensor, utils/multimodal.py:97-153)."""
    cols = [c.to(torch.int64) for c in cols]
    if cols[0].shape[0] == 0:
        return cols[0].clone()
    maxes = [int(c.abs().max()) + 1 for c in cols]
    total = 1
    for m in maxes:
        total *= m
    assert total < torch.iinfo(torch.int64).max, \
        'The dtype of at least one of the input tensors must allow the ' \
        'composite computation.'
    key = torch.zeros_like(cols[0])
    for i, c in enumerate(cols):
        base = 1
        for m in maxes[i + 1:]:
            base *= m
        key = key + c * base
    return key


def lexargsort(*cols):
    """Indices sorting the columns lexicographically.  Ties keep their input
    order (the reference's sort is unstable: any tie order is valid there)."""
    return torch.sort(composite_key(*cols), stable=True).indices


def lexargunique(*cols):
    """Index of the first occurrence of each distinct row, rows in
    lexicographic order (cpu_lex_op 'argunique', utils/multimodal.py:308-312)."""
    key = composite_key(*cols)
    if key.shape[0] == 0:
        return key
    order = torch.sort(key, stable=True).indices
    ks = key[order]
    first = torch.ones_like(ks, dtype=torch.bool)
    first[1:] = ks[1:] != ks[:-1]
    return order[first]


def lexunique(*cols):
    """Distinct rows of the columns in lexicographic order: one tensor for one
    column, a list of columns otherwise (cpu_lex_op 'unique',
    utils/multimodal.py:313-317)."""
    if len(cols) == 1:
        return torch.unique(cols[0], sorted=True)
    first = lexargunique(*cols)
    return [c[first] for c in cols]


def _csr_from_dense_rows(point_ids, image_ids, pixels, features, num_points, num_images,
                         dedupe=False, row2view=None):
    """Nested CSR arrays from dense rows on the device (``mvf_mapping_from_dense``:
    one key per row, one radix sort).  ``point_ids`` / ``image_ids`` / ``features``
    are per row, or per source view when ``row2view`` (int32 [rows]) is given.
    Returns pointers, images, view pointers, pixels, features (or None)."""
    import ctypes as C  # noqa: F401
    from . import _lib as L
    dev = pixels.device
    lib = L.load()
    rows = pixels.shape[0]
    point_ids = point_ids.to(torch.int64).contiguous()
    image_ids = image_ids.to(torch.int64).contiguous()
    pixels = pixels.contiguous()
    F = 0 if features is None else int(features.shape[1])
    feats = None if features is None else features.to(torch.float32).contiguous()
    pointers = torch.empty(num_points + 1, dtype=torch.int64, device=dev)
    images = torch.empty(rows, dtype=torch.int64, device=dev)
    view_ptr = torch.empty(rows + 1, dtype=torch.int64, device=dev)
    out_pix = torch.empty((rows, 2), dtype=torch.int16, device=dev)
    out_feat = torch.empty((rows, F), dtype=torch.float32, device=dev) if F else None
    counters = torch.zeros(2, dtype=torch.int64, device=dev)
    ws = L.workspace(lib.mvf_mapping_from_dense_workspace_bytes(rows, num_points), dev)
    with torch.cuda.device(dev):
        L.check(lib.mvf_mapping_from_dense(
            L.ptr(point_ids), L.ptr(image_ids), L.ptr(row2view), L.ptr(pixels), L.ptr(feats), F,
            rows, num_points, num_images, int(bool(dedupe)), L.ptr(pointers), L.ptr(images),
            L.ptr(view_ptr), L.ptr(out_pix), L.ptr(out_feat), L.ptr(counters), L.ptr(ws),
            ws.numel(), L.stream_ptr(dev)))
    n_views, n_pix = counters.tolist()
    return (pointers, images[:n_views], view_ptr[:n_views + 1], out_pix[:n_pix],
            out_feat[:n_views] if F else None)


def _device_assembly_fits(pixels, rows, num_points, num_images):
    return (pixels.is_cuda and pixels.dtype == torch.int16 and 0 < rows < 2 ** 32 - 1
            and num_points * num_images < 2 ** 32)


class ImageMapping(CSRData):

    @staticmethod
    def from_dense(point_ids, image_ids, pixels, features, num_points=None):
        """Build the nested CSR from one row per (point, image, pixel)
        (core/multimodal/image.py:1738-1805).  Pixels of one (point, image)
        group come out in (x, y) order."""
        assert point_ids.ndim == 1, 'point_ids and image_ids must be 1D tensors'
        assert point_ids.shape == image_ids.shape, \
            'point_ids and image_ids must have the same shape'
        assert point_ids.shape[0] == pixels.shape[0], \
            'pixels and indices must have the same shape'
        assert features is None or point_ids.shape[0] == features.shape[0], \
            'point_ids and features must have the same shape'
        in_device = point_ids.device
        if not in_device.type == 'cuda':
            from . import _lib as L
            dev = L.require_device()
            point_ids, image_ids, pixels = point_ids.to(dev), image_ids.to(dev), pixels.to(dev)
            features = features.to(dev) if features is not None else None

        rows = point_ids.shape[0]
        if rows > 0 and pixels.dtype == torch.int16 and (features is None or features.dim() == 2):
            # one launch set on the device: key per row, radix sort, CSR assembly
            # one read-back: the ranges of the ids and of the pixel coordinates.  Ids and
            # coordinates are unsigned fields of the sort keys (here and, through
            # max(abs) + 1 digits, in the composite keys of the reference, utils/multimodal.py:13-48)
            lims = torch.stack([*torch.aminmax(point_ids), *torch.aminmax(image_ids),
                                pixels.min().to(torch.int64)]).tolist()
            p_min, p_max, i_min, i_max, pix_min = (int(v) for v in lims)
            if min(p_min, i_min, pix_min) < 0:
                raise ValueError("ImageMapping.from_dense: point ids, image ids and pixel coordinates "
                                 f"must not be negative (minima {p_min}, {i_min}, {pix_min})")
            n_pts = max(int(num_points or 0), p_max + 1)
            if _device_assembly_fits(pixels, rows, n_pts, i_max + 1):
                arrays = _csr_from_dense_rows(
                    point_ids, image_ids, pixels, features, n_pts, i_max + 1)
                feats = arrays[4]
                if feats is not None and features.dtype != feats.dtype and features.dtype.is_floating_point:
                    feats = feats.to(features.dtype)       # computed in float32, returned as given
                return ImageMapping.from_csr_arrays(
                    arrays[0], arrays[1], arrays[3], feats, arrays[2]).to(in_device)

        order = lexargsort(point_ids, image_ids, pixels[:, 0], pixels[:, 1])
        point_ids, image_ids, pixels = point_ids[order], image_ids[order], pixels[order]
        if features is not None:
            features = features[order]

        view_ids = composite_key(point_ids, image_ids)
        image_pixel_mappings = CSRData(view_ids, pixels, dense=True)
        last = image_pixel_mappings.pointers[1:] - 1
        image_ids = image_ids[last]
        point_ids = point_ids[last]
        if features is not None:
            in_dtype = features.dtype
            features = _pooling.segment_csr(
                features.float(), image_pixel_mappings.pointers, reduce='mean')
            if in_dtype.is_floating_point and in_dtype != features.dtype:
                features = features.to(in_dtype)
            mapping = ImageMapping(
                point_ids, image_ids, image_pixel_mappings, features, dense=True,
                is_index_value=[True, False, False])
        else:
            mapping = ImageMapping(
                point_ids, image_ids, image_pixel_mappings, dense=True,
                is_index_value=[True, False])

        if num_points is None or num_points < point_ids.max() + 1:
            num_points = int(point_ids.max()) + 1
        point_ids = point_ids[mapping.pointers[1:] - 1]
        mapping = mapping.insert_empty_groups(point_ids, num_groups=num_points)
        return mapping.to(in_device)

    @staticmethod
    def from_csr_arrays(pointers, images, pixels, features, view_pointers=None):
        """Wrap already-built CSR arrays (the fused MapImages path)."""
        if view_pointers is None:
            view_pointers = torch.arange(
                images.shape[0] + 1, dtype=torch.int64, device=images.device)
        nested = CSRData(view_pointers, pixels, dense=False)
        if features is None:
            return ImageMapping(pointers, images, nested, dense=False,
                                is_index_value=[True, False])
        return ImageMapping(pointers, images, nested, features, dense=False,
                            is_index_value=[True, False, False])

    def _from_kept_views(self, keep, values):
        """Mapping made of the views selected by ``keep`` (index or mask over
        the view-level rows): pointers rebuilt from the surviving views' point
        ids, empty points re-inserted (core/multimodal/image.py:2084-2099)."""
        sizes = self.pointers[1:] - self.pointers[:-1]
        point_ids = torch.arange(self.num_groups, device=self.device).repeat_interleave(sizes)
        point_ids = point_ids[keep]
        pointers = CSRData._sorted_indices_to_pointers(point_ids)
        out = self.__class__(pointers, *values, dense=False, is_index_value=self.is_index_value)
        point_ids = point_ids[out.pointers[1:] - 1]
        return out.insert_empty_groups(point_ids, num_groups=self.num_groups)

    def select_images(self, idx):
        """Keep the mappings to the images ``idx`` (no duplicates) and renumber
        them idx[i] -> i (core/multimodal/image.py:2039-2103)."""
        idx = tensor_idx(idx).to(self.device)
        assert idx.unique().numel() == idx.shape[0], "Index must not contain duplicates."
        if self.num_items == 0:
            return self.clone()
        view_idx = torch.where((self.images[..., None] == idx).any(-1))[0]
        values = [val[view_idx] for val in self.values]
        if idx.shape[0] == 0:
            return self.__class__(torch.zeros_like(self.pointers), *values, dense=False,
                                  is_index_value=self.is_index_value)
        idx_gen = torch.full((int(idx.max()) + 1,), -1, dtype=torch.int64, device=self.device)
        idx_gen = idx_gen.scatter_(0, idx, torch.arange(idx.shape[0], device=self.device))
        values[0] = idx_gen[values[0]]
        if view_idx.shape[0] == 0:
            return self.__class__(torch.zeros_like(self.pointers), *values, dense=False,
                                  is_index_value=self.is_index_value)
        return self._from_kept_views(view_idx, values)

    def select_views(self, view_mask):
        """Keep the views of the boolean ``view_mask`` and renumber the images
        that are left; returns (mapping, kept image indices or None when all
        images remain) (core/multimodal/image.py:2105-2175)."""
        if isinstance(view_mask, np.ndarray):
            view_mask = torch.from_numpy(view_mask)
        assert view_mask.dtype == torch.bool and view_mask.dim() == 1 \
            and view_mask.shape[0] == self.num_items, \
            f"view_mask must be a torch.BoolTensor of size {self.num_items}."
        if self.num_items == 0:
            return self.clone()
        view_mask = view_mask.to(self.device)
        values = [val[view_mask] for val in self.values]
        if not torch.any(view_mask):
            out = self.__class__(torch.zeros_like(self.pointers), *values, dense=False,
                                 is_index_value=self.is_index_value)
            return out, torch.LongTensor([])
        img_idx = values[0].unique()
        if img_idx.numel() < int(self.images.max()) + 1:
            idx_gen = torch.full((int(img_idx.max()) + 1,), -1, dtype=torch.int64,
                                 device=self.device)
            idx_gen = idx_gen.scatter_(
                0, img_idx, torch.arange(img_idx.shape[0], device=self.device))
            values[0] = idx_gen[values[0]]
        else:
            img_idx = None
        return self._from_kept_views(view_mask, values), img_idx

    def select_points(self, idx, mode='pick'):
        """Update the 3D points sampling (core/multimodal/image.py:2177-2287):
        'pick' keeps the points ``idx``; 'merge' agglomerates point i into point
        ``idx[i]``, combining the mappings and dropping duplicate pixels (the
        first row of each duplicate in (point, image, x, y) order keeps its
        view-averaged features).  Returns a new ImageMapping."""
        MODES = ['pick', 'merge']
        assert mode in MODES, \
            f"Unknown mode '{mode}'. Supported modes are {MODES}."
        idx = tensor_idx(idx).to(self.device)
        if idx is None or idx.shape[0] == 0 or self.num_groups == 0:
            return self.clone()
        if self.num_items == 0:
            out = self.clone()
            out.pointers = torch.zeros(idx.shape[0] + 1, dtype=torch.int64, device=self.device)
            return out
        if mode == 'pick':
            return self[idx]

        if not idx.shape[0] == self.num_groups > 0:
            return self.clone()
        if not torch.arange(int(idx.max()) + 1, device=self.device).equal(idx.unique()):
            return self.clone()

        n_new = int(idx.max()) + 1
        i_max, i_min, pix_min = torch.stack(
            [self.images.max(), self.images.min(), self.pixels.min().to(torch.int64)]).tolist()
        n_img = int(i_max) + 1
        # (the device keys hold image ids and pixel coordinates as unsigned fields)
        if min(int(i_min), int(pix_min)) >= 0 \
                and _device_assembly_fits(self.pixels, self.pixels.shape[0], n_new, n_img) \
                and self.num_items < 2 ** 31:
            return self._merge_points_device(idx, n_new, n_img)

        # dense view-level format
        view_sizes = self.pointers[1:] - self.pointers[:-1]
        point_ids = idx.repeat_interleave(view_sizes)
        image_ids = self.images
        if not self.has_features:
            features = None
        elif self.num_items <= 1:
            features = self.features
        else:
            # mean of the features of all views that land on the same (point,
            # image) pair, redistributed to the source views; rows are summed in
            # their input order (a stable sort feeds the segment kernel)
            view_ids = composite_key(point_ids, image_ids)
            order = torch.sort(view_ids, stable=True).indices
            ks = view_ids[order]
            first = torch.ones_like(ks, dtype=torch.bool)
            first[1:] = ks[1:] != ks[:-1]
            starts = torch.where(first)[0]
            ptr = torch.cat([starts, torch.tensor([ks.shape[0]], device=ks.device)])
            mean = _pooling.segment_csr(self.features.float()[order].contiguous(), ptr, reduce='mean')
            group = torch.cumsum(first.to(torch.int64), 0) - 1
            features = torch.empty_like(mean[group])
            features[order] = mean[group]
        # dense atomic-level format
        pixel_sizes = self.values[1].pointers[1:] - self.values[1].pointers[:-1]
        point_ids = point_ids.repeat_interleave(pixel_sizes)
        image_ids = image_ids.repeat_interleave(pixel_sizes)
        if features is not None:
            features = features.repeat_interleave(pixel_sizes, dim=0)
        pixels = self.pixels
        keep = lexargunique(point_ids, image_ids, pixels[:, 0], pixels[:, 1])
        return ImageMapping.from_dense(
            point_ids[keep], image_ids[keep], pixels[keep],
            features[keep] if features is not None else None, num_points=n_new)

    def _merge_points_device(self, idx, n_new, n_img):
        """select_points(mode='merge') on the device: per view the new point id and
        the feature mean over the views that land on the same (point, image) pair
        (``mvf_group_mean``), then one ``mvf_mapping_from_dense`` pass over the
        pixels, which reads the view-level arrays through the pixels' view index
        and drops duplicate pixels."""
        from . import _lib as L
        from . import dense as _dense
        dev = self.device
        lib = L.load()
        n_views = self.num_items
        view_point = _dense.csr_row_segments(self.pointers.contiguous(), n_views)       # int32 [Vw]
        new_point = idx.to(torch.int64)[view_point.long()].contiguous()
        images = self.images.to(torch.int64).contiguous()
        feats = None
        if self.has_features:
            feats = self.features.to(torch.float32).contiguous()
            if n_views > 1:
                keys = (new_point * n_img + images).contiguous()
                out = torch.empty_like(feats)
                ws = L.workspace(lib.mvf_mapping_from_dense_workspace_bytes(n_views, 1), dev)
                with torch.cuda.device(dev):
                    L.check(lib.mvf_group_mean(
                        L.ptr(keys), L.ptr(feats), feats.shape[1], n_views, n_new * n_img, L.ptr(out),
                        L.ptr(ws), ws.numel(), L.stream_ptr(dev)))
                feats = out
        view_ptr = self.values[1].pointers.contiguous()
        row2view = _dense.csr_row_segments(view_ptr, self.pixels.shape[0])
        arrays = _csr_from_dense_rows(new_point, images, self.pixels, feats, n_new, n_img,
                                      dedupe=True, row2view=row2view)
        return ImageMapping.from_csr_arrays(arrays[0], arrays[1], arrays[3], arrays[4], arrays[2])

    def rescale_images(self, ratio):
        """Mappings at a new image resolution: downscaled when ``ratio < 1``,
        upscaled otherwise (core/multimodal/image.py:1911-1924).  Returns a new
        ImageMapping."""
        if ratio < 1:
            return self.downscale_images(1 / ratio)
        return self.upscale_images(ratio)

    def _with_pixels(self, pixels):
        """Copy of self whose nested pixel CSR holds ``pixels`` (the pointer and
        value tensors of the other levels are shared)."""
        out = self.clone()
        nested = self.values[1].clone()
        nested.values[0] = pixels
        out.values[1] = nested
        return out

    def downscale_images(self, ratio):
        """Pixel coordinates floor-divided by ``ratio >= 1``
        (core/multimodal/image.py:1926-1990).  The reference de-duplicates on
        (atomic row id, x, y); row ids are distinct, so every row is kept and
        the nested pointers do not change."""
        assert ratio >= 1, \
            f"Invalid image subsampling ratio: {ratio}. Must be larger than 1."
        if ratio == 1:
            return self.clone()
        pix = self.pixels
        return self._with_pixels((pix // ratio).long().type(pix.dtype))

    def upscale_images(self, ratio, center=True):
        """Pixel coordinates multiplied by ``ratio >= 1``; with ``center`` the
        high-resolution pixel closest to the centre of the low-resolution one
        (core/multimodal/image.py:1992-2037).  Unlike the reference, whose
        shallow clone shares the nested pixel CSR with its source and therefore
        rescales the source too, the result here is independent of ``self``."""
        assert ratio >= 1, \
            f"Invalid image upsampling ratio: {ratio}. Must be larger than 1."
        if ratio == 1:
            return self.clone()
        pix = self.pixels
        if center:
            up = (pix.float() * ratio + ratio / 2).long()
        else:
            up = (pix.float() * ratio).long()
        return self._with_pixels(up.type(pix.dtype))

    def crop(self, crop_size, crop_offsets):
        """Pixel coordinates after cropping every image at its own offset
        [n_images, 2] to a common ``crop_size`` (W, H); mappings that fall
        outside are dropped and the nested CSR is rebuilt
        (core/multimodal/image.py:2289-2352).  As in the reference, the pixels
        take the integer type of the offsets, and when *no* pixel is left inside
        the box the mappings are returned shifted but otherwise whole."""
        assert tuple(crop_offsets.shape) == (self.images.unique().numel(), 2), \
            f"Expected crop_offsets to have shape {(self.images.unique().numel(), 2)} " \
            f"but got shape {tuple(crop_offsets.shape)} instead."
        sizes = self.values[1].pointers[1:] - self.values[1].pointers[:-1]
        image_ids = self.images.repeat_interleave(sizes)
        pixels = self.pixels - crop_offsets.to(self.device)[image_ids]
        box = torch.tensor(list(crop_size), device=self.device)
        inside = torch.where((pixels >= 0).all(dim=1) & (pixels < box).all(dim=1))[0]
        if inside.shape[0] == 0:
            return self._with_pixels(pixels)
        point_ids = torch.arange(self.num_groups, device=self.device).repeat_interleave(
            self.pointers[1:] - self.pointers[:-1]).repeat_interleave(sizes)
        features = self.features.repeat_interleave(sizes, dim=0)[inside] \
            if self.has_features else None
        kept = pixels[inside]
        if kept.is_cuda and max(crop_size) < 2 ** 15 and kept.dtype != torch.int16:
            # the kept pixels lie inside the box: 16 bits hold them, which is what
            # the device CSR assembly sorts on
            out = ImageMapping.from_dense(point_ids[inside], image_ids[inside], kept.to(torch.int16),
                                          features, num_points=self.num_groups)
            return out._with_pixels(out.pixels.to(kept.dtype))
        return ImageMapping.from_dense(
            point_ids[inside], image_ids[inside], kept, features, num_points=self.num_groups)

    def debug(self):
        super(ImageMapping, self).debug()
        assert len(self.values) == 2 or self.has_features
        assert isinstance(self.values[1], CSRData)
        assert len(self.values[1].values) == 1

    @property
    def points(self):
        return torch.arange(self.num_groups, device=self.device)

    @property
    def images(self):
        return self.values[0]

    @images.setter
    def images(self, images):
        self.values[0] = images.to(self.device)

    @property
    def has_features(self):
        return len(self.values) == 3

    @property
    def features(self):
        return self.values[2] if self.has_features else None

    @features.setter
    def features(self, features):
        if self.has_features:
            if features is None:
                self.values.pop(-1)
            else:
                self.values[2] = features.to(self.device)
        elif features is not None:
            self.values.append(features.to(self.device))

    @property
    def pixels(self):
        return self.values[1].values[0]

    @pixels.setter
    def pixels(self, pixels):
        self.values[1].values[0] = pixels.to(self.device)

    @staticmethod
    def get_batch_type():
        return ImageMappingBatch

    def image_stats(self, num_images=None):
        """int64 [num_images, 5]: mapped pixels, x min, x max, y min, y max of
        every image id (zeros for an image without pixels).  ``num_images``
        defaults to the largest id present + 1.  On a HIP device this is one
        pass over the views (``mvf_image_stats``); on the host a scatter
        composition of the same reductions."""
        if num_images is None:
            num_images = int(self.images.max()) + 1 if self.images.numel() else 0
        view_ptr, pix = self.values[1].pointers, self.pixels
        stats = torch.zeros((num_images, 5), dtype=torch.int64, device=self.device)
        if num_images == 0 or self.images.numel() == 0:
            return stats
        if pix.is_cuda and pix.dtype in (torch.int16, torch.int64):
            from . import _lib as L
            pix = pix.contiguous()
            with torch.cuda.device(self.device):
                L.check(L.load().mvf_image_stats(
                    L.ptr(view_ptr.contiguous()), L.ptr(self.images.contiguous()), L.ptr(pix),
                    pix.element_size(), self.images.shape[0], num_images, L.ptr(stats),
                    L.stream_ptr(self.device)))
            return stats
        ids = self.images.repeat_interleave(view_ptr[1:] - view_ptr[:-1])
        keep = (ids >= 0) & (ids < num_images)
        ids, pix = ids[keep], pix[keep].long()
        stats[:, 0].index_add_(0, ids, torch.ones_like(ids))
        idx2 = ids[:, None].expand(-1, 2)
        lo = torch.zeros((num_images, 2), dtype=torch.int64, device=self.device).scatter_reduce_(
            0, idx2, pix, 'amin', include_self=False)
        hi = torch.zeros((num_images, 2), dtype=torch.int64, device=self.device).scatter_reduce_(
            0, idx2, pix, 'amax', include_self=False)
        stats[:, 1], stats[:, 2], stats[:, 3], stats[:, 4] = lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1]
        return stats

    @property
    def bounding_boxes(self):
        """(w_min, w_max, h_min, h_max) of the mapped pixels of every image, one
        entry per image id up to the largest one present; an id without
        mappings gets zeros (core/multimodal/image.py:1869-1879, whose
        torch_scatter reductions size their output by the largest index and fill
        empty slots with 0)."""
        stats = self.image_stats().to(self.pixels.dtype)
        return stats[:, 1], stats[:, 2], stats[:, 3], stats[:, 4]

    @property
    def feature_map_indexing(self):
        """Index tuple for ``X[idx]`` on ``X`` of shape [B, C, H, W]
        (core/multimodal/image.py:1881-1895)."""
        sizes = self.values[1].pointers[1:] - self.values[1].pointers[:-1]
        idx_batch = self.images.repeat_interleave(sizes)
        return (idx_batch.long(), ..., self.pixels[:, 1].long(), self.pixels[:, 0].long())

    @property
    def atomic_csr_indexing(self):
        """Pointers for the pixel -> view pooling (:1897-1902)."""
        return self.values[1].pointers

    @property
    def view_csr_indexing(self):
        """Pointers for the view -> point pooling (:1904-1909)."""
        return self.pointers


class ImageMappingBatch(ImageMapping, CSRBatch):
    """Batch of ImageMappings (core/multimodal/image.py:2355-2400)."""
    __csr_type__ = ImageMapping

"""``MapImages``: point -> image -> pixel mapping of one sample, with the
interface of ``torch_points3d/core/data_transform/multimodal/image.py``
(``ImageTransform`` :29-61, ``MapImages`` :162-428).

Two execution plans, same results as the reference's CPU path:

* fused (pinhole cameras, distinct mapping ids; ``SplattingVisibility`` with
  ``exact=True`` and integer ``proj_upscale`` / ``downscale`` -- the shipped
  KITTI-360 / WildScenes configuration --, with ``exact=False`` at the images'
  own resolution, or ``DepthBasedVisibility`` with decoded depth maps): every
  view of the sample goes through one launch set (``mvf_map_images_exact``,
  ``mvf_map_images_splat``, ``mvf_map_images_depth``) which also assembles the
  nested CSR on the device; nothing is sorted.
* per view (anything else the HIP kernels support): the reference's own loop --
  visibility model, pixel conversion, duplicate removal, ``from_dense`` -- with
  each step on the device.

Local point order.  The reference hands each view the points returned by an
sklearn ``KDTree(leaf_size=50).query_radius`` (transforms.py:301-405), i.e. in
the tree's traversal order; depth ties and shared centre pixels are resolved on
that order.  ``point_order='kdtree'`` (default) rebuilds that order on the host
for bit-identical results; ``point_order='index'`` uses the array order and
needs no host work; a precomputed permutation may be supplied as
``data.mvf_point_order``.
"""
import ctypes as C
import threading
import time

import numpy as np
import torch

from . import _lib as L
from . import visibility as visibility_module
from .images import ImageData, SameSettingImageData
from .mapping import ImageMapping, lexargunique, lexunique

MAPPING_KEY = 'mapping_index'
ORDER_KEY = 'mvf_point_order'


class ImageTransform:
    """Transforms on ``Data`` and ``ImageData`` / ``SameSettingImageData``
    (image.py:29-61)."""

    _PROCESS_IMAGE_DATA = False

    def _process(self, data, images):
        raise NotImplementedError

    def __call__(self, data, images):
        if isinstance(data, list):
            assert isinstance(images, list) and len(data) == len(images), \
                f"List(Data) items and List(SameSettingImageData) must " \
                f"have the same lengths."
            out = [self.__call__(da, im) for da, im in zip(data, images)]
            data_out, images_out = [list(x) for x in zip(*out)]
        elif isinstance(images, ImageData) and not self._PROCESS_IMAGE_DATA:
            out = [self.__call__(data.clone(), im) for im in images]
            images_out = ImageData([im for _, im in out])
            data_out = out[0][0] if len(out) > 0 else data
        else:
            data_out, images_out = self._process(data, images)
        return data_out, images_out

    def __repr__(self):
        attr_repr = ', '.join([f'{k}={v}' for k, v in self.__dict__.items()])
        return f'{self.__class__.__name__}({attr_repr})'


_NO_MAPPING_MSG = (
    "No mappings were found between the 3D points and any "
    "of the provided images. This will cause errors in the "
    "subsequent operations. Make sure your images are "
    "located in the vicinity of your point cloud and that "
    "the projection parameters allow for at least one "
    "point-image-pixel mapping before re-running this "
    "transformation.")


_PINNED = threading.local()


def _pinned(tag, shape, dtype):
    """A page-locked host buffer kept for the life of the process (one per purpose and
    size): freshly allocated host memory costs a page fault per 4 KB on first touch --
    several milliseconds for a few megabytes in a process that has mapped a GPU -- and
    page-locked memory also copies to and from the device at full rate.  The event is
    recorded after the last device copy out of the buffer: wait for it before writing."""
    cache = _PINNED.__dict__.setdefault('buffers', {})       # one set per calling thread
    key = (tag, tuple(shape), dtype)
    hit = cache.get(key)
    if hit is None:
        if len(cache) > 16:
            cache.clear()
        hit = cache[key] = [torch.empty(shape, dtype=dtype).pin_memory(), None]
    if hit[1] is not None:
        hit[1].synchronize()
        hit[1] = None
    return hit


def kdtree_order(pos, cylinder, threads=0, device=None):
    """Traversal order of the KD-tree the reference's radius samplers build
    (transforms.py:326,380): ``query_radius`` returns the points of that order
    that fall inside the radius.  Computed by the library's own host routine
    (``mvf_kdtree_order``: the same permutation as
    ``sklearn.neighbors.KDTree(leaf_size=50).idx_array``, without sklearn).
    ``device``: return the order as a tensor on that HIP device (staged through
    page-locked buffers, nothing allocated on the host per call); otherwise a host
    tensor."""
    p = pos.detach()
    n = p.shape[0]
    staged = device is not None and torch.device(device).type == 'cuda'
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=device if staged else None)
    if p.is_cuda:
        # through a page-locked buffer (a pageable copy allocates and faults in fresh memory)
        buf = _pinned('kd_pos', p.shape, torch.float32)[0]
        buf.copy_(p.to(torch.float32), non_blocking=True)
        torch.cuda.current_stream(p.device).synchronize()
        p = buf
    p = np.ascontiguousarray(np.asarray(p, dtype=np.float32))
    if staged:
        slot = _pinned('kd_order', (n,), torch.int64)
        order = np.asarray(slot[0])
    else:
        order = np.empty(n, dtype=np.int64)
    L.check(L.load().mvf_kdtree_order(
        p.ctypes.data, n, 2 if cylinder else 3, p.shape[1], 50, order.ctypes.data, int(threads)))
    if not staged:
        return torch.from_numpy(order)
    out = slot[0].to(device, non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))
    return out


def _source_key(tag, tensors):
    """What a staging cache remembers of its sources: the tensor objects
    themselves (so that their storage cannot be freed and handed to another
    tensor behind the cache's back) with the version counter they had."""
    return tag, [(t, t._version) if torch.is_tensor(t) else t for t in tensors]


def _same_sources(key, tag, tensors):
    """True when `tensors` are the very objects `key` was taken from, unmodified
    since (in-place writes bump the version counter; a replaced attribute is
    another object)."""
    if key[0] != tag or len(key[1]) != len(tensors):
        return False
    for old, new in zip(key[1], tensors):
        if torch.is_tensor(new):
            if not (isinstance(old, tuple) and old[0] is new and old[1] == new._version):
                return False
        elif isinstance(old, tuple) or old != new:
            return False
    return True


_VIEW_DTYPE = np.dtype({
    'names': ['img_xyz', 'fx', 'fy', 'cx', 'cy', 'cam_to_world', 'centre', 'crop_off_x', 'crop_off_y'],
    'formats': [('f4', 3), 'f4', 'f4', 'f4', 'f4', ('f4', 16), ('f8', 3), 'i4', 'i4'],
    'offsets': [0, 12, 16, 20, 24, 28, 96, 120, 124], 'itemsize': 128})


def pack_views(images, camera):
    """[V] mvf_view structs of the views as a host uint8 tensor (the layout of
    ``_lib.MvfView``; what ``visibility.make_view`` fills for one view)."""
    V = images.num_views
    if camera in ('kitti360_fisheye', 's3dis_equirectangular'):
        # few views, many camera parameters: one make_view per view
        fish = images.intrinsic_fisheye.float() if getattr(images, 'is_fisheye', False) else None
        opk = images.opk.float() if getattr(images, 'has_opk', False) else None
        parts = []
        for v in range(V):
            view = visibility_module.make_view(
                images.pos[v], None, images.extrinsic[v].float() if images.has_extrinsic else None, camera,
                centre=images.pos[v].double(), crop_offsets=images.crop_offsets[v].tolist(),
                img_opk=None if opk is None else opk[v],
                img_intrinsic_fisheye=None if fish is None else fish[v])
            parts.append(np.frombuffer(bytes(view), dtype=np.uint8))
        return torch.from_numpy(np.concatenate(parts).copy())
    assert camera in ('kitti360_perspective', 'scannet')
    arr = np.zeros(V, dtype=_VIEW_DTYPE)
    pos = np.asarray(images.pos.detach().cpu())
    arr['img_xyz'] = pos.astype(np.float32)
    arr['centre'] = pos.astype(np.float64)
    for name, key in (('fx', 'fx'), ('fy', 'fy'), ('cx', 'mx'), ('cy', 'my')):
        arr[name] = np.asarray(torch.as_tensor(getattr(images, key)).detach().cpu().float())
    E = np.ascontiguousarray(np.asarray(images.extrinsic.detach().cpu().float()))
    if camera == 'scannet':
        E = np.linalg.inv(E)             # the reference inverts the float32 matrix (visibility.py:233)
    arr['cam_to_world'] = E.reshape(V, 16).astype(np.float32)
    off = np.asarray(images.crop_offsets.detach().cpu()).astype(np.int32)
    arr['crop_off_x'], arr['crop_off_y'] = off[:, 0], off[:, 1]
    return torch.from_numpy(arr.view(np.uint8).reshape(V * 128).copy())


def _stack_depth_maps(images, device):
    """[V, W, H] float32 device tensor of the decoded depth maps of the views
    (a list of [W, H] maps, or the stacked tensor itself)."""
    dm = images.depth_maps
    if not torch.is_tensor(dm):
        dm = torch.stack([torch.as_tensor(d) for d in dm])
    dm = dm.to(device=device, dtype=torch.float32).contiguous()
    assert tuple(dm.shape) == (images.num_views,) + tuple(images.proj_size), \
        f"depth maps must be [V, W, H] = {(images.num_views,) + tuple(images.proj_size)}"
    return dm


class FusedMapPlan:
    """One sample staged on the device for ``mvf_map_images_exact`` plus the
    output / workspace buffers, which are reused from run to run and grown when
    the library reports an overflow.

    ``run()`` is the synchronous form (launch, read the data-dependent sizes
    back, return the sliced outputs).  ``run_async()`` only enqueues the pass
    and an asynchronous copy of the sizes to pinned host memory and returns a
    ``PendingMap``; outputs alternate between two buffer sets, so that the pass
    of the next sample can be enqueued before the sizes of the current one are
    waited for and the device never idles on the host round trip."""

    SLOTS = 2

    def __init__(self, pos, ids, inv, attrs, views, mask, params, cylinder, upscale,
                 downscale, crop_size, V, num_points, device, pair_cap=None, out_cap=None,
                 lane=0, centre_slots=None, depth_maps=None, pix_cap=None, window=None,
                 view_first=None, views_per_point=None, mask_cap=None):
        self.pos, self.ids, self.inv, self.attrs = pos, ids, inv, attrs
        # a batch of samples in one pass (mvf_map_images_exact_batch): window
        # [N, 2] int32 = (first view, number of views) of every point position,
        # view_first [V] int32 = first view of the sample of every view
        self.window, self.view_first = window, view_first
        self.views_per_point = views_per_point or V
        # [V, W, H] float32 depth maps: the pass then runs DepthBasedVisibility
        # (mvf_map_images_depth) instead of the exact splat visibility
        self.depth_maps = depth_maps
        self.views, self.mask, self.params = views, mask, params
        self.cylinder, self.upscale, self.downscale, self.crop_size = \
            cylinder, upscale, downscale, crop_size
        self.N, self.V, self.num_points, self.device = pos.shape[0], V, num_points, device
        self.F = 2 + sum(a is not None for a in attrs)
        pairs = self.N * self.views_per_point            # (point, view) pairs of the pass
        self.pair_cap = pair_cap or (pairs // 4 + 4096)
        self.out_cap = out_cap or (pairs // 4 + 4096)
        # slots of the table that resolves shared centre pixels (the library
        # rounds up to a power of two); fitted to the mappings after a run
        self.centre_slots = centre_slots or max(pairs // 8, 65536)
        # plain splat visibility (params.exact == 0): every pixel a point wins is
        # a mapping; room for that many pixels (at most one per image pixel)
        self.splat = depth_maps is None and not params.exact
        self.pix_cap = pix_cap or (V * params.W * params.H // 2)
        # ... and for the pass's verdict bits: one 8-pixel entry per column and group of
        # 8 image rows of the splat boxes of the pairs that win a pixel
        self.mask_cap = mask_cap or (V * params.W * params.H // 4)
        self._slots = [None] * self.SLOTS
        self._ws = None
        self._turn = 0
        self.last_counts = None
        self.lane = lane

    @staticmethod
    def batched(plans):
        """One plan over the samples of ``plans`` (exact splat visibility, same
        image settings): clouds and views concatenated, every point facing the
        views of its own sample (``mvf_map_images_exact_batch``).  The outputs are
        the CSR of the collated batch.  None when the samples cannot share a
        pass."""
        first = plans[0]
        key = lambda p: (bytes(p.params), p.cylinder, p.upscale, p.downscale, tuple(p.crop_size),
                         [a is not None for a in p.attrs], p.device)
        def same_mask(p):
            if p.mask is None or first.mask is None:
                return p.mask is first.mask
            return p.mask is first.mask or torch.equal(p.mask, first.mask)
        if any(p.depth_maps is not None or p.splat or p.window is not None or not same_mask(p)
               or key(p) != key(first) for p in plans):
            return None
        n_views = [p.V for p in plans]
        V = sum(n_views)
        if V > 480 or V * first.params.W * first.params.H >= 2 ** 32 - 1:
            return None
        dev = first.device
        n_pos = [p.N for p in plans]
        pos0 = np.concatenate(([0], np.cumsum(n_pos)))
        view0 = np.concatenate(([0], np.cumsum(n_views)))
        inv = torch.cat([torch.where(p.inv >= 0, p.inv + int(o), p.inv) for p, o in zip(plans, pos0)])
        attrs = [torch.cat([p.attrs[k] for p in plans]) if first.attrs[k] is not None else None
                 for k in range(len(first.attrs))]
        window = torch.from_numpy(np.repeat(
            np.stack([view0[:-1], n_views], axis=1).astype(np.int32), n_pos, axis=0)).to(dev)
        view_first = torch.from_numpy(np.repeat(view0[:-1].astype(np.int32), n_views)).to(dev)
        return FusedMapPlan(
            torch.cat([p.pos for p in plans]), None, inv.contiguous(), attrs,
            torch.cat([p.views for p in plans]), first.mask, first.params, first.cylinder,
            first.upscale, first.downscale, first.crop_size, V, sum(p.num_points for p in plans), dev,
            window=window, view_first=view_first, views_per_point=max(n_views))

    def twin(self, lane):
        """A second plan over the same staged sample with its own outputs,
        and workspace, to be driven on another stream."""
        return FusedMapPlan(
            self.pos, self.ids, self.inv, self.attrs, self.views, self.mask, self.params,
            self.cylinder, self.upscale, self.downscale, self.crop_size, self.V,
            self.num_points, self.device, self.pair_cap, self.out_cap, lane=lane,
            centre_slots=self.centre_slots, depth_maps=self.depth_maps, pix_cap=self.pix_cap,
            window=self.window, view_first=self.view_first, views_per_point=self.views_per_point,
            mask_cap=self.mask_cap)

    def _slot(self, k):
        """Output buffers of slot k: (capacities, counters, host counters, event,
        pointers, view_seen, images, pixels, features, view pointers)."""
        sl = self._slots[k]
        caps = (self.out_cap, self.pix_cap if self.splat else 0)
        if sl is None or sl[0] != caps:
            d = self.device
            keep = sl[1:4] if sl is not None else (
                torch.zeros(8, dtype=torch.int64, device=d),
                torch.zeros(8, dtype=torch.int64).pin_memory(),
                torch.cuda.Event())
            n_pix = self.pix_cap if self.splat else self.out_cap
            sl = (caps,) + tuple(keep) + (
                torch.empty(self.num_points + 1, dtype=torch.int64, device=d),
                torch.empty(self.V, dtype=torch.int32, device=d),
                torch.empty(self.out_cap, dtype=torch.int64, device=d),
                torch.empty((n_pix, 2), dtype=torch.int16, device=d),
                torch.empty((self.out_cap, self.F), dtype=torch.float32, device=d),
                torch.empty(self.out_cap + 1, dtype=torch.int64, device=d) if self.splat else None)
            self._slots[k] = sl
        return sl

    def _workspace(self):
        key = (self.pair_cap, self.centre_slots, self.mask_cap if self.splat else 0)
        if self._ws is None or self._ws[0] != key:
            lib = L.load()
            if self.depth_maps is not None:
                ws_bytes = lib.mvf_map_images_depth_workspace_bytes(self.num_points, self.V)
            elif self.splat:
                ws_bytes = lib.mvf_map_images_splat_workspace_bytes(
                    self.N, self.V, self.params.W, self.params.H, self.num_points, self.pair_cap,
                    self.mask_cap)
            elif self.window is not None:
                ws_bytes = lib.mvf_map_images_batch_workspace_bytes(
                    self.N, self.V, self.views_per_point, self.params.W, self.params.H,
                    self.num_points, self.pair_cap, self.centre_slots)
            else:
                ws_bytes = lib.mvf_map_images_workspace_bytes(
                    self.N, self.V, self.params.W, self.params.H, self.num_points, self.pair_cap,
                    self.centre_slots)
            self._ws = (key, L.workspace(ws_bytes, self.device))
        return self._ws[1]

    def launch(self, slot=0):
        """Enqueue one fused mapping pass into the buffers of `slot` (no
        synchronisation).  Returns the slot's buffers."""
        lib = L.load()
        P = self.params
        sl = self._slot(slot)
        _, counters, _, _, pointers, view_seen, img_ids, pixels, feats, view_ptr = sl
        ws = self._workspace()
        if self.splat:
            assert self.upscale == 1 and self.downscale == 1
            with torch.cuda.device(self.device):
                L.check(lib.mvf_map_images_splat(
                    L.ptr(self.pos), L.ptr(self.inv), self.N, self.num_points, L.ptr(self.views),
                    self.V, C.byref(P), int(self.cylinder), L.ptr(self.mask),
                    *[L.ptr(a) for a in self.attrs], int(self.crop_size[0]), int(self.crop_size[1]),
                    self.pair_cap, self.mask_cap, L.ptr(pointers), L.ptr(img_ids), L.ptr(view_ptr),
                    L.ptr(pixels),
                    L.ptr(feats), self.out_cap, self.pix_cap, L.ptr(view_seen), L.ptr(counters),
                    L.ptr(ws), ws.numel(), L.stream_ptr(self.device)))
            return sl
        if self.depth_maps is not None:
            with torch.cuda.device(self.device):
                L.check(lib.mvf_map_images_depth(
                    L.ptr(self.pos), L.ptr(self.inv), self.N, self.num_points,
                    L.ptr(self.views), self.V, C.byref(P), int(self.cylinder), L.ptr(self.mask),
                    L.ptr(self.depth_maps), *[L.ptr(a) for a in self.attrs], self.upscale,
                    self.downscale, int(self.crop_size[0]), int(self.crop_size[1]),
                    L.ptr(pointers), L.ptr(img_ids), L.ptr(pixels), L.ptr(feats), self.out_cap,
                    L.ptr(view_seen), L.ptr(counters), L.ptr(ws), ws.numel(),
                    L.stream_ptr(self.device)))
            return sl
        if self.window is not None:
            with torch.cuda.device(self.device):
                L.check(lib.mvf_map_images_exact_batch(
                    L.ptr(self.pos), L.ptr(self.inv), L.ptr(self.window), self.N, self.num_points,
                    L.ptr(self.views), L.ptr(self.view_first), self.V, self.views_per_point,
                    C.byref(P), int(self.cylinder), L.ptr(self.mask),
                    *[L.ptr(a) for a in self.attrs], self.upscale, self.downscale,
                    int(self.crop_size[0]), int(self.crop_size[1]), self.centre_slots,
                    self.pair_cap, L.ptr(pointers), L.ptr(img_ids), L.ptr(pixels),
                    L.ptr(feats), self.out_cap, L.ptr(view_seen), L.ptr(counters),
                    L.ptr(ws), ws.numel(), L.stream_ptr(self.device)))
            return sl
        with torch.cuda.device(self.device):
            L.check(lib.mvf_map_images_exact(
                L.ptr(self.pos), L.ptr(self.ids), L.ptr(self.inv), self.N, self.num_points,
                L.ptr(self.views), self.V, C.byref(P), int(self.cylinder), L.ptr(self.mask),
                *[L.ptr(a) for a in self.attrs], self.upscale, self.downscale,
                int(self.crop_size[0]), int(self.crop_size[1]), self.centre_slots,
                self.pair_cap, L.ptr(pointers), L.ptr(img_ids), L.ptr(pixels),
                L.ptr(feats), self.out_cap, L.ptr(view_seen), L.ptr(counters),
                L.ptr(ws), ws.numel(), L.stream_ptr(self.device)))
        return sl

    def run_async(self):
        """Enqueue a mapping pass and the copy of its sizes to pinned host
        memory.  The returned PendingMap owns one of the two output buffer sets
        until the PendingMap after the next one is created."""
        slot = self._turn % self.SLOTS
        self._turn += 1
        sl = self.launch(slot)
        with torch.cuda.device(self.device):
            sl[2].copy_(sl[1], non_blocking=True)
            sl[3].record(torch.cuda.current_stream(self.device))
        return PendingMap(self, sl)

    def _fit(self, n_map):
        """Sizes the centre-pixel table to the mappings seen (about 2.5 slots
        per mapping before the library rounds up to a power of two)."""
        def pow2(x):
            return 1 << max(int(x) - 1, 1).bit_length()
        target = max(int(2.5 * n_map), 4096)
        if pow2(target) != pow2(self.centre_slots):
            self.centre_slots = target

    def _grow(self, n_map, pairs, overflow=1, n_pix=0, n_mask=0):
        """Larger buffers after an overflow (bit 0: pairs / outputs, bit 1: the
        centre-pixel table, bit 2: pixels of the plain splat mode, bit 3: its
        verdict entries)."""
        if overflow & 1:
            self.pair_cap = max(self.pair_cap, int(pairs * 1.1) + 4096)
            self.out_cap = max(self.out_cap, int(n_map * 1.1) + 4096)
        if overflow & 2:
            self.centre_slots = max(4 * self.centre_slots, 4 * n_map)
        if overflow & 4:
            self.pix_cap = max(self.pix_cap, int(n_pix * 1.1) + 4096)
        if overflow & 8:
            self.mask_cap = max(self.mask_cap, int(n_mask * 1.1) + 4096)

    def _outputs(self, sl, n_map, n_pix):
        """The sliced outputs of a finished pass: pointers, images, pixels,
        features, view_seen (and, in the plain splat mode, the view pointers)."""
        _, _, _, _, pointers, view_seen, img_ids, pixels, feats, view_ptr = sl
        if self.splat:
            return (pointers, img_ids[:n_map], pixels[:n_pix], feats[:n_map], view_seen,
                    view_ptr[:n_map + 1])
        return pointers, img_ids[:n_map], pixels[:n_map], feats[:n_map], view_seen

    def run(self):
        """Mapping pass + read-back of the data-dependent sizes (one
        synchronisation); repeats with larger buffers after an overflow.
        Returns pointers, images, pixels, features, view_seen (and the view
        pointers in the plain splat mode)."""
        for _ in range(4):
            sl = self.launch(0)
            n_map, pairs, overflow, n_seen, n_pix, n_mask, n_chunks = sl[1].tolist()[:7]
            self.last_counts = dict(mappings=n_map, pairs=pairs, seen_views=n_seen, pixels=n_pix,
                                    verdict_entries=n_mask, chunks=n_chunks)
            if not overflow:
                self._fit(n_map)
                return self._outputs(sl, n_map, n_pix)
            self._grow(n_map, pairs, overflow, n_pix, n_mask)
        raise L.MvfError("MapImages: buffers still too small after retries")


class PendingMap:
    """A mapping pass in flight (FusedMapPlan.run_async)."""

    def __init__(self, plan, slot):
        self.plan, self._slot = plan, slot

    def result(self):
        """Waits for the sizes of this pass (not for later work on the stream)
        and returns what FusedMapPlan.run returns.  After an overflow the pass
        is repeated synchronously with larger buffers."""
        host, event = self._slot[2], self._slot[3]
        event.synchronize()
        n_map, pairs, overflow, n_seen, n_pix, n_mask, n_chunks = host.tolist()[:7]
        if overflow:
            self.plan._grow(n_map, pairs, overflow, n_pix, n_mask)
            return self.plan.run()
        self.plan.last_counts = dict(mappings=n_map, pairs=pairs, seen_views=n_seen, pixels=n_pix,
                                     verdict_entries=n_mask, chunks=n_chunks)
        self.plan._fit(n_map)
        return self.plan._outputs(self._slot, n_map, n_pix)


class MapImages(ImageTransform):
    """Computes the mappings between individual 3D points and image pixels
    (image.py:162-428).  Returns the input data and the images augmented with
    the point-image-pixel ``ImageMapping``; views without any mapping are
    dropped."""

    def __init__(
            self, method='SplattingVisibility', proj_upscale=None,
            ref_size=None, use_cuda=False, verbose=False, cylinder=False,
            **kwargs):
        self.key = MAPPING_KEY
        self.verbose = verbose
        self.cylinder = cylinder
        self.ref_size = ref_size
        self.proj_upscale = proj_upscale
        self.method = method
        # computation always runs on the HIP device; kept for signature parity
        self.use_cuda = use_cuda
        self.point_order = kwargs.pop('point_order', 'kdtree')
        assert self.point_order in ('kdtree', 'index')
        # streams a list of samples is spread over (fused path only)
        self.lanes = int(kwargs.pop('lanes', 3))
        # where the mappings go: None = the device of the images (what the
        # reference returns); a HIP device keeps them where they were computed
        # (the small per-view tensors of the images stay where they are)
        self.output_device = kwargs.pop('output_device', None)
        self.kwargs = kwargs
        self._pair_cap = None
        self._out_cap = None
        self._centre_slots = None
        self.timings = {}

    # ------------------------------------------------------------------ #
    def _order(self, data, device=None):
        """Local point order of the sample; the KD-tree order is kept on
        ``data`` for the next call, as the reference keeps the tree itself
        (transforms.py:325-329,379-383).  ``device``: where a newly built order
        is wanted."""
        order = getattr(data, ORDER_KEY, None)
        if order is not None and order.shape[0] == data.pos.shape[0]:
            return order
        if self.point_order == 'index':
            return None
        order = kdtree_order(data.pos, self.cylinder, device=device)
        try:
            setattr(data, ORDER_KEY, order)
        except Exception:
            pass
        return order

    def _fusable(self, images, visi_model):
        """Can all views of these images go through one launch set?  Splat
        visibility (exact, or plain at the images' own resolution) or depth-map
        visibility (with the decoded maps at hand as ``images.depth_maps``),
        integer scale factors, any of the four cameras."""
        ints = all(float(s).is_integer() and s >= 1
                   for s in (images.proj_upscale, images.downscale))
        if visi_model.camera in ('kitti360_perspective', 'scannet'):
            camera_ok = images.is_pinhole and images.has_extrinsic
        else:
            camera_ok = visi_model.camera in ('kitti360_fisheye', 's3dis_equirectangular')
        common = (camera_ok and ints and images.pixel_dtype == torch.int16
                  and images.num_views <= 480)
        if self.method == 'SplattingVisibility':
            small = images.num_views * images.proj_size[0] * images.proj_size[1] < 2 ** 32 - 1
            if visi_model.exact:
                return common and small
            # plain splat visibility: other scale factors merge pixels of a point,
            # which the per-view path resolves with its duplicate removal
            return common and small and images.proj_upscale == 1 and images.downscale == 1
        if self.method == 'DepthBasedVisibility':
            return common and getattr(images, 'depth_maps', None) is not None
        return False

    def _settle(self, data, images):
        """Checks and image settings of image.py:192-215; returns the
        visibility model of these images."""
        assert hasattr(data, self.key)
        assert images.num_views >= 1, "At least one image must be provided."
        if self.ref_size is not None:
            images.ref_size = self.ref_size
        if self.proj_upscale is not None:
            images.proj_upscale = self.proj_upscale
        if images.mask is not None:
            assert tuple(images.mask.shape) == tuple(images.proj_size)
        visi_cls = getattr(visibility_module, self.method)
        return visi_cls(img_size=images.proj_size, **self.kwargs)

    def __call__(self, data, images):
        if isinstance(data, list) and len(data) > 1 and self.lanes > 1 \
                and isinstance(images, list) and len(data) == len(images) \
                and not any(isinstance(im, ImageData) for im in images):
            out = self._process_list_fused(data, images)
            if out is not None:
                return out
        return super(MapImages, self).__call__(data, images)

    # ------------------------------------------------------------------ #
    def prepare_batch(self, data_list, images_list, device=None):
        """Stage a list of samples for one batched pass (``mvf_map_images_exact_batch``):
        clouds and views concatenated, every point facing the views of its own
        sample.  Returns (plan, per-sample bookkeeping), or None when the samples
        cannot share a pass (different image settings, a mode other than the
        exact splat visibility, repeated mapping ids, too many views)."""
        device = device or L.require_device(data_list[0].pos)
        parts, models = [], []
        for data, images in zip(data_list, images_list):
            if isinstance(images, ImageData):
                return None
            visi_model = self._settle(data, images)
            if not self._fusable(images, visi_model) or self.method != 'SplattingVisibility' \
                    or not visi_model.exact:
                return None
            plan = self.prepare_fused(data, images, visi_model, device)
            if plan is None:
                return None
            parts.append(plan)
            models.append(visi_model)
        first = images_list[0]
        if not all(im.mask is first.mask for im in images_list):
            return None
        plan = FusedMapPlan.batched(parts)
        if plan is None:
            return None
        n_ids, n_views = [p.num_points for p in parts], [p.V for p in parts]
        return plan, dict(n_ids=n_ids, n_views=n_views, models=models)

    def process_batch(self, data_list, images_list):
        """MapImages on every sample of a list and the collation of the results,
        as one pass on the device: returns ``(data_list, SameSettingImageBatch)``
        whose ``mappings`` are the ``ImageMappingBatch`` that
        ``SameSettingImageBatch.from_data_list`` builds from the samples mapped
        one by one (image.py:162-428 per sample, then core/multimodal/image.py:1329-1405
        and core/multimodal/csr.py:347-416).  Samples that cannot share a pass
        are mapped one by one and collated."""
        from .images import SameSettingImageBatch
        from .csr import CSRBatch, CSRData
        from .mapping import ImageMappingBatch
        staged = self.prepare_batch(data_list, images_list)
        if staged is None:
            out = [super(MapImages, self).__call__(d, im) for d, im in zip(data_list, images_list)]
            return [o[0] for o in out], SameSettingImageBatch.from_data_list([o[1] for o in out])
        plan, info = staged
        pointers, img_ids, pixels, feats, view_seen = plan.run()[:5]
        id0 = np.concatenate(([0], np.cumsum(info['n_ids'])))
        # rows of every sample, and which of its views were seen: one small read-back
        bounds = pointers[torch.from_numpy(id0).to(pointers.device)].tolist()
        rows = np.diff(bounds)
        if (rows == 0).any():
            raise ValueError(_NO_MAPPING_MSG)
        seen = (view_seen > 0).cpu()
        kept, v0 = [], 0
        for images, visi_model, nv in zip(images_list, info['models'], info['n_views']):
            s = torch.where(seen[v0:v0 + nv])[0]
            v0 += nv
            images = images[s.to(images.device)] if s.shape[0] != nv else images
            images.mappings = None
            images.visibility = visi_model
            kept.append(images)
        batch = SameSettingImageBatch.from_data_list(kept)
        nested = CSRBatch(torch.arange(img_ids.shape[0] + 1, dtype=torch.int64, device=pointers.device),
                          pixels, dense=False)
        nested.__sizes__ = torch.from_numpy(rows.astype(np.int64))
        nested.__csr_type__ = CSRData
        mappings = ImageMappingBatch(pointers, img_ids, nested, feats, dense=False,
                                     is_index_value=[True, False, False])
        mappings.__sizes__ = torch.tensor(info['n_ids'], dtype=torch.int64)
        mappings.__csr_type__ = ImageMappingBatch.__csr_type__
        batch.mappings = mappings.to(
            kept[0].device if self.output_device is None else torch.device(self.output_device))
        batch.visibility = info['models'][0]
        return list(data_list), batch

    def _process_list_fused(self, data_list, images_list):
        """A list of independent samples on the fused path: the passes are
        spread over ``lanes`` streams so that the latency-bound kernels of one
        sample overlap with those of the others; results are those of the
        one-by-one loop.  Returns None when a sample cannot take the fused
        path (the caller then loops)."""
        device = L.require_device(data_list[0].pos)
        staged = []
        for data, images in zip(data_list, images_list):
            visi_model = self._settle(data, images)
            if not self._fusable(images, visi_model):
                return None
            plan = self.prepare_fused(data, images, visi_model, device)
            if plan is None:
                return None
            staged.append((data, images, visi_model, plan))
        n_lanes = min(self.lanes, len(staged))
        current = torch.cuda.current_stream(device)
        streams = [torch.cuda.Stream(device) for _ in range(n_lanes)]
        pending = []
        for k, (_, _, _, plan) in enumerate(staged):
            plan.lane = k % n_lanes
            s = streams[plan.lane]
            plan._slot(0)                    # buffers belong to the caller's stream
            plan._workspace()
            s.wait_stream(current)           # inputs were staged on the caller's stream
            with torch.cuda.stream(s):
                pending.append(plan.run_async())
        results = []
        for (_, _, _, plan), p in zip(staged, pending):
            with torch.cuda.stream(streams[plan.lane]):
                res = p.result()
            for t in res:                    # consumed on the caller's stream from here on
                t.record_stream(current)
            results.append(res)
            self._pair_cap = max(self._pair_cap or 0, plan.pair_cap)
            self._out_cap = max(self._out_cap or 0, plan.out_cap)
            self._centre_slots = max(self._centre_slots or 0, plan.centre_slots)
        for s in streams:
            current.wait_stream(s)
        data_out, images_out = [], []
        for (data, images, visi_model, _), res in zip(staged, results):
            pointers, img_ids, pixels, feats, view_seen = res[:5]
            if img_ids.shape[0] == 0:
                raise ValueError(_NO_MAPPING_MSG)
            in_device = images.device
            mappings = ImageMapping.from_csr_arrays(pointers, img_ids, pixels, feats, *res[5:])
            seen = torch.where(view_seen > 0)[0]
            if seen.shape[0] != images.num_views:
                images = images[seen.to(in_device)]
            images.mappings = mappings.to(in_device)
            images.visibility = visi_model
            data_out.append(data)
            images_out.append(images)
        return data_out, images_out

    def _process(self, data, images):
        in_device = images.device
        device = L.require_device(data.pos)
        visi_model = self._settle(data, images)
        self.timings = dict(t_sphere_sampling=0.0, t_stage=0.0, t_visibility=0.0,
                            t_ImageMapping_init=0.0)

        mappings = seen = None
        if self._fusable(images, visi_model):
            res = self._process_fused(data, images, visi_model, device)
            if res is not None:
                mappings, seen = res
        if mappings is None:
            t0 = time.perf_counter()
            mappings, seen = self._process_per_view(data, images, visi_model, device)
            self.timings['t_visibility'] = (time.perf_counter() - t0) * 1e3

        t0 = time.perf_counter()
        if seen.shape[0] != images.num_views:
            images = images[seen.to(in_device)]
        out_device = in_device if self.output_device is None else torch.device(self.output_device)
        images.mappings = mappings.to(out_device)
        images.visibility = visi_model
        self.timings['t_ImageMapping_init'] += (time.perf_counter() - t0) * 1e3
        return data, images

    # ------------------------------------------------------------------ #
    def _stage_points(self, data, order, device):
        """The per-point inputs of the fused path on the device: points (and
        attributes) permuted into the local order, their mapping ids, the
        position of every id.  Kept on ``data`` (next to the point order) for
        the next call on the same, unmodified tensors -- a sample is usually
        mapped once per image setting.  None when mapping ids repeat or are
        negative."""
        N = data.pos.shape[0]
        ids = getattr(data, self.key)
        names = ('linearity', 'planarity', 'scattering', 'norm')
        src = [data.pos, ids, order] + [getattr(data, k, None) for k in names]
        cached = getattr(data, '_mvf_staged', None)
        if cached is not None and _same_sources(cached[0], str(device), src):
            return cached[1]
        if N == 0:
            return None
        if not ids.is_cuda:
            # host ids: their range is settled here, without a device round trip
            ids_h = np.asarray(ids.detach())
            lo, hi = int(ids_h.min()), int(ids_h.max())
        else:
            lo, hi = (int(v) for v in torch.aminmax(ids))
        if lo < 0:
            return None
        num_points = hi + 1
        ids = ids.to(device=device, dtype=torch.int64, non_blocking=True)
        pos = data.pos.to(device=device, dtype=torch.float32, non_blocking=True)
        attrs = [getattr(data, k, None) for k in names]
        attrs = [a.to(device=device, dtype=torch.float32, non_blocking=True) if a is not None else None
                 for a in attrs]
        if order is not None:
            order = order.to(device, non_blocking=True)
            pos, ids = pos[order], ids[order]
            attrs = [a[order] if a is not None else None for a in attrs]
        # position of every mapping id in the local order; ids must be distinct (repeated
        # ids mean several pixels per (point, view), which the per-view path handles)
        inv = torch.full((num_points,), -1, dtype=torch.int64, device=device)
        inv[ids] = torch.arange(N, dtype=torch.int64, device=device)
        if int((inv >= 0).sum()) != N:
            return None
        staged = (pos.contiguous(), ids.contiguous(), inv,
                  [a.contiguous() if a is not None else None for a in attrs], num_points)
        try:
            data._mvf_staged = (_source_key(str(device), src), staged)
        except Exception:
            pass
        return staged

    @staticmethod
    def _stage_views(images, camera, device):
        """The packed views on the device, kept on ``images`` for the next call on
        the same, unmodified camera tensors."""
        src = [images.pos, images.extrinsic, images.crop_offsets] + [
            getattr(images, k, None) for k in ('fx', 'fy', 'mx', 'my', 'opk', 'xi', 'k1', 'k2',
                                                'gamma1', 'gamma2', 'u0', 'v0')]
        tag = (str(device), camera)
        cached = getattr(images, '_mvf_views', None)
        if cached is not None and _same_sources(cached[0], tag, src):
            return cached[1]
        views = pack_views(images, camera).to(device)
        try:
            images._mvf_views = (_source_key(tag, src), views)
        except Exception:
            pass
        return views

    def prepare_fused(self, data, images, visi_model, device):
        """Stage one sample on the device for the fused path: points permuted
        into the sampler's order, mapping ids and their inverse, packed views.
        Returns a FusedMapPlan, or None when mapping ids repeat."""
        N = data.pos.shape[0]
        depth = self.method == 'DepthBasedVisibility'
        t0 = time.perf_counter()
        # the depth test has no ties: any point order gives the same mappings
        order = None if depth else self._order(data, device)
        self.timings['t_sphere_sampling'] = (time.perf_counter() - t0) * 1e3
        staged = self._stage_points(data, order, device)
        if staged is None:
            return None
        pos, ids, inv, attrs, num_points = staged
        return FusedMapPlan(
            pos, ids, inv, attrs, self._stage_views(images, visi_model.camera, device),
            images.mask.to(device=device, dtype=torch.uint8).contiguous()
            if images.mask is not None else None,
            visibility_module.make_params(**visi_model.__dict__), bool(self.cylinder),
            int(images.proj_upscale), int(images.downscale), tuple(images.crop_size),
            images.num_views, num_points, device, self._pair_cap, self._out_cap,
            centre_slots=self._centre_slots,
            depth_maps=_stack_depth_maps(images, device) if depth else None)

    def _process_fused(self, data, images, visi_model, device):
        """The phases are host wall clock (ms): work enqueued by one phase may
        run during the next; ``t_visibility`` ends with the read-back of the
        sizes, which waits for everything before it."""
        t0 = time.perf_counter()
        plan = self.prepare_fused(data, images, visi_model, device)
        if plan is None:
            return None
        t1 = time.perf_counter()
        res = plan.run()
        pointers, img_ids, pixels, feats, view_seen = res[:5]
        t2 = time.perf_counter()
        self._pair_cap, self._out_cap = plan.pair_cap, plan.out_cap
        self._centre_slots = plan.centre_slots
        if img_ids.shape[0] == 0:
            raise ValueError(_NO_MAPPING_MSG)
        mappings = ImageMapping.from_csr_arrays(pointers, img_ids, pixels, feats, *res[5:])
        if plan.last_counts['seen_views'] == images.num_views:
            seen = torch.arange(images.num_views)          # every view kept: nothing to look up
        else:
            seen = torch.where(view_seen > 0)[0]
        self.timings['t_stage'] = (t1 - t0) * 1e3 - self.timings['t_sphere_sampling']
        self.timings['t_visibility'] = (t2 - t1) * 1e3
        self.timings['t_ImageMapping_init'] = (time.perf_counter() - t2) * 1e3
        return mappings, seen

    # ------------------------------------------------------------------ #
    def _process_per_view(self, data, images, visi_model, device):
        N = data.pos.shape[0]
        order = self._order(data)
        order = torch.arange(N, device=device) if order is None else order.to(device)
        pos = data.pos.to(device)
        pos64 = pos.double()[order]
        pos32 = pos.float()
        ids = getattr(data, self.key).to(device=device, dtype=torch.int64)
        attrs = {k: getattr(data, a, None) for k, a in (
            ('linearity', 'linearity'), ('planarity', 'planarity'),
            ('scattering', 'scattering'), ('normals', 'norm'))}
        attrs = {k: (a.to(device).float() if a is not None else None) for k, a in attrs.items()}
        K = images.intrinsic_pinhole.float() if images.is_pinhole else None
        fish = images.intrinsic_fisheye.float() if getattr(images, 'is_fisheye', False) else None
        opk = images.opk.float() if getattr(images, 'has_opk', False) else None
        r = float(visi_model.r_max)

        image_ids, point_ids, features, pixels = [], [], [], []
        for v in range(images.num_views):
            c = images.pos[v].double().to(device)
            d2 = (pos64[:, 0] - c[0]) ** 2 + (pos64[:, 1] - c[1]) ** 2
            if not self.cylinder:
                d2 = d2 + (pos64[:, 2] - c[2]) ** 2
            local = order[d2 <= r * r]
            extra = {}
            if getattr(images, 'depth_maps', None) is not None:
                extra['depth_map'] = images.depth_maps[v]
            out_vm = visi_model(
                pos32[local], images.pos[v].float(),
                img_opk=opk[v] if opk is not None else None,
                img_intrinsic_pinhole=K[v] if K is not None else None,
                img_intrinsic_fisheye=fish[v] if fish is not None else None,
                img_extrinsic=images.extrinsic[v].float() if images.has_extrinsic else None,
                img_mask=images.mask if images.mask is not None else None,
                **{k: (a[local] if a is not None else None) for k, a in attrs.items()},
                **extra)
            if out_vm['idx'].shape[0] == 0:
                continue
            point_ids_ = ids[local][out_vm['idx']]
            pix_x_ = out_vm['x'].long()
            pix_y_ = out_vm['y'].long()
            features_ = out_vm['features'].float()
            pix_x_ = pix_x_ // images.proj_upscale
            pix_y_ = pix_y_ // images.proj_upscale
            pix_x_ = pix_x_ - images.crop_offsets[v][0].item()
            pix_y_ = pix_y_ - images.crop_offsets[v][1].item()
            in_crop = torch.where(
                (pix_x_ >= 0) & (pix_y_ >= 0) & (pix_x_ < images.crop_size[0])
                & (pix_y_ < images.crop_size[1]))
            pix_x_, pix_y_ = pix_x_[in_crop], pix_y_[in_crop]
            point_ids_, features_ = point_ids_[in_crop], features_[in_crop]
            pix_x_ = (pix_x_ // images.downscale).long()
            pix_y_ = (pix_y_ // images.downscale).long()
            unique_idx = lexargunique(point_ids_, pix_x_, pix_y_)
            pixels_ = torch.stack((pix_x_[unique_idx], pix_y_[unique_idx]), dim=1).type(
                images.pixel_dtype)
            image_ids.append(v)
            point_ids.append(point_ids_[unique_idx])
            features.append(features_[unique_idx])
            pixels.append(pixels_)

        image_ids = torch.tensor(image_ids, dtype=torch.int64)
        if image_ids.shape[0] == 0:
            raise ValueError(_NO_MAPPING_MSG)
        seen = lexunique(image_ids)
        ranks = torch.bucketize(image_ids, seen)
        dense_img = ranks.repeat_interleave(
            torch.tensor([x.shape[0] for x in point_ids])).to(device)
        mappings = ImageMapping.from_dense(
            torch.cat(point_ids), dense_img, torch.cat(pixels), torch.cat(features),
            num_points=int(ids.max().item()) + 1)
        return mappings, seen


def _morton_order(pos):
    """Permutation that sorts the points along a Z-order curve (10 bits per
    axis of the bounding cube)."""
    lo = pos.min(dim=0).values
    ext = (pos.max(dim=0).values - lo).max().clamp(min=1e-20)
    cell = ((pos - lo) / ext * 1023.0).long().clamp_(0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    key = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
    return torch.argsort(key)


def knn(xyz, k):
    """Indices [N, k] of the k nearest neighbours of every point among the
    points themselves (the point first), ascending in distance; exact
    (float64 distances, equal distances by ascending index).  The kernel gets
    the points in Z-order (it is several times faster on spatially coherent
    rows); the result does not depend on that order."""
    device = L.require_device(xyz)
    pos = xyz.to(device=device, dtype=torch.float32)
    n = pos.shape[0]
    perm = _morton_order(pos)
    pos_s = pos[perm].contiguous()
    nbr_s = torch.empty((n, k), dtype=torch.int64, device=device)
    # bounding boxes of the chunks of 256 rows (the last one padded with its own rows)
    pad = (-n) % 256
    rows = torch.cat([pos_s, pos_s[-1:].expand(pad, 3)]) if pad else pos_s
    rows = rows.view(-1, 256, 3)
    box_lo, box_hi = rows.amin(dim=1).contiguous(), rows.amax(dim=1).contiguous()
    with torch.cuda.device(device):
        L.check(L.load().mvf_knn(L.ptr(pos_s), L.ptr(perm), n, int(k), L.ptr(box_lo),
                                 L.ptr(box_hi), L.ptr(nbr_s), L.stream_ptr(device)))
    out = torch.empty_like(nbr_s)
    out[perm] = nbr_s
    return out


class NeighborhoodBasedMappingFeatures(ImageTransform):
    """Appends neighbourhood-based mapping features to ``images.mappings``
    (core/data_transform/multimodal/image.py:431-612): the local surface
    density from the radius of the k-NN ball, and the occlusion, i.e. the share
    of the k nearest neighbours of a point that the image of a mapping also
    sees.  ``k`` may be a list: one column per value and feature.

    The neighbour search is exact (``knn``); ``use_cuda``, ``use_faiss``,
    ``ncells`` and ``nprobes`` are accepted for signature parity (the
    reference searches approximately with FAISS or exactly with KeOps)."""

    def __init__(self, k=20, voxel=None, density=True, occlusion=True, use_cuda=False,
                 use_faiss=True, ncells=None, nprobes=10, verbose=False):
        self.k_list = sorted(k) if isinstance(k, list) else [k]
        self.voxel = voxel if voxel is not None else 1
        self.compute_density = density
        self.compute_occlusion = occlusion
        self.use_faiss = use_faiss
        self.use_cuda = use_cuda
        self.ncells = ncells
        self.nprobes = nprobes
        self.verbose = verbose
        assert density or occlusion, \
            "At least one of `density` or `occlusion` must be True."

    def _process(self, data, images):
        assert images.mappings is not None
        in_device = images.device
        device = L.require_device(data.pos)
        m = images.mappings
        xyz = data.pos.to(device=device, dtype=torch.float32).contiguous()
        neighbors = knn(xyz, self.k_list[-1])
        pointers = m.pointers.to(device)
        sizes = pointers[1:] - pointers[:-1]
        new_cols = []

        if self.compute_density:
            cols = []
            for k in self.k_list:
                # squared radius of the k-NN ball, density of a 2D manifold inside it
                d2_max = ((xyz - xyz[neighbors[:, k - 1]]) ** 2).sum(dim=1)
                v_sphere = 3.1416 * d2_max
                voxel_density = 1 / self.voxel ** 2
                density = ((k + 1) / v_sphere) / voxel_density
                density[torch.where(density.isnan())] = 1
                cols.append(density.view(-1, 1))
            new_cols.append(torch.cat(cols, dim=1).repeat_interleave(sizes, 0))

        if self.compute_occlusion:
            n_points = xyz.shape[0]
            image_ids = m.images.to(device).contiguous()
            n_images = int(image_ids.max()) + 1 if image_ids.numel() else 1
            point_ids = torch.arange(n_points, device=device).repeat_interleave(sizes).contiguous()
            seen = torch.zeros((n_points, n_images), dtype=torch.uint8, device=device)
            seen[point_ids, image_ids] = 1
            rows = image_ids.shape[0]
            occ = torch.empty((rows, len(self.k_list)), dtype=torch.float32, device=device)
            lib = L.load()
            with torch.cuda.device(device):
                for c, k in enumerate(self.k_list):
                    L.check(lib.mvf_mapping_occlusion(
                        L.ptr(neighbors), neighbors.shape[1], int(k), L.ptr(point_ids),
                        L.ptr(image_ids), rows, L.ptr(seen), n_images, L.ptr(occ[:, c:]),
                        occ.shape[1], L.stream_ptr(device)))
            new_cols.append(occ)

        feats = torch.cat(new_cols, dim=1).to(in_device)
        if not m.has_features:
            m.features = feats
        else:
            m.features = torch.cat([m.features, feats.to(m.features.device)], dim=1)
        return data, images


MAPPING_KEY = 'mapping_index'


class SelectMappingFromPointId(ImageTransform):
    """Keeps the mappings of the points of ``data`` (their ``mapping_index``)
    and the images those points are seen in, then renumbers the point ids
    (core/data_transform/multimodal/image.py:615-644)."""

    def __init__(self):
        self.key = MAPPING_KEY

    def _process(self, data, images):
        assert hasattr(data, self.key)
        assert images.mappings is not None
        images = images.select_points(getattr(data, self.key), mode='pick')
        setattr(data, self.key, torch.arange(data.pos.shape[0], device=images.device))
        return data, images


class DropImagesOutsideDataBoundingBox(ImageTransform):
    """Drops the images whose position is outside the bounding box of the
    points (+ margin) (core/data_transform/multimodal/image.py:647-666)."""

    def __init__(self, margin=0, ignore_z=False):
        self.margin = margin
        self.ignore_z = ignore_z

    def _process(self, data, images):
        lo, hi = torch.aminmax(data.pos, dim=0)
        pos = images.pos.to(lo.device)
        axes = slice(0, 2) if self.ignore_z else slice(0, 3)
        # strictly inside the box grown by margin / 2 on every side
        inside = ((pos > lo - self.margin / 2) & (pos < hi + self.margin / 2))[:, axes].all(dim=1)
        return data, images[inside]


class PickKImages(ImageTransform):
    """Keeps K images, at random or one every K in their order
    (core/data_transform/multimodal/image.py:689-710)."""

    def __init__(self, k, random=False, replace=False):
        self.k = k
        self.random = random
        self.replace = replace

    def _process(self, data, images):
        if self.random:
            idx = torch.from_numpy(np.random.choice(
                range(images.num_views), size=self.k, replace=self.replace))
        else:
            idx = slice(0, images.num_views, self.k)
        return data, images[idx]


class PickImagesFromMappingArea(ImageTransform):
    """Drops the images whose mappings cover less than ``area_ratio`` of the
    image (pixel count, or bounding box of the mapped pixels with
    ``use_bbox``), keeps at most ``n_max`` of the others, largest first
    (core/data_transform/multimodal/image.py:713-762)."""

    def __init__(self, area_ratio=0.02, n_max=None, n_min=0, use_bbox=False):
        self.area_ratio = area_ratio
        self.n_max = n_max if n_max is not None and n_max >= 1 else None
        self.n_min = n_min if n_max is not None and n_min >= 0 else 0
        self.use_bbox = use_bbox

    def _process(self, data, images):
        assert images.mappings is not None, "No mappings found in images."
        # one reduction over the views gives every image's pixel count and box
        stats = images.mappings.image_stats(images.num_views)
        if self.use_bbox:
            areas = (stats[:, 2] - stats[:, 1]) * (stats[:, 4] - stats[:, 3])
        else:
            areas = stats[:, 0]
        threshold = images.img_size[0] * images.img_size[1] * self.area_ratio
        ranked = areas.argsort(stable=True).flip(0)
        ranked = ranked[areas[ranked] > threshold]
        if self.n_max is not None:
            ranked = ranked[:self.n_max]
        return data, images[ranked]


class PickImagesFromMemoryCredit(ImageTransform):
    """Draws images from an ``ImageData`` until a budget of pixels (``credit``, or
    ``n_img`` images of ``img_size``) is spent.  An image is drawn with a
    probability proportional to its pixel area relative to the largest
    candidate plus, with ``k_coverage`` > 0, ``k_coverage`` times the share of
    still unseen points it would add; candidates larger than the remaining
    credit leave the draw (core/data_transform/multimodal/image.py:765-874).
    Draws come from ``np.random`` in the reference's order, one
    ``np.random.choice`` per picked image."""

    _PROCESS_IMAGE_DATA = True

    def __init__(self, credit=None, img_size=[], k_coverage=0, n_img=0):
        if credit is not None:
            self.credit = credit
        elif len(img_size) == 2 and n_img > 0:
            self.credit = img_size[0] * img_size[1] * n_img
        else:
            raise ValueError("Either credit or img_size and n_img must be provided.")
        self.use_coverage = k_coverage > 0
        self.k_coverage = k_coverage

    def _process(self, data, images):
        if images.num_views == 0:
            return data, images
        # candidates in (setting, view) order: owner setting, view, pixel area
        owner = [(i, j) for i, im in enumerate(images) for j in range(im.num_views)]
        area = [images[i].img_size[0] * images[i].img_size[1] for i, _ in owner]
        unseen = None
        if self.use_coverage:
            # points every candidate sees, as rows of a boolean matrix
            sees = torch.zeros(images.num_views, data.num_nodes, dtype=torch.bool)
            first = 0
            for im in images:
                m = im.mappings
                rows = (m.images + first).cpu()
                pts = m.points.repeat_interleave(m.pointers[1:] - m.pointers[:-1]).cpu()
                sees[rows, pts] = True
                first += im.num_views
            unseen = [r.numpy() for r in sees]
        credit = self.credit
        assert credit > 0 and credit >= min(area), \
            f"Insufficient credit={credit} to pick any of the provided images with min_size={min(area)}."
        picked = [[] for _ in range(len(images))]
        while credit > 0 and len(owner) > 0 and credit >= min(area):
            fits = [k for k, a in enumerate(area) if a <= credit]
            owner = [owner[k] for k in fits]
            area = [area[k] for k in fits]
            if unseen is not None:
                unseen = [unseen[k] for k in fits]
                gain = np.array([u.sum() for u in unseen])
                gain = self.k_coverage * gain / (gain.max() + 1)
            else:
                gain = np.zeros(len(owner))
            weights = np.array(area) / np.array(area).max() + gain
            k = np.random.choice(np.arange(weights.shape[0]), p=weights / weights.sum())
            i, j = owner.pop(k)
            credit -= area.pop(k)
            picked[i].append(j)
            if unseen is not None:
                new = unseen.pop(k)
                unseen = [np.logical_and(u, ~new) for u in unseen]
        return data, ImageData([im[torch.LongTensor(idx)] for im, idx in zip(images, picked) if len(idx) > 0])


class CenterRoll(ImageTransform):
    """Rolls spherical images (and their mappings) along the width so that the
    mapped pixels sit as close to the centre column as possible: of
    ``angular_res`` evenly spaced rolls every image takes the one that
    minimises (span of its mapped columns + distance of their middle from the
    centre), columns measured on a 256-step circle
    (core/data_transform/multimodal/image.py:962-1037)."""

    def __init__(self, angular_res=16):
        assert isinstance(angular_res, int)
        assert angular_res <= 256
        self.angular_res = angular_res

    def _process(self, data, images):
        who = f"{self.__class__.__name__} cannot operate if images and mappings underwent prior cropping or resizing."
        assert images.mappings is not None, "No mappings found in images."
        assert images.ref_size[0] == images.img_size[0], who
        assert images.crop_size is None or images.crop_size[0] == images.ref_size[0], who
        assert images.downscale is None or images.downscale == 1, who
        m = images.mappings
        if m.images.shape[0] == 0:
            return data, images
        # mapped columns on the 256-step circle, one row per distinct (image, column)
        img = m.images.repeat_interleave(m.values[1].pointers[1:] - m.values[1].pointers[:-1])
        col = (m.pixels[:, 0].float() * 256 / images.ref_size[0]).long()
        img, col = lexunique(img, col)
        rolls = torch.arange(0, 256, int(256 / self.angular_res), device=col.device)
        rolled = (col.view(-1, 1) + rolls.view(1, -1)) % 256           # uint8 wrap-around of the reference
        index = img.view(-1, 1).expand_as(rolled)
        shape = (images.num_views, rolls.shape[0])
        lo = torch.full(shape, 255, dtype=torch.int64, device=col.device).scatter_reduce_(0, index, rolled, 'amin')
        hi = torch.zeros(shape, dtype=torch.int64, device=col.device).scatter_reduce_(0, index, rolled, 'amax')
        off_centre = ((hi.float() + lo) / 2. - 128).abs().int()
        cost = (hi - lo).int() + off_centre
        best = cost.min(dim=1).indices                                   # (an image without mappings keeps roll 0)
        rollings = (rolls[best] / 256. * images.ref_size[0]).long()
        images.update_rollings(rollings)
        return data, images


def _loaded(images, who):
    """The reference's image-side transforms read the files on demand
    (``images.load()``); file IO is not part of this package: the feature maps
    must be there."""
    if images.x is None:
        raise ValueError(f"{who}: the images hold no feature maps (x is None); load them first")
    return images.x


class AddPixelHeightFeature(ImageTransform):
    """Appends the pixel's height in [0, 1] to the image features
    (core/data_transform/multimodal/image.py:1163-1176)."""

    def _process(self, data, images):
        x = _loaded(images, "AddPixelHeightFeature")
        batch, _, height, width = x.shape
        feat = torch.linspace(0, 1, height).float().to(x.device)      # (the host's values, as the reference)
        images.x = torch.cat((x, feat.view(1, 1, height, 1).repeat(batch, 1, 1, width)), 1)
        return data, images


class AddPixelWidthFeature(ImageTransform):
    """Appends the pixel's width in [0, 1] to the image features
    (core/data_transform/multimodal/image.py:1179-1192)."""

    def _process(self, data, images):
        x = _loaded(images, "AddPixelWidthFeature")
        batch, _, height, width = x.shape
        feat = torch.linspace(0, 1, width).float().to(x.device)       # (the host's values, as the reference)
        images.x = torch.cat((x, feat.view(1, 1, 1, width).repeat(batch, 1, height, 1)), 1)
        return data, images


class RandomHorizontalFlip(ImageTransform):
    """With probability ``p`` (one ``torch.rand(1)`` draw per setting, as the
    reference) flips the feature maps along the width and mirrors the pixel
    columns of the mappings (core/data_transform/multimodal/image.py:1195-1218)."""

    def __init__(self, p=0.50):
        self.p = p

    def _process(self, data, images):
        x = _loaded(images, "RandomHorizontalFlip")
        if torch.rand(1) <= self.p:
            images.x = torch.flip(x, [3])
            width = x.shape[3]
            images.mappings.pixels[:, 0] = width - 1 - images.mappings.pixels[:, 0]
        return data, images


class ToFloatImage(ImageTransform):
    """[0, 255] uint8 feature maps to [0, 1] float32
    (core/data_transform/multimodal/image.py:1221-1232)."""

    def _process(self, data, images):
        x = _loaded(images, "ToFloatImage")
        # a tensor divisor: the device's division by a Python scalar multiplies by the rounded
        # reciprocal, which is one unit in the last place off the reference's quotient for some values
        images.x = x.float() / torch.full((1,), 255., device=x.device)
        return data, images


class Normalize(ImageTransform):
    """Per-channel ``(x - mean) / std`` of float feature maps
    (core/data_transform/multimodal/image.py:1271-1284, a wrapper of
    ``torchvision.transforms.Normalize``, which is restated here: torchvision is
    not a dependency of this package)."""

    def __init__(self, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]):
        self.mean = mean
        self.std = std

    def _process(self, data, images):
        x = _loaded(images, "Normalize")
        if not x.is_floating_point():
            raise TypeError(f"Input tensor should be a float tensor. Got {x.dtype}.")
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device)
        if (std == 0).any():
            raise ValueError(f"std evaluated to zero after conversion to {x.dtype}, leading to division by zero.")
        images.x = (x - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1)
        return data, images

    def __repr__(self):
        return f"Normalize(mean={self.mean}, std={self.std})"


class PickMappingsFromMappingFeatures(ImageTransform):
    """Drops the views whose mapping features fall outside per-feature lower /
    upper bounds, both exclusive (core/data_transform/multimodal/image.py:877-931)."""

    def __init__(self, feat=None, lower=None, upper=None):
        self.feat = self.sanitize(feat)
        self.lower = self.sanitize(lower)
        self.upper = self.sanitize(upper)
        if len(self.lower) == 0:
            self.lower = [None] * len(self.feat)
        if len(self.upper) == 0:
            self.upper = [None] * len(self.feat)
        for x in [self.lower, self.upper]:
            assert len(x) == len(self.feat), \
                f"{x} has {len(x)} elements but {len(self.feat)} were expected."

    @staticmethod
    def sanitize(x):
        if x is None:
            return []
        if isinstance(x, (str, bytes)) or not hasattr(x, '__iter__'):
            return [x]
        return x

    def _process(self, data, images):
        m = images.mappings
        if m is None or not m.has_features or len(self.feat) == 0:
            return data, images
        assert max(self.feat) == 0 or max(self.feat) < m.features.shape[1], \
            f"Out of bounds feature id {max(self.feat)}."
        # one comparison of the selected columns against [lower, upper] rows, a
        # missing bound being infinite
        inf = float('inf')
        cols = m.features.view(m.num_items, -1)[:, list(self.feat)]
        lower = cols.new_tensor([-inf if v is None else v for v in self.lower])
        upper = cols.new_tensor([inf if v is None else v for v in self.upper])
        return data, images.select_views(((cols > lower) & (cols < upper)).all(dim=1))


class JitterMappingFeatures(ImageTransform):
    """Adds clamped gaussian noise to the mapping features
    (core/data_transform/multimodal/image.py:934-959)."""

    def __init__(self, sigma=0.02, clip=0.03):
        self.sigma = sigma
        self.clip = clip

    def _process(self, data, images):
        m = images.mappings
        if m is None or not m.has_features:
            return data, images
        noise = torch.randn_like(m.features).mul_(self.sigma).clamp_(-self.clip, self.clip)
        m.features = m.features + noise
        return data, images


class CropImageGroups(ImageTransform):
    """Greedy cropping of the images around their mappings: every image goes to
    the smallest size of the family (min_size, min_size), (2 min_size, min_size),
    (2 min_size, 2 min_size), ... , img_size that holds the padded bounding box of
    its mapped pixels, the box is centred in the crop as far as the image
    borders allow, and the images of one size come back as one
    SameSettingImageData of an ImageData
    (core/data_transform/multimodal/image.py:1040-1141)."""

    def __init__(self, padding=0, min_size=64):
        assert padding >= 0, f"Expected a positive scalar but got {padding} instead."
        assert ((min_size & (min_size - 1)) == 0) & (min_size != 0), \
            f"Expected a power of two but got {min_size} instead."
        self.padding = padding
        self.min_size = min_size

    def crop_sizes(self, img_size):
        """The family of crop sizes, smallest first, ending with the full image:
        starting from a square of min_size the width and the height double in
        turn, each capped by the image."""
        W, H = img_size
        if self.min_size > W or self.min_size > H:
            return [(W, H)]
        sizes, w, h, widen = [], self.min_size, self.min_size, True
        while (w, h) != (W, H):
            sizes.append((w, h))
            w, h = (min(2 * w, W), h) if widen else (w, min(2 * h, H))
            widen = not widen
        sizes.append((W, H))
        return sizes

    def _process(self, data, images):
        from .images import ImageData
        assert images.mappings is not None, "No mappings found in images."
        if images.num_views == 0:
            return data, ImageData([images])
        W, H = (int(v) for v in images.img_size)
        dev = images.mappings.device
        pix_dtype = images.mappings.pixels.dtype
        box = images.mappings.image_stats(images.num_views)[:, 1:].to(pix_dtype)
        x_lo = (box[:, 0] - self.padding).clamp(min=0)
        x_hi = (box[:, 1] + self.padding).clamp(0, W)
        y_lo = (box[:, 2] - self.padding).clamp(min=0)
        y_hi = (box[:, 3] + self.padding).clamp(0, H)
        width, height = x_hi - x_lo, y_hi - y_lo

        # every image takes the first size of the family that holds its box: one
        # comparison against the whole family instead of a loop over the sizes
        sizes = torch.tensor(self.crop_sizes((W, H)), dtype=torch.int64, device=dev)
        holds = (width[:, None] <= sizes[None, :, 0]) & (height[:, None] <= sizes[None, :, 1])
        holds[:, -1] = True
        family = holds.to(torch.uint8).argmax(dim=1)
        crop_w, crop_h = sizes[family, 0], sizes[family, 1]
        # the box centred in its crop, as far as the image borders allow
        off_x = torch.minimum((x_lo - (crop_w - width) / 2.).long().clamp(min=0), W - crop_w)
        off_y = torch.minimum((y_lo - (crop_h - height) / 2.).long().clamp(min=0), H - crop_h)
        offsets = torch.stack((off_x, off_y), dim=1)

        family = family.cpu()
        out = []
        for k in family.unique().tolist():
            idx = torch.where(family == k)[0].to(dev)
            size = tuple(int(v) for v in sizes[k].tolist())
            out.append(images[idx].update_cropping(size, offsets[idx]))
        return data, ImageData(out)

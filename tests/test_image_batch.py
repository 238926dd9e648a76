"""Collation of images and mappings into per-GPU batches, and the view / image
selections of ImageMapping, against outputs of the reference's own classes
(fixture image_batch.npz).  Every case runs on host tensors (index logic, no
GPU needed) and, marked ``gpu``, on device tensors, where the per-image
reductions and the CSR re-assembly are HIP kernels."""
import numpy as np
import pytest
import torch

from mvfusion.images import ImageBatch, ImageData, SameSettingImageBatch, SameSettingImageData
from mvfusion.mapping import ImageMapping, ImageMappingBatch

DEV = "cpu"


@pytest.fixture(autouse=True, params=["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def dev(request):
    """Device of the tensors the helpers below create."""
    global DEV
    DEV = request.param
    yield request.param
    DEV = "cpu"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.cpu().numpy()


def mapping(g, prefix):
    return ImageMapping.from_csr_arrays(
        T(g[prefix + "pointers"]), T(g[prefix + "images"]), T(g[prefix + "pixels"]),
        T(g[prefix + "features"]), view_pointers=T(g[prefix + "view_pointers"]))


def same(m, g, prefix):
    assert m.device.type == DEV
    assert np.array_equal(N(m.pointers), g[prefix + "pointers"]), prefix
    assert np.array_equal(N(m.images), g[prefix + "images"]), prefix
    assert np.array_equal(N(m.values[1].pointers), g[prefix + "view_pointers"]), prefix
    assert np.array_equal(N(m.pixels), g[prefix + "pixels"]), prefix
    assert np.array_equal(N(m.features), g[prefix + "features"]), prefix


def items(g, tag, size):
    out = []
    for i in range(3):
        p = f"{tag}{i}_"
        n = g[p + "pos"].shape[0]
        out.append(SameSettingImageData(
            path=[f"{tag}{i}_{k}" for k in range(n)], pos=T(g[p + "pos"]),
            extrinsic=torch.eye(4, device=DEV).repeat(n, 1, 1), fx=torch.ones(n, device=DEV),
            fy=torch.ones(n, device=DEV),
            mx=torch.zeros(n, device=DEV), my=torch.zeros(n, device=DEV), ref_size=(size, size), proj_upscale=1,
            downscale=1, x=T(g[p + "x"]), mappings=mapping(g, p)))
    return out


def test_same_setting_image_batch(golden):
    g = golden("image_batch")
    big = items(g, "big", 64)
    b = SameSettingImageBatch.from_data_list(big)
    assert isinstance(b.mappings, ImageMappingBatch)
    same(b.mappings, g, "ssb_")
    assert np.array_equal(N(b.pos), g["ssb_pos"])
    assert np.array_equal(N(b.x), g["ssb_x"])
    assert np.array_equal(b.batch_items_sizes, g["ssb_sizes"])
    assert b.num_views == sum(im.num_views for im in big) and b.num_batch_items == 3
    back = b.to_data_list()
    assert [im.num_views for im in back] == [im.num_views for im in big]
    same(back[1].mappings, g, "ssb_item1_")
    assert np.array_equal(N(back[1].pos), N(big[1].pos))
    assert torch.equal(back[2].x, big[2].x)
    # different settings cannot be batched together
    small = items(g, "small", 32)
    try:
        SameSettingImageBatch.from_data_list([big[0], small[0]])
        raise RuntimeError("expected an AssertionError")
    except AssertionError:
        pass


def test_select_images_and_views(golden):
    g = golden("image_batch")
    m = mapping(g, "big1_")
    same(m.select_images(T(g["sel_idx"])), g, "sel_")
    out, img_idx = m.select_views(T(g["view_mask"]))
    same(out, g, "views_")
    assert np.array_equal(N(img_idx), g["views_img_idx"])
    # all views kept: nothing to renumber
    out, img_idx = m.select_views(torch.ones(m.num_items, dtype=torch.bool, device=DEV))
    assert img_idx is None and torch.equal(out.pointers, m.pointers)
    # no image kept: every point becomes empty
    empty = m.select_images(torch.zeros(0, dtype=torch.int64, device=DEV))
    assert empty.num_items == 0 and int(empty.pointers.max()) == 0
    # selecting views of a SameSettingImageData re-indexes its mappings too
    im = items(g, "big", 64)[1][T(g["sel_idx"])]
    same(im.mappings, g, "sel_")
    assert im.num_views == 2


def test_image_batch(golden):
    g = golden("image_batch")
    big, small = items(g, "big", 64), items(g, "small", 32)
    ib = ImageBatch.from_data_list([ImageData([a, c]) for a, c in zip(big, small)])
    assert len(ib) == 2
    by_size = {int(g[f"ib{k}_ref_size"][0]): k for k in range(2)}
    n_total = int(g["n_pts"].sum())
    for ssb in ib:
        k = by_size[ssb.ref_size[0]]
        same(ssb.mappings, g, f"ib{k}_")
        assert np.array_equal(N(ssb.pos), g[f"ib{k}_pos"])
        assert ssb.mappings.num_groups == n_total
    back = ib.to_data_list()
    for i, il in enumerate(back):
        for k, im in enumerate(il):
            src = (big, small)[k][i]
            assert torch.equal(im.mappings.pointers, src.mappings.pointers)
            assert torch.equal(im.mappings.images, src.mappings.images)
            assert torch.equal(im.mappings.pixels, src.mappings.pixels)
            assert torch.equal(im.mappings.features, src.mappings.features)
            assert torch.equal(im.pos, src.pos)


# --------------------------------------------------------- image selection
class _Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _sample(g):
    n_img = g["pos_img"].shape[0]
    size = int(g["size"])
    return SameSettingImageData(
        path=[f"v{i}" for i in range(n_img)], pos=T(g["pos_img"]), ref_size=(size, size),
        proj_upscale=1, downscale=1, mappings=mapping(g, "in_"))


def _same_images(im, g, tag):
    assert np.array_equal(N(im.pos), g[tag + "_pos"]), tag
    assert [str(p) for p in im.path] == [str(p) for p in g[tag + "_path"]], tag
    same(im.mappings, g, tag + "_")


def test_select_points_and_views_on_images(golden):
    """SameSettingImageData.select_points ('pick' drops unseen images, 'merge'
    keeps them) and select_views.  The mappings' own select_points runs on the
    HIP device, so this test only covers 'pick' and the views here; 'merge'
    is covered on the GPU."""
    g = golden("image_selection")
    _same_images(_sample(g).select_points(T(g["pick"]), mode="pick"), g, "pick")
    _same_images(_sample(g).select_views(T(g["view_mask"])), g, "views")
    im = _sample(g)
    assert im.select_views(torch.ones(im.mappings.num_items, dtype=torch.bool, device=DEV)).num_views == im.num_views
    both = ImageData([_sample(g), _sample(g)]).select_points(T(g["pick"]), mode="pick")
    assert len(both) == 2
    _same_images(both[1], g, "pick")


def test_image_selection_transforms(golden):
    from mvfusion.transforms import (DropImagesOutsideDataBoundingBox, PickImagesFromMappingArea,
                                     PickKImages, SelectMappingFromPointId)
    g = golden("image_selection")
    n = g["xyz"].shape[0]
    sub = g["sub"]
    data = _Data(pos=T(g["xyz"][sub]), mapping_index=T(sub))
    d2, out = SelectMappingFromPointId()(data, _sample(g))
    assert np.array_equal(N(d2.mapping_index), g["smfp_index"])
    _same_images(out, g, "smfp")
    data = _Data(pos=T(g["xyz"]), mapping_index=torch.arange(n, device=DEV))
    _, out = DropImagesOutsideDataBoundingBox(margin=0.5)(data, _sample(g))
    _same_images(out, g, "bbox")
    _, out = PickImagesFromMappingArea(area_ratio=0.02, n_max=3)(data, _sample(g))
    _same_images(out, g, "area")
    _, out = PickImagesFromMappingArea(area_ratio=0.3, use_bbox=True)(data, _sample(g))
    _same_images(out, g, "area_bbox")
    _, out = PickKImages(2)(data, _sample(g))
    assert out.num_views == 3 and [str(p) for p in out.path] == ["v0", "v2", "v4"]


# ------------------------------------------------------------- crop groups
def _crop_sample(g):
    n_img, (W, H) = int(g["n_img"]), (int(v) for v in g["size"])
    x = torch.arange(n_img * 2 * H * W, dtype=torch.float32, device=DEV).view(n_img, 2, H, W)
    return SameSettingImageData(
        path=[f"v{i}" for i in range(n_img)], pos=T(g["pos_img"]), ref_size=(W, H),
        proj_upscale=1, downscale=1, mappings=mapping(g, "in_"), x=x)


def test_bounding_boxes_and_feature_thresholds(golden):
    """ImageMapping.bounding_boxes, PickMappingsFromMappingFeatures and
    JitterMappingFeatures against the reference's own (CropImageGroups, which
    rebuilds the mappings on the HIP device, is checked in the GPU tests)."""
    from mvfusion.transforms import JitterMappingFeatures, PickMappingsFromMappingFeatures
    g = golden("crop_groups")
    boxes = _crop_sample(g).mappings.bounding_boxes
    for k, v in zip(("w_min", "w_max", "h_min", "h_max"), boxes):
        assert np.array_equal(N(v), g["bbox_" + k]) and N(v).dtype == g["bbox_" + k].dtype
    data = _Data(pos=torch.zeros(3, 3, device=DEV))
    _, out = PickMappingsFromMappingFeatures(feat=[0, 2], lower=[0.2, None], upper=[None, 0.7])(
        data, _crop_sample(g))
    assert [str(p) for p in out.path] == [str(p) for p in g["pmf_path"]]
    same(out.mappings, g, "pmf_")
    im = _crop_sample(g)
    before = im.mappings.features.clone()
    _, im = JitterMappingFeatures(sigma=0.5, clip=0.03)(data, im)
    d = (im.mappings.features - before).abs()
    assert 0 < float(d.max()) <= 0.03 + 1e-7


def _roll_sample(g, with_x=True, views=None, size=None):
    """The spherical sample of roll_and_credit.npz, or the views `views` of it at image size `size`."""
    n_img, (W, H) = int(g["n_img"]), (int(v) for v in g["size"])
    size = (W, H) if size is None else size
    sel = np.arange(n_img) if views is None else np.asarray(views)
    m = mapping(g, "in_")
    if views is not None:
        m = m.select_images(torch.as_tensor(sel, device=DEV))
    px = m.pixels.clone()                  # (update_rollings edits the pixels in place)
    px[:, 0] %= size[0]
    px[:, 1] %= size[1]
    m.pixels = px
    x = None
    if with_x:
        x = (torch.arange(n_img * 2 * H * W, dtype=torch.float32, device=DEV).view(n_img, 2, H, W)
             % int(g["x_mod"]))[torch.as_tensor(sel, device=DEV)][:, :, :size[1], :size[0]].contiguous()
    return SameSettingImageData(
        path=[f"s{size[0]}_v{i}" for i in sel], pos=T(g["pos_img"][sel]), ref_size=size,
        proj_upscale=1, downscale=1, mappings=m, x=x)


def test_center_roll(golden):
    """CenterRoll + SameSettingImageData.update_rollings against the reference's
    own run: rollings, rolled mappings (width coordinates modulo the image
    width), rolled feature maps; rollings follow an image through indexing and
    batching."""
    from mvfusion.transforms import CenterRoll
    g = golden("roll_and_credit")
    data = _Data(pos=torch.zeros(3, 3, device=DEV))
    for res in (16, 4, 256):
        _, out = CenterRoll(angular_res=res)(data, _roll_sample(g))
        t = f"roll{res}"
        assert np.array_equal(N(out.rollings), g[t + "_rollings"]) and out.rollings.dtype == torch.int64
        same(out.mappings, g, t + "_")
        assert np.array_equal(N(out.x[:, 0, 0, :8]), g[t + "_x_head"])
        assert np.array_equal(N(out.x.double().sum(dim=(1, 2)))[:, ::37], g[t + "_x_sum_cols"])
    assert np.array_equal(N(out[torch.tensor([4, 1])].rollings), g["roll256_rollings"][[4, 1]])
    batch = SameSettingImageBatch.from_data_list([out[torch.tensor([0, 1])], out[torch.tensor([5])]])
    assert np.array_equal(N(batch.rollings), g["roll256_rollings"][[0, 1, 5]])
    # not after a crop along the width
    im = _roll_sample(g)
    im.crop_size = (256, 256)
    with pytest.raises(AssertionError):
        CenterRoll()(data, im)


def test_pick_images_from_memory_credit(golden):
    """PickImagesFromMemoryCredit on an ImageData of two settings, with the
    reference's seeded draws: same images in the same order, same mappings."""
    from mvfusion.transforms import PickImagesFromMemoryCredit
    g = golden("roll_and_credit")
    W, H = (int(v) for v in g["size"])
    n = mapping(g, "in_").num_groups
    data = _Data(pos=torch.zeros(n, 3, device=DEV), num_nodes=n)
    for tag, kw, seed in (("credit", dict(credit=3 * W * H), 5),
                          ("credit_cov", dict(img_size=[W, H], n_img=2, k_coverage=2), 6),
                          ("credit_small", dict(credit=W * H // 4 + 5, k_coverage=0.5), 7)):
        ims = ImageData([_roll_sample(g, False, [0, 2, 3, 5]), _roll_sample(g, False, [1, 4], size=(W // 2, H // 2))])
        np.random.seed(seed)
        _, out = PickImagesFromMemoryCredit(**kw)(data, ims)
        assert len(out) == int(g[tag + "_n"]), tag
        for j, im in enumerate(out):
            assert [str(p) for p in im.path] == [str(p) for p in g[f"{tag}_{j}_path"]], tag
            same(im.mappings, g, f"{tag}_{j}_")
    with pytest.raises(ValueError):
        PickImagesFromMemoryCredit()
    with pytest.raises(AssertionError):
        PickImagesFromMemoryCredit(credit=10)(data, ImageData([_roll_sample(g, False)]))


def _side_sample(g):
    W, H = (int(v) for v in g["size"])
    m = mapping(g, "in_")
    m.pixels = m.pixels.clone()            # (the flip edits the pixels in place)
    n_img = g["x"].shape[0]
    return SameSettingImageData(
        path=[f"v{i}" for i in range(n_img)], pos=torch.zeros(n_img, 3, device=DEV), ref_size=(W, H),
        proj_upscale=1, downscale=1, mappings=m, x=T(g["x"]))


def test_image_side_transforms(golden):
    """ToFloatImage, AddPixelHeightFeature, AddPixelWidthFeature and RandomHorizontalFlip (the
    flip with torch's seeded draw) against the reference's own run; Normalize against the formula
    of torchvision's transform, which the reference wraps (torchvision is not installed here: no
    fixture, the formula is the pin)."""
    from mvfusion import transforms as tr
    g = golden("image_side")
    data = _Data(pos=torch.zeros(int(g["n"]), 3, device=DEV))
    _, f = tr.ToFloatImage()(data, _side_sample(g))
    assert f.x.dtype == torch.float32 and np.array_equal(N(f.x), g["float_x"])
    _, out = tr.AddPixelHeightFeature()(data, tr.ToFloatImage()(data, _side_sample(g))[1])
    assert np.array_equal(N(out.x), g["height_x"])
    _, out = tr.AddPixelWidthFeature()(data, tr.ToFloatImage()(data, _side_sample(g))[1])
    assert np.array_equal(N(out.x), g["width_x"])
    flipped = 0
    for tag in ("flip1", "flip0", "flip_s3", "flip_s4", "flip_s5", "flip_s6"):
        torch.manual_seed(int(g[tag + "_seed"]))
        _, out = tr.RandomHorizontalFlip(p=float(g[tag + "_p"]))(data, _side_sample(g))
        assert np.array_equal(N(out.x), g[tag + "_x"]), tag
        same(out.mappings, g, tag + "_")
        flipped += int(not np.array_equal(g[tag + "_x"], g["x"]))
    assert 1 < flipped < 6                 # both outcomes of the seeded draw are in the fixture
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    _, out = tr.Normalize(mean, std)(data, f)
    want = (g["float_x"] - np.asarray(mean, np.float32).reshape(1, 3, 1, 1)) / np.asarray(std, np.float32).reshape(1, 3, 1, 1)
    assert np.array_equal(N(out.x), want.astype(np.float32))
    with pytest.raises(TypeError):
        tr.Normalize()(data, _side_sample(g))                  # uint8 maps
    empty = _side_sample(g)
    empty.x = None
    with pytest.raises(ValueError):
        tr.ToFloatImage()(data, empty)

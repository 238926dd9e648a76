"""Host logic of MapImages' staging caches: a cache entry belongs to the very
tensor objects it was built from, in the state they had then."""
import torch

from mvfusion.transforms import _same_sources, _source_key


def test_same_tensors_hit():
    a, b = torch.zeros(4, 3), torch.arange(4)
    key = _source_key("cuda:0", [a, None, b, 2.5])
    assert _same_sources(key, "cuda:0", [a, None, b, 2.5])
    assert not _same_sources(key, "cuda:1", [a, None, b, 2.5])


def test_in_place_write_misses():
    a = torch.zeros(4, 3)
    key = _source_key("d", [a])
    a.add_(1.0)
    assert not _same_sources(key, "d", [a])


def test_replaced_tensor_misses_even_at_the_same_address():
    a = torch.zeros(4, 3)
    key = _source_key("d", [a])
    # an out-of-place update gives another object; the cache keeps `a` alive, so the new
    # tensor cannot land on its storage and be mistaken for it
    b = a @ torch.eye(3)
    assert b.data_ptr() != a.data_ptr()
    assert not _same_sources(key, "d", [b])
    assert not _same_sources(key, "d", [None])
    assert not _same_sources(key, "d", [a, a])


def test_cached_order_of_another_length_is_not_reused():
    from mvfusion.transforms import MapImages

    class Data:
        pass
    d = Data()
    d.pos = torch.rand(10, 3)
    d.mvf_point_order = torch.arange(7)
    t = MapImages(point_order="index")
    assert t._order(d) is None
    d.mvf_point_order = torch.arange(10)
    assert t._order(d) is d.mvf_point_order

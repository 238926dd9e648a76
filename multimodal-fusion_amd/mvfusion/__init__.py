"""mvfusion: MI355X-native multi-view image->point fusion (DeepViewAgg path).

Sub-modules mirror the reference's module layout for the hot path:
``visibility`` (camera projection, splat / depth-map visibility, visibility
models), ``transforms`` (``MapImages``), ``csr`` / ``mapping`` (CSR containers,
``ImageMapping``), ``pooling`` (``BimodalCSRPool``, ``QKVBimodalCSRPool`` ...).
All compute runs in libmvfusion.so (HIP, gfx950) through a C ABI; see
``include/mvfusion.h``.
"""
__version__ = "0.1.0"

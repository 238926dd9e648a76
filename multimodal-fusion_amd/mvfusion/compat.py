"""Drop-in registration: makes the reference's module paths resolve to this
package, so that code which looks the hot path up by name keeps working:

* ``getattr(visibility_module, self.method)`` in ``MapImages``
  (core/data_transform/multimodal/image.py:214-215),
* ``getattr(importlib.import_module("torch_points3d.modules.multimodal.pooling"), name)``
  in the model factory (models/base_architectures/unet.py:62-101),
* ``from torch_points3d.core.multimodal.csr import CSRData`` etc.

Call ``mvfusion.compat.install()`` before the reference modules are imported
(only the listed sub-modules are replaced; the rest of torch_points3d is left
alone).
"""
import importlib
import sys
import types

ALIASES = {
    "torch_points3d.core.multimodal.visibility": "mvfusion.visibility",
    "torch_points3d.core.multimodal.csr": "mvfusion.csr",
    "torch_points3d.modules.multimodal.pooling": "mvfusion.pooling",
}
# names the reference imports from modules we only partly replace
PARTIAL = {
    "torch_points3d.core.multimodal.image": ("mvfusion.mapping", ["ImageMapping", "ImageMappingBatch"]),
    "torch_points3d.core.data_transform.multimodal.image": (
        "mvfusion.transforms",
        ["MapImages", "ImageTransform", "NeighborhoodBasedMappingFeatures",
         "SelectMappingFromPointId", "DropImagesOutsideDataBoundingBox", "PickKImages",
         "PickImagesFromMappingArea", "PickMappingsFromMappingFeatures", "JitterMappingFeatures",
         "CropImageGroups"]),
    "torch_points3d.utils.multimodal": ("mvfusion.mapping", ["lexargsort", "lexargunique", "lexunique"]),
}


def install(replace_partial=False):
    """Register the aliases.  With ``replace_partial`` the listed names are also
    patched into reference modules that are already imported."""
    done = []
    for ref_name, ours in ALIASES.items():
        sys.modules[ref_name] = importlib.import_module(ours)
        done.append(ref_name)
        parent, _, leaf = ref_name.rpartition(".")
        if parent in sys.modules and isinstance(sys.modules[parent], types.ModuleType):
            setattr(sys.modules[parent], leaf, sys.modules[ref_name])
    if replace_partial:
        for ref_name, (ours, names) in PARTIAL.items():
            if ref_name in sys.modules:
                mod = importlib.import_module(ours)
                for n in names:
                    setattr(sys.modules[ref_name], n, getattr(mod, n))
                done.append(ref_name)
    return done

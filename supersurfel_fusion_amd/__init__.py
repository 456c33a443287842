"""supersurfel_fusion_amd -- MI355X-native per-frame hot path of supersurfel_fusion
(extract | ICP | fuse) behind the reference's Frame-in -> Pose+Model-out call surface.

The product is the C-ABI library supersurfel_fusion_amd/csrc/libssf_hip.so (hand-written HIP for
gfx950, header include/ssf.h).  This package holds only the host-side mirror of that interface
(binding.py), the sharded multi-GPU driver (sharded.py) and the synthetic RGB-D scene used by the
tests and the bench (synthetic.py)."""
from .binding import (Fusion, Library, SsfConfig, SsfError, load_product, ABI_SYMBOLS,  # noqa: F401
                      ICP_RECORD, PRODUCT_LIB, SURFEL_FIELDS)

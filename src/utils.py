"""`src.utils` IS tinydiffusionmodels_amd.utils (the same module object, not a re-export): the reference's
tests patch names like `src.utils.download_from_gcs` / `src.utils.Path` (tests/test_utils.py:94-222 there),
and a patch only reaches `load_checkpoint` / `save_samples` if it lands in the namespace those functions
look their globals up in."""
import sys

import tinydiffusionmodels_amd.utils as _impl

sys.modules[__name__] = _impl

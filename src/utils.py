"""Alias of tinydiffusionmodels_amd.utils (same nine functions as the reference's src/utils.py)."""
from tinydiffusionmodels_amd.utils import *  # noqa: F401,F403
from tinydiffusionmodels_amd.utils import storage  # noqa: F401

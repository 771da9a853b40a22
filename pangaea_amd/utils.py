"""The reference's module name for ``pangaea_amd.runtime`` (``from utils import ...`` in src/pangaea.py:17)."""
from .runtime import CustomWeightedRandomSampler, EarlyStopping, init_all, run_cmd  # noqa: F401

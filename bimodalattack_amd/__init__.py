"""MI355X-native joint GCG+PGD attack engine.

Drop-in for ``bimodalattack`` (reference ``bimodalattack/__init__.py:1``): the package
root exports exactly ``BimodalAttackConfig``, ``run`` and ``BimodalAttackResult`` --
plus ``GCGConfig``, nanoGCG's name for the same config.

Importing the package binds ``libbma_hip.so``; if the HIP library has not been built
the import fails loudly (there is no CPU or PyTorch fallback for the attack kernels).
"""

from .config import BimodalAttackConfig, BimodalAttackResult, GCGConfig  # noqa: F401
from .attack import run  # noqa: F401

__all__ = ["BimodalAttackConfig", "run", "BimodalAttackResult", "GCGConfig"]

"""MI355X-native ResNet-26 + attention-MIL hot path.

Importable as `mil_amd` (repo-root shim `mil_amd.py`); the on-disk package directory keeps the
project's hyphenated name."""
from ._lib import BF16X3, LIB_PATH, MilLibraryError, build_library, lib  # noqa: F401
from . import alt_resnet  # noqa: F401
from .encoder import BasicResBlock, ResNet, invalidate_packed_weights  # noqa: F401
from .dist import FlatAdam, FlatParams, gather_features, shard_bags  # noqa: F401
from .train import BagTrainer, load_checkpoint, save_checkpoint, set_stage, stage_for_epoch, visualize_terms, write_attention_map, write_map  # noqa: F401
from .preprocess import S2dTiles, TilePreprocessor  # noqa: F401
from .model import Attention, ContextLayer, CrossEntropyWithProbs, TileParallel  # noqa: F401

__all__ = ["alt_resnet", "Attention", "ResNet", "BasicResBlock", "ContextLayer", "CrossEntropyWithProbs", "TileParallel",
           "FlatParams", "FlatAdam", "shard_bags", "gather_features", "BagTrainer", "set_stage", "stage_for_epoch", "write_attention_map", "write_map", "visualize_terms", "save_checkpoint", "load_checkpoint", "TilePreprocessor", "S2dTiles", "build_library", "lib", "MilLibraryError", "LIB_PATH", "BF16X3", "invalidate_packed_weights"]

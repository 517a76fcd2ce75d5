"""The part of `pig.evaluation` that does not need the Peppa dataset: `load_best_model` (pig/evaluation.py:42-53).
The scoring pipelines of that module (`score`, `full_score`, ...) iterate the private dataset through moviepy and are
out of scope."""
from .checkpoint import load_best_model  # noqa: F401

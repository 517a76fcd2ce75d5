"""Drop-in alias: `import pig.models`, `pig.loss`, `pig.triplet`, ... resolve to the MI355X
implementation in `peppa_amd` (same names and signatures as gchrupala/peppa's `pig` package)."""
import importlib
import sys

for _name in ("util", "loss", "metrics", "triplet", "optimization", "transforms", "execution", "data", "models", "evaluation",
              "targeted_triplets"):
    _mod = importlib.import_module(f"peppa_amd.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    globals()[_name] = _mod

"""`utility` alias of `utility1`: BASELINE.json's north star says `utility/dataloader.Loader`, the reference's
package is `utility1`.  Both import paths resolve to the same modules."""
import importlib
import sys

for _name in ("dataloader", "model", "batch_test", "metrics", "utils", "gpuutil", "Logging"):
    _mod = importlib.import_module("utility1." + _name) if _name not in ("batch_test",) else None
    if _mod is not None:
        sys.modules[__name__ + "." + _name] = _mod
        globals()[_name] = _mod


def __getattr__(name):  # batch_test parses argv at import (like the reference), so bind it lazily
    if name == "batch_test":
        mod = importlib.import_module("utility1.batch_test")
        sys.modules[__name__ + ".batch_test"] = mod
        return mod
    raise AttributeError(name)

"""Runtime caches of the module mirror (packed weight images, codebook images, activation workspaces, ctypes structs)
live HERE, keyed weakly on the owning nn.Module -- never in the module's __dict__.  The reference's modules are plain
nn.Modules that pickle and deep-copy (torch.save(model), EMA copies, checkpoint-by-module); a ctypes struct or a local
lambda in __dict__ breaks both.  A copy simply starts with empty caches and rebuilds them on its first forward."""
from __future__ import annotations

import weakref

_SIDE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def side(mod) -> dict:
    d = _SIDE.get(mod)
    if d is None:
        d = _SIDE[mod] = {}
    return d


def drop(mod, *keys) -> None:
    d = _SIDE.get(mod)
    if d is None:
        return
    if not keys:
        d.clear()
    for k in keys:
        d.pop(k, None)

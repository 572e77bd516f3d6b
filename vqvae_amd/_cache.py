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


# ---- packed images are produced asynchronously on the stream that first misses the cache; a forward on ANOTHER stream that
# hits the cache right afterwards must not read them before the pack kernels finished: every cache entry carries
# the event recorded behind its pack launches, and a hit from a different stream waits on it.
def mark_ready(device):
    """-> (event, stream id) recorded on the current stream of `device` (None on the CPU)."""
    import torch
    if device.type != "cuda":
        return None
    ev = torch.cuda.Event()
    st = torch.cuda.current_stream(device)
    ev.record(st)
    return (ev, st.cuda_stream)


def wait_ready(mark, device):
    import torch
    if mark is None:
        return
    ev, sid = mark
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream != sid:
        cur.wait_event(ev)


def lru_put(table: dict, key, value, cap: int):
    """Insert at the most-recent end; evict only the least recently used entries beyond `cap` (never the whole table: the hot
    entry of a model that alternates two batch sizes must survive)."""
    table.pop(key, None)
    table[key] = value
    while len(table) > cap:
        table.pop(next(iter(table)))


def lru_get(table: dict, key):
    v = table.pop(key, None)
    if v is not None:
        table[key] = v
    return v

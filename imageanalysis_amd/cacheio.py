"""Host I/O around the GPU feature detector: with SIFT at a few
milliseconds per frame the reference's cache formats and the JPEG decode are what an image
costs -- gzip(level 6) of a 50 k x 128 float32 `.desc` takes ~2 s on one core, a 20 MP JPEG
~0.2-0.4 s to decode (scripts/lib/image.py:99-121,140-217).

* `write_gzip(path, payload)`: the file is written by background threads as a MULTI-MEMBER
  gzip stream (1 MiB of payload per member, members compressed in parallel; zlib releases the
  GIL).  gzip.open() -- what the reference's loaders use -- reads such a file exactly like a
  single-member one, and the decompressed bytes are identical to what the reference writes.
* `wait(path)`: readers of a path first wait for its pending write; everything is flushed at
  interpreter exit.
* `Prefetch`: decodes / cache-loads the next few images on worker threads while the GPU works
  on the current one (bounded window: a decoded 20 MP frame is 60 MB).
"""
import atexit
import os
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor

MEMBER_BYTES = 1 << 22               # (few, large zlib calls: every call is a GIL hand-over for the
                                     #  thread that drives the detector)
GZIP_LEVEL = 6                        # the reference's compresslevel (image.py:201,213)

_lock = threading.Lock()
_jobs = None                          # one job per file (serialise + orchestrate)
_workers = None                       # member compression / decode
_pending = {}                         # path -> Future


def _pools():
    global _jobs, _workers
    with _lock:
        if _jobs is None:
            n = max(2, min(96, os.cpu_count() or 2))
            _jobs = ThreadPoolExecutor(max_workers=max(4, min(32, n // 4)), thread_name_prefix='iamx-cache')
            _workers = ThreadPoolExecutor(max_workers=n, thread_name_prefix='iamx-io')
    return _jobs, _workers


def _member(chunk, level, strategy=0):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)            # wbits 31: gzip container
    return c.compress(chunk) + c.flush()


NATIVE_THREADS = 8                    # zlib threads inside one iamx_gzip_members call


def _native_members(bufs, level, member_bytes, strategy=0):
    """all members of one file in ONE call into libiamx (zlib on threads of its own, no GIL): with
    a python future per member the hand-overs of the interpreter lock, not the compression,
    bounded a detection loop on a many-core host (tools/detect_stages.py).  None if the library
    is not there (the pure-python members below are the same stream)."""
    try:
        import ctypes
        import numpy as np
        from . import _lib
        L = _lib.lib()
    except Exception:                                         # noqa: BLE001
        return None
    views = [np.frombuffer(memoryview(b).cast('B'), np.uint8) for b in bufs]
    views = [v for v in views if len(v)]
    lens = np.array([len(v) for v in views], np.int64)
    ptrs = (ctypes.c_void_p * max(len(views), 1))(*[v.ctypes.data for v in views])
    total = int(lens.sum())
    n_members = int(sum((int(n) + member_bytes - 1) // member_bytes for n in lens)) or 1
    cap = int(L.iamx_gzip_records_bound(total, n_members) if isinstance(strategy, tuple)
              else L.iamx_gzip_members_bound(total, n_members))
    out = np.empty(cap, np.uint8)
    if isinstance(strategy, tuple):
        # ('records', width): streams of fixed-width records (the .feat pickle) through libiamx's
        # own DEFLATE encoder, which only looks one record back (iamx_gzip_records)
        n = L.iamx_gzip_records(ptrs, lens.ctypes.data_as(ctypes.c_void_p), len(views), int(member_bytes),
                                int(strategy[1]), NATIVE_THREADS, out.ctypes.data_as(ctypes.c_void_p), cap)
        if n < 0:
            _lib.check(int(n), 'iamx_gzip_records')
        return memoryview(out)[:int(n)]
    n = L.iamx_gzip_members(ptrs, lens.ctypes.data_as(ctypes.c_void_p), len(views), int(member_bytes),
                            int(level), int(strategy), NATIVE_THREADS, out.ctypes.data_as(ctypes.c_void_p), cap)
    if n < 0:
        _lib.check(int(n), 'iamx_gzip_members')
    return memoryview(out)[:int(n)]


def gzip_member_list(raw, level=GZIP_LEVEL, member_bytes=None, strategy=0):
    """bytes-like, or a sequence of bytes-likes that are to follow each other (e.g. an .npy
    header and the array's own memory: nothing is copied together first) -> list of gzip members
    whose concatenation decompresses to those bytes (members compressed in parallel)"""
    bufs = [raw] if isinstance(raw, (bytes, bytearray, memoryview)) else list(raw)
    native = _native_members(bufs, level, member_bytes or MEMBER_BYTES, strategy)
    if native is not None:
        return [native]
    if isinstance(strategy, tuple):
        strategy = 1                          # (no library: zlib's filtered strategy, the same payload)
    chunks = []
    for b in bufs:
        b = memoryview(b).cast('B')
        mb = member_bytes or MEMBER_BYTES
        chunks.extend(b[i:i + mb] for i in range(0, len(b), mb))
    if not chunks:
        chunks = [memoryview(b'')]
    if len(chunks) == 1:
        return [_member(chunks[0], level, strategy)]
    _j, workers = _pools()
    try:
        return list(workers.map(lambda c: _member(c, level, strategy), chunks))
    except RuntimeError:
        # "cannot schedule new futures after interpreter shutdown": the process is exiting while
        # this file is still queued -- compress it right here, the file must not be lost
        return [_member(c, level, strategy) for c in chunks]


def gzip_members(raw, level=GZIP_LEVEL):
    """... the same as ONE bytes object"""
    return b''.join(gzip_member_list(raw, level))


def _write_job(path, payload, on_error=None, level=GZIP_LEVEL, member_bytes=None, strategy=0):
    try:
        raw = payload() if callable(payload) else payload
        members = gzip_member_list(raw, level, member_bytes, strategy)
        # pid + thread id: two ranks may write the same boundary image's cache at the same time
        tmp = '%s.tmp%d.%d' % (path, os.getpid(), threading.get_ident())
        with open(tmp, 'wb') as f:
            f.writelines(members)
        os.replace(tmp, path)                                 # readers never see half a file
    except Exception as e:                                    # noqa: BLE001
        if on_error is None:
            raise
        on_error(e)


def _write_raw_job(path, payload, on_error=None):
    try:
        raw = payload() if callable(payload) else payload
        tmp = '%s.tmp%d.%d' % (path, os.getpid(), threading.get_ident())
        with open(tmp, 'wb') as f:
            if isinstance(raw, (bytes, bytearray, memoryview)):
                f.write(raw)
            else:
                f.writelines(raw)
        os.replace(tmp, path)
    except Exception as e:                                    # noqa: BLE001
        if on_error is None:
            raise
        on_error(e)


def write_raw(path, payload, background=True, on_error=None):
    """uncompressed twin of write_gzip (same ordering / wait() / atomic-replace rules)"""
    if not background:
        _write_raw_job(path, payload, on_error)
        return None
    jobs, _w = _pools()
    wait(path)
    fut = jobs.submit(_write_raw_job, path, payload, on_error)
    with _lock:
        _pending[path] = fut
    return fut


def write_gzip(path, payload, background=True, on_error=None, level=GZIP_LEVEL, member_bytes=None,
               strategy=0):
    """payload: bytes or a callable returning bytes (run on the job thread, e.g. np.save into a
    buffer).  Errors go to `on_error(exc)` if given, else surface in wait().  `level`: zlib level
    of the members (any level reads back the same bytes)."""
    if not background:
        # (the caller waits for this one file: many small members, all workers on it)
        _write_job(path, payload, on_error, level, 1 << 20, strategy)
        return None
    jobs, _w = _pools()
    wait(path)                                                # keep two writes of a path ordered
    fut = jobs.submit(_write_job, path, payload, on_error, level, member_bytes, strategy)
    with _lock:
        _pending[path] = fut
    return fut


def wait(path=None):
    """Block until the pending write of `path` (all paths if None) is on disk; re-raises the
    write's exception."""
    with _lock:
        if path is None:
            futs = list(_pending.items())
        else:
            futs = [(path, _pending[path])] if path in _pending else []
    for p, fut in futs:
        try:
            fut.result()
        finally:
            with _lock:
                if _pending.get(p) is fut:
                    del _pending[p]


def _flush_at_exit():
    try:
        wait()
    except Exception as e:                                    # noqa: BLE001
        print("cache write failed at exit:", e)


atexit.register(_flush_at_exit)
try:
    # runs at threading shutdown, BEFORE concurrent.futures stops accepting work (callbacks run
    # in reverse order of registration and concurrent.futures registered first): queued files
    # are still compressed in parallel when the process ends right after a save
    threading._register_atexit(_flush_at_exit)
except Exception:                                             # noqa: BLE001 (private API)
    pass


class Prefetch(object):
    """Runs `fn(item)` for a sequence of items on the worker threads, at most `depth` results
    outstanding; `take(item)` returns the result (computing it inline if it was never
    scheduled) and schedules the next one."""

    def __init__(self, fn, items, depth=6):
        self.fn = fn
        self.todo = list(items)
        self.depth = depth
        self.next = 0
        self.futs = {}
        self.taken = set()
        # own threads: a job may wait for a pending cache write, which needs the shared workers
        self.workers = ThreadPoolExecutor(max_workers=max(1, depth), thread_name_prefix='iamx-pre')
        for _ in range(min(depth, len(self.todo))):
            self._schedule()

    def _schedule(self):
        while self.next < len(self.todo):
            item = self.todo[self.next]
            self.next += 1
            if id(item) in self.taken:               # (asked for ahead of its turn: done inline)
                continue
            self.futs[id(item)] = (item, self.workers.submit(self.fn, item))
            break

    def take(self, item):
        ent = self.futs.pop(id(item), None)
        if ent is None:
            # not in flight: never listed, or asked for ahead of its turn -- computed inline, and
            # not a second time when its turn comes
            self.taken.add(id(item))
            return self.fn(item)
        self._schedule()
        return ent[1].result()

    def pending(self, item):
        return id(item) in self.futs

    def close(self):
        for _item, fut in self.futs.values():
            fut.cancel()
        self.futs = {}
        self.workers.shutdown(wait=False)

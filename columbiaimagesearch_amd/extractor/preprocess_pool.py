"""Host preprocessing pool that can feed the GPU: N worker processes decode and preprocess image buffers straight into
slots of a shared, page-locked float32 ring; the parent hands whole batches of the ring to the CNN forward.

Counterpart of the reference's per-image fan-out -- DaemonBatchExtractor.run, one image per call in `nb_threads`
processes, each with its own network (cufacesearch/cufacesearch/extractor/generic_extractor.py:49-162, spawned by
cufacesearch/cufacesearch/updater/extraction_processor.py:688-696).  Here the processes only do what must stay host code
(decode, bytescale, LANCZOS resize, crop, mean subtraction: the featurizer's own `preprocess_img`, through its picklable
`preprocess_spec()`), the one network lives on the GPU and sees batches.

Workers are started with the "spawn" method: the parent usually holds a HIP context, which a forked child must not
inherit.  Nothing big is pickled: the encoded images are copied into a shared byte inbox (the pool's one task pipe carries
(slot, offset, length) only -- with the buffers themselves in it, 64 workers starve at ~3.5 k images/s), the preprocessed
tensor (618 KB for DeepSentibank) is written in place into the shared ring.
"""
import multiprocessing as mp
from multiprocessing import shared_memory

import numpy as np

_W = {}  # per-worker state


def cpu_allowance():
    """CPU cores this process may really use: the affinity mask, cut by the container's cgroup-v2 time allowance
    (cpu.max = "<quota us> <period us>"; a GPU box may show 256 CPUs and grant 16 cores of time)."""
    import os
    n = float(len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, float(q) / float(per))
    except Exception:
        pass
    return max(n, 1.0)


def _worker_init(shm_name, inbox_name, n_slots, item_shape, fn, consts):
    shm = shared_memory.SharedMemory(name=shm_name)
    _W["shm"] = shm
    _W["inbox"] = shared_memory.SharedMemory(name=inbox_name)
    _W["ring"] = np.ndarray((n_slots,) + tuple(item_shape), dtype=np.float32, buffer=shm.buf)
    _W["fn"], _W["consts"] = fn, consts


def _worker_task(args):
    slot, buf, off, length = args
    try:
        if buf is None:  # the encoded image waits in the shared inbox
            buf = bytes(_W["inbox"].buf[off:off + length])
        _W["fn"](buf, *_W["consts"], out=_W["ring"][slot])
        return slot, True
    except Exception:
        return slot, False


class PreprocessPool(object):
    """pool = PreprocessPool(featurizer, workers=32, slots=1024); ok = pool.run(buffers) -> (ring view [n, ...], ok flags)."""

    def __init__(self, featurizer, workers=None, slots=512):
        import os
        self.fn, self.consts = featurizer.preprocess_spec()
        probe_shape = tuple(np.asarray(featurizer.mu).shape)  # the tensor a preprocessed image becomes
        self.item_shape = probe_shape
        self.slots = int(slots)
        # default: 1.5 workers per granted core (measured on a 16-core allowance: 16 workers 5.4 k, 24 workers 5.7 k, 64 workers
        # 4.8 k images/s -- profiles/r03l_ingest_buffers.txt)
        self.workers = int(workers or max(1, min(int(cpu_allowance() * 1.5 + 0.5), 96)))
        nbytes = self.slots * int(np.prod(self.item_shape)) * 4
        self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
        self.ring = np.ndarray((self.slots,) + self.item_shape, dtype=np.float32, buffer=self._shm.buf)
        self.inbox_per_slot = 192 * 1024  # bytes of encoded image per slot on average; larger ones travel by pickle
        self._inbox = shared_memory.SharedMemory(create=True, size=self.slots * self.inbox_per_slot)
        self._pinned = False
        try:  # page-lock the ring so that the host-to-device copy of a batch runs at PCIe speed
            import torch
            if torch.cuda.is_available() and not os.environ.get("CIS_POOL_NO_PIN"):
                rc = torch.cuda.cudart().cudaHostRegister(self.ring.ctypes.data, nbytes, 0)
                self._pinned = int(rc) == 0
        except Exception:
            self._pinned = False
        ctx = mp.get_context("spawn")
        self._pool = ctx.Pool(self.workers, initializer=_worker_init,
                              initargs=(self._shm.name, self._inbox.name, self.slots, self.item_shape, self.fn, self.consts))

    def start(self, buffers, first_slot=0, chunksize=8):
        """Hand `buffers` to the workers, slot first_slot + k for buffer k; returns a handle for finish().  Several
        batches may be in flight in disjoint slot ranges (the caller overlaps the GPU work of one with the decoding of
        the next)."""
        n = len(buffers)
        if first_slot < 0 or first_slot + n > self.slots:
            raise ValueError("slots %d..%d outside the ring of %d" % (first_slot, first_slot + n, self.slots))
        # the batch's share of the inbox: the bytes behind its own slots
        pos, end = first_slot * self.inbox_per_slot, (first_slot + n) * self.inbox_per_slot
        tasks = []
        for k, b in enumerate(buffers):
            ln = len(b) if isinstance(b, (bytes, bytearray, memoryview)) else -1
            if 0 <= ln <= end - pos:
                self._inbox.buf[pos:pos + ln] = b
                tasks.append((first_slot + k, None, pos, ln))
                pos += ln
            else:
                tasks.append((first_slot + k, b, 0, 0))
        res = self._pool.map_async(_worker_task, tasks, chunksize=chunksize)
        return res, first_slot, n

    def finish(self, handle):
        """Wait for a start() handle: (ring[first_slot : first_slot + n], ok flags) -- a view, valid until those slots are reused."""
        res, first_slot, n = handle
        ok = np.zeros(n, dtype=bool)
        for slot, good in res.get():
            ok[slot - first_slot] = good
        return self.ring[first_slot:first_slot + n], ok

    def run(self, buffers, chunksize=8):
        """Preprocess up to `slots` buffers; returns (ring[:n], ok) -- a view, valid until the next call."""
        if len(buffers) > self.slots:
            raise ValueError("at most %d buffers per call" % self.slots)
        return self.finish(self.start(buffers, 0, chunksize))

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None
        if getattr(self, "_shm", None) is not None:
            if self._pinned:
                try:
                    import torch
                    torch.cuda.cudart().cudaHostUnregister(self.ring.ctypes.data)
                except Exception:
                    pass
            self.ring = None
            for m in (self._shm, self._inbox):
                try:
                    m.close()
                    m.unlink()
                except Exception:
                    pass
            self._shm = self._inbox = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

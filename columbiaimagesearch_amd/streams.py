"""Lane streams for batches in flight.

The GPU dispatches kernels from four hardware pipes.  HIP gives every stream a hardware queue when the stream first submits work, and a
queue lands on pipe (order of acquisition) mod 4, whatever GPU_MAX_HW_QUEUES says (measured: tools/r06_queue_probe.py).  Two streams on
one pipe do not overlap, and a stream that holds an event wait blocks its pipe for the others on it.  So the streams of batches that are
meant to overlap (index views, CNN views: INTEGRATION.md) should be made back to back, submit something at once, and be reused for the
life of the process -- streams made later, after a varying number of others, land on whatever pipe comes next.

    from columbiaimagesearch_amd.streams import lane_streams
    s0, s1, s2 = lane_streams(3)          # three streams on three different pipes (the default stream sits on a fourth)
"""
_POOL = {}


def lane_streams(n, device=None):
    """The first `n` lane streams of `device` (torch.cuda.Stream objects; the pool grows in fours, each new stream submits a tiny fill at
    once so that it takes its hardware queue now, next to its siblings).  More than four lanes share pipes: lanes i and i + 4 do not
    overlap with each other."""
    import torch
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    pool = _POOL.setdefault(str(device), [])
    want = max(int(n), 1)
    want = (want + 3) // 4 * 4
    while len(pool) < want:
        s = torch.cuda.Stream(device=device)
        with torch.cuda.stream(s):
            torch.zeros(1, device=device)
        pool.append(s)
    return pool[:int(n)]

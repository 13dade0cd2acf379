"""Fork safety (include/cis_hip.h: "HIP is initialised lazily on first use, so a process may fork before touching the library"): the
reference's callers fork -- gunicorn workers, multiprocessing.Process per extractor.  The parent imports the package and
builds the model OBJECT (no device work), forks two children that each encode + search on the GPU and report checksums, then does the same
itself: all three must agree with each other and with the golden codes.  Usage: python tests/tools/fork_check.py"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conftest import load_golden
from test_lopq_hip_parity import hip_model
from columbiaimagesearch_amd.lopq import LOPQSearcherHIP

z, X, Q = load_golden("c2")
m = hip_model(z)  # parameters only: the device handle is created on first use


def work():
    coarse, fine = m.predict_batch(X[:5000])
    assert (coarse == z["coarse"][:5000]).all() and (fine == z["fine"][:5000]).all()
    s = LOPQSearcherHIP(m)
    s.add_codes_array(coarse, fine)
    r = s.search_batch(Q[:8], quota=500, limit=20)
    return hashlib.sha1(r["ids"].tobytes() + r["dists"].tobytes()).hexdigest()


pipes = []
for _ in range(2):
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(rd)
        try:
            os.write(wr, work().encode())
            os._exit(0)
        except BaseException as e:  # noqa
            os.write(wr, ("ERR %r" % (e,)).encode())
            os._exit(1)
    os.close(wr)
    pipes.append((pid, rd))
got = []
for pid, rd in pipes:
    data = os.read(rd, 4096).decode()
    _, status = os.waitpid(pid, 0)
    assert status == 0, data
    got.append(data)
mine = work()  # the parent touches the device only now, after its children did
assert got[0] == got[1] == mine, (got, mine)
print("fork check ok: two forked children and the parent agree (%s)" % mine[:12])

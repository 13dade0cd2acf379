"""Runs the host side of libcis_hip.so under AddressSanitizer (`make -C columbiaimagesearch_amd/csrc asan`; this process is started
with LD_PRELOAD = the clang ASan runtime and CIS_LIB_PATH = the instrumented library -- tests/test_abi.py does that).
Without a GPU: the argument-validation and error paths of the entry points (every call must come back with an error code, nothing
may touch freed or foreign memory).  With a GPU (argument `gpu`): a small encode / insert / search / async-search / view life cycle
as well, so that the workspace bookkeeping of real calls is covered."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from columbiaimagesearch_amd import _lib  # noqa: E402

L = _lib.lib()
assert "asan" in _lib.LIB_PATH, _lib.LIB_PATH
bad = 0


def expect_error(rc, what):
    global bad
    if rc == 0:
        bad += 1
        print("NO ERROR from", what)


out = ctypes.c_void_p()
# model: bad shapes, NULL parameters
expect_error(L.cis_model_create(ctypes.byref(out), 8, 8, 4, 3, 16, 8, None, None, None, None, None, None, 8, 0), "cis_model_create(M odd)")
expect_error(L.cis_model_create(ctypes.byref(out), 8, 8, 0, 4, 16, 8, None, None, None, None, None, None, 8, 0), "cis_model_create(V = 0)")
expect_error(L.cis_model_create(None, 8, 8, 4, 4, 16, 8, None, None, None, None, None, None, 8, 0), "cis_model_create(out NULL)")
Cs = np.zeros((2, 4, 4)); Rs = np.zeros((2, 4, 4, 4)); mus = np.zeros((2, 4, 4)); subs = np.zeros((4, 16, 2))
rc = L.cis_model_create(ctypes.byref(out), 8, 8, 4, 4, 16, 8, _lib.ptr(Cs), _lib.ptr(Rs), _lib.ptr(mus), _lib.ptr(subs), None, None, 8, 0)
have_gpu = rc == 0
model = out.value if have_gpu else None
if not have_gpu:
    assert rc == _lib.CIS_ENODEVICE, (rc, _lib.last_error())
# index: NULL handles and arguments
ix = ctypes.c_void_p()
expect_error(L.cis_index_create(ctypes.byref(ix), None), "cis_index_create(model NULL)")
expect_error(L.cis_index_create(None, None), "cis_index_create(out NULL)")
expect_error(L.cis_index_create_view(ctypes.byref(ix), None), "cis_index_create_view(base NULL)")
expect_error(L.cis_index_set_scan_mode(None, 0), "cis_index_set_scan_mode(NULL)")
expect_error(L.cis_index_search(None, None, 8, 1, 10, 10, None, None, None, None, None, None), "cis_index_search(NULL)")
expect_error(L.cis_index_search_async(None, None, 8, 1, 10, 10, None, None, None, None, None, None), "cis_index_search_async(NULL)")
expect_error(L.cis_index_search_wait(None), "cis_index_search_wait(NULL)")
expect_error(L.cis_index_add(None, None, None, None, 1, 1, None), "cis_index_add(NULL)")
expect_error(L.cis_index_stream_counters(None, None), "cis_index_stream_counters(NULL)")
expect_error(L.cis_index_insert_counters(None, None), "cis_index_insert_counters(NULL)")
expect_error(L.cis_index_last_stats(None, None), "cis_index_last_stats(NULL)")
expect_error(L.cis_index_read_profile(None, None, None), "cis_index_read_profile(NULL)")
L.cis_index_destroy(None)
L.cis_model_destroy(None)
L.cis_cnn_destroy(None)
L.cis_host_free(None)
expect_error(L.cis_host_alloc(None, 16), "cis_host_alloc(out NULL)")
n, b = ctypes.c_int64(), ctypes.c_int64()
assert L.cis_alloc_stats(ctypes.byref(n), ctypes.byref(b)) == 0 and L.cis_alloc_stats(None, None) == 0
expect_error(L.cis_cnn_create(ctypes.byref(out), 2, None, 117), "cis_cnn_create(tensors NULL)")
expect_error(L.cis_cnn_create(ctypes.byref(out), 7, None, 0), "cis_cnn_create(arch 7)")
expect_error(L.cis_multisequence(None, 8, None, None, 8, 1, 4, 4, 4, None, None, None), "cis_multisequence(NULL)")
x = np.zeros((1, 8)); cells = np.zeros((1, 4, 2), dtype=np.int32); dd = np.zeros((1, 4)); dt = ctypes.c_int(0)
rc = L.cis_multisequence(_lib.ptr(x), 8, _lib.ptr(Cs[0]), _lib.ptr(Cs[1]), 8, 1, 4, 4, 4, _lib.ptr(cells), _lib.ptr(dd), ctypes.byref(dt))
assert rc == 0 if have_gpu else rc == _lib.CIS_ENODEVICE, rc
assert isinstance(_lib.last_error(), str)

if have_gpu and len(sys.argv) > 1 and sys.argv[1] == "gpu":
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
    from conftest import load_golden
    from test_lopq_hip_parity import hip_model
    from columbiaimagesearch_amd.lopq import LOPQSearcherHIP
    z, X, Q = load_golden("c2")
    m = hip_model(z)
    co, fi = m.predict_batch(X[:5000])
    assert (co == z["coarse"][:5000]).all() and (fi == z["fine"][:5000]).all()
    s = LOPQSearcherHIP(m)
    s.add_codes_array(co, fi)
    s.add_codes_array(co[:100], fi[:100])                       # duplicates: in-place insert path with the dedup walk
    s.add_codes_array(co[:300], fi[:300], ids=np.arange(10 ** 6, 10 ** 6 + 300))
    v = s.view()
    for mode in (0, 1, 2, 3, 5, 6, 7):
        s.set_scan_mode(mode=mode)
        r = s.search_batch(Q[:9], quota=800, limit=30)
        assert (r["n_found"] > 0).all()
    s.set_scan_mode(mode=0)
    r1 = s.search_batch(Q[:20], quota=5000, limit=None)        # limit = quota: the segmented sort
    o = v.search_batch_async(Q[:20], quota=5000, limit=100)
    v.search_wait()
    assert (o["ids"][:, :100] == r1["ids"][:, :100]).all()
    res, visited = s.search(Q[0], quota=100, limit=10, with_dists=True)
    assert len(res) == 10
    v.close()
    s.close()
    print("gpu life cycle ok")
if model:
    L.cis_model_destroy(model)
print("asan abi paths ok" if bad == 0 else "FAILED: %d calls returned no error" % bad)
sys.stdout.flush()
# Leave without the interpreter's and the HIP runtime's exit-time teardown: on some boxes of the pool the sanitizer's own device-allocator
# hook aborts there ("sanitizer_allocator_device.h: dev_runtime_unloaded_") after every handle of this library has been destroyed above and
# hangs the process -- the library has no static object that owns device memory, so nothing of it runs at exit.
__import__("os")._exit(1 if bad else 0)

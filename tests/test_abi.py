"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/cis_hip.h declares, the ctypes table matches the header, and compute entry points fail
loudly (never fall back to a CPU path) when no MI355X is visible."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO, has_gpu, load_golden

HEADER = os.path.join(REPO, "include", "cis_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cis_[a-z0-9_]+)\s*\(", src)))


def test_header_and_ctypes_table_agree():
    from columbiaimagesearch_amd import _lib
    assert declared_symbols() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    from columbiaimagesearch_amd import _lib
    L = _lib.lib()
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert L.cis_version() >= 100
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (cis_[a-z0-9_]+)", out))
    assert set(declared_symbols()) <= exported


def test_hit_struct_layout():
    from columbiaimagesearch_amd import _lib
    assert ctypes.sizeof(_lib.cis_hit) == 32
    assert _lib.HIT_DTYPE.fields["id"][1] == 16 and _lib.HIT_DTYPE.fields["cell"][1] == 24


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    from columbiaimagesearch_amd import _lib
    from test_lopq_hip_parity import hip_model
    z, X, Q = load_golden("tiny")
    m = hip_model(z)
    with pytest.raises(_lib.HipError):
        m.predict(X[0])
    assert "no CPU fallback" in _lib.last_error() or "HIP" in _lib.last_error()


def test_argument_errors_are_python_exceptions():
    from columbiaimagesearch_amd import _lib
    from columbiaimagesearch_amd.lopq import LOPQModel
    with pytest.raises(ValueError):
        LOPQModel(V=4, M=4).predict(np.zeros(8))  # no parameters yet
    L = _lib.lib()
    out = ctypes.c_void_p()
    rc = L.cis_model_create(ctypes.byref(out), 8, 8, 4, 3, 16, 8, None, None, None, None, None, None, 8, 0)
    assert rc == _lib.CIS_EINVAL and out.value is None
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_hand_counted_waits_behind_inline_asm_loads(tmp_path):
    """tools/check_asm_waits.py on the device assembly of csrc/cnn.hip and csrc/lopq_search.hip (`make check-asm`): nothing the
    compiler schedules may touch the destination registers of an inline-assembly load before an inline wait.  The checker itself is
    exercised on a hand-made listing first (a copy of a loaded register before the wait must be reported)."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(repo, "tools", "check_asm_waits.py")
    bad = tmp_path / "bad.s"
    bad.write_text("_Z3foov:\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[4:7], v[2:3], off\n\t;;#ASMEND\n\tv_mov_b32_e32 v9, v5\n"
                   "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b32_e32 v10, v5\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, tool, str(bad)], stdout=subprocess.PIPE)
    assert r.returncode == 1 and b"1 violations" in r.stdout and b"v_mov_b32_e32 v9, v5" in r.stdout
    good = tmp_path / "good.s"
    good.write_text(bad.read_text().replace("\tv_mov_b32_e32 v9, v5\n", ""))
    assert subprocess.run([sys.executable, tool, str(good)], stdout=subprocess.PIPE).returncode == 0
    r = subprocess.run(["make", "-C", os.path.join(repo, "columbiaimagesearch_amd", "csrc"), "check-asm"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1200)
    assert r.returncode == 0 and b" 0 violations" in r.stdout, r.stdout.decode()[-2000:]


def _asan_env():
    import glob
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    lib = os.path.join(REPO, "columbiaimagesearch_amd", "lib", "libcis_hip_asan.so")
    if not rt:
        pytest.skip("no clang AddressSanitizer runtime in this image")
    return rt[-1], lib


def test_asan_build_of_the_host_shim():
    """`make asan`: the library with its HOST side under AddressSanitizer (device code as usual); tests/tools/asan_abi_paths.py then
    drives the argument-validation and error paths of the entry points through it in a fresh interpreter (LD_PRELOAD of the ASan
    runtime).  No GPU needed: every call must come back with an error code and ASan must stay silent."""
    import subprocess, sys
    rt, lib = _asan_env()
    r = subprocess.run(["make", "-C", os.path.join(REPO, "columbiaimagesearch_amd", "csrc"), "asan", "-j4"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0 and os.path.exists(lib), r.stdout.decode()[-2000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23", CIS_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "tools", "asan_abi_paths.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    text = r.stdout.decode()
    assert r.returncode == 0 and "asan abi paths ok" in text and "AddressSanitizer" not in text, text[-3000:]


@pytest.mark.gpu
def test_asan_host_shim_through_a_real_life_cycle():
    """The same library on a GPU: encode, inserts (bulk, duplicates, in place), every scan route, limit = quota, the asynchronous host
    entry points on a view, close order -- real workspace bookkeeping under AddressSanitizer."""
    import subprocess, sys
    rt, lib = _asan_env()
    if not os.path.exists(lib):
        pytest.skip("libcis_hip_asan.so was not built (make -C columbiaimagesearch_amd/csrc asan)")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23:protect_shadow_gap=0", CIS_LIB_PATH=lib)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "tools", "asan_abi_paths.py"), "gpu"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    text = r.stdout.decode()
    assert r.returncode == 0 and "gpu life cycle ok" in text and "asan abi paths ok" in text and "AddressSanitizer" not in text, text[-3000:]

"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/blight_mi355x.h
declares, and parameter validation reproduces the reference's ctor errors (blight.h:75-92).  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "blight_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blmx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import blight_amd
    from blight_amd import build
    build.build()
    lib = C.CDLL(blight_amd.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s
    lib.blmx_version.restype = C.c_char_p
    assert b"gfx950" in lib.blmx_version()


def test_ctor_validation_messages():
    import blight_amd
    lib = blight_amd.load_library()
    cases = [((33, 7, 5, 3, 1, 6, 8), "kmer size too large"), ((31, 10, 5, 3, 1, 6, 8), "minimizer_length must be odd"),
             ((31, 17, 5, 3, 1, 6, 8), "minimizer_length size too large"),
             ((31, 7, 14, 3, 1, 6, 8), "log2_mphfs_number must not be larger than 2*minimizer_length - 1"),
             ((31, 7, 5, 6, 1, 6, 8), "log2_superbuckets_number must not be larger than log2_mphfs_number"),
             ((64, 9, 7, 3, 1, 8, 16), "kmer size too large")]
    for (k, m, n, s, t, b, kb), msg in cases:
        h = C.c_void_p()
        p = blight_amd.blmx_params(k, m, n, s, t, b, kb, 0)
        rc = lib.blmx_create(C.byref(p), C.byref(h))
        assert rc == blight_amd.BLMX_EINVAL and not h.value
        assert lib.blmx_create_error().decode() == msg
        with pytest.raises(ValueError, match=msg.replace("*", r"\*")):
            blight_amd.kmer_Set_Light(k, m, n, s, t, b, kmer_bytes=kb)


def test_peek_params_rejects_foreign_files(tmp_path):
    import blight_amd
    lib = blight_amd.load_library()
    p = blight_amd.blmx_params()
    assert lib.blmx_peek_params(os.fsencode(str(tmp_path / "missing.blmx")), C.byref(p)) == blight_amd.BLMX_EIO
    bogus = tmp_path / "bogus.blmx"
    bogus.write_bytes(b">seq\nACGT\n" * 100)
    assert lib.blmx_peek_params(os.fsencode(str(bogus)), C.byref(p)) == blight_amd.BLMX_EIO
    with pytest.raises(OSError):
        blight_amd.kmer_Set_Light.from_file(str(bogus))


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product must fail loudly (BLMX_EHIP), never compute on the host."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    import blight_amd
    with pytest.raises(blight_amd.BlightError) as e:
        blight_amd.kmer_Set_Light(31, 7, 5, 3, 1, 6)
    assert e.value.code == blight_amd.BLMX_EHIP


def test_product_never_references_the_oracle():
    """The shipped sources must not import / include / link anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "blight_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|oracle/|liborc|blight_oracle)", txt):
                    bad.append(f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        if re.search(r"(liborc|blight_oracle)", open(os.path.join(ROOT, "include", f)).read()):
            bad.append(f)
    assert not bad, bad


def test_missing_rccl_is_an_error_code_not_a_crash(tmp_path):
    """the loader's error text used dlerror() twice (the second call returns NULL -> std::string + NULL).  With the library
    name pointed at a file that does not exist, blmx_comm_unique_id must return BLMX_ERCCL and leave a message -- on any host, GPU or
    not (no device is touched before the library is loaded)."""
    import subprocess
    import sys
    code = (
        "import os, sys, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "os.environ['BLIGHT_NO_TORCH'] = '1'\n"
        "os.environ['BLIGHT_RCCL_LIB'] = %r\n"
        "import blight_amd as B\n"
        "lib = B.load_library()\n"
        "buf = C.create_string_buffer(B.COMM_ID_BYTES)\n"
        "rc = lib.blmx_comm_unique_id(buf)\n"
        "msg = lib.blmx_last_error(None).decode()\n"
        "assert rc == B.BLMX_ERCCL, rc\n"
        "assert 'librccl_missing' in msg and len(msg) > 20, msg\n"
        "lb = C.create_string_buffer(B.COMM_ID_BYTES)\n"
        "assert lib.blmx_comm_loopback_id(lb) == 0 and lb.raw[:13] == b'BLMX-LOOPBACK'\n"
        "print('ERCCL-OK')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "librccl_missing.so")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ERCCL-OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])

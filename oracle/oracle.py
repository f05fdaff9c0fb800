"""ctypes bindings for the CPU oracle (oracle/liborc.so) and for the real reference (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (blight_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBORC = os.path.join(HERE, "liborc.so")
REF64 = os.path.join(HERE, "_ref", "libblight_ref64.so")
REF128 = os.path.join(HERE, "_ref", "libblight_ref128.so")

NOT_FOUND = 0xFFFFFFFFFFFFFFFF
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


def build(force=False):
    """(Re)build liborc.so; build _ref/ too when /root/reference is present."""
    if force or not os.path.exists(LIBORC) or os.path.getmtime(LIBORC) < max(
        os.path.getmtime(os.path.join(HERE, f)) for f in ("blight_oracle.c", "oracle_impl.h")
    ):
        subprocess.check_call(["make", "-s", "-C", HERE, "liborc.so"])
    if os.path.isdir("/root/reference") and (force or not (os.path.exists(REF64) and os.path.exists(REF128))):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _ptr(a, t):
    return a.ctypes.data_as(t)


def kmers_to_array(kmers, kmer_bytes):
    """list/array of python ints -> little-endian (nq, kmer_bytes/8) uint64 array"""
    if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint64:
        return np.ascontiguousarray(kmers)
    kmers = [int(x) for x in kmers]
    if kmer_bytes == 8:
        return np.array(kmers, dtype=np.uint64)
    out = np.empty((len(kmers), 2), dtype=np.uint64)
    for i, x in enumerate(kmers):
        out[i, 0] = x & 0xFFFFFFFFFFFFFFFF
        out[i, 1] = x >> 64
    return out


def array_to_kmers(arr, kmer_bytes):
    if kmer_bytes == 8:
        return [int(x) for x in arr.reshape(-1)]
    arr = arr.reshape(-1, 2)
    return [int(arr[i, 0]) | (int(arr[i, 1]) << 64) for i in range(arr.shape[0])]


def concat_sequences(seqs):
    """list of str/bytes -> (bytes buffer, uint64 offsets[n+1])"""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), offs


def read_fasta(path):
    """The reference's 2-line reader (blight.cpp:212-229): returns list of sequence strings."""
    lib = _orc()
    buf = C.c_void_p()
    offs = u64p()
    n = C.c_uint64()
    rc = lib.orc_read_fasta(path.encode(), C.byref(buf), C.byref(offs), C.byref(n))
    if rc:
        raise IOError(path)
    total = offs[n.value]
    data = C.string_at(buf, total)
    o = [offs[i] for i in range(n.value + 1)]
    lib.orc_free_buf(buf)
    lib.orc_free_buf(offs)
    return [data[o[i]:o[i + 1]].decode() for i in range(n.value)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("number_kmer", "number_super_kmer", "total_nt", "largest_mphf",
                                          "largest_bucket", "positions_bits", "seq_words")] + \
               [("nbuckets", C.c_uint32), ("ngroups", C.c_uint32)] + \
               [(n, C.c_int) for n in ("k", "m", "n", "b", "kmer_bytes")]


_orc_lib = None


def _orc():
    global _orc_lib
    if _orc_lib is None:
        build()
        lib = C.CDLL(LIBORC)
        lib.orc_build.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, u64p,
                                  C.c_uint64]
        lib.orc_free.argtypes = [C.c_void_p]
        lib.orc_import.argtypes = [C.POINTER(C.c_void_p)] + [C.c_int] * 5 + [C.c_uint64, C.c_uint64, u32p, u64p, u64p, u64p, C.c_uint64,
                                   u64p, u64p, u64p, u32p, u64p, u64p, u64p, u64p, u64p, u32p, C.c_uint64]
        lib.orc_query_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, i64p, C.c_int]
        lib.orc_query_kmers_mini.argtypes = [C.c_void_p, C.c_void_p, u32p, C.c_uint64, i64p, C.c_int]
        lib.orc_query_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, i64p, C.c_void_p]
        lib.orc_query_sequence.restype = C.c_int64
        lib.orc_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
        for name, rt in (("orc_bucket_nt", u32p), ("orc_bucket_start", u64p), ("orc_seq", u64p), ("orc_positions", u64p),
                         ("orc_group_nkeys", u64p), ("orc_group_prefix", u64p), ("orc_group_posstart", u64p),
                         ("orc_group_nbits", u8p)):
            getattr(lib, name).argtypes = [C.c_void_p]
            getattr(lib, name).restype = rt
        lib.orc_mphf_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(u64p), C.POINTER(u64p), u64p, C.POINTER(C.c_int)]
        lib.orc_mphf_info.restype = C.c_uint64
        lib.orc_mphf_fallback.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, u64p]
        lib.orc_mphf_lookup.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.orc_mphf_lookup.restype = C.c_uint64
        lib.orc_check_params.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_char_p)]
        lib.orc_read_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(u64p), u64p]
        lib.orc_free_buf.argtypes = [C.c_void_p]
        lib.orc_p_nuc2int.argtypes = [C.c_char]
        lib.orc_p_revhash32.argtypes = [C.c_uint32]
        lib.orc_p_revhash32.restype = C.c_int32
        lib.orc_p_rc32.argtypes = [C.c_uint32, C.c_int]
        lib.orc_p_rc32.restype = C.c_uint32
        lib.orc_p_parity_canon.argtypes = [C.c_uint32, C.c_int]
        lib.orc_p_parity_canon.restype = C.c_uint32
        lib.orc_p_rc64.argtypes = [C.c_uint64, C.c_int]
        lib.orc_p_rc64.restype = C.c_uint64
        lib.orc_p_rc128.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.orc_p_minimizer64.argtypes = [C.c_uint64, C.c_int, C.c_int]
        lib.orc_p_minimizer64.restype = C.c_uint32
        lib.orc_p_minimizer128.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.orc_p_minimizer128.restype = C.c_uint32
        lib.orc_p_hash_bis.argtypes = [C.c_uint64, C.c_uint64]
        lib.orc_p_hash_bis.restype = C.c_uint64
        lib.orc_p_level_hashes.argtypes = [C.c_uint64, C.c_uint, u64p]
        lib.orc_p_mulhi64.argtypes = [C.c_uint64, C.c_uint64]
        lib.orc_p_mulhi64.restype = C.c_uint64
        lib.orc_p_configure_levels.argtypes = [C.c_uint64, u64p]
        lib.orc_p_configure_levels.restype = C.c_uint64
        lib.orc_p_murmur3_x64_128.argtypes = [C.c_char_p, C.c_int, C.c_uint32, u64p]
        lib.orc_p_murmur3_x64_128.restype = None
        lib.orc_p_str2num64.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        lib.orc_p_str2num64.restype = C.c_uint64
        lib.orc_p_chop.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, u32p, u32p, u32p, C.c_uint64]
        lib.orc_p_chop.restype = C.c_int64
        _orc_lib = lib
    return _orc_lib


class OracleError(Exception):
    def __init__(self, code):
        super().__init__({-1: "invalid argument", -2: "sequence shorter than k", -3: "Invalid char in DNA"}.get(code, str(code)))
        self.code = code


class Oracle:
    """CPU restatement of kmer_Set_Light (index build + queries)."""

    def __init__(self, k, m, n, b, seqs, kmer_bytes=None):
        self.lib = _orc()
        self.kmer_bytes = kmer_bytes or (8 if k <= 32 else 16)
        self.k, self.m, self.n, self.b = k, m, n, b
        buf, offs = concat_sequences(seqs)
        h = C.c_void_p()
        rc = self.lib.orc_build(C.byref(h), k, m, n, b, self.kmer_bytes, buf, _ptr(offs, u64p), len(offs) - 1)
        if rc:
            raise OracleError(rc)
        self.h = h
        st = _Stats()
        self.lib.orc_stats(self.h, C.byref(st))
        self.stats = {f[0]: getattr(st, f[0]) for f in _Stats._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.lib.orc_free(self.h)
            self.h = None

    __del__ = close

    def query_kmers(self, kmers, threads=1):
        arr = kmers_to_array(kmers, self.kmer_bytes)
        nq = arr.shape[0]
        ids = np.empty(nq, dtype=np.int64)
        self.lib.orc_query_kmers(self.h, arr.ctypes.data, nq, _ptr(ids, i64p), threads)
        return ids

    def query_kmers_mini(self, kmers, minis, want_pos=False):
        arr = kmers_to_array(kmers, self.kmer_bytes)
        minis = np.ascontiguousarray(minis, dtype=np.uint32)
        nq = arr.shape[0]
        out = np.empty(nq, dtype=np.int64)
        self.lib.orc_query_kmers_mini(self.h, arr.ctypes.data, _ptr(minis, u32p), nq, _ptr(out, i64p), int(want_pos))
        return out

    def query_sequence(self, seq, return_kmers=False):
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        nk = max(0, len(s) - self.k + 1)
        ids = np.empty(nk, dtype=np.int64)
        km = np.empty((nk, self.kmer_bytes // 8), dtype=np.uint64) if return_kmers else None
        r = self.lib.orc_query_sequence(self.h, s, len(s), _ptr(ids, i64p), km.ctypes.data if return_kmers else None)
        if r < 0:
            raise OracleError(int(r))
        return (ids, km) if return_kmers else ids

    def canonical_kmers(self, seq):
        """canonical k-mers of a sequence as a (n, kmer_bytes/8) uint64 array (no index lookups)"""
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        nk = max(0, len(s) - self.k + 1)
        km = np.empty((nk, self.kmer_bytes // 8), dtype=np.uint64)
        r = self.lib.orc_query_sequence(self.h, s, len(s), None, km.ctypes.data)
        if r < 0:
            raise OracleError(int(r))
        return km if self.kmer_bytes == 16 else km.reshape(-1)

    # ---- internals, for array-level parity ----
    def arrays(self):
        st = self.stats
        nb, ng = st["nbuckets"], st["ngroups"]

        def cp(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)

        out = {
            "bucket_nt": cp(self.lib.orc_bucket_nt(self.h), nb, np.uint32),
            "bucket_start": cp(self.lib.orc_bucket_start(self.h), nb, np.uint64),
            "seq": cp(self.lib.orc_seq(self.h), (st["total_nt"] + 31) // 32, np.uint64),
            "positions": cp(self.lib.orc_positions(self.h), (st["positions_bits"] + 63) // 64, np.uint64),
            "g_nkeys": cp(self.lib.orc_group_nkeys(self.h), ng, np.uint64),
            "g_prefix": cp(self.lib.orc_group_prefix(self.h), ng, np.uint64),
            "g_posstart": cp(self.lib.orc_group_posstart(self.h), ng, np.uint64),
            "g_nbits": cp(self.lib.orc_group_nbits(self.h), ng, np.uint8),
        }
        return out

    def mphf(self, g):
        sizes = u64p()
        bits = u64p()
        nfb = C.c_uint64()
        lv = C.c_int()
        nwords = self.lib.orc_mphf_info(self.h, g, C.byref(sizes), C.byref(bits), C.byref(nfb), C.byref(lv))
        res = {"nwords": nwords, "levels_used": lv.value, "sizes": np.array([sizes[i] for i in range(16)], dtype=np.uint64),
               "bits": np.ctypeslib.as_array(bits, shape=(nwords,)).copy() if nwords else np.zeros(0, np.uint64)}
        fk, fr = [], []
        for i in range(nfb.value):
            key = (C.c_uint64 * 2)()
            rank = C.c_uint64()
            self.lib.orc_mphf_fallback(self.h, g, i, key, C.byref(rank))
            fk.append(int(key[0]) | ((int(key[1]) << 64) if self.kmer_bytes == 16 else 0))
            fr.append(rank.value)
        res["fallback_keys"], res["fallback_ranks"] = fk, fr
        return res

    def mphf_lookup(self, g, key):
        arr = kmers_to_array([key], self.kmer_bytes)
        return self.lib.orc_mphf_lookup(self.h, g, arr.ctypes.data)


class ImportedOracle(Oracle):
    """The CPU query path of the oracle over an index image exported from elsewhere (bench.py cpu_baseline leg).

    `arrays` = dict of numpy arrays named like blight_amd.EXPORT (bucket_nt, bucket_start, seq (zero-padded), positions,
    g_nkeys, g_prefix, g_posstart, g_nbits(u32), level_sizes, g_wbase, mphf_bits, fb_keys, fb_ranks, fb_group)."""

    def __init__(self, k, m, n, b, total_nt, number_kmer, positions_bits, arrays, kmer_bytes=8):
        self.lib = _orc()
        self.kmer_bytes = kmer_bytes
        self.k, self.m, self.n, self.b = k, m, n, b
        self._keep = {key: np.ascontiguousarray(v) for key, v in arrays.items()}
        a = self._keep
        h = C.c_void_p()
        rc = self.lib.orc_import(C.byref(h), k, m, n, b, kmer_bytes, total_nt, number_kmer, _ptr(a["bucket_nt"], u32p),
                                 _ptr(a["bucket_start"], u64p), _ptr(a["seq"], u64p), _ptr(a["positions"], u64p), positions_bits,
                                 _ptr(a["g_nkeys"], u64p), _ptr(a["g_prefix"], u64p), _ptr(a["g_posstart"], u64p),
                                 _ptr(a["g_nbits"], u32p), _ptr(a["level_sizes"], u64p), _ptr(a["g_wbase"], u64p),
                                 _ptr(a["mphf_bits"], u64p), _ptr(a["fb_keys"], u64p), _ptr(a["fb_ranks"], u64p),
                                 _ptr(a["fb_group"], u32p), a["fb_ranks"].size)
        if rc:
            raise OracleError(rc)
        self.h = h
        self.stats = {"number_kmer": number_kmer, "total_nt": total_nt}


# ---- primitives ----
def p_nuc2int(c):
    return _orc().orc_p_nuc2int(c.encode() if isinstance(c, str) else c)


def p_revhash32(x):
    return _orc().orc_p_revhash32(x)


def p_rc32(x, m):
    return _orc().orc_p_rc32(x, m)


def p_parity_canon(x, m):
    return _orc().orc_p_parity_canon(x, m)


def p_rc(x, k, kmer_bytes=8):
    if kmer_bytes == 8:
        return _orc().orc_p_rc64(x, k)
    a = kmers_to_array([x], 16)
    o = np.empty_like(a)
    _orc().orc_p_rc128(a.ctypes.data, k, o.ctypes.data)
    return array_to_kmers(o, 16)[0]


def p_minimizer(x, k, m, kmer_bytes=8):
    if kmer_bytes == 8:
        return _orc().orc_p_minimizer64(x, k, m)
    a = kmers_to_array([x], 16)
    return _orc().orc_p_minimizer128(a.ctypes.data, k, m)


def p_hash_bis(key, seed):
    return _orc().orc_p_hash_bis(key, seed)


def p_level_hashes(key, n):
    out = (C.c_uint64 * n)()
    _orc().orc_p_level_hashes(key, n, out)
    return [int(x) for x in out]


def p_murmur3_x64_128(data, seed=0):
    """oracle restatement of MurmurHash3_x64_128 (reference MurmurHash3.cpp:255) -> (h1, h2)"""
    b = bytes(data)
    out = (C.c_uint64 * 2)()
    _orc().orc_p_murmur3_x64_128(b, len(b), C.c_uint32(seed), out)
    return int(out[0]), int(out[1])


MURMUR_REF = os.path.join(HERE, "_ref", "libmurmur3_ref.so")


def ref_murmur3_x64_128(data, seed=0):
    """the reference's own MurmurHash3.cpp (oracle/_ref/libmurmur3_ref.so)"""
    lib = C.CDLL(MURMUR_REF)
    b = bytes(data)
    out = (C.c_uint64 * 2)()
    lib.ref_murmur3_x64_128(b, len(b), C.c_uint32(seed), out)
    return int(out[0]), int(out[1])


def p_configure_levels(n):
    out = (C.c_uint64 * 16)()
    tot = _orc().orc_p_configure_levels(n, out)
    return tot, [int(x) for x in out]


def p_str2num(s, k=None):
    s = s.encode() if isinstance(s, str) else s
    err = C.c_int()
    v = _orc().orc_p_str2num64(s, k or len(s), C.byref(err))
    if err.value:
        raise OracleError(-3)
    return v


def p_chop(seq, k, m):
    s = seq.encode() if isinstance(seq, str) else seq
    cap = max(1, len(s))
    o = np.empty(cap, np.uint32)
    l = np.empty(cap, np.uint32)
    mi = np.empty(cap, np.uint32)
    n = _orc().orc_p_chop(s, len(s), k, m, _ptr(o, u32p), _ptr(l, u32p), _ptr(mi, u32p), cap)
    if n < 0:
        raise OracleError(int(n))
    return list(zip(o[:n].tolist(), l[:n].tolist(), mi[:n].tolist()))


def check_params(k, m, n, s, kmer_bytes=8):
    msg = C.c_char_p()
    rc = _orc().orc_check_params(k, m, n, s, kmer_bytes, C.byref(msg))
    return rc, (msg.value or b"").decode()


# ---- the real reference (oracle/_ref) ----
class _RefStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("number_kmer", "number_super_kmer", "total_nt", "largest_mphf", "largest_bucket",
                                          "positions_bits", "number_query")] + [("nbuckets", C.c_uint32), ("ngroups", C.c_uint32)]


_ref_libs = {}


def ref_available():
    return os.path.exists(REF64) and os.path.exists(REF128)


def _ref(kmer_bytes):
    if kmer_bytes not in _ref_libs:
        lib = C.CDLL(REF64 if kmer_bytes == 8 else REF128)
        assert lib.ref_kmer_bytes() == kmer_bytes
        lib.ref_last_error.restype = C.c_char_p
        lib.ref_create.argtypes = [C.c_uint] * 6
        lib.ref_create.restype = C.c_void_p
        lib.ref_destroy.argtypes = [C.c_void_p]
        lib.ref_construct.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        lib.ref_query_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, i64p]
        lib.ref_query_kmers_mini.argtypes = [C.c_void_p, C.c_void_p, u32p, C.c_uint64, i64p, C.c_int]
        lib.ref_query_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, i64p]
        lib.ref_query_sequence.restype = C.c_int64
        lib.ref_query_sequence_bool.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, u32p, u32p]
        lib.ref_stats.argtypes = [C.c_void_p, C.POINTER(_RefStats)]
        lib.ref_buckets.argtypes = [C.c_void_p, u32p, u64p]
        lib.ref_groups.argtypes = [C.c_void_p, u64p, u64p, u64p, u8p]
        lib.ref_seq_words.argtypes = [C.c_void_p, u64p, C.c_uint64]
        lib.ref_positions_words.argtypes = [C.c_void_p, u64p, C.c_uint64]
        lib.ref_mphf_info.argtypes = [C.c_void_p, C.c_uint32, u64p, u64p, u64p]
        lib.ref_mphf_info.restype = C.c_uint64
        lib.ref_mphf_bits.argtypes = [C.c_void_p, C.c_uint32, u64p, u64p]
        lib.ref_mphf_fallback.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, u64p]
        lib.ref_mphf_lookup.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.ref_mphf_lookup.restype = C.c_uint64
        lib.ref_p_nuc2int.argtypes = [C.c_char]
        lib.ref_p_revhash32.argtypes = [C.c_uint32]
        lib.ref_p_revhash32.restype = C.c_int32
        lib.ref_p_rc32.argtypes = [C.c_uint32, C.c_int]
        lib.ref_p_rc32.restype = C.c_uint32
        lib.ref_p_parity_canon.argtypes = [C.c_uint32, C.c_int]
        lib.ref_p_parity_canon.restype = C.c_uint32
        lib.ref_p_rc.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.ref_p_minimizer.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.ref_p_minimizer.restype = C.c_uint32
        lib.ref_p_hash.argtypes = [C.c_void_p, C.c_uint64]
        lib.ref_p_hash.restype = C.c_uint64
        lib.ref_p_level_hashes.argtypes = [C.c_void_p, C.c_uint, u64p]
        lib.ref_p_str2num.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        lib.ref_p_chop.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, u32p, u32p, u32p, C.c_uint64]
        lib.ref_p_chop.restype = C.c_int64
        _ref_libs[kmer_bytes] = lib
    return _ref_libs[kmer_bytes]


class Reference:
    """The real (patched) reference kmer_Set_Light, always driven with -t 1."""

    def __init__(self, k, m, n, b, seqs=None, fasta=None, s=None, kmer_bytes=None):
        self.kmer_bytes = kmer_bytes or (8 if k <= 32 else 16)
        self.lib = _ref(self.kmer_bytes)
        self.k, self.m, self.n, self.b = k, m, n, b
        s = min(n, 3) if s is None else s
        self.h = self.lib.ref_create(k, m, n, s, 1, b)
        if not self.h:
            raise ValueError(self.lib.ref_last_error().decode())
        with tempfile.TemporaryDirectory() as td:
            if fasta is None:
                fasta = os.path.join(td, "in.fa")
                with open(fasta, "w") as f:
                    for i, sq in enumerate(seqs):
                        f.write(">%d\n%s\n" % (i, sq))
            # the reference prints its recap to stdout; silence it at the fd level
            devnull = os.open(os.devnull, os.O_WRONLY)
            saved = os.dup(1)
            os.dup2(devnull, 1)
            try:
                rc = self.lib.ref_construct(self.h, os.path.abspath(fasta).encode(), td.encode())
            finally:
                os.dup2(saved, 1)
                os.close(saved)
                os.close(devnull)
            if rc:
                raise RuntimeError(self.lib.ref_last_error().decode())
        st = _RefStats()
        self.lib.ref_stats(self.h, C.byref(st))
        self.stats = {f[0]: getattr(st, f[0]) for f in _RefStats._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.lib.ref_destroy(self.h)
            self.h = None

    __del__ = close

    def query_kmers(self, kmers):
        arr = kmers_to_array(kmers, self.kmer_bytes)
        nq = arr.shape[0]
        ids = np.empty(nq, dtype=np.int64)
        self.lib.ref_query_kmers(self.h, arr.ctypes.data, nq, _ptr(ids, i64p))
        return ids

    def query_kmers_mini(self, kmers, minis, want_pos=False):
        arr = kmers_to_array(kmers, self.kmer_bytes)
        minis = np.ascontiguousarray(minis, dtype=np.uint32)
        nq = arr.shape[0]
        out = np.empty(nq, dtype=np.int64)
        self.lib.ref_query_kmers_mini(self.h, arr.ctypes.data, _ptr(minis, u32p), nq, _ptr(out, i64p), int(want_pos))
        return out

    def query_sequence(self, seq):
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        nk = max(0, len(s) - self.k + 1)
        ids = np.empty(max(nk, 1), dtype=np.int64)
        r = self.lib.ref_query_sequence(self.h, s, len(s), _ptr(ids, i64p))
        if r < 0:
            raise ValueError(self.lib.ref_last_error().decode())
        return ids[:r].copy()

    def query_sequence_bool(self, seq):
        s = seq.encode() if isinstance(seq, str) else bytes(seq)
        a, b = C.c_uint32(), C.c_uint32()
        if self.lib.ref_query_sequence_bool(self.h, s, len(s), C.byref(a), C.byref(b)):
            raise ValueError(self.lib.ref_last_error().decode())
        return a.value, b.value

    def arrays(self):
        st = self.stats
        nb, ng = st["nbuckets"], st["ngroups"]
        out = {"bucket_nt": np.zeros(nb, np.uint32), "bucket_start": np.zeros(nb, np.uint64),
               "g_nkeys": np.zeros(ng, np.uint64), "g_prefix": np.zeros(ng, np.uint64), "g_posstart": np.zeros(ng, np.uint64),
               "g_nbits": np.zeros(ng, np.uint8),
               "seq": np.zeros((st["total_nt"] + 31) // 32, np.uint64),
               "positions": np.zeros((st["positions_bits"] + 63) // 64, np.uint64)}
        self.lib.ref_buckets(self.h, _ptr(out["bucket_nt"], u32p), _ptr(out["bucket_start"], u64p))
        self.lib.ref_groups(self.h, _ptr(out["g_nkeys"], u64p), _ptr(out["g_prefix"], u64p), _ptr(out["g_posstart"], u64p),
                            _ptr(out["g_nbits"], u8p))
        if out["seq"].size:
            self.lib.ref_seq_words(self.h, _ptr(out["seq"], u64p), out["seq"].size)
        if out["positions"].size:
            self.lib.ref_positions_words(self.h, _ptr(out["positions"], u64p), out["positions"].size)
        return out

    def mphf(self, g):
        sizes = np.zeros(16, np.uint64)
        nfb, nr = C.c_uint64(), C.c_uint64()
        nwords = self.lib.ref_mphf_info(self.h, g, _ptr(sizes, u64p), C.byref(nfb), C.byref(nr))
        bits = np.zeros(nwords, np.uint64)
        ranks = np.zeros(nr.value, np.uint64)
        if nwords:
            self.lib.ref_mphf_bits(self.h, g, _ptr(bits, u64p), _ptr(ranks, u64p) if nr.value else None)
        fk = np.zeros((nfb.value, self.kmer_bytes // 8), np.uint64)
        fr = np.zeros(nfb.value, np.uint64)
        if nfb.value:
            self.lib.ref_mphf_fallback(self.h, g, fk.ctypes.data, _ptr(fr, u64p))
        return {"nwords": nwords, "sizes": sizes, "bits": bits, "ranks": ranks,
                "fallback_keys": array_to_kmers(fk, self.kmer_bytes), "fallback_ranks": [int(x) for x in fr]}

    def mphf_lookup(self, g, key):
        arr = kmers_to_array([key], self.kmer_bytes)
        return self.lib.ref_mphf_lookup(self.h, g, arr.ctypes.data)

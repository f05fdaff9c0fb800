"""blight_amd -- MI355X-native Blight k-mer index (hot path only): Python mirror of the reference's `kmer_Set_Light`.

The product is the C-ABI library blight_amd/libblight_mi355x.so (hand-written HIP for gfx950, include/blight_mi355x.h);
this module is a thin ctypes binding with the reference's method names, argument meaning and error behaviour
(reference blight.h:15-136).  There is no CPU fallback: every call that needs the GPU raises when the library or a
device is missing.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libblight_mi355x.so")

ABI_VERSION = 11  # must equal BLMX_ABI_VERSION of include/blight_mi355x.h (struct layouts mirrored below)
BLMX_OK, BLMX_EINVAL, BLMX_ESHORT, BLMX_EDNA, BLMX_EIO, BLMX_EHIP, BLMX_ENOMEM, BLMX_ESTATE, BLMX_ELIMIT, BLMX_ERCCL = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9
COMM_ID_BYTES = 128

_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_u32p = C.POINTER(C.c_uint32)


class blmx_params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("m", C.c_uint32), ("log2_mphf", C.c_uint32), ("log2_superbuckets", C.c_uint32),
                ("threads", C.c_uint32), ("bits_saved", C.c_uint32), ("kmer_bytes", C.c_uint32), ("device", C.c_int32)]


class blmx_stats_t(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("number_kmer", "number_super_kmer", "largest_mphf", "positions_total_size",
                                          "number_query", "total_nt", "largest_bucket", "mphf_bytes", "fallback_keys",
                                          "device_bytes")] + \
               [(n, C.c_uint32) for n in ("nbuckets", "ngroups", "k", "m", "log2_mphf", "bits_saved", "kmer_bytes",
                                          "max_levels_used")] + [("build_seconds", C.c_double), ("local_number_kmer", C.c_uint64)] + \
               [(n, C.c_uint32) for n in ("shard_rank", "shard_world", "shard_group_lo", "shard_group_hi", "mphf_groups_lds_built",
                                          "reserved0")] + [("build_scratch_peak_bytes", C.c_uint64)]


_lib = None


def load_library():
    """Load libblight_mi355x.so (fails loudly when it has not been built: `python -m blight_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("BLIGHT_LIB_PATH", LIB_PATH)  # (experiments: a variant build of the same sources, blight_amd.build --variant)
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -m blight_amd.build` (hipcc, gfx950)" % path)
    # One HIP runtime per process: torch bundles its own libamdhip64 (same soname); whichever is loaded first is shared.
    # Import torch first when it is available so that torch tensors / streams and this library use the same runtime.
    if "torch" not in sys.modules and os.environ.get("BLIGHT_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional plumbing
            pass
    lib = C.CDLL(path)
    lib.blmx_version.restype = C.c_char_p
    lib.blmx_create_error.restype = C.c_char_p
    lib.blmx_last_error.restype = C.c_char_p
    lib.blmx_last_error.argtypes = [C.c_void_p]
    lib.blmx_create.argtypes = [C.POINTER(blmx_params), C.POINTER(C.c_void_p)]
    lib.blmx_destroy.argtypes = [C.c_void_p]
    lib.blmx_build_from_fasta.argtypes = [C.c_void_p, C.c_char_p]
    lib.blmx_build_from_sequences.argtypes = [C.c_void_p, C.c_char_p, _u64p, C.c_uint64]
    lib.blmx_build_from_device_sequences.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    lib.blmx_query_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, _i64p]
    lib.blmx_query_kmers_with_minimizer.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.c_uint64, _i64p]
    lib.blmx_query_positions.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.c_uint64, C.POINTER(C.c_int32)]
    lib.blmx_query_membership.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint8)]
    lib.blmx_query_sequences.argtypes = [C.c_void_p, C.c_char_p, _u64p, C.c_uint64, _i64p]
    lib.blmx_query_file.argtypes = [C.c_void_p, C.c_char_p, _u64p, _u64p]
    lib.blmx_save.argtypes = [C.c_void_p, C.c_char_p]
    lib.blmx_payload_create.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.blmx_payload_destroy.argtypes = [C.c_void_p]
    lib.blmx_payload_destroy.restype = None
    lib.blmx_payload_add_file.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, _u64p, _u64p]
    lib.blmx_payload_add_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.blmx_payload_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.blmx_payload_export.restype = C.c_int64
    lib.blmx_partition_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.blmx_query_partitioned_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.blmx_gather_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.blmx_load.argtypes = [C.c_void_p, C.c_char_p]
    lib.blmx_peek_params.argtypes = [C.c_char_p, C.POINTER(blmx_params)]
    lib.blmx_sequence_kmer_count.argtypes = [C.c_void_p, _u64p, C.c_uint64]
    lib.blmx_sequence_kmer_count.restype = C.c_uint64
    lib.blmx_query_kmers_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.blmx_query_sequences_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.blmx_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.blmx_last_profile.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    lib.blmx_kmer_groups_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.blmx_kmers_at_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.blmx_canonicalize_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.blmx_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    lib.blmx_shard_bounds.argtypes = [C.c_void_p, _u32p]
    lib.blmx_route_count_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.blmx_route_scatter_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
    lib.blmx_unpermute_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.blmx_set_query_binning.argtypes = [C.c_void_p, C.c_int]
    lib.blmx_comm_unique_id.argtypes = [C.c_void_p]
    lib.blmx_comm_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    lib.blmx_comm_destroy.argtypes = [C.c_void_p]
    lib.blmx_comm_destroy.restype = None
    lib.blmx_query_kmers_sharded_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.blmx_build_sharded_from_sequences.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, _u64p, C.c_uint64]
    lib.blmx_build_sharded_from_fasta.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    lib.blmx_query_kmers_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, _i64p]
    lib.blmx_query_sequences_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, _u64p, C.c_uint64, _i64p]
    lib.blmx_query_sequences_sharded_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                                        C.c_void_p, C.c_void_p]
    lib.blmx_build_sharded_from_device_sequences.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    lib.blmx_shard_build_begin.argtypes = [C.c_void_p, C.c_char_p, _u64p, C.c_uint64, _u64p]
    lib.blmx_shard_build_begin_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, _u64p]
    lib.blmx_shard_build_plan.argtypes = [C.c_void_p, _u64p, C.c_uint32, _u64p, _u64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.blmx_shard_halo_words.argtypes = [C.c_void_p]
    lib.blmx_shard_halo_words.restype = C.c_uint64
    lib.blmx_shard_halo_get.argtypes = [C.c_void_p, _u64p]
    lib.blmx_shard_halo_set.argtypes = [C.c_void_p, _u64p, C.c_uint32]
    lib.blmx_comm_loopback_id.argtypes = [C.c_void_p]
    lib.blmx_comm_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.blmx_comm_transport.argtypes = [C.c_void_p]
    lib.blmx_comm_transport.restype = C.c_char_p
    lib.blmx_shard_build_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _u64p, _u64p]
    lib.blmx_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.blmx_stats.argtypes = [C.c_void_p, C.POINTER(blmx_stats_t)]
    lib.blmx_partition_stats.argtypes = [C.c_void_p, _u64p, C.c_int]
    lib.blmx_serve_stats.argtypes = [C.c_void_p, _u64p]
    lib.blmx_release_scratch.argtypes = [_u64p, _u64p]
    lib.blmx_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
    lib.blmx_export.restype = C.c_int64
    lib.blmx_primitive.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    if lib.blmx_abi_version() != ABI_VERSION:
        raise ImportError("libblight_mi355x.so ABI version %d != binding %d: rebuild (python -m blight_amd.build)" % (
            lib.blmx_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class BlightError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (blmx status %d)" % (msg, code))
        self.code = code


def _raise(code, msg):
    # the reference's exception types: std::invalid_argument -> ValueError, std::domain_error -> ValueError,
    # std::runtime_error (file) -> OSError
    if code == BLMX_EINVAL:
        raise ValueError(msg)
    if code == BLMX_EDNA:
        raise ValueError(msg or "Invalid char in DNA")
    if code == BLMX_EIO:
        raise OSError(msg or "Problem with files opening")
    raise BlightError(code, msg)


def _kmers_array(kmers, kmer_bytes):
    """ints / uint64 array -> contiguous little-endian array of kmer_bytes per k-mer"""
    if isinstance(kmers, np.ndarray) and kmers.dtype == np.uint64:
        a = np.ascontiguousarray(kmers)
        if a.size % (kmer_bytes // 8):
            raise ValueError("a %d-byte k-mer array needs a multiple of %d uint64 words" % (kmer_bytes, kmer_bytes // 8))
        return a, a.size // (kmer_bytes // 8)
    kl = [int(x) for x in (kmers if hasattr(kmers, "__iter__") else [kmers])]
    if kmer_bytes == 8:
        return np.array(kl, dtype=np.uint64), len(kl)
    a = np.empty((len(kl), 2), dtype=np.uint64)
    for i, x in enumerate(kl):
        a[i, 0] = x & 0xFFFFFFFFFFFFFFFF
        a[i, 1] = x >> 64
    return a, len(kl)


def _concat(seqs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return b"".join(bs), offs


EXPORT = {"bucket_nt": (0, np.uint32), "bucket_start": (1, np.uint64), "seq": (2, np.uint64), "positions": (3, np.uint64),
          "g_nkeys": (4, np.uint64), "g_prefix": (5, np.uint64), "g_posstart": (6, np.uint64), "g_nbits": (7, np.uint32),
          "level_sizes": (8, np.uint64), "g_wbase": (9, np.uint64), "mphf_bits": (10, np.uint64), "fb_keys": (11, np.uint64),
          "fb_ranks": (12, np.uint64), "fb_group": (13, np.uint32)}


class kmer_Set_Light:
    """Drop-in mirror of the reference class (blight.h:15-136).  Scalars keep the reference's per-k-mer signatures; every
    query method also accepts a batch (sequence / uint64 array) and then returns a numpy array."""

    def __init__(self, k, minimizer_length, log2_mphfs_number, log2_superbuckets_number, cores_number, bits_to_save,
                 device=0, kmer_bytes=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        p = blmx_params(k, minimizer_length, log2_mphfs_number, log2_superbuckets_number, cores_number, bits_to_save,
                        kmer_bytes, device)
        rc = self._lib.blmx_create(C.byref(p), C.byref(self._h))
        if rc:
            self._h = None
            _raise(rc, self._lib.blmx_create_error().decode())
        self._k, self._minimizer_length = k, minimizer_length
        self.log2_mphf = log2_mphfs_number
        self.kmer_bytes = kmer_bytes or (8 if k <= 32 else 16)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.blmx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            _raise(rc, self._lib.blmx_last_error(self._h).decode())

    # -- construction (blight.cpp:108-125)
    def construct_index(self, input_file):
        self._check(self._lib.blmx_build_from_fasta(self._h, os.fsencode(input_file)))

    def construct_index_from_sequences(self, seqs):
        buf, offs = _concat(seqs)
        self._check(self._lib.blmx_build_from_sequences(self._h, buf, offs.ctypes.data_as(_u64p), len(offs) - 1))

    def construct_index_from_device(self, d_concat_ptr, d_offsets_ptr, nseq, total_len):
        self._check(self._lib.blmx_build_from_device_sequences(self._h, d_concat_ptr, d_offsets_ptr, nseq, total_len))

    # -- index image on disk (mainline Blight's dump_disk / load constructor, absent from this fork)
    def dump_disk(self, path):
        self._check(self._lib.blmx_save(self._h, os.fsencode(path)))

    def load_disk(self, path):
        """Fill this (unbuilt) object from an image written with the same k, m, log2_mphfs_number, bits_to_save."""
        self._check(self._lib.blmx_load(self._h, os.fsencode(path)))

    @classmethod
    def from_file(cls, path, device=0):
        lib = load_library()
        p = blmx_params()
        rc = lib.blmx_peek_params(os.fsencode(path), C.byref(p))
        if rc:
            _raise(rc, "not a Blight MI355X index file: %s" % path)
        self = cls(p.k, p.m, p.log2_mphf, p.log2_superbuckets, 1, p.bits_saved, device=device, kmer_bytes=p.kmer_bytes)
        self.load_disk(path)
        return self

    # -- counters (blight.h:52-59)
    def stats(self):
        st = blmx_stats_t()
        self._check(self._lib.blmx_stats(self._h, C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in blmx_stats_t._fields_}

    def serve_stats(self):
        """the bounded serving burst of small calls: {kernels, calls, relaunches_after_deadline, alive}"""
        a = np.zeros(4, dtype=np.uint64)
        self._check(self._lib.blmx_serve_stats(self._h, a.ctypes.data_as(_u64p)))
        return {"kernels": int(a[0]), "calls": int(a[1]), "relaunches_after_deadline": int(a[2]), "alive": bool(a[3])}

    def partition_stats(self, reset=False):
        """one-pass partition: {batches, kmers, overflow_kmers, max_overflow_of_a_batch} since the last reset"""
        a = np.zeros(4, dtype=np.uint64)
        self._check(self._lib.blmx_partition_stats(self._h, a.ctypes.data_as(_u64p), int(reset)))
        return {"batches": int(a[0]), "kmers": int(a[1]), "overflow_kmers": int(a[2]), "max_overflow_of_a_batch": int(a[3])}

    number_kmer = property(lambda self: self.stats()["number_kmer"])
    number_super_kmer = property(lambda self: self.stats()["number_super_kmer"])
    largest_MPHF = property(lambda self: self.stats()["largest_mphf"])
    number_query = property(lambda self: self.stats()["number_query"])
    positions_total_size = property(lambda self: self.stats()["positions_total_size"])

    # -- queries
    @staticmethod
    def _out_array(out, nq, dtype):
        """the caller's result buffer (like the C ABI: a caller that reuses it does not pay for fresh pages on every call) or a new one"""
        if out is None:
            return np.empty(nq, dtype=dtype)
        if not isinstance(out, np.ndarray) or out.dtype != dtype or out.size != nq or not out.flags.c_contiguous:
            raise ValueError("out must be a contiguous %s array of %d elements" % (np.dtype(dtype).name, nq))
        return out

    def query_kmer_hash(self, canon, out=None):
        """blight.cpp:545-550; scalar int -> int, batch -> int64 array (written into `out` when given)"""
        scalar = not hasattr(canon, "__iter__")
        a, nq = _kmers_array(canon, self.kmer_bytes)
        out = self._out_array(out, nq, np.int64)
        self._check(self._lib.blmx_query_kmers(self._h, a.ctypes.data, nq, out.ctypes.data_as(_i64p)))
        return int(out[0]) if scalar else out

    def query_kmer_bool(self, canon, out=None):
        """blight.cpp:537-541; batch -> bool array (one byte per k-mer; written into the uint8 array `out` when given)"""
        scalar = not hasattr(canon, "__iter__")
        a, nq = _kmers_array(canon, self.kmer_bytes)
        out = self._out_array(out, nq, np.uint8)
        self._check(self._lib.blmx_query_membership(self._h, a.ctypes.data, nq, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return bool(out[0]) if scalar else out.view(np.bool_)

    def query_get_hash(self, canon, minimizer):
        """blight.cpp:716-742"""
        scalar = not hasattr(canon, "__iter__")
        a, nq = _kmers_array(canon, self.kmer_bytes)
        mi = np.ascontiguousarray(np.atleast_1d(minimizer), dtype=np.uint32)
        if mi.size != nq:
            raise ValueError("%d minimizers for %d k-mers" % (mi.size, nq))
        out = np.empty(nq, dtype=np.int64)
        self._check(self._lib.blmx_query_kmers_with_minimizer(self._h, a.ctypes.data, mi.ctypes.data_as(_u32p), nq,
                                                              out.ctypes.data_as(_i64p)))
        return int(out[0]) if scalar else out

    def query_get_pos_unitig(self, canon, minimizer):
        """blight.cpp:686-712"""
        scalar = not hasattr(canon, "__iter__")
        a, nq = _kmers_array(canon, self.kmer_bytes)
        mi = np.ascontiguousarray(np.atleast_1d(minimizer), dtype=np.uint32)
        if mi.size != nq:
            raise ValueError("%d minimizers for %d k-mers" % (mi.size, nq))
        out = np.empty(nq, dtype=np.int32)
        self._check(self._lib.blmx_query_positions(self._h, a.ctypes.data, mi.ctypes.data_as(_u32p), nq,
                                                   out.ctypes.data_as(C.POINTER(C.c_int32))))
        return int(out[0]) if scalar else out

    def single_query(self, minimizer, kastor):
        """blight.cpp:658-660"""
        r = self.query_get_pos_unitig(kastor, minimizer)
        return (r >= 0) if not hasattr(r, "__iter__") else (r >= 0)

    def query_sequences_hash(self, seqs):
        """query_sequence_hash over many sequences in one launch; returns one flat int64 array"""
        buf, offs = _concat(seqs)
        nk = self._lib.blmx_sequence_kmer_count(self._h, offs.ctypes.data_as(_u64p), len(offs) - 1)
        out = np.empty(nk, dtype=np.int64)
        self._check(self._lib.blmx_query_sequences(self._h, buf, offs.ctypes.data_as(_u64p), len(offs) - 1,
                                                   out.ctypes.data_as(_i64p)))
        return out

    def query_sequence_hash(self, query):
        """blight.cpp:575-591"""
        return self.query_sequences_hash([query])

    def query_sequence_bool(self, query):
        """blight.cpp:554-571 -> (found, not found)"""
        ids = self.query_sequences_hash([query])
        found = int((ids >= 0).sum())
        return found, int(ids.size - found)

    def file_query(self, query_file):
        """blight.cpp:746-799: streams the 2-line FASTA records through the GPU (blmx_query_file), skips sequences shorter
        than k, prints the reference's recap and returns (good, erroneous)."""
        import time
        t0 = time.time()
        tp, fp = C.c_uint64(), C.c_uint64()
        self._check(self._lib.blmx_query_file(self._h, os.fsencode(query_file), C.byref(tp), C.byref(fp)))
        print("-----------------------QUERY RECAP 2----------------------------")
        print("Good kmer: %d" % tp.value)
        print("Erroneous kmers: %d" % fp.value)
        print("Query performed: %d" % self.number_query)
        print("The whole QUERY took me %g seconds." % (time.time() - t0))
        return tp.value, fp.value

    # -- device-resident batch API (bench / pipelines): raw device pointers + hipStream_t
    def query_kmers_device(self, d_kmers_ptr, nq, d_ids_ptr, stream=None):
        self._check(self._lib.blmx_query_kmers_device(self._h, d_kmers_ptr, nq, d_ids_ptr, stream))

    def query_sequences_device(self, d_concat_ptr, d_offsets_ptr, nseq, total_len, nk, d_ids_ptr, d_err_ptr=None, stream=None):
        self._check(self._lib.blmx_query_sequences_device(self._h, d_concat_ptr, d_offsets_ptr, nseq, total_len, nk, d_ids_ptr, d_err_ptr, stream))

    PROFILE_STAGES = ("front", "k_bin_hist", "scan", "k_bin_scatter", "k_query", "k_gather_ids")

    def set_profiling(self, on):
        self._check(self._lib.blmx_set_profiling(self._h, int(on)))

    def last_profile(self):
        """stage -> milliseconds of the last device-resident query call (HIP events on its stream)"""
        ms = (C.c_float * 8)()
        n = self._lib.blmx_last_profile(self._h, ms, 8)
        if n < 0:
            self._check(n)
        return {self.PROFILE_STAGES[i]: float(ms[i]) for i in range(n)}

    def partition_device(self, d_kmers_ptr, nq, d_ordered_ptr, d_inv_ptr, d_rank_counts_ptr=None, stream=None):
        self._check(self._lib.blmx_partition_device(self._h, d_kmers_ptr, nq, d_ordered_ptr, d_inv_ptr, d_rank_counts_ptr, stream))

    def query_partitioned_device(self, d_ordered_ptr, nq, d_ids_ptr, stream=None):
        self._check(self._lib.blmx_query_partitioned_device(self._h, d_ordered_ptr, nq, d_ids_ptr, stream))

    def gather_ids_device(self, d_ids_ordered_ptr, d_inv_ptr, nq, d_ids_out_ptr, stream=None):
        self._check(self._lib.blmx_gather_ids_device(self._h, d_ids_ordered_ptr, d_inv_ptr, nq, d_ids_out_ptr, stream))

    def kmer_groups_device(self, d_kmers_ptr, nq, d_groups_ptr, stream=None):
        self._check(self._lib.blmx_kmer_groups_device(self._h, d_kmers_ptr, nq, d_groups_ptr, stream))

    def kmers_at_device(self, d_ascii_ptr, d_positions_ptr, nq, d_kmers_out_ptr, d_err_ptr=None, stream=None):
        self._check(self._lib.blmx_kmers_at_device(self._h, d_ascii_ptr, d_positions_ptr, nq, d_kmers_out_ptr, d_err_ptr, stream))

    def canonicalize_device(self, d_kmers_ptr, nq, stream=None):
        self._check(self._lib.blmx_canonicalize_device(self._h, d_kmers_ptr, nq, stream))

    def set_query_binning(self, policy):
        """0 automatic, 1 always partition large batches by minimizer prefix before querying, -1 never"""
        self._check(self._lib.blmx_set_query_binning(self._h, policy))

    def set_tuning(self, key, value):
        """per-index tuning knob (see blmx_set_tuning in include/blight_mi355x.h); results never depend on it"""
        self._check(self._lib.blmx_set_tuning(self._h, key.encode(), int(value)))

    # -- multi-GPU sharding
    def set_shard(self, rank, world):
        self._check(self._lib.blmx_set_shard(self._h, rank, world))

    # -- sharded construction from this rank's own slice of the input, phase by phase (bring your own transport)
    def shard_build_begin(self, seqs):
        """chop the slice; returns (keys per MPHF group, super-k-mers) of it -- sum both over all ranks for shard_build_plan"""
        buf, offs = _concat(seqs)
        hist = np.zeros((1 << self.log2_mphf) + 1, dtype=np.uint64)
        self._check(self._lib.blmx_shard_build_begin(self._h, buf, offs.ctypes.data_as(_u64p), len(offs) - 1, hist.ctypes.data_as(_u64p)))
        return hist

    def shard_build_plan(self, global_hist, world):
        """-> (send_sk[world], send_words[world], device pointer of the meta words, device pointer of the packed nucleotides)"""
        gh = np.ascontiguousarray(global_hist, dtype=np.uint64)
        if gh.size != (1 << self.log2_mphf) + 1:
            raise ValueError("global_hist must hold 2^log2_mphf + 1 words")
        sk, words = np.zeros(world, np.uint64), np.zeros(world, np.uint64)
        pm, pw = C.c_void_p(), C.c_void_p()
        self._check(self._lib.blmx_shard_build_plan(self._h, gh.ctypes.data_as(_u64p), world, sk.ctypes.data_as(_u64p),
                                                    words.ctypes.data_as(_u64p), C.byref(pm), C.byref(pw)))
        return sk, words, pm.value or 0, pw.value or 0

    def shard_halo_get(self):
        """after shard_build_finish: the head of this shard's packed sequence (one record: valid nucleotides, then 2-bit words)"""
        rec = np.zeros(int(self._lib.blmx_shard_halo_words(self._h)), dtype=np.uint64)
        self._check(self._lib.blmx_shard_halo_get(self._h, rec.ctypes.data_as(_u64p)))
        return rec

    def shard_halo_set(self, records):
        """records of ALL ranks in rank order (world x shard_halo_words): what follows this shard in the single index goes behind it"""
        r = np.ascontiguousarray(records, dtype=np.uint64)
        world = r.size // int(self._lib.blmx_shard_halo_words(self._h))
        self._check(self._lib.blmx_shard_halo_set(self._h, r.ctypes.data_as(_u64p), world))

    def shard_build_finish(self, d_meta_ptr, d_words_ptr, recv_sk, recv_words):
        rs, rw = np.ascontiguousarray(recv_sk, dtype=np.uint64), np.ascontiguousarray(recv_words, dtype=np.uint64)
        self._check(self._lib.blmx_shard_build_finish(self._h, d_meta_ptr, d_words_ptr, rs.ctypes.data_as(_u64p), rw.ctypes.data_as(_u64p)))

    def shard_bounds(self):
        world = self.stats()["shard_world"]
        b = np.zeros(world + 1, dtype=np.uint32)
        self._check(self._lib.blmx_shard_bounds(self._h, b.ctypes.data_as(_u32p)))
        return b

    def route_count_device(self, d_kmers, nq, d_bounds, world, d_dest, d_counts, stream=None):
        self._check(self._lib.blmx_route_count_device(self._h, d_kmers, nq, d_bounds, world, d_dest, d_counts, stream))

    def route_scatter_device(self, d_kmers, d_dest, nq, d_offsets, world, d_sorted, d_order, d_cursor, stream=None):
        self._check(self._lib.blmx_route_scatter_device(self._h, d_kmers, d_dest, nq, d_offsets, world, d_sorted, d_order, d_cursor, stream))

    def unpermute_device(self, d_vals, d_order, nq, d_out, stream=None):
        self._check(self._lib.blmx_unpermute_device(self._h, d_vals, d_order, nq, d_out, stream))

    # -- array export (tests)
    def export(self, name):
        which, dt = EXPORT[name]
        n = self._lib.blmx_export(self._h, which, None, 0)
        if n < 0:
            _raise(int(n), self._lib.blmx_last_error(self._h).decode())
        width = (self.kmer_bytes // 8) if name == "fb_keys" else 1
        out = np.zeros(int(n) * width, dtype=dt)
        if n:
            r = self._lib.blmx_export(self._h, which, out.ctypes.data, int(n))
            if r < 0:
                _raise(int(r), self._lib.blmx_last_error(self._h).decode())
        return out


def release_scratch():
    """blmx_release_scratch: build scratch and pinned buffers the library keeps for the next build -> back to the driver;
    returns (device bytes, pinned bytes) released"""
    d, p = C.c_uint64(0), C.c_uint64(0)
    load_library().blmx_release_scratch(C.byref(d), C.byref(p))
    return int(d.value), int(p.value)


def comm_unique_id():
    """rank 0: the 128-byte id every rank passes to Comm() (ncclGetUniqueId)"""
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.blmx_comm_unique_id(buf)
    if rc:
        _raise(rc, lib.blmx_last_error(None).decode() or "blmx_comm_unique_id failed")
    return buf.raw


def comm_loopback_id():
    """an id for `world` communicators inside ONE process (one thread per rank, one GPU): the loopback transport"""
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.blmx_comm_loopback_id(buf)
    if rc:
        _raise(rc, lib.blmx_last_error(None).decode() or "blmx_comm_loopback_id failed")
    return buf.raw


class Comm:
    """Native communicator of the library: RCCL (one process per GPU; collective construction like ncclCommInitRank) or, with an
    id from comm_loopback_id(), the in-process loopback transport."""

    def __init__(self, unique_id, rank, world, device=0):
        self._lib = load_library()
        self._c = C.c_void_p()
        self.rank, self.world = rank, world
        rc = self._lib.blmx_comm_create(unique_id, rank, world, device, C.byref(self._c))
        if rc:
            self._c = None
            _raise(rc, self._lib.blmx_last_error(None).decode() or "blmx_comm_create failed")

    @property
    def transport(self):
        return self._lib.blmx_comm_transport(self._c).decode()

    def set_tuning(self, key, value):
        """"flow_words" / "pieces" / "ship_minimizers" (see blmx_comm_set_tuning); the same on every rank"""
        rc = self._lib.blmx_comm_set_tuning(self._c, key.encode(), int(value))
        if rc:
            _raise(rc, self._lib.blmx_last_error(None).decode() or "blmx_comm_set_tuning failed")

    def query_kmers_sharded_device(self, index, d_kmers_ptr, nq, d_ids_ptr, stream=None):
        """collective: every rank calls it with its own batch; IDs in the caller's order"""
        index._check(self._lib.blmx_query_kmers_sharded_device(index._h, self._c, d_kmers_ptr, nq, d_ids_ptr, stream))

    def build_sharded_from_sequences(self, index, seqs):
        """collective: every rank passes ITS OWN consecutive slice of the input (rank order = input order)"""
        buf, offs = _concat(seqs)
        index._check(self._lib.blmx_build_sharded_from_sequences(index._h, self._c, buf, offs.ctypes.data_as(_u64p), len(offs) - 1))

    def build_sharded_from_fasta(self, index, path):
        """collective: every rank names the same file and indexes the rank-th consecutive part of its sequences"""
        index._check(self._lib.blmx_build_sharded_from_fasta(index._h, self._c, os.fsencode(path)))

    def query_kmers_sharded(self, index, kmers):
        """collective, host buffers: canonical k-mers (uint64, two words per k-mer for k > 32) -> IDs"""
        a, nq = _kmers_array(np.ascontiguousarray(kmers, dtype=np.uint64), index.kmer_bytes)
        ids = np.empty(nq, dtype=np.int64)
        index._check(self._lib.blmx_query_kmers_sharded(index._h, self._c, a.ctypes.data, nq, ids.ctypes.data_as(_i64p)))
        return ids

    def query_sequences_sharded(self, index, seqs):
        """collective: query_sequence_hash of this rank's sequences against the sharded index"""
        buf, offs = _concat(seqs)
        nk = int(self._lib.blmx_sequence_kmer_count(index._h, offs.ctypes.data_as(_u64p), len(offs) - 1))
        ids = np.empty(nk, dtype=np.int64)
        index._check(self._lib.blmx_query_sequences_sharded(index._h, self._c, buf, offs.ctypes.data_as(_u64p), len(offs) - 1, ids.ctypes.data_as(_i64p)))
        return ids

    def build_sharded_from_device(self, index, d_concat_ptr, d_offsets_ptr, nseq, total_len):
        index._check(self._lib.blmx_build_sharded_from_device_sequences(index._h, self._c, d_concat_ptr, d_offsets_ptr, nseq, total_len))

    def close(self):
        if getattr(self, "_c", None):
            self._lib.blmx_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def primitive(op, data, arg=0, arg2=0, kmer_bytes=8):
    """device KAT entry point (tests): see blmx_primitive in include/blight_mi355x.h"""
    lib = load_library()
    a = np.ascontiguousarray(data)
    in_elem = 4 if (op <= 2 or op == 8) else (kmer_bytes if op in (3, 4, 7) else (8 if op == 5 else 1))
    n = a.nbytes // in_elem
    if op in (0,):
        out = np.empty(n, np.int32)
    elif op in (1, 2, 4):
        out = np.empty(n, np.uint32)
    elif op == 3:
        out = np.empty(n * (kmer_bytes // 8), np.uint64)
    elif op == 5:
        out = np.empty((n, 16), np.uint64)
    elif op == 7:
        out = np.empty((n, 2), np.uint64)
    elif op == 8:
        out = np.empty((2, n), np.uint32)  # sorted keys, then the order
    else:
        out = np.empty(n, np.uint8)
    rc = lib.blmx_primitive(op, kmer_bytes, a.ctypes.data, n, arg, arg2, out.ctypes.data)
    if rc:
        _raise(rc, "blmx_primitive failed")
    return out


class Payload:
    """Per-k-mer payload in HBM keyed by the IDs of an index (reference Colored_/Abundance_De_Bruijn_graph_snippet.cpp):
    kind "abundance" = uint8 counters (wrap at 256 like the reference), kind "colors" = ncolors bits per k-mer."""

    def __init__(self, index, kind, ncolors=1):
        self._ix = index
        self._lib = index._lib
        self.kind = {"abundance": 0, "colors": 1}[kind]
        self.ncolors = 1 if self.kind == 0 else int(ncolors)
        self._p = C.c_void_p()
        index._check(self._lib.blmx_payload_create(index._h, self.kind, self.ncolors, C.byref(self._p)))

    def close(self):
        if getattr(self, "_p", None):
            self._lib.blmx_payload_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_file(self, path, color=0):
        """every line of `path` that starts with A/C/G/T is a sequence; returns (k-mers found, k-mers absent)"""
        f, a = C.c_uint64(), C.c_uint64()
        self._ix._check(self._lib.blmx_payload_add_file(self._p, os.fsencode(path), color, C.byref(f), C.byref(a)))
        return f.value, a.value

    def add_ids_device(self, d_ids_ptr, n, color=0, stream=None):
        self._ix._check(self._lib.blmx_payload_add_ids_device(self._p, d_ids_ptr, n, color, stream))

    def export(self):
        """abundance: uint8[number_kmer]; colors: bool[number_kmer, ncolors]"""
        nbytes = self._lib.blmx_payload_export(self._p, None, 0)
        if nbytes < 0:
            self._ix._check(int(nbytes))
        raw = np.empty(nbytes, dtype=np.uint8)
        self._lib.blmx_payload_export(self._p, raw.ctypes.data, nbytes)
        if self.kind == 0:
            return raw
        n = self._ix.number_kmer
        return np.unpackbits(raw, bitorder="little")[: n * self.ncolors].reshape(n, self.ncolors).astype(bool)

"""Loader for tests/golden/*.npz (vectors produced from the real reference by oracle/make_golden.py)."""
import glob
import hashlib
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAMBDA = os.path.join(GOLD, "lambda_virus.unitigs.fa")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")) if "primitives" not in p)


class Case:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLD, name + ".npz"))
        self.z = z
        self.k, self.m, self.n, self.b, self.kmer_bytes = (int(x) for x in z["params"])
        self.seqs = bytes(z["seqs"]).decode().split("\n")
        self.ids = z["ids"]
        self.neg_kmers = z["neg_kmers"]
        self.neg_ids = z["neg_ids"]
        (self.number_kmer, self.number_super_kmer, self.total_nt, self.largest_mphf, self.largest_bucket,
         self.positions_bits) = (int(x) for x in z["stats"])

    def __getitem__(self, key):
        return self.z[key]

    def extra_queries(self):
        for key in self.z.files:
            if key.startswith("xq_") and not key.endswith("_ids"):
                yield key[3:], bytes(self.z[key]).decode(), self.z[key + "_ids"]


def primitives():
    return np.load(os.path.join(GOLD, "primitives.npz"))


def to_ints(arr, kmer_bytes):
    arr = np.asarray(arr)
    if kmer_bytes == 8:
        return [int(x) for x in arr.reshape(-1)]
    arr = arr.reshape(-1, 2)
    return [int(arr[i, 0]) | (int(arr[i, 1]) << 64) for i in range(arr.shape[0])]

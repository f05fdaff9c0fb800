#!/usr/bin/env python3
"""make_golden.py -- generate tests/golden/*.npz from the REAL reference (oracle/_ref, built by build_ref.sh).

Run in the dev container (needs /root/reference):   python oracle/make_golden.py
Every vector stored is an INPUT or an OUTPUT of the (patched, -t 1) reference; no reference source text is stored.
The only reference file copied is its data fixture lambda_virus.unitigs.fa (4 unitigs; data, not code).
"""
import hashlib
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LAMBDA_SRC = "/root/reference/lambda_virus.unitigs.fa"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def synthetic_unitigs(seed, n_unitigs, k, mean_extra, lowcomplex=0.0):
    """seeded random unitigs: length k-1+Geometric(mean_extra); a fraction drawn from a 2-letter source (bucket skew)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_unitigs):
        L = k - 1 + int(rng.geometric(1.0 / mean_extra))
        if rng.random() < lowcomplex:
            out.append("".join("AC"[i] for i in rng.integers(0, 2, L)))
        else:
            out.append(rand_seq(rng, L))
    return out


def random_canonical_kmers(rng, k, nq, kmer_bytes):
    xs = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(nq)]
    return [min(x, O.p_rc(x, k, kmer_bytes)) for x in xs]


def index_case(name, k, m, n, b, seqs, seed, nneg=2000, extra_queries=()):
    """Build with the reference; store ids of all k-mers in file order + negatives + internals digests."""
    ref = O.Reference(k, m, n, b, seqs)
    kb = ref.kmer_bytes
    rng = np.random.default_rng(seed)
    ids = np.concatenate([ref.query_sequence(s) for s in seqs]) if seqs else np.zeros(0, np.int64)
    neg = random_canonical_kmers(rng, k, nneg, kb)
    neg_arr = O.kmers_to_array(neg, kb)
    neg_ids = ref.query_kmers(neg)
    arrs = ref.arrays()
    sizes = np.zeros((ref.stats["ngroups"], 16), np.uint64)
    bits_h = hashlib.sha256()
    fb_keys, fb_ranks, fb_group = [], [], []
    for g in range(ref.stats["ngroups"]):
        mp = ref.mphf(g)
        if mp["nwords"]:
            sizes[g] = mp["sizes"]
            bits_h.update(mp["bits"].tobytes())
        for kk, rr in zip(mp["fallback_keys"], mp["fallback_ranks"]):
            fb_keys.append(kk)
            fb_ranks.append(rr)
            fb_group.append(g)
    # query_get_pos_unitig on the first 4000 k-mers of the input (with their true minimizers) and on the negatives
    out = {
        "params": np.array([k, m, n, b, kb], np.int64),
        "seqs": np.frombuffer("\n".join(seqs).encode(), np.uint8),  # newline-joined input sequences
        "ids": ids.astype(np.int64),
        "neg_kmers": neg_arr,
        "neg_ids": neg_ids,
        "stats": np.array([ref.stats[f] for f in ("number_kmer", "number_super_kmer", "total_nt", "largest_mphf",
                                                   "largest_bucket", "positions_bits")], np.uint64),
        "bucket_nt_sha": sha(arrs["bucket_nt"]), "bucket_start_sha": sha(arrs["bucket_start"]),
        "seq_sha": sha(arrs["seq"]), "positions_sha": sha(arrs["positions"]),
        "g_nkeys": arrs["g_nkeys"], "g_prefix": arrs["g_prefix"], "g_posstart": arrs["g_posstart"], "g_nbits": arrs["g_nbits"],
        "level_sizes": sizes, "mphf_bits_sha": bits_h.hexdigest(),
        "fb_keys": O.kmers_to_array(fb_keys, kb) if fb_keys else np.zeros((0, kb // 8), np.uint64),
        "fb_ranks": np.array(fb_ranks, np.uint64), "fb_group": np.array(fb_group, np.uint32),
    }
    # positional variant + caller-supplied (possibly wrong) minimizers
    orc = O.Oracle(k, m, n, b, seqs)
    kmers0 = orc.canonical_kmers(seqs[0][: min(len(seqs[0]), k + 1500)])
    kl = O.array_to_kmers(np.asarray(kmers0), kb)
    minis = np.array([O.p_minimizer(x, k, m, kb) for x in kl], np.uint32)
    out["pos_kmers"] = O.kmers_to_array(kl, kb)
    out["pos_minis"] = minis
    out["pos_out"] = ref.query_kmers_mini(kl, minis, want_pos=True)
    wrong = (minis + np.uint32(1)) % np.uint32(1 << (2 * m - 1))
    out["wrongmini_out"] = ref.query_kmers_mini(kl, wrong, want_pos=False)
    out["wrongmini"] = wrong
    for qname, q in extra_queries:
        out["xq_" + qname] = np.frombuffer(q.encode(), np.uint8)
        out["xq_" + qname + "_ids"] = ref.query_sequence(q)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-28s N=%d sk=%d found=%d distinct=%d fallback=%d negpos=%d" % (
        name, ref.stats["number_kmer"], ref.stats["number_super_kmer"], int((ids >= 0).sum()), len(set(ids.tolist())),
        len(fb_keys), int((neg_ids >= 0).sum())))
    ref.close()


def primitive_kats():
    rng = np.random.default_rng(7)
    r64 = O._ref(8)
    r128 = O._ref(16)
    import ctypes as C
    out = {}
    xs = rng.integers(0, 1 << 32, 512, dtype=np.uint64).astype(np.uint32)
    xs[:4] = [0, 1, 4660, 8191]
    out["revhash_in"] = xs
    out["revhash_out"] = np.array([r64.ref_p_revhash32(int(x)) for x in xs], np.int32)
    for m in (5, 7, 9, 11, 13, 15):
        v = (xs & np.uint32((1 << (2 * m)) - 1)).astype(np.uint32)
        out["mmer_in_m%d" % m] = v
        out["rc32_m%d" % m] = np.array([r64.ref_p_rc32(int(x), m) for x in v], np.uint32)
        out["parity_m%d" % m] = np.array([r64.ref_p_parity_canon(int(x), m) for x in v], np.uint32)
    out["nuc2int"] = np.array([r64.ref_p_nuc2int(bytes([c])) for c in range(256)], np.int32)
    # k-mers: rc + minimizer, both widths
    for kb, lib, ks, ms in ((8, r64, (15, 21, 25, 31, 32), (5, 7, 9)), (16, r128, (33, 47, 63), (7, 9, 11))):
        for k in ks:
            vals = [int.from_bytes(rng.bytes(16), "little") & ((1 << (2 * k)) - 1) for _ in range(128)]
            arr = O.kmers_to_array(vals, kb)
            o = np.empty_like(arr).reshape(len(vals), -1)
            arr2 = arr.reshape(len(vals), -1)
            for i in range(len(vals)):
                lib.ref_p_rc(arr2[i].ctypes.data, k, o[i].ctypes.data)
            out["kmer_in_k%d" % k] = arr
            out["kmer_rc_k%d" % k] = o.reshape(arr.shape)
            if k == 32:
                continue  # k=32 minimizer: offsetUpdateAnchor UB in the reference (SURVEY D-3); rc only
            for m in ms:
                out["mini_k%d_m%d" % (k, m)] = np.array(
                    [lib.ref_p_minimizer(arr2[i].ctypes.data, k, m) for i in range(len(vals))], np.uint32)
    # hashes (64-bit keys; the 128-bit functor ignores the high half: stored as a separate vector)
    keys = rng.integers(0, 1 << 63, 256, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 256, dtype=np.uint64)
    keys[:3] = [0, 1, 0x0123456789ABCDEF]
    lv = np.zeros((len(keys), 16), np.uint64)
    for i, kx in enumerate(keys):
        a = np.array([kx], np.uint64)
        buf = (C.c_uint64 * 16)()
        r64.ref_p_level_hashes(a.ctypes.data, 16, buf)
        lv[i] = list(buf)
    out["hash_keys"] = keys
    out["hash_levels"] = lv
    lv128 = np.zeros((32, 16), np.uint64)
    k128 = np.zeros((32, 2), np.uint64)
    for i in range(32):
        k128[i] = [keys[i], rng.integers(0, 1 << 62, dtype=np.uint64)]
        buf = (C.c_uint64 * 16)()
        r128.ref_p_level_hashes(k128[i].ctypes.data, 16, buf)
        lv128[i] = list(buf)
    out["hash_keys128"] = k128
    out["hash_levels128"] = lv128
    # level sizes = mphf::configureLevels: taken from real MPHFs in the index cases (level_sizes there); plus str2num
    strs = [rand_seq(rng, 31) for _ in range(64)] + ["ACGTTGCAACGTTGCAACGTTGCAACGTTGC", "acgttgcaacgttgcaacgttgcaacgttgc"]
    vals = []
    for s in strs:
        o = np.zeros(1, np.uint64)
        assert r64.ref_p_str2num(s.encode(), 31, o.ctypes.data) == 0
        vals.append(o[0])
    out["str2num_in"] = np.array(strs)
    out["str2num_out"] = np.array(vals, np.uint64)
    # chopper
    chops = []
    for i, (k, m) in enumerate(((31, 7), (31, 9), (21, 7), (25, 5), (63, 9), (31, 11))):
        s = rand_seq(rng, 400 + 50 * i)
        cap = len(s)
        o, l, mi = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        n = r64.ref_p_chop(s.encode(), len(s), k, m, O._ptr(o, O.u32p), O._ptr(l, O.u32p), O._ptr(mi, O.u32p), cap)
        out["chop%d_seq" % i] = np.array(s)
        out["chop%d_km" % i] = np.array([k, m])
        out["chop%d_out" % i] = np.stack([o[:n], l[:n], mi[:n]], axis=1)
    np.savez_compressed(os.path.join(GOLD, "primitives.npz"), **out)
    print("primitives.npz written")


def main():
    if not O.ref_available():
        O.build(force=True)
    assert O.ref_available(), "oracle/_ref missing: run oracle/build_ref.sh in the dev container"
    os.makedirs(GOLD, exist_ok=True)
    shutil.copyfile(LAMBDA_SRC, os.path.join(GOLD, "lambda_virus.unitigs.fa"))
    lam = O.read_fasta(LAMBDA_SRC)
    primitive_kats()
    lam_lc = lam[2].lower()
    lam_rc = revcomp(lam[2])
    index_case("lambda_k31_m7_n5_b6", 31, 7, 5, 6, lam, 101, extra_queries=(("lower", lam_lc), ("revcomp", lam_rc)))
    index_case("lambda_k31_m9_n7_b6", 31, 9, 7, 6, lam, 102)
    index_case("lambda_k31_m9_n14_b6", 31, 9, 14, 6, lam, 103)
    index_case("lambda_k31_m7_n10_b6", 31, 7, 10, 6, lam, 104)
    index_case("lambda_k31_m11_n7_b6", 31, 11, 7, 6, lam, 105)
    index_case("lambda_k21_m7_n5_b4", 21, 7, 5, 4, lam, 106)
    index_case("lambda_k25_m5_n3_b3", 25, 5, 3, 3, lam, 107)
    index_case("lambda_k31_m7_n5_b0", 31, 7, 5, 0, lam, 108)
    index_case("lambda_k31_m7_n0_b6", 31, 7, 0, 6, lam, 109)
    index_case("lambda_k63_m7_n5_b6", 63, 7, 5, 6, lam, 110)
    index_case("lambda_k63_m9_n7_b8", 63, 9, 7, 8, lam, 111)
    index_case("lambda_k15_m5_n4_b3", 15, 5, 4, 3, lam, 112)  # short k: duplicate k-mers + phantom positives
    index_case("dup_k31_m7_n5_b6", 31, 7, 5, 6, [lam[0], lam[1], lam[0]], 113)
    index_case("dup_k63_m9_n7_b8", 63, 9, 7, 8, [lam[0], lam[1], lam[0]], 114)
    syn = synthetic_unitigs(12345, 400, 31, 80)
    index_case("syn400_k31_m9_n7_b6", 31, 9, 7, 6, syn, 115)
    syn2 = synthetic_unitigs(777, 300, 31, 60, lowcomplex=0.15)
    index_case("synskew_k31_m7_n5_b6", 31, 7, 5, 6, syn2, 116)
    syn3 = synthetic_unitigs(4242, 200, 63, 120)
    index_case("syn200_k63_m9_n7_b8", 63, 9, 7, 8, syn3, 117)
    index_case("single_k31_m7_n5_b6", 31, 7, 5, 6, [lam[0][:31]], 118)  # exactly one k-mer


if __name__ == "__main__":
    main()

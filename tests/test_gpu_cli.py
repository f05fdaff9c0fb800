"""The reference's CLI harness (bench_blight.cpp) rebuilt on the C++ facade include/blight.h: construct + self-query of
the bundled lambda file must print the reference's recap (Good 48,462 / Erroneous 0)."""
import os
import subprocess

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_blight_cli_self_query():
    exe = os.path.join(ROOT, "tools", "bench_blight")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "bench_blight"])
    out = subprocess.run([exe, "-g", G.LAMBDA, "-k", "31", "-m", "7", "-n", "5", "-s", "3", "-t", "1", "-b", "6"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Kmer in graph: 48,462" in out.stdout
    assert "Super Kmer in graph: 3,708" in out.stdout
    assert "Largest MPHF: 3,036" in out.stdout and "Largest Bucket: 1,337" in out.stdout
    assert "Good kmer: 48,462" in out.stdout and "Erroneous kmers: 0" in out.stdout
    assert "Query performed: 48,462" in out.stdout
    # ctor errors surface as the reference's std::invalid_argument text
    bad = subprocess.run([exe, "-g", G.LAMBDA, "-k", "31", "-m", "10", "-n", "5", "-s", "3"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "minimizer_length must be odd" in bad.stderr


def test_bench_blight_cli_dump_then_load(tmp_path):
    """-d writes the index image, a second process -l loads it and answers the same file identically."""
    exe = os.path.join(ROOT, "tools", "bench_blight")
    img = str(tmp_path / "lambda.blmx")
    common = ["-k", "31", "-m", "7", "-n", "5", "-s", "3", "-t", "1", "-b", "6"]
    a = subprocess.run([exe, "-g", G.LAMBDA, "-d", img] + common, capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr
    b = subprocess.run([exe, "-q", G.LAMBDA, "-l", img] + common, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr
    assert "Kmer in graph" not in b.stdout  # nothing was constructed
    assert "Good kmer: 48,462" in b.stdout and "Erroneous kmers: 0" in b.stdout
    wrong = subprocess.run([exe, "-q", G.LAMBDA, "-l", img, "-k", "31", "-m", "7", "-n", "6", "-s", "3", "-b", "6"],
                           capture_output=True, text=True, timeout=60)
    assert wrong.returncode != 0 and "differ" in wrong.stderr

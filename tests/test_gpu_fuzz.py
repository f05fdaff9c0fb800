"""A short slice of the randomised parity sweep (tests/fuzz_parity.py; 3000 configurations were swept once by hand in r01, about 11600 with the
absent-key filter, the one-pass partition and the sharded construction from slices in r02):
random (k, m, n, b, k-mer width) and random inputs with low-complexity, tandem and duplicated stretches; counters, exported
arrays, IDs (direct and partitioned pipeline), negatives, near misses and EVERY k-mer readable anywhere in the packed sequence
(junctions of super-k-mers, of buckets and the zero padding behind the last one) against the CPU oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_parameter_sweep():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_parity
    fuzz_parity.sweep(80, 20261004, verbose=False)

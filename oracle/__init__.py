"""CPU oracle for the Blight hot path -- TEST INFRASTRUCTURE (see oracle/blight_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
"""

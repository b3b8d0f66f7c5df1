"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/planning_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity-sample legs (bench.py + benchmarks/) may import this.
"""

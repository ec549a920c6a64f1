"""CPU oracle of the subgraph-sketching hot path -- TEST INFRASTRUCTURE (see sketch_oracle.c header).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""

"""CPU oracle: test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline); never imported by the product."""

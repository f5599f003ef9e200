"""Parts of the benchmark driver (`bench.py` at the repository root is the entry point and the contract):
configs (BASELINE.json's workloads), work (algorithmic / executed flop and byte accounting, counter-traffic lookup),
sample (the reverse-diffusion sampling measurement), train (the DP training step), cpu_baseline (the oracle timed on the
host cores -- the only place outside tests/ and smoke() that imports oracle/)."""

"""Parts of bench.py (repo root): shared timing helpers, the decode measurement, the optional objects of the default line, the
launch plumbing.  bench.py keeps the contract, the primary train measurement and the CPU-oracle baselines."""

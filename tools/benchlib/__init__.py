"""Pieces of bench.py (the driver's entry point stays at the repo root): common = workloads / calibrations / helpers, run = one rank's
run, legs = single-GPU extras, sharded_legs = N > 1 report and verification, cpu = the CPU baseline."""

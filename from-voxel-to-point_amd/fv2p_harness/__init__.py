"""Synthetic-workload harness (benchmark/test plumbing, not part of the pcdet.ops boundary)."""

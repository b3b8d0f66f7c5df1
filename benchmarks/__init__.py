"""Workload implementations behind bench.py (repo root), split by planner family."""

"""Runners driving HBM-resident shards (mirror of wenbo_engine.runner)."""

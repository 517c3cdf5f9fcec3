"""Circuit front end and planners (mirror of wenbo_engine.circuit)."""

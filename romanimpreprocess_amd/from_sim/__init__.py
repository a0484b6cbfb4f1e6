"""Simulation side: the per-pixel part of the reference's ``from_sim`` package, on the GPU."""

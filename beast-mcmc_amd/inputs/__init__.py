"""Inputs of the hot path (models, rates, patterns, trees) — host-side, never on the GPU."""

"""Minimal ``basicsr`` namespace: only the arch modules on the Shift-Net inference hot path exist here."""

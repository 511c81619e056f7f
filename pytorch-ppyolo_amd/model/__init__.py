"""Reference-compatible module names (`from model.ppyolo import PPYOLO`, ...)."""

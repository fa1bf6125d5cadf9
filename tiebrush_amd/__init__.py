"""tiebrush_amd — MI355X-native tiebrush collapse / tiecov coverage hot path."""
__version__ = "0.1.0"

"""canonicalvoting_amd: MI355X-native hot path of CanonicalVoting."""
__version__ = "0.1.0"

"""MI355X-native streaming ASR engine behind speechcatcher's native-decoder API."""
__version__ = "0.1.0"

"""MI355X-native DiT denoise engine behind the LightX2V operator API."""
__version__ = "0.1.0"

"""torch_detection_amd — MI355X-native ResNet/FPN + box-op hot path behind the Torch_Detection registry."""
__version__ = "0.1.0"

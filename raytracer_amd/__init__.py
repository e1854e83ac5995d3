"""raytracer_amd — MI355X-native render path for lerouxrgd/raytracer's Camera::render."""
__version__ = "0.1.0"

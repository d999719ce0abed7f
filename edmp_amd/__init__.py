"""edmp_amd — MI355X-native guided reverse-diffusion sampler (EDMP hot path)."""
__version__ = "0.1.0"

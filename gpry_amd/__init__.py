"""gpry_amd -- MI355X-native GP regression + NORA acquisition hot path for GPry."""
__version__ = "0.1.0"

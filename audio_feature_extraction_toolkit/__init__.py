"""Import-name alias for drop-in use: ``from audio_feature_extraction_toolkit import AudioFeatureExtractor,
FeatureEvaluator`` (what the reference's examples/basic_usage.py:3 and its own package __init__ spell) resolves
to the MI355X engine in ``audio_feature_extraction_amd``.  This is an alias, not a second implementation: both
names are the same class objects."""
from audio_feature_extraction_amd import AudioFeatureExtractor, FeatureEvaluator, __version__

__all__ = ["AudioFeatureExtractor", "FeatureEvaluator"]

"""audio_feature_extraction_amd -- MI355X-native drop-in for
``audio_feature_extraction_toolkit`` (reference __init__.py:1-6): same two exports."""
from .core.feature_extractor import AudioFeatureExtractor
from .evaluation.evaluator import FeatureEvaluator

__version__ = '0.1.0'

__all__ = ['AudioFeatureExtractor', 'FeatureEvaluator']

"""Host-side behaviour of the drop-in classes that needs no GPU: constructor contract, public
attributes, the override detection that gates the fused path, f0 stand-in schema."""
import inspect

import numpy as np

import audio_feature_extraction_amd as pkg
from audio_feature_extraction_amd import AudioFeatureExtractor


def test_package_exports_match_reference():
    assert pkg.__all__ == ["AudioFeatureExtractor", "FeatureEvaluator"] and pkg.__version__ == "0.1.0"


def test_constructor_signature_and_attributes():
    sig = inspect.signature(AudioFeatureExtractor.__init__)
    names = list(sig.parameters)[1:8]
    assert names == ["sr", "frame_length", "hop_length", "n_mfcc", "f0_min", "f0_max", "pre_emphasis"]   # feature_extractor.py:10-17
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["sr"], d["frame_length"], d["hop_length"], d["n_mfcc"], d["pre_emphasis"]) == (22050, 1024, 256, 13, 0.97)
    assert abs(d["f0_min"] - 65.40639132514966) < 1e-12 and abs(d["f0_max"] - 2093.004522404789) < 1e-9   # C2, C7
    ex = AudioFeatureExtractor(16000, 512, 128, 40, 50.0, 500.0, 0.95)
    assert (ex.sr, ex.frame_length, ex.hop_length, ex.n_mfcc, ex.f0_min, ex.f0_max, ex.pre_emphasis) == \
        (16000, 512, 128, 40, 50.0, 500.0, 0.95)
    assert ex.logger.name.startswith("audio_feature_extraction_amd")
    for m in ("load_audio", "preprocess_audio", "extract_f0", "extract_mfcc", "extract_energy",
              "extract_features", "batch_process"):
        assert callable(getattr(ex, m))


def test_override_detection_gates_fused_path():
    ex = AudioFeatureExtractor()
    assert ex._uses_reference_stages()
    ex.preprocess_audio = lambda y: y                       # README.md:135-136 style monkey-patch
    assert not ex._uses_reference_stages()

    class Sub(AudioFeatureExtractor):
        def extract_energy(self, y):
            return {"energy_mean": 0.0, "energy_std": 0.0, "energy_range": 0.0}
    assert not Sub()._uses_reference_stages()


def test_f0_dict_schema_and_override_detection():
    out = AudioFeatureExtractor._f0_to_dict(np.array([220.0, 1.5, 0.25, 0.75]))
    assert list(out) == ["f0_mean", "f0_std", "f0_missing_rate", "f0_quality"]      # feature_extractor.py:109-114
    assert out == {"f0_mean": 220.0, "f0_std": 1.5, "f0_missing_rate": 0.25, "f0_quality": 0.75}
    assert all(type(v) is float for v in out.values())
    ex = AudioFeatureExtractor()
    assert ex._f0_on_gpu()
    ex.extract_f0 = lambda y: {"f0_mean": 1.0, "f0_std": 0.0, "f0_missing_rate": 0.0, "f0_quality": 1.0}
    assert not ex._f0_on_gpu() and not ex._uses_reference_stages()


def test_batch_process_empty_dir_returns_empty_list(tmp_path):
    assert AudioFeatureExtractor().batch_process(str(tmp_path)) == []

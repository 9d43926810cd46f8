"""FeatureEvaluator: the same four cases (and the same hand-written feature dicts) as the
reference's tests/test_evaluator.py:7-104, against this package's class."""
import json

import pytest

from audio_feature_extraction_amd import FeatureEvaluator

FEATURES = [
    {"file_path": "test1.wav", "f0_mean": 440.0, "f0_std": 1.0, "f0_missing_rate": 0.1, "f0_quality": 0.9,
     "mfcc_mean": [1.0, 2.0, 3.0], "mfcc_std": [0.1, 0.2, 0.3], "energy_mean": 0.8, "energy_std": 0.05},
    {"file_path": "test2.wav", "f0_mean": 880.0, "f0_std": 2.0, "f0_missing_rate": 0.2, "f0_quality": 0.8,
     "mfcc_mean": [2.0, 3.0, 4.0], "mfcc_std": [0.2, 0.3, 0.4], "energy_mean": 0.9, "energy_std": 0.06},
]


def test_calculate_feature_statistics():
    st = FeatureEvaluator().calculate_feature_statistics(FEATURES)
    assert isinstance(st, dict)
    for k in ("f0_mean_mean", "mfcc_mean_mean", "energy_mean_mean"):
        assert k in st
    assert st["f0_mean_mean"] == pytest.approx(660.0)          # test_evaluator.py:50
    assert st["mfcc_mean_mean"] == pytest.approx(2.5) and st["mfcc_mean_min"] == 1.0 and st["mfcc_mean_max"] == 4.0
    assert FeatureEvaluator().calculate_feature_statistics([]) == {}


def test_evaluate_feature_quality():
    q = FeatureEvaluator().evaluate_feature_quality(FEATURES)
    for k in ("total_files", "feature_integrity_rate", "f0_quality_rate", "mfcc_stability_rate", "energy_stability_rate"):
        assert k in q
    assert q["total_files"] == 2                                 # test_evaluator.py:64
    assert q["f0_quality_rate"] == pytest.approx(85.0)
    assert q["mfcc_stability_rate"] == 100.0 and q["energy_stability_rate"] == 100.0


def test_generate_evaluation_report(tmp_path):
    out = tmp_path / "test_output"
    rep = FeatureEvaluator().generate_evaluation_report(FEATURES, output_dir=str(out))
    assert set(rep) == {"statistics", "quality_metrics", "features_list"}
    assert (out / "evaluation_detailed.json").exists() and (out / "evaluation_summary.csv").exists()
    assert json.load(open(out / "evaluation_detailed.json"))["quality_metrics"]["total_files"] == 2
    assert (out / "evaluation_summary.csv").read_text().splitlines()[0] == "Metric,Value"


def test_analyze_feature_distribution():
    d = FeatureEvaluator().analyze_feature_distribution(FEATURES)
    for k in ("f0_distribution", "mfcc_distribution", "energy_distribution"):
        assert k in d
    f0 = d["f0_distribution"]
    assert f0["mean"] == pytest.approx(660.0) and set(f0["percentiles"]) == {"25", "50", "75"}

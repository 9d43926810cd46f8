"""Host-side consumer of the per-file feature dicts; mirrors the public behaviour of
the reference's ``FeatureEvaluator``
(audio_feature_extraction_toolkit/evaluation/evaluator.py:8-206): same method names,
arguments, result keys and report files, so the extractor's dict schema is pinned by
the same consumer.  Pure numpy/pandas post-processing of ~60 floats per file -- not
part of the GPU hot path."""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Any, Dict, List

import numpy as np


class FeatureEvaluator:
    MFCC_STD_THRESHOLD = 0.5       # evaluator.py:84
    ENERGY_STD_THRESHOLD = 0.1     # evaluator.py:92

    def __init__(self):
        logging.basicConfig(level=logging.INFO)
        self.logger = logging.getLogger(__name__)

    def calculate_feature_statistics(self, features_list: List[Dict[str, Any]]) -> Dict[str, Any]:
        """min/max/mean/std per feature name; list-valued features are flattened (evaluator.py:16-55)."""
        if not features_list:
            return {}
        names = set()
        for feats in features_list:
            names.update(feats.keys())
        names.discard("file_path")
        stats: Dict[str, Any] = {}
        for name in names:
            pool: List[float] = []
            for feats in features_list:
                if name not in feats:
                    continue
                v = feats[name]
                if isinstance(v, list):
                    pool.extend(v)
                else:
                    pool.append(v)
            if pool:
                arr = np.array(pool)
                stats[f"{name}_min"] = float(np.min(arr))
                stats[f"{name}_max"] = float(np.max(arr))
                stats[f"{name}_mean"] = float(np.mean(arr))
                stats[f"{name}_std"] = float(np.std(arr))
        return stats

    def evaluate_feature_quality(self, features_list: List[Dict[str, Any]]) -> Dict[str, float]:
        """Threshold rates in percent (evaluator.py:57-99)."""
        if not features_list:
            return {}
        total = len(features_list)
        f0_sum = sum(f.get("f0_quality", 0) for f in features_list)
        mfcc_ok = sum(1 for f in features_list
                      if np.mean(f.get("mfcc_std", [1.0])) < self.MFCC_STD_THRESHOLD)
        energy_ok = sum(1 for f in features_list
                        if f.get("energy_std", 1.0) < self.ENERGY_STD_THRESHOLD)
        return {
            "total_files": total,
            "feature_integrity_rate": 100.0,
            "f0_quality_rate": (f0_sum / total) * 100,
            "mfcc_stability_rate": (mfcc_ok / total) * 100,
            "energy_stability_rate": (energy_ok / total) * 100,
        }

    def generate_evaluation_report(self, features_list: List[Dict[str, Any]],
                                   output_dir: str = "feature_evaluation") -> Dict[str, Any]:
        """Writes evaluation_detailed.json + evaluation_summary.csv (evaluator.py:101-147)."""
        try:
            out = Path(output_dir)
            out.mkdir(parents=True, exist_ok=True)
            quality = self.evaluate_feature_quality(features_list)
            report = {
                "statistics": self.calculate_feature_statistics(features_list),
                "quality_metrics": quality,
                "features_list": features_list,
            }
            with open(out / "evaluation_detailed.json", "w", encoding="utf-8") as f:
                json.dump(report, f, indent=2, ensure_ascii=False)
            import pandas as pd
            pd.DataFrame({"Metric": list(quality.keys()), "Value": list(quality.values())}).to_csv(
                out / "evaluation_summary.csv", index=False)
            self.logger.info("評估報告生成完成")
            return report
        except Exception as e:
            self.logger.error(f"生成評估報告失敗: {str(e)}")
            raise

    @staticmethod
    def _dist(values) -> Dict[str, Any]:
        return {
            "mean": float(np.mean(values)),
            "std": float(np.std(values)),
            "percentiles": {q: float(np.percentile(values, int(q))) for q in ("25", "50", "75")},
        }

    def analyze_feature_distribution(self, features_list: List[Dict[str, Any]]) -> Dict[str, Any]:
        """mean/std/quartiles of f0_mean (>0 only), all mfcc_mean entries, energy_mean (evaluator.py:149-206)."""
        if not features_list:
            return {}
        res: Dict[str, Any] = {}
        f0 = [f.get("f0_mean", 0) for f in features_list if f.get("f0_mean", 0) > 0]
        if f0:
            res["f0_distribution"] = self._dist(f0)
        mf: List[float] = []
        for f in features_list:
            if "mfcc_mean" in f:
                mf.extend(f["mfcc_mean"])
        if mf:
            res["mfcc_distribution"] = self._dist(mf)
        en = [f.get("energy_mean", 0) for f in features_list]
        if en:
            res["energy_distribution"] = self._dist(en)
        return res

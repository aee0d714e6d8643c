"""Post-path output cleaning and scoring (SURVEY.md §8 f1 — the step right after the hot path).

``clean_prediction`` restates utils/evaluation_utils.py:469-595 of the reference for the three tasks in
scope (VOXCELEB single label; HVB multi label; VOXPOPULI multi label with 'none') and is pinned to the
reference function by tests/golden/clean_prediction.json.  ``evaluate_predictions`` reports the headline
numbers of utils/evaluation_utils.py:106-337 (accuracy / macro-F1 for VOXCELEB, sample-averaged
precision/recall/F1 + exact match for the multi-label tasks); the reference's per-class breakdown tables
are not reproduced yet (round-1 scope: DESIGN.md §6).
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Set

from ..data.task_configs import DatasetType, get_dataset_config


def clean_prediction(prediction: str, dataset_type: Optional[DatasetType] = None) -> str:
    text = re.sub(r"\s+", " ", prediction.replace("\\", ""))
    if "\n" in text:  # unreachable after whitespace normalisation; kept for parity with the reference's order
        text = text.split("\n")[0]
    text = re.sub(r",\s*,", ",", text)
    text = re.sub(r",\s*$", "", text)
    text = re.sub(r"^\s*,", "", text)
    valid: Optional[Set[str]] = None
    if dataset_type is not None:
        try:
            labels = get_dataset_config(dataset_type).valid_labels
            valid = {l.lower() for l in labels} if labels else None
        except Exception:
            valid = None
    dt = DatasetType(dataset_type) if dataset_type is not None else None
    if dt == DatasetType.VOXCELEB:
        words = [w.strip().lower() for w in re.split(r"[^a-zA-Z]", text)]
        words = [w for w in words if w]
        if valid and words:
            for w in words:
                if w in valid:
                    return w
            return words[0]
        return words[0] if words else text.lower()
    if dt in (DatasetType.HVB, DatasetType.VOXPOPULI):
        if dt == DatasetType.VOXPOPULI and text.lower().strip() == "none":
            return "none"
        labels = [l.strip().lower() for l in text.split(",")]
        labels = [l for l in labels if l and "(" not in l]
        if valid:
            ok = set(valid) | ({"none"} if dt == DatasetType.VOXPOPULI else set())
            found = [l for l in labels if l in ok]
            return ", ".join(found) if found else text
        return ", ".join(labels) if labels else text
    return text.lower().strip()


def _label_set(s: str) -> Set[str]:
    return {p.strip().lower() for p in s.split(",") if p.strip() and p.strip().lower() != "none"}


def evaluate_predictions(predictions: List[Dict[str, Any]], dataset_type: DatasetType) -> Dict[str, Any]:
    if not predictions:
        return {"error": "Empty predictions list", "accuracy": 0.0}
    dt = DatasetType(dataset_type)
    gt = [str(p.get("true_label", "")).lower().strip() for p in predictions]
    pd_ = [clean_prediction(str(p.get("predicted_label", "")), dt) for p in predictions]
    n = len(gt)
    if dt == DatasetType.VOXCELEB:
        classes = sorted(set(gt) | set(get_dataset_config(dt).valid_labels))
        acc = sum(a == b for a, b in zip(gt, pd_)) / n
        f1s = []
        for c in classes:
            tp = sum(a == c and b == c for a, b in zip(gt, pd_))
            fp = sum(a != c and b == c for a, b in zip(gt, pd_))
            fn = sum(a == c and b != c for a, b in zip(gt, pd_))
            f1s.append(2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) else 0.0)
        return {"accuracy": acc, "macro_f1": sum(f1s) / len(f1s), "total_samples": n,
                "invalid_predictions": sum(b not in classes for b in pd_)}
    pr = rc = f1 = em = 0.0
    for a, b in zip(gt, pd_):
        sa, sb = _label_set(a), _label_set(b)
        inter = len(sa & sb)
        p_ = inter / len(sb) if sb else float(not sa)
        r_ = inter / len(sa) if sa else float(not sb)
        pr, rc = pr + p_, rc + r_
        f1 += 2 * p_ * r_ / (p_ + r_) if (p_ + r_) else 0.0
        em += float(sa == sb)
    return {"precision": pr / n, "recall": rc / n, "f1": f1 / n, "exact_match": em / n, "accuracy": em / n,
            "total_samples": n}

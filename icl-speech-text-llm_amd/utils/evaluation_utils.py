"""Post-path output cleaning and scoring (SURVEY.md §8 f1 — the step right after the hot path).

Restates, table for table, the reference's ``utils/evaluation_utils.py``:
  ``clean_prediction``      :469-595   (pinned by tests/golden/clean_prediction.json)
  ``evaluate_predictions``  :16-104    dispatch on the task family, swap families score against swap entry 1
  ``evaluate_voxceleb``     :106-211   single label: filtered accuracy / macro-F1 / per-class P,R,F1 / confusion matrix,
                                       plus macro-F1 with out-of-vocabulary predictions counted as 'invalid'
  ``evaluate_hvb``          :213-274   multi label indicator matrices: macro / micro / weighted F1, per class, exact match
  ``evaluate_voxpopuli``    :276-337   same with the extra 'none' class
  ``evaluate_vp_nel``       :356-467   entity spans with time alignment: word-level P/R/F1 at 6 overlap tolerances, frame-level
  ``evaluate_sqa``          :832-957   normalised exact match, token F1, sentence BLEU (method-1 smoothing)
The reference computes these through pandas + scikit-learn; here they are plain numpy counts with scikit-learn's
conventions (``zero_division=0``; macro = unweighted mean over the listed classes; weighted = support-weighted mean),
pinned to the reference's outputs by tests/golden/metrics.json.  ``evaluate_sqa`` is unpinned for its BLEU term (the
reference imports nltk, which is absent here and makes its own SQA scoring fail at import).
"""
from __future__ import annotations

import logging
import math
import re
from collections import Counter
from typing import Any, Dict, List, Optional, Sequence, Set

import numpy as np

from ..data.task_configs import DatasetType, get_dataset_config, get_swap_config, is_swap_type

logger = logging.getLogger(__name__)

_SINGLE_CLEAN = ("VOXCELEB", "VOXCELEB_GREEK", "MELD_EMOTION", "MELD_EMOTION_GREEK")
_SINGLE_EVAL = ("VOXCELEB", "VOXCELEB_SWAP", "VOXCELEB_GREEK", "MELD", "MELD_GREEK", "MELD_EMOTION", "MELD_EMOTION_GREEK")
_HVB_EVAL = ("HVB", "HVB_SWAP", "HVB_GREEK")
_VP_EVAL = ("VOXPOPULI", "VOXPOPULI_SWAP", "VOXPOPULI_GREEK")


def _name(dataset_type) -> Optional[str]:
    if dataset_type is None:
        return None
    try:
        return DatasetType(dataset_type).name
    except ValueError:
        return None


def clean_prediction(prediction: str, dataset_type: Optional[DatasetType] = None) -> str:
    text = re.sub(r"\s+", " ", prediction.replace("\\", ""))
    if "\n" in text:  # unreachable after whitespace normalisation; kept for parity with the reference's order
        text = text.split("\n")[0]
    text = re.sub(r",\s*,", ",", text)
    text = re.sub(r",\s*$", "", text)
    text = re.sub(r"^\s*,", "", text)
    valid: Optional[Set[str]] = None
    if dataset_type:
        try:
            cfg = get_dataset_config(dataset_type)
            if cfg is not None and cfg.valid_labels:
                valid = {l.lower() for l in cfg.valid_labels}
        except Exception:
            valid = None
    dt = _name(dataset_type)
    if dt in _SINGLE_CLEAN:
        words = [w.strip().lower() for w in re.split(r"[^a-zA-Z]", text)]
        words = [w for w in words if w]
        if valid and words:
            for w in words:
                if w in valid:
                    return w
            return words[0]
        return words[0] if words else text.lower()
    if dt in ("HVB", "HVB_GREEK", "VOXPOPULI", "VOXPOPULI_GREEK"):
        with_none = dt.startswith("VOXPOPULI")
        if with_none and text.lower().strip() == "none":
            return "none"
        labels = [l.strip().lower() for l in text.split(",")]
        labels = [l for l in labels if l and "(" not in l]
        if valid:
            ok = set(valid) | ({"none"} if with_none else set())
            found = [l for l in labels if l in ok]
            return ", ".join(found) if found else text
        return ", ".join(labels) if labels else text
    if dt == "SQA":
        text = text.strip()
        try:
            start, end = map(float, text.split())
            return f"{start:.2f} {end:.2f}"
        except Exception:
            return text
    if dt == "VOXPOPULI_NEL":
        if text.lower() == "none":
            return "none"
        spans = []
        for span in text.split(";"):
            span = span.strip()
            if ":" in span:
                entity, times = span.split(":", 1)
                try:
                    start, end = map(float, times.strip().split())
                    spans.append(f"{entity.strip()}: {start:.2f} {end:.2f}")
                except Exception:
                    spans.append(span)
        return "; ".join(spans)
    return text.lower().strip()


# ---------------------------------------------------------------------------------------------------------------------
# counting helpers (scikit-learn conventions)
# ---------------------------------------------------------------------------------------------------------------------
def _div0(num: np.ndarray, den: np.ndarray) -> np.ndarray:
    num, den = np.asarray(num, np.float64), np.asarray(den, np.float64)
    out = np.zeros_like(num)
    np.divide(num, den, out=out, where=den != 0)
    return out


def _prf(tp, fp, fn):
    p, r = _div0(tp, tp + fp), _div0(tp, tp + fn)
    return p, r, _div0(2 * tp, 2 * tp + fp + fn)


def _single_label_counts(gt: Sequence[str], pd_: Sequence[str], classes: Sequence[str]):
    tp = np.array([sum(g == c and p == c for g, p in zip(gt, pd_)) for c in classes], np.float64)
    fp = np.array([sum(g != c and p == c for g, p in zip(gt, pd_)) for c in classes], np.float64)
    fn = np.array([sum(g == c and p != c for g, p in zip(gt, pd_)) for c in classes], np.float64)
    return tp, fp, fn


def evaluate_voxceleb(gt: Sequence[str], pd_: Sequence[str], valid_classes: List[str]) -> Dict[str, Any]:
    total = len(gt)
    rows = [(g.lower(), p.lower()) for g, p in zip(gt, pd_)]
    rows = [(g, p) for g, p in rows if g in valid_classes]
    n_gt = len(rows)
    invalid = sum(p not in valid_classes for _, p in rows)
    g_all = [g for g, _ in rows]
    p_all = [p if p in valid_classes else "invalid" for _, p in rows]
    macro_with_invalid = float(_prf(*_single_label_counts(g_all, p_all, valid_classes))[2].mean()) if valid_classes else 0.0
    kept = [(g, p) for g, p in rows if p in valid_classes]
    if not kept:
        return {"macro_f1_filtered": 0.0, "macro_f1_with_invalid": 0.0, "invalid_predictions": invalid,
                "total_samples": total, "valid_gt_samples": n_gt, "valid_samples": 0}
    g, p = [a for a, _ in kept], [b for _, b in kept]
    index = {c: i for i, c in enumerate(valid_classes)}
    cm = np.zeros((len(valid_classes),) * 2, np.int64)
    for a, b in kept:
        cm[index[a], index[b]] += 1
    with np.errstate(divide="ignore", invalid="ignore"):
        class_acc = cm.diagonal() / cm.sum(axis=1)          # NaN for a class without samples, as numpy gives the reference
    prec, rec, f1 = _prf(*_single_label_counts(g, p, valid_classes))
    return {
        "accuracy": float(np.mean([a == b for a, b in kept])),
        "macro_f1_filtered": float(f1.mean()),
        "class_accuracy_filtered": class_acc.tolist(),
        "class_precision": prec.tolist(), "class_recall": rec.tolist(), "class_f1": f1.tolist(),
        "confusion_matrix_filtered": cm.tolist(),
        "valid_samples": len(kept),
        "macro_f1_with_invalid": macro_with_invalid,
        "invalid_predictions": invalid,
        "total_samples": total, "valid_gt_samples": n_gt, "valid_classes": valid_classes,
    }


def _multilabel(gt: Sequence[Any], pd_: Sequence[Any], classes: List[str], reported_classes: List[str], strip: bool):
    def split(x):
        if isinstance(x, str):
            x = x.split(",")
        return [(l.strip() if strip else l).lower() for l in x]

    total = len(gt)
    rows = [(split(g), split(p)) for g, p in zip(gt, pd_)]
    rows = [(g, p) for g, p in rows if any(l in classes for l in g)]
    invalid = sum(not any(l in classes for l in p) for _, p in rows)

    def vec(labels):
        if not any(l in classes for l in labels):
            return np.zeros(len(classes))
        return np.array([1.0 if c in labels else 0.0 for c in classes])

    if not rows:
        raise ValueError("no sample with a valid ground-truth label")     # scikit-learn rejects the empty matrices too
    yt, yp = np.stack([vec(g) for g, _ in rows]), np.stack([vec(p) for _, p in rows])
    tp, fp, fn = (yt * yp).sum(0), ((1 - yt) * yp).sum(0), (yt * (1 - yp)).sum(0)
    prec, rec, f1 = _prf(tp, fp, fn)
    support = yt.sum(0)
    micro = _prf(tp.sum(), fp.sum(), fn.sum())[2]
    weighted = float((f1 * support).sum() / support.sum()) if support.sum() else 0.0
    return {
        "exact_match": float(sum(np.array_equal(a, b) for a, b in zip(yt, yp)) / max(1, len(yt))),
        "macro_f1": float(f1.mean()), "micro_f1": float(micro), "weighted_f1": weighted,
        "class_precision": prec.tolist(), "class_recall": rec.tolist(), "class_f1": f1.tolist(),
        "support": support.tolist(),
        "total_samples": total, "valid_gt_samples": len(rows), "invalid_samples": invalid,
        "valid_classes": reported_classes,
    }


def evaluate_hvb(gt, pd_, valid_classes: List[str]) -> Dict[str, Any]:
    return _multilabel(gt, pd_, valid_classes, valid_classes, strip=False)   # the reference does not strip HVB labels (:217-218)


def evaluate_voxpopuli(gt, pd_, valid_classes: List[str]) -> Dict[str, Any]:
    classes = valid_classes if "none" in valid_classes else valid_classes + ["none"]
    return _multilabel(gt, pd_, classes, valid_classes, strip=True)


def parse_entities(entity_string: str):
    """``"TYPE: start end; TYPE2: start end"`` → [(type, start, end)]; malformed spans are skipped (:339-354)."""
    out = []
    if not entity_string or entity_string.strip() == "":
        return out
    for ent in entity_string.split(";"):
        if ent.strip():
            try:
                etype, times = ent.strip().split(":")
                start, end = map(float, times.strip().split())
                out.append((etype.strip(), start, end))
            except Exception as e:
                logger.warning("Error parsing entity: %s, Error: %s", ent, e)
    return out


def evaluate_vp_nel(gt: Sequence[str], pd_: Sequence[str], valid_classes=None) -> Dict[str, Any]:
    G = [parse_entities(g.lower()) for g in gt]
    P = [parse_entities(p.lower()) for p in pd_]
    word = {}
    for tol in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5):
        correct = n_pred = n_gt = 0
        for ge, pe in zip(G, P):
            n_gt, n_pred = n_gt + len(ge), n_pred + len(pe)
            used: Set[int] = set()
            for ptype, ps, pe_ in pe:
                best, best_i = 0.0, None
                for i, (gtype, gs, ge_) in enumerate(ge):
                    if i in used or ptype.upper() != gtype.upper():
                        continue
                    lo, hi = max(ps, gs), min(pe_, ge_)
                    if hi > lo:
                        ov = (hi - lo) / (ge_ - gs)
                        if ov >= tol and ov > best:
                            best, best_i = ov, i
                if best_i is not None:
                    correct += 1
                    used.add(best_i)
        prec, rec = correct / max(n_pred, 1), correct / max(n_gt, 1)
        word[str(tol)] = {"precision": prec, "recall": rec, "f1": 2 * (prec * rec) / max(prec + rec, 1e-6)}
    fp_ = fg = fc = 0
    for ge, pe in zip(G, P):
        for ptype, ps, pe_ in pe:
            fp_ += int((pe_ - ps) * 100)
            for gtype, gs, ge_ in ge:
                if ptype.upper() == gtype.upper():
                    lo, hi = max(ps, gs), min(pe_, ge_)
                    if hi > lo:
                        fc += int((hi - lo) * 100)
        for _, gs, ge_ in ge:
            fg += int((ge_ - gs) * 100)
    prec, rec = fc / max(fp_, 1), fc / max(fg, 1)
    return {
        "word_metrics": word,
        "frame_metrics": {"precision": prec, "recall": rec, "f1": 2 * (prec * rec) / max(prec + rec, 1e-6)},
        "total_samples": len(gt),
        "total_gt_entities": sum(len(e) for e in G), "total_pred_entities": sum(len(e) for e in P),
        "total_frames": {"gt": fg, "pred": fp_, "correct": fc},
    }


def _sentence_bleu_method1(reference: List[str], hypothesis: List[str], max_n: int = 4, eps: float = 0.1) -> float:
    """Sentence BLEU-4 with uniform weights and Chen & Cherry (2014) smoothing 1 (zero n-gram matches count as eps), the
    formula nltk's ``sentence_bleu(..., smoothing_function=SmoothingFunction().method1)`` publishes."""
    if not hypothesis:
        return 0.0
    logs = 0.0
    for n in range(1, max_n + 1):
        hyp = Counter(tuple(hypothesis[i:i + n]) for i in range(len(hypothesis) - n + 1))
        ref = Counter(tuple(reference[i:i + n]) for i in range(len(reference) - n + 1))
        num, den = sum((hyp & ref).values()), max(1, sum(hyp.values()))
        if n == 1 and num == 0:
            return 0.0
        logs += math.log((num if num else eps) / den) / max_n
    bp = 1.0 if len(hypothesis) > len(reference) else math.exp(1 - len(reference) / len(hypothesis))
    return bp * math.exp(logs)


def evaluate_sqa(gt: Sequence[Any], pd_: Sequence[Any], valid_classes=None) -> Dict[str, Any]:
    def norm(t):
        if t is None:
            return ""
        t = re.sub(r"[^\w\s]", " ", str(t).lower())
        return re.sub(r"\s+", " ", t).strip()

    f1s, bleus, em = [], [], 0
    for g, p in zip(gt, pd_):
        em += int(norm(g) == norm(p))
        gt_tok, pd_tok = norm(g).split(), norm(p).split()
        if not gt_tok and not pd_tok:
            f1 = 1.0
        elif not gt_tok or not pd_tok:
            f1 = 0.0
        else:
            common = sum((Counter(gt_tok) & Counter(pd_tok)).values())
            prec, rec = common / max(len(pd_tok), 1), common / max(len(gt_tok), 1)
            f1 = 2 * (prec * rec) / max(prec + rec, 1e-6)
        f1s.append(f1)
        bleus.append(_sentence_bleu_method1(gt_tok, pd_tok) if gt_tok else (0.0 if pd_tok else 1.0))
    n = len(f1s)
    return {"exact_match": em / max(len(gt), 1), "f1_score": sum(f1s) / max(n, 1), "bleu_score": sum(bleus) / max(n, 1),
            "total_samples": len(gt), "samples_evaluated": n,
            "sample_metrics": {"exact_match": [1 if f == 1.0 else 0 for f in f1s], "f1_scores": f1s, "bleu_scores": bleus}}


def evaluate_predictions(predictions: List[Dict[str, Any]], dataset_type: DatasetType) -> Dict[str, Any]:
    if not predictions:
        logger.warning("Empty predictions list provided for evaluation")
        return {"error": "Empty predictions list", "accuracy": 0.0}
    try:
        dt = _name(dataset_type)
        swap_scored = dt in ("VOXCELEB_SWAP", "HVB_SWAP", "VOXPOPULI_SWAP")       # MELD_EMOTION_SWAP is not in the list (:34)
        config = get_swap_config(dataset_type) if swap_scored else get_dataset_config(dataset_type)
        if not config:
            logger.warning("No config found for dataset type: %s", dataset_type)
            return {"error": "Invalid dataset type"}
        gt = [p.get("true_label", "") for p in predictions]
        pd_ = [clean_prediction(p.get("predicted_label", ""), dataset_type) for p in predictions]
        valid = [l.lower() for l in config.valid_labels] if config.valid_labels is not None else None
        if dt in _SINGLE_EVAL:
            return evaluate_voxceleb(gt, pd_, valid)
        if dt in _HVB_EVAL:
            return evaluate_hvb(gt, pd_, valid)
        if dt in _VP_EVAL:
            return evaluate_voxpopuli(gt, pd_, valid)
        if dt == "VOXPOPULI_NEL":
            return evaluate_vp_nel(gt, pd_, valid)
        if dt == "SQA":
            return evaluate_sqa(gt, pd_)
        logger.warning("Unsupported dataset type for evaluation: %s", dataset_type)
        return {"accuracy": 0.0}
    except Exception as e:
        logger.error("Error in evaluate_predictions: %s", e)
        return {"error": str(e), "accuracy": 0.0}

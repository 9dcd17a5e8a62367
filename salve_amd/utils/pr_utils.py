"""Precision / recall / mean accuracy of binary edge classifications (class 1 = correctly aligned = positive).

Mirror of salve/utils/pr_utils.py:34-103 (compute_tp_fp_fn_tn_counts, compute_precision_recall); pinned by the
reference's tests/test_pr_utils.py cases, restated in tests/test_dataset_contract.py.
"""

from typing import Tuple

import numpy as np

EPS = 1e-7


def compute_tp_fp_fn_tn_counts(y_true: np.ndarray, y_pred: np.ndarray) -> Tuple[int, int, int, int]:
    agree = y_true == y_pred
    tp = int(np.logical_and(agree, y_pred == 1).sum())
    fp = int(np.logical_and(~agree, y_pred == 1).sum())
    fn = int(np.logical_and(~agree, y_pred == 0).sum())
    tn = int(np.logical_and(agree, y_pred == 0).sum())
    return tp, fp, fn, tn


def compute_precision_recall(y_true: np.ndarray, y_pred: np.ndarray) -> Tuple[float, float, float]:
    """(precision, recall, mean accuracy): rows of the confusion matrix (actual P, actual N) are normalised with an
    epsilon of 1e-7 and the mean accuracy is the mean of its diagonal."""
    tp, fp, fn, tn = compute_tp_fp_fn_tn_counts(y_true, y_pred)
    acc_pos = tp / (tp + fn + EPS)
    acc_neg = tn / (fp + tn + EPS)
    return tp / (tp + fp + EPS), tp / (tp + fn + EPS), float(np.mean([acc_pos, acc_neg]))

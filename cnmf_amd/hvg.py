"""High-variance-gene selection from per-gene moments.

The O(cells x genes) part of the reference's ``get_highvar_genes`` / ``get_highvar_genes_sparse``
(cnmf.py:192-246, 136-188) is the per-gene mean and population variance of the TPM matrix; on the
device that is ``Engine.col_mean_var()`` (float64, two passes).  Everything after it is O(genes) and is
restated here: the Fano factor, the expected-Fano line ``A^2 * mean + B^2`` (A from the 20 most expressed
genes, B from the median Fano factor inside the 10 %-90 % winsor box), and either the ``numgenes`` largest
Fano ratios or the ``T``-threshold rule.
"""
import numpy as np
import pandas as pd


def highvar_genes_from_moments(mean, var, numgenes=None, expected_fano_threshold=None, minimal_mean=0.5):
    """Returns ``(stats DataFrame, params dict)`` with the reference's columns / keys."""
    mean = pd.Series(np.asarray(mean, dtype=np.float64))
    var = pd.Series(np.asarray(var, dtype=np.float64))
    fano = var / mean
    # coefficient of variation of the 20 most expressed genes: the smallest one anchors the line's slope
    top = mean.sort_values(ascending=False).index[:20]
    A = (np.sqrt(var) / mean)[top].min()
    m_lo, m_hi = mean.quantile([0.10, 0.90])
    f_lo, f_hi = fano.quantile([0.10, 0.90])
    box = (fano > f_lo) & (fano < f_hi) & (mean > m_lo) & (mean < m_hi)
    B = np.sqrt(fano[box].median())
    expected = (A ** 2) * mean + (B ** 2)
    ratio = fano / expected
    if numgenes is not None:
        chosen = ratio.sort_values(ascending=False).index[:numgenes]
        high = ratio.index.isin(chosen)
        T = None
    else:
        T = expected_fano_threshold if expected_fano_threshold else 1.0 + fano[box].std()
        high = (ratio > T) & (mean > minimal_mean)
    stats = pd.DataFrame({"mean": mean, "var": var, "fano": fano, "expected_fano": expected,
                          "high_var": np.asarray(high), "fano_ratio": ratio})
    return stats, {"A": A, "B": B, "T": T, "minimal_mean": minimal_mean}

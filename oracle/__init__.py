"""CPU oracle for the cNMF hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  The product
path (``cnmf_amd``) never imports this package and fails loudly when the HIP
extension is missing.

Contents
--------
nmf_cd.py        numpy restatement of scikit-learn's coordinate-descent NMF
                 (sklearn 1.7.2 ``decomposition/_nmf.py`` + ``_cdnmf_fast.pyx``),
                 the arithmetic behind ``cNMF._nmf`` (reference cnmf.py:661-674).
consensus.py     numpy restatement of the consensus core (reference
                 cnmf.py:871-936) incl. sklearn's euclidean_distances, KMeans
                 (k-means++ / Lloyd), silhouette, and pandas' groupby-median.
sklearn_ref.py   thin wrapper that calls the REAL scikit-learn functions the
                 reference calls (the arithmetic lives in that third-party
                 dependency, pinned only as ``scikit-learn>=1.0`` in the
                 reference's setup.py:41; 1.7.2 is installed in this image and
                 on the GPU box).  Used to pin the restatements and as the
                 ``cpu_baseline`` of kind "reference".
scanpy_shim.py   ~60-line stand-in for the parts of scanpy the reference
                 imports, so the UNMODIFIED reference (``/root/reference/src``)
                 can run in the build container to generate golden vectors
                 (tests/golden/, see tools/make_golden.py).  Never used on the
                 GPU box (the reference tree does not exist there).
mt19937.c        C restatement of numpy's legacy MT19937 + polar ``gauss`` used
                 by ``RandomState(seed).standard_normal`` -- validates the
                 on-device random-init kernel bit-for-bit against numpy.

Parity pinning: the reference's own golden files are download-only (no
network), and its tests never pin ``factorize`` numerics
(tests/test_reproducibility.py:85-89 copies merged spectra instead).  The
restatements are therefore pinned against (a) the live scikit-learn/numpy/pandas
functions on seeded inputs (tests/test_oracle_*.py), (b) known-answer vectors
from SURVEY.md section 8c, and (c) fixtures under tests/golden/ produced by the
unmodified reference run through scanpy_shim here (tools/make_golden.py).
"""

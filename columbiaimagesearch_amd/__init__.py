"""columbiaimagesearch_amd -- MI355X-native embed-then-index hot path of ColumbiaImageSearch.

Sub-packages mirror the reference's plugin surfaces for this path only:

* ``columbiaimagesearch_amd.lopq``  -- ``LOPQModel`` / ``LOPQModelPCA`` / ``LOPQSearcherHIP``
  (reference: the vendored ``lopq`` package, lopq/lopq/{model,search,utils}.py)
* ``columbiaimagesearch_amd.featurizer`` -- batched CNN featurizers behind the cufacesearch
  ``GenericFeaturizer`` shape (reference: cufacesearch/cufacesearch/featurizer/)

All arithmetic runs in libcis_hip.so (HIP kernels for gfx950, C ABI in include/cis_hip.h); there is
no CPU fallback.
"""
__version__ = "0.1.0"

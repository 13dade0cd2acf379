"""Drop-in for the reference's vendored ``lopq`` package (lopq/lopq/__init__.py)."""
from . import model, search, utils
from .model import LOPQCode, LOPQModel, LOPQModelPCA
from .search import LOPQSearcher, LOPQSearcherHIP, LOPQSearcherLMDB, multisequence

__all__ = ["LOPQModel", "LOPQModelPCA", "LOPQSearcher", "LOPQSearcherHIP", "LOPQSearcherLMDB", "LOPQCode", "multisequence", "model", "search",
           "utils"]


def install_as_lopq():
    """Register this package under the name ``lopq`` so that ``import lopq`` in cufacesearch
    (cufacesearch/searcher/searcher_lopqhbase.py:14-16) and pickles that reference
    ``lopq.model.LOPQModelPCA`` (cufacesearch/storer/local.py:58,75) resolve to the HIP classes."""
    import sys
    sys.modules["lopq"] = sys.modules[__name__]
    sys.modules["lopq.model"] = model
    sys.modules["lopq.search"] = search
    sys.modules["lopq.utils"] = utils

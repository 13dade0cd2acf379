"""Batched counterpart of cufacesearch.extractor.generic_extractor for the full-image DeepSentibank path."""
from .generic_extractor import (GenericExtractor, build_extr_str, build_extr_str_failed, build_extr_str_processed)

__all__ = ["GenericExtractor", "build_extr_str", "build_extr_str_processed", "build_extr_str_failed"]

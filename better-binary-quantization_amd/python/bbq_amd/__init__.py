"""bbq_amd - Python plumbing over libbbq (the MI355X-native scan + top-k path).  The drop-in host API of the
reference lives in ../../js (JavaScript + .d.ts over the N-API addon); api.py mirrors the same surface for
pytest / bench.py."""
from .capi import BBQError, Index, SIMS, Vectors, search_rerank_batch, file_info, file_shards, centroid_dp, device_count, quantize_query, quantize_queries, quantize_vectors, replay, replay_batch, merge_answers, key_of_score  # noqa: F401
from .api import (BinaryQuantizationFormat, DEFAULT_CONFIG, VectorSimilarityFunction, createBinaryQuantizationFormat,  # noqa: F401
                  quickQuantize, quickSearch, createDeviceVectors, getOversampledTopKWithHeap, getOversampledTopKWithSort)

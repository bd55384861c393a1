from lcm.datasets.batch import EmbeddingsBatch, PaddingMask  # noqa: F401

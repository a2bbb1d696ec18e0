from . import anchor_free_head, maskformer_head  # noqa: F401

from openpvsg_amd.relation import PositionalEncoding, TemporalTransformer  # noqa: F401

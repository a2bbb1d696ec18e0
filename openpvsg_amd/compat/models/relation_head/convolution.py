from openpvsg_amd.relation import HandcraftedFilter, Learnable1DConv  # noqa: F401

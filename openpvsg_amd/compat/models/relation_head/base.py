from openpvsg_amd.relation import ObjectEncoder, PairProposalNetwork, VanillaModel  # noqa: F401

from openpvsg_amd.relation import concatenate_sub_obj  # noqa: F401

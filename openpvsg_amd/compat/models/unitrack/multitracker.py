from openpvsg_amd.unitrack import AssociationTracker, class_aware_distance  # noqa: F401

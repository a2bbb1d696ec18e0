from openpvsg_amd.unitrack import remove_duplicated_box, tlbr_to_tlwh, tlwh_to_tlbr, tlwh_to_xyah  # noqa: F401

import os

from openpvsg_amd.tubes import write_mots_results  # noqa: F401


def mkdir_if_missing(d):
    os.makedirs(d, exist_ok=True)

from .builder import DATASETS, PIPELINES, build_dataloader, build_dataset  # noqa: F401
from . import pipelines  # noqa: F401


def replace_ImageToTensor(pipelines_cfg):
    return pipelines_cfg

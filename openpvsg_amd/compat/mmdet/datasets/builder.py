from openpvsg_amd import registry as R
from openpvsg_amd.compat._policy import protect

DATASETS = protect(R.DATASETS)
PIPELINES = protect(R.PIPELINES)


def build_dataset(cfg, default_args=None):
    return R.build_from_cfg(cfg, DATASETS, default_args)


def _concat_dataset(cfg, default_args=None):
    raise NotImplementedError('dataset concatenation is a training utility')


def build_dataloader(dataset, samples_per_gpu=1, workers_per_gpu=0, num_gpus=1, dist=False, shuffle=False, seed=None, **kwargs):
    import torch
    kwargs.pop('persistent_workers', None)
    return torch.utils.data.DataLoader(dataset, batch_size=samples_per_gpu, shuffle=shuffle, num_workers=workers_per_gpu,
                                       collate_fn=lambda b: b[0] if samples_per_gpu == 1 else b)

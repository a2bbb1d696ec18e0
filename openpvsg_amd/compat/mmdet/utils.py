import logging

import torch


def get_root_logger(log_file=None, log_level=logging.INFO):
    logger = logging.getLogger('mmdet')
    if not logger.handlers:
        logging.basicConfig(level=log_level)
    return logger


def replace_cfg_vals(cfg):
    return cfg


def update_data_root(cfg, logger=None):
    return cfg


def compat_cfg(cfg):
    return cfg


def setup_multi_processes(cfg):
    return None


def get_device():
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def build_dp(model, device='cuda', dim=0, *args, **kwargs):
    from mmcv.parallel import MMDataParallel
    return MMDataParallel(model.to(device) if device != 'cpu' else model, **kwargs)


def build_ddp(model, device='cuda', *args, **kwargs):
    return build_dp(model, device)

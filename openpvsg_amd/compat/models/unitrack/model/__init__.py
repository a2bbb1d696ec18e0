from openpvsg_amd.unitrack import AppearanceModel  # noqa: F401


def partial_load(pretrained_dict, model, skip_keys=[], log=False):
    """models/unitrack/model/model.py:23-41: load the keys the model has, minus skip_keys."""
    own = model.state_dict()
    own.update({k: v for k, v in pretrained_dict.items() if k in own and not any(sk in k for sk in skip_keys)})
    model.load_state_dict(own)

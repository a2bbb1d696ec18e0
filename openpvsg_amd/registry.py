"""Registry + build_from_cfg: the reference builds every block from `dict(type=..., **kwargs)`
through mmcv registries (models/mask2former/mask2former_head.py:93-95,108;
models/mask2former/mask2former.py:32-43; tools/test.py:229).  This is a clean-room equivalent with
the same call surface (`@X.register_module()`, `X.build(cfg)`, `build_from_cfg(cfg, X, defaults)`)."""
import inspect


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self.build_func = build_func or build_from_cfg
        self.parent = parent

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return 'Registry(name=%s, items=%s)' % (self._name, sorted(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError('%s is already registered in %s' % (n, self._name))
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type) or inspect.isfunction(name):  # used as a bare decorator
            self._register(name)
            return name
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict, got %s' % type(cfg))
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError('`cfg` or `default_args` must contain the key "type", got %s' % (cfg,))
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        cls = registry.get(obj_type)
        if cls is None:
            raise KeyError('%s is not in the %s registry' % (obj_type, registry.name))
    elif inspect.isclass(obj_type) or inspect.isfunction(obj_type):
        cls = obj_type
    else:
        raise TypeError('type must be a str or class, got %s' % type(obj_type))
    try:
        return cls(**args)
    except Exception as e:
        raise type(e)('%s: %s' % (cls.__name__, e))


# the registries the reference's configs and in-repo classes use
ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('feed-forward Network')
POSITIONAL_ENCODING = Registry('position encoding')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
PLUGIN_LAYERS = Registry('plugin layer')
BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
DETECTORS = Registry('detector')
LOSSES = Registry('loss')
DATASETS = Registry('dataset')
PIPELINES = Registry('pipeline')


def build_attention(cfg, default_args=None):
    return build_from_cfg(cfg, ATTENTION, default_args)


def build_feedforward_network(cfg, default_args=None):
    return build_from_cfg(cfg, FEEDFORWARD_NETWORK, default_args)


def build_positional_encoding(cfg, default_args=None):
    return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)


def build_transformer_layer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER, default_args)


def build_transformer_layer_sequence(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER_LAYER_SEQUENCE, default_args)


def build_plugin_layer(cfg, postfix='', **kwargs):
    """mmcv.cnn.build_plugin_layer: returns (name, layer)."""
    args = dict(cfg)
    layer = build_from_cfg(dict(args, **kwargs), PLUGIN_LAYERS)
    return str(args['type']).lower() + str(postfix), layer


def build_backbone(cfg):
    return build_from_cfg(cfg, BACKBONES)


def build_neck(cfg):
    return build_from_cfg(cfg, NECKS)


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)


def build_loss(cfg):
    """Inference backend: losses are not part of the hot path."""
    return None


def build_detector(cfg, train_cfg=None, test_cfg=None):
    args = dict(cfg)
    if train_cfg is not None:
        args.setdefault('train_cfg', train_cfg)
    if test_cfg is not None:
        args.setdefault('test_cfg', test_cfg)
    return build_from_cfg(args, DETECTORS)
